#!/bin/bash
# Round 3, session T: which of a stream's buffers decides the scan's launch-time level?  (kr_debug_stream_move) + syn10000 on the final code
ulimit -c 0
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
for rep in 1 2; do
  python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --stream-variance 5 --stream-variance-move 4,2,1,0,3 \
     > gpurun_out/r3t_move$rep.json 2> gpurun_out/r3t_move$rep.err
  grep "stream-variance" gpurun_out/r3t_move$rep.err
done
rm -rf /tmp/krepp_bench_*
python3 bench.py --workload syn10000 --steps 10 --warmup 2 --no-cpu-baseline --no-host-inclusive > gpurun_out/r3t_syn10000.json 2> gpurun_out/r3t_syn10000.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r3t_syn10000.json').read().strip().splitlines()[-1])
print('syn10000:', round(d['value'] / 1e6, 2), round(d['ms_per_step'], 2), {k: round(v, 2) for k, v in d['kernel_ms'].items() if isinstance(v, float)}, d['check']['rows_equal'])
PY
