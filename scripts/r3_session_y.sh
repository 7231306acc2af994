#!/bin/bash
# Round 3, session Y: select kernel three loads deep across reads, dedup kernel with one 16-byte probe per slot; larger batches
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py -x -q -k "golden or report_modes or large_batch or device_brent or where_a_streams or lanes" 2>&1 | tail -3
python3 -m pytest tests/test_gpu_syn1000.py -x -q -k "slotted" 2>&1 | tail -3
rm -rf /tmp/pytest-of-* /tmp/krepp_*
trace() {
  OUT=$PWD/gpurun_out/r3y_$1; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --distinct-batches 1 > $OUT/bench.log 2>&1
  echo "== $1"; python3 scripts/kstats.py $OUT | grep -E "select|dedup_kernel|llh_kernel|acc_kernel_t<true, 5, false, 7>|scan_pipe" | cut -c1-200
  rm -rf /tmp/krepp_bench_*
}
trace new
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
cp krepp_amd/lib/variants/probe8/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
trace probe8
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
for n in 8000000 12000000 16000000; do
  python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --reads-per-step $n > gpurun_out/r3y_n$n.json 2> gpurun_out/r3y_n$n.err
  python3 - gpurun_out/r3y_n$n.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
n = int(sys.argv[2]) / 1e6
print(sys.argv[2], 'reads per launch:', round(d['value'] / 1e6, 2), 'M reads/s', {k: round(v / n, 3) for k, v in d['kernel_ms'].items() if isinstance(v, float) and k in ('scan', 'accumulate', 'llh_select')}, 'ms per million reads', d['check']['rows_equal'])
PY
  rm -rf /tmp/krepp_bench_*
done
