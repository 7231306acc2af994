#!/bin/bash
# round 5, session 1: the three verification tests of the round-4 review + this box's baseline lines + the accumulate kernel's
# per-read statistics on both synthetic indexes (the KR_STATS build: scripts/build_variant.sh stats -DKR_STATS=1)
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_syn1000.py tests/test_gpu_bench.py::test_bench_eight_ranks_share_one_gpu -x -q -s --durations=8 > gpurun_out/r5_s1_tests.txt 2>&1
tail -25 gpurun_out/r5_s1_tests.txt
B="--no-cpu-baseline --no-host-inclusive --steps 8 --warmup 2 --check-reads 2000 --skip-host-path-check"
python bench.py $B > gpurun_out/r5_s1_syn1000.json 2> gpurun_out/r5_s1_syn1000.err
python bench.py --workload syn10000 $B > gpurun_out/r5_s1_syn10000.json 2> gpurun_out/r5_s1_syn10000.err
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
cp krepp_amd/lib/variants/stats/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
S="--no-cpu-baseline --no-host-inclusive --steps 1 --warmup 0 --check-reads 1000 --skip-host-path-check --distinct-batches 1"
KR_ITEM_PLACEMENT_TRIALS=0 KR_DEBUG_SKIP=512 python bench.py --workload syn10000 $S > gpurun_out/r5_s1_stats10000.json 2> gpurun_out/r5_s1_stats10000.err
KR_ITEM_PLACEMENT_TRIALS=0 KR_DEBUG_SKIP=512 python bench.py $S > gpurun_out/r5_s1_stats1000.json 2> gpurun_out/r5_s1_stats1000.err
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
for f in syn1000 syn10000; do python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r5_s1_$f.json') if l.startswith('{')][-1]); print('$f', round(d['value']/1e6,2), {k:(round(x,2) if isinstance(x,float) else x) for k,x in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"; done
grep -h "kr stats" gpurun_out/r5_s1_stats10000.err | tail -8
grep -h "kr stats" gpurun_out/r5_s1_stats1000.err | tail -8
