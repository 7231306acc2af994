#!/bin/bash
# round 3, session J: the two-segment accumulate instantiation (parity, 250-bp rate); place with compacted candidates; what the device phase of place spends its time on
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_place.py tests/test_gpu_place_k27.py tests/test_gpu_syn1000.py::test_place_on_the_1000_genome_tree_never_leaves_the_device -m gpu -x -q > gpurun_out/r3j_tests.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r3j_tests.log; tail -5 gpurun_out/r3j_tests.log
if [ $rc -ne 0 ]; then grep -n "^E " gpurun_out/r3j_tests.log | head -20; exit 0; fi
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 4000 --distinct-batches 1"
one() { name=$1; shift; echo -n "$name: "; env "$@" 2>gpurun_out/r3j_$name.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check'])"; rm -rf /tmp/krepp_bench_*; }
one len250_small X=1 $B --read-len 250 --reads-per-step 200000
one len250 X=1 $B --read-len 250 --reads-per-step 4000000
one len200 X=1 $B --read-len 200 --reads-per-step 4000000
one len300 X=1 $B --read-len 300 --reads-per-step 4000000
one len150 X=1 $B
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=$PWD/gpurun_out/r3j_place_trace
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/time_place_big.py 400000 > $OUT/run.log 2>&1
tail -3 $OUT/run.log
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r3j_place_trace/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'kr_' in r['Name']:
        print(r['Name'][:80], r['Calls'], 'avg_ms', round(float(r['AverageNs'])/1e6, 3), 'max_ms', round(float(r['MaxNs'])/1e6, 3))
PY
find $OUT -name "*.csv" -size +5M -delete
