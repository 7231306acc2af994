#!/bin/bash
# round 3, session K: place with precomputed (leaf, ancestor) weights; event region of the two-segment accumulate instantiation; the 10,000-genome workload
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_place.py tests/test_gpu_place_k27.py tests/test_gpu_syn1000.py::test_place_on_the_1000_genome_tree_never_leaves_the_device -m gpu -x -q > gpurun_out/r3k_tests.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r3k_tests.log; tail -4 gpurun_out/r3k_tests.log
if [ $rc -ne 0 ]; then grep -n "^E " gpurun_out/r3k_tests.log | head -20; fi
KR_PLACE_TIMING=1 python scripts/time_place_big.py 400000 > gpurun_out/r3k_place.log 2>&1; grep -v "^\[place" gpurun_out/r3k_place.log | tail -3; grep "place/device" gpurun_out/r3k_place.log | tail -6
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 4000 --distinct-batches 1"
one() { name=$1; shift; echo -n "$name: "; env "$@" 2>gpurun_out/r3k_$name.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"; rm -rf /tmp/krepp_bench_*; }
one len250_ev4 X=1 $B --read-len 250 --reads-per-step 4000000
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for v in ev3 ev6; do cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; one len250_$v X=1 $B --read-len 250 --reads-per-step 4000000; done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
one syn10000 X=1 python bench.py --workload syn10000 --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --check-reads 4000 --distinct-batches 1
