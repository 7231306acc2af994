#!/bin/bash
# round 3, session L: long sequences across waves (parity, rate), the whole GPU suite, 250-bp and 10,000-genome rates after the fixes
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_long_sequences.py -m gpu -x -q > gpurun_out/r3l_tiles.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r3l_tiles.log; tail -4 gpurun_out/r3l_tiles.log
if [ $rc -ne 0 ]; then grep -n "^E " gpurun_out/r3l_tiles.log | head -30; fi
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_long_sequences.py > gpurun_out/r3l_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r3l_pytest_gpu.log; tail -6 gpurun_out/r3l_pytest_gpu.log
if [ $rc -eq 0 ]; then python scripts/time_contigs.py 400000 8 2>&1 | tail -4; python scripts/time_contigs.py 5000 2000 2>&1 | tail -4; fi
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 4000 --distinct-batches 1"
one() { name=$1; shift; echo -n "$name: "; env "$@" 2>gpurun_out/r3l_$name.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"; rm -rf /tmp/krepp_bench_*; }
one syn10000 X=1 python bench.py --workload syn10000 --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --check-reads 4000 --distinct-batches 1
one len250 X=1 $B --read-len 250 --reads-per-step 4000000
one len150 X=1 $B
