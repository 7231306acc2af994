#!/bin/bash
# round 3, session H: the item list out of 2 MB pieces (HIP virtual-memory API) against the scan's launch-time levels
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or lanes or long_reads or export" > gpurun_out/r3h_tests.log 2>&1; tail -3 gpurun_out/r3h_tests.log
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1"
one() { name=$1; shift; echo -n "$name: "; env "$@" 2>gpurun_out/r3h_$name.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'], round(d['setup_s'],1))"; grep stream-variance gpurun_out/r3h_$name.err | sed 's/\[stream-variance\] //'; }
for i in 1 2 3 4; do one vmm2_$i X=1 $B --stream-variance 4; done
for i in 1 2; do one vmm0_$i KR_ITEMS_VMM=0 $B --stream-variance 4; done
for i in 1 2; do one vmm32_$i KR_ITEMS_VMM=16 $B --stream-variance 4; done
