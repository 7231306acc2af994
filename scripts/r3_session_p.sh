#!/bin/bash
# round 3, session P: does the third accumulate launch cost the 150-bp workload anything?  (accumulate 21.1 ms before it, 22.1 after)
ulimit -c 0
mkdir -p gpurun_out
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1"
one() { name=$1; shift; echo -n "$name: "; env "$@" 2>gpurun_out/r3p_$name.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"; rm -rf /tmp/krepp_bench_*; }
one lean2_on_a X=1 $B
one lean2_off_a KR_DEBUG_NO_LEAN2=1 $B
one lean2_on_b X=1 $B
one lean2_off_b KR_DEBUG_NO_LEAN2=1 $B
