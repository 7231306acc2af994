#!/bin/bash
# round 3, session Q: accumulate kernel before / after the two-segment instantiation (same box)
ulimit -c 0
mkdir -p gpurun_out
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1"
one() { name=$1; shift; echo -n "$name: "; env "$@" 2>gpurun_out/r3q_$name.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"; rm -rf /tmp/krepp_bench_*; }
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
one head_a X=1 $B
for v in old40d8 mid95f4; do cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; one $v X=1 $B; done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
one head_b X=1 $B
