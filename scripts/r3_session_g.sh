#!/bin/bash
# round 3, session G: is the scan's launch-time level a property of the hardware queue or of the buffers?  place throughput on the 1000-genome tree; 250-bp reads
mkdir -p gpurun_out
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1"
one() { name=$1; shift; echo -n "$name: "; env "$@" 2>gpurun_out/r3g_$name.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"; grep stream-variance gpurun_out/r3g_$name.err | sed 's/\[stream-variance\] //'; }
one q0_a KR_DEBUG_EXTRA_STREAMS=0 $B --stream-variance 5
one q0_b KR_DEBUG_EXTRA_STREAMS=0 $B --stream-variance 5
one q1_a KR_DEBUG_EXTRA_STREAMS=1 $B --stream-variance 5
one q1_b KR_DEBUG_EXTRA_STREAMS=1 $B --stream-variance 5
one q3_a KR_DEBUG_EXTRA_STREAMS=3 $B --stream-variance 5
one hwq1 GPU_MAX_HW_QUEUES=1 $B --stream-variance 5
python scripts/time_place_big.py 400000 2>&1 | tail -4
one len250 X=1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1 --read-len 250 --reads-per-step 4000000
