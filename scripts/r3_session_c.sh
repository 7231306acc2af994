#!/bin/bash
# round 3, session C: scan of batch i+1 beside accumulate / likelihood of batch i (two kernel chains)
mkdir -p gpurun_out
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1"
one() { # name, args..., env via leading VAR=VAL words
  name=$1; shift
  echo -n "$name: "
  env "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"
}
one serial X=1 $B
one serial_2streams X=1 $B --pipeline-streams 2
one acc8 KR_DEBUG_ACC_WAVES=8 $B
one acc12 KR_DEBUG_ACC_WAVES=12 $B
one acc16 KR_DEBUG_ACC_WAVES=16 $B
one ovl_d2_s2_a8 KR_OVERLAP=1 $B --pipeline-streams 2
one ovl_d2_s2_a12 KR_OVERLAP=1 KR_OVERLAP_ACC_WAVES=12 $B --pipeline-streams 2
one ovl_d2_s2_a6 KR_OVERLAP=1 KR_OVERLAP_ACC_WAVES=6 $B --pipeline-streams 2
one ovl_d1_s2_a12 KR_OVERLAP=1 KR_OVERLAP_SCAN_D1=1 KR_OVERLAP_ACC_WAVES=12 $B --pipeline-streams 2
one ovl_d1_s2_a8 KR_OVERLAP=1 KR_OVERLAP_SCAN_D1=1 KR_OVERLAP_ACC_WAVES=8 $B --pipeline-streams 2
one ovl_d1_s3_a4 KR_OVERLAP=1 KR_OVERLAP_SCAN_D1=1 KR_OVERLAP_SCAN_BLOCKS=3 KR_OVERLAP_ACC_WAVES=4 $B --pipeline-streams 2
one ovl_d2_s3_a8 KR_OVERLAP=1 KR_OVERLAP_SCAN_BLOCKS=3 KR_OVERLAP_ACC_WAVES=8 $B --pipeline-streams 2
one ovl_d1_s4_a24 KR_OVERLAP=1 KR_OVERLAP_SCAN_D1=1 KR_OVERLAP_SCAN_BLOCKS=4 KR_OVERLAP_ACC_WAVES=24 $B --pipeline-streams 2
one ovl_d2_s2_a8_3streams KR_OVERLAP=1 $B --pipeline-streams 3
