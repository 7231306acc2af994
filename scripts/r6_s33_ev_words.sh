#!/bin/bash
# round 6, session 33: the accumulate kernel's LDS layout on the 10,000-genome index -- 8,256 B a wave (19 waves per CU) against
# 8,192 (20), 7,168 (22: 768 events in LDS) and 6,656 (24: 640 events).  Variants built by scripts/build_variant.sh:
#   ev1200 -DKR_ACC_LEAN_EV_WORDS=1200 ; ev944 -DKR_ACC_LEAN_EV_WORDS=944 -DKR_ACC_EV_CAP=768 ; ev816 -DKR_ACC_LEAN_EV_WORDS=816 -DKR_ACC_EV_CAP=640
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s33
cp krepp_amd/lib/libkrepp_amd.so /tmp/base.so
for v in base ev1200 ev944 ev816; do
  if [ $v = base ]; then cp /tmp/base.so krepp_amd/lib/libkrepp_amd.so; else cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; fi
  timeout 600 python bench.py --workload syn10000 --steps 8 --warmup 3 --no-cpu-baseline --no-host-inclusive > gpurun_out/s33/$v.json 2> gpurun_out/s33/$v.err
  python - <<PY
import json
try:
    o=json.loads([l for l in open("gpurun_out/s33/$v.json") if l.startswith("{")][0])
    k={x["stage"]:x["avg_launch_ms"] for x in o["roofline"]["kernels"]}
    print("$v", round(o["value"]/1e6,2), k, o["check"]["rows_equal"], o["check"]["whole_launch"]["equal_on_an_independent_stream"])
except Exception as e: print("$v failed", e)
PY
done
cp /tmp/base.so krepp_amd/lib/libkrepp_amd.so
