#!/bin/bash
# round 6, session 44: the select kernel's 64-read chunks handed out by a counter (HEAD) against by grid position (variants/dyn0)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s44
cp krepp_amd/lib/libkrepp_amd.so /tmp/base.so
for wl in syn10000 syn1000; do
for v in dyn0 base dyn0 base; do
  if [ $v = base ]; then cp /tmp/base.so krepp_amd/lib/libkrepp_amd.so; else cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; fi
  timeout 600 python bench.py --workload $wl --steps 8 --warmup 3 --no-cpu-baseline --no-host-inclusive --no-whole-launch-check > gpurun_out/s44/${wl}_$v.json 2> gpurun_out/s44/${wl}_$v.err
  python - <<PY
import json
try:
    o=json.loads([l for l in open("gpurun_out/s44/${wl}_$v.json") if l.startswith("{")][0])
    k={x["stage"]:round(x["avg_launch_ms"],2) for x in o["roofline"]["kernels"]}
    print("$wl $v", round(o["value"]/1e6,2), k, o["check"]["rows_equal"])
except Exception as e: print("$wl $v failed", e)
PY
done; done
cp /tmp/base.so krepp_amd/lib/libkrepp_amd.so
