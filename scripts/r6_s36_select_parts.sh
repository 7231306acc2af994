#!/bin/bash
# round 6, session 36: where the select kernel's time goes -- variants that leave a class of reads out (-DKR_SELECT_SKIP=k, wrong
# results, timing only): 1 the lanes' own walks (reads of at most 8 records), 2 / 4 the groups of 16 / 32 lanes, 8 the reads of
# 33..64 records, 16 the larger ones, 31 all of them (the stage's other kernels remain).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s36
cp krepp_amd/lib/libkrepp_amd.so /tmp/base.so
for wl in syn1000; do
for v in base skip1 skip2 skip4 skip8 skip16 skip31; do
  if [ $v = base ]; then cp /tmp/base.so krepp_amd/lib/libkrepp_amd.so; else cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; fi
  timeout 600 python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --no-whole-launch-check --check-reads 1000 > gpurun_out/s36/${wl}_$v.json 2> gpurun_out/s36/${wl}_$v.err
  python - <<PY
import json
try:
    o=json.loads([l for l in open("gpurun_out/s36/${wl}_$v.json") if l.startswith("{")][0])
    k={x["stage"]:round(x["avg_launch_ms"],2) for x in o["roofline"]["kernels"]}
    print("$wl $v", round(o["value"]/1e6,2), k, o["check"]["rows_equal"])
except Exception as e: print("$wl $v failed", e)
PY
done; done
cp /tmp/base.so krepp_amd/lib/libkrepp_amd.so
