#!/bin/bash
# round 6, session 39: the de-duplication table's size now that four records in five go to the direct part: KR_DD_SHIFT (slots = record
# slots >> shift, 16 B each, cleared every batch; 1 since round 4, measured then with every record in the table)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s39
for wl in syn1000 syn10000; do
for sh in 1 3 4 5 6; do
  KR_DD_SHIFT=$sh timeout 600 python bench.py --workload $wl --steps 8 --warmup 3 --no-cpu-baseline --no-host-inclusive > gpurun_out/s39/${wl}_$sh.json 2> gpurun_out/s39/${wl}_$sh.err
  python - <<PY
import json
try:
    o=json.loads([l for l in open("gpurun_out/s39/${wl}_$sh.json") if l.startswith("{")][0])
    k={x["stage"]:round(x["avg_launch_ms"],2) for x in o["roofline"]["kernels"]}
    print("$wl shift $sh", round(o["value"]/1e6,2), k, o["check"]["rows_equal"], o["check"]["whole_launch"]["equal_on_an_independent_stream"])
except Exception as e: print("$wl $sh failed", e)
PY
done; done
