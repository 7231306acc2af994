#!/bin/bash
# round 4: the same, trials first on a fresh box
ulimit -c 0
mkdir -p gpurun_out
for k in 1 2 3; do
  for t in 3 0; do
    KR_ITEM_PLACEMENT_VERBOSE=1 KR_ITEM_PLACEMENT_TRIALS=$t python3 bench.py --steps 6 --warmup 9 --no-cpu-baseline --no-host-inclusive --skip-host-path-check > gpurun_out/tr2_${t}_$k.json 2> gpurun_out/tr2_${t}_$k.err
    python3 - <<PY
import json,re
d=json.loads([l for l in open("gpurun_out/tr2_${t}_$k.json") if l.startswith("{\"metric\"")][-1])
tr=[re.sub(r"^\[krepp_amd\] item list","",l.strip()) for l in open("gpurun_out/tr2_${t}_$k.err") if "item list" in l]
print("trials $t run $k:", round(d["value"]/1e6,1), "M reads/s; scan per launch", [round(x,1) for x in d["kernel_ms"]["scan_per_launch"]], d["config"]["item_list_placement"]["tried"], d["config"]["item_list_placement"]["kept"], "|", "; ".join(tr)[:330])
PY
  done
done
