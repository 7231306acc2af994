#!/bin/bash
# round 4: does memory held between the index and the stream (KR_BENCH_BALLAST_GB) decide the scan's launch-time level?  Alternating runs.
ulimit -c 0
mkdir -p gpurun_out
B="--steps 10 --warmup 3 --no-cpu-baseline --no-host-inclusive --skip-host-path-check"
for k in 1 2 3 4; do
  for g in 0 48 96; do
    KR_BENCH_BALLAST_GB=$g python3 bench.py $B > gpurun_out/ball_${g}_$k.json 2> gpurun_out/ball_${g}_$k.err
    python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ball_${g}_$k.json") if l.startswith("{\"metric\"")][-1])
print("ballast $g GB run $k:", round(d["value"]/1e6,1), "M reads/s; scan per launch", [round(x,1) for x in d["kernel_ms"]["scan_per_launch"]], d["config"]["item_list_placement"]["kept"])
PY
  done
done
