#!/bin/bash
# round 6, session 46: per-kernel times of the host-inclusive leg's row kernels (rocprofv3 --kernel-trace --stats over a bench run with that leg)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s46
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 KR_ITEM_PLACEMENT_TRIALS=0
D=$(mktemp -d /tmp/krprof.XXXXXX)
rocprofv3 --kernel-trace --stats --output-format csv -d $D/p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --read-procs 1 --distinct-batches 1 --skip-host-path-check --no-whole-launch-check > gpurun_out/s46/bench.log 2>&1
f=$(grep -l kr_scan_pipe $(find $D/p -name "*kernel_stats.csv") | head -n 1)
python3 - "$f" <<'PY' | tee gpurun_out/s46/kernel_times.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "kr_" not in n: continue
    short = n.split("kr_")[1].split("(")[0]
    print(f"  kr_{short[:60]:60s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e6:8.3f} ms  min {float(r['MinNs'])/1e6:8.3f}")
PY
