#!/bin/bash
# round 6, session 40: per-kernel times of both workloads at HEAD (rocprofv3 --kernel-trace --stats only)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s40
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 KR_ITEM_PLACEMENT_TRIALS=0
D=$(mktemp -d /tmp/krprof.XXXXXX)
for wl in syn1000 syn10000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof_$wl -- python3 bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --distinct-batches 1 --skip-host-path-check > gpurun_out/s40/bench_$wl.log 2>&1
  f=$(grep -l kr_scan_pipe $(find $D/prof_$wl -name "*kernel_stats.csv") | head -n 1) # (the bench's own process, not its helpers')
  python3 - "$f" $wl <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("==", sys.argv[2])
for r in rows:
    n = r["Name"]
    if "kr_" not in n: continue
    short = n.split("kr_")[1].split("(")[0].split("<")[0]
    print(f"  kr_{short:28s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e6:8.3f} ms")
PY
done > gpurun_out/s40/kernel_times.txt 2>&1
cat gpurun_out/s40/kernel_times.txt
