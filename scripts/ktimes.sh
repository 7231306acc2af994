#!/bin/bash
# Per-kernel launch durations of one short bench run (kernel trace only): max / median per kr_* kernel.  usage: scripts/ktimes.sh <tag> [bench args...]
TAG=$1; shift
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/kt_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --distinct-batches 1 --skip-host-path-check "$@" > $OUT/bench.log 2>&1
python3 - <<PY
import csv,glob,collections,re,statistics
d=collections.defaultdict(list)
for f in glob.glob("$OUT/**/*kernel_trace.csv",recursive=True):  # (bench.py inflates its table in a child process: one trace per process)
  for r in csv.DictReader(open(f)):
    m=re.search(r'(kr_\w+)(<[^>]*>)?', r["Kernel_Name"])
    if m: d[m.group(1)+(m.group(2) or "")].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
tot=0
for k,v in sorted(d.items(), key=lambda kv:-max(kv[1])):
    print(f"{k:60s} n={len(v):3d} max={max(v):8.3f} ms  median={statistics.median(v):8.3f}")
    tot+=max(v)
print("sum of max:", round(tot,2))
PY
find $OUT -name "*.csv" -delete
