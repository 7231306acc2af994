#!/usr/bin/env python3
"""Text summary of one `scripts/profile.sh <tag>` run for profiles/: the kr_* rows of kernel_stats.csv, per-kernel durations from
kernel_trace.csv, and the per-dispatch averages of every PMC pass.  usage: rocprof_summary.py gpurun_out/prof_<tag> [header file] > out.txt"""
import collections, csv, glob, os, sys

d = sys.argv[1].rstrip("/")
if len(sys.argv) > 2:
    sys.stdout.write(open(sys.argv[2]).read().rstrip("\n") + "\n\n")
print("== rocprofv3 --kernel-trace --stats (kernel_stats.csv) ==")
first = True
for f in sorted(glob.glob(d + "/trace/**/*kernel_stats.csv", recursive=True)):
    rows = list(csv.reader(open(f)))
    if first:
        print(",".join(rows[0]))
        first = False
    for r in rows[1:]:
        if "kr_" in r[0]:
            print(",".join(r))
print("\n== per-kernel durations from kernel_trace.csv (ms): name, calls, avg, min, max ==")
dur = collections.defaultdict(list)
for f in sorted(glob.glob(d + "/trace/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "kr_" in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[:100]}, {len(v)}, {sum(v) / len(v):.4f}, {min(v):.4f}, {max(v):.4f}")
for p in sorted(glob.glob(d + "/pmc_*")):
    print(f"\n== {os.path.basename(p)}: per-dispatch counter averages ==")
    acc = collections.defaultdict(list)
    for f in sorted(glob.glob(p + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if "kr_" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"], r["Counter_Name"])].append((r["Dispatch_Id"], float(r["Counter_Value"])))
    for (k, c), v in acc.items():
        per = collections.defaultdict(float)
        for did, x in v:
            per[did] += x
        xs = list(per.values())
        print(f"{k[:90]}, {c}, n={len(xs)}, avg={sum(xs) / len(xs):.6g}, max={max(xs):.6g}")
