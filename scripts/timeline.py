#!/usr/bin/env python3
"""Merged timeline of kernels and memory copies from a rocprofv3 --kernel-trace --memory-copy-trace run.
usage: timeline.py <dir> [last_ms]   (prints the last `last_ms` milliseconds of activity, default 150)"""
import csv, glob, os, sys
d = sys.argv[1]
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"]
        nm = nm.split("(anonymous namespace)::")[-1][:28] if "kr_" in nm else nm[:28]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", nm, r.get("Queue_Id", "")))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "")[:20] + " " + r.get("Bytes", r.get("Size", "")), ""))
ev.sort()
t_end = max(e[1] for e in ev)
t0 = t_end - int(last_ms * 1e6)
for s, e, k, nm, q in ev:
    if e < t0 or (e - s) < 20000:
        continue
    print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f} {k} q{q:>3} {nm}")
