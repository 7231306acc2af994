#!/usr/bin/env python3
"""Merged timeline of kernels and memory copies from a rocprofv3 --kernel-trace --memory-copy-trace run.
usage: timeline.py <dir> [last_ms | "copies"]
  last_ms  : print the last `last_ms` milliseconds of activity (default 150)
  "copies" : print the span that holds the large (> 1 ms) host<->device copies -- the host-inclusive leg of bench.py --
             and say how much of the copy time ran beside a kernel"""
import csv, glob, os, sys
d = sys.argv[1]
mode = sys.argv[2] if len(sys.argv) > 2 else "150"
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"]
        nm = nm.split("(anonymous namespace)::")[1][:34] if "(anonymous namespace)::kr_" in nm else nm[:34]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", nm, r.get("Queue_Id", "")))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "").replace("MEMORY_COPY_", ""), "s" + r.get("Stream_Id", "")))
ev.sort()
if mode == "copies":
    big = [e for e in ev if e[2] == "C" and e[1] - e[0] > 1_000_000]
    t0, t1 = big[0][0] - 2_000_000, big[-1][1] + 2_000_000
    ks = [(s, e) for s, e, k, _, _ in ev if k == "K" and e > t0 and s < t1]
    tot = ov = 0
    for s, e, k, _, _ in big:
        tot += e - s
        ov += sum(max(0, min(e, ke) - max(s, ks_)) for ks_, ke in ks)
    print(f"# {len(big)} copies of more than 1 ms, {tot / 1e6:.1f} ms in all, {ov / 1e6:.1f} ms of it ({100.0 * ov / max(1, tot):.0f} %) beside a running kernel")
    n = 0
    for s, e, k, nm, q in ev:
        if e < t0 or s > t1 or (e - s) < 300_000:
            continue
        print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f} {k} {q:>4} {nm}")
        n += 1
        if n >= 140:
            print("# ...")
            break
else:
    last_ms = float(mode)
    t_end = max(e[1] for e in ev)
    t0 = t_end - int(last_ms * 1e6)
    for s, e, k, nm, q in ev:
        if e < t0 or (e - s) < 20000:
            continue
        print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f} {k} {q:>4} {nm}")
