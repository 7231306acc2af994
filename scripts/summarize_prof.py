#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output of scripts/profile.sh into profiles-ready files."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]


def find_all(sub, pat):
    """every file of a pass: the profiler writes one set per PROCESS, and bench.py has a child (krepp_amd.inflate_worker)"""
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


def find(sub, pat):
    r = find_all(sub, pat)
    return r[0] if r else None


def rows_of(sub, pat):
    for f in find_all(sub, pat):
        yield from csv.DictReader(open(f))


summary = {"tag": tag}
lines = []
st = find("trace", "*kernel_stats.csv")
if st:
    rows = [r for r in rows_of("trace", "*kernel_stats.csv")]
    lines.append("== rocprofv3 --kernel-trace --stats (kernel_stats.csv) ==")
    lines.append(",".join(rows[0].keys()) if rows else "")
    for r in rows:
        lines.append(",".join(r.values()))
    summary["kernel_stats"] = rows
kt = find("trace", "*kernel_trace.csv")
if kt:
    d = defaultdict(list)
    for r in rows_of("trace", "*kernel_trace.csv"):
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    lines.append("\n== per-kernel durations from kernel_trace.csv (ms): name, calls, avg, min, max ==")
    for k_, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"{k_[:110]}, {len(v)}, {sum(v)/len(v):.4f}, {min(v):.4f}, {max(v):.4f}")
for sub in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for r in rows_of(sub, "*counter_collection.csv"):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines.append(f"\n== {sub}: per-dispatch counter averages ==")
    for k_, cs in acc.items():
        if "kr_scan" not in k_ and "kr_acc" not in k_ and "llh" not in k_ and "select" not in k_ and "kr_dedup" not in k_ and "kr_rows" not in k_:
            continue
        for c, v in cs.items():
            lines.append(f"{k_[:90]}, {c}, n={len(v)}, avg={sum(v)/len(v):.6g}, max={max(v):.6g}")
            kk = k_.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]  # kernel name + template arguments
            summary.setdefault("pmc", {}).setdefault(kk, {})[c] = sum(v) / len(v)
            summary.setdefault("pmc_max", {}).setdefault(kk, {})[c] = max(v)  # a full-size launch (the parity-check launch is small)
open(os.path.join(out, f"summary_{tag}.txt"), "w").write("\n".join(lines) + "\n")
json.dump(summary, open(os.path.join(out, f"summary_{tag}.json"), "w"), indent=1)
print("\n".join(lines))
