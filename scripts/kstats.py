#!/usr/bin/env python3
"""Print the kr_* rows of a rocprofv3 kernel_stats.csv (name truncated).  usage: kstats.py <prof dir>"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kr_" in r["Name"]:
            n = r["Name"].split("kr_")[1][:40]
            print(f"kr_{n:40s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:10.1f} max_us {float(r['MaxNs'])/1e3:10.1f}")
