#!/usr/bin/env python3
"""HBM traffic of kr_scan_kernel per launch from the rocprofv3 PMC passes of scripts/profile.sh.

usage: traffic.py <gpurun_out/prof_TAG dir> <bench json line file> <out json>

Method (MI355X_MICROARCH.md §HBM + profiles/round1_h_fetch_calibration.txt): FETCH_SIZE and WRITE_SIZE come from
separate --pmc passes, in KB.  On gfx950 FETCH_SIZE = (L2 -> fabric read requests) x 64 B: a request for a
full or half 128-byte line is tallied as 64 B.  The scan kernel's bucket reads are 16 B per lane over
contiguous 64-byte pieces (lines are fetched whole: x2), its descriptor (8 B) and colour (4 B) gathers touch one
line each (counted as the 64 B the counter reports).  The request mix comes from the oracle's exact counts per
read (bench.py roofline.oracle_counts_per_read): probes -> descriptor requests, hits -> colour requests, the
rest of the requests are bucket lines.
"""
import json, sys
prof, benchf, outf = sys.argv[1:4]
summ = json.load(open(f"{prof}/summary_{prof.rstrip('/').split('prof_')[-1]}.json"))
line = [l for l in open(benchf) if l.startswith('{"metric"')][-1]
b = json.loads(line)
pmc = {k: v for k, v in summ["pmc_max"].items() if "kr_scan" in k}  # max over dispatches = a full-size launch
(name, c), = pmc.items()
fetch_b, write_b = c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
n = b["roofline"]["reads_per_launch"]
per = b["roofline"]["oracle_counts_per_read"]
req = fetch_b / 64.0
small = min(req, n * (per["probes"] + per["hits"]))  # isolated gathers: one request, 64 B each
lines = req - small                                   # bucket lines (+ the read bases): 128 B each
out = {
    "kernel": "kr_scan_kernel", "workload": b["config"]["workload"], "reads_per_launch": n,
    "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
    "read_requests": req, "isolated_gather_requests": small, "line_requests": lines,
    "hbm_bytes_lower_bound_all_64B": fetch_b + write_b,
    "hbm_bytes_upper_bound_all_128B": 2 * fetch_b + write_b,
    "hbm_bytes_per_launch": lines * 128.0 + small * 64.0 + write_b,
    "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_read"] * n,
    "method": __doc__.split("Method", 1)[1].strip(),
}
json.dump(out, open(outf, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "read_requests")}))
