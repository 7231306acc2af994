#!/usr/bin/env python3
"""HBM traffic of kr_scan_kernel per launch from the rocprofv3 PMC passes of scripts/profile.sh.

usage: traffic.py <gpurun_out/prof_TAG dir> <bench json line file> <out json>

Method (MI355X_MICROARCH.md "HBM" + profiles/round1_h_fetch_calibration.txt): FETCH_SIZE and WRITE_SIZE come from
separate --pmc passes, in KB.  On gfx950 FETCH_SIZE = (L2 -> fabric read requests) x 64 B with ONE request per
128-byte line touched (calibrated on this kernel's access shapes): a request for a whole line is tallied at half
its bytes, so line reads are doubled.  WRITE_SIZE reads exactly.

Slotted table (the default on dense tables): every bucket probe reads one aligned slot = whole 128-byte lines and
nothing else (no descriptor gather, no colour gather: that moved to kr_acc_kernel), the read bases stream as whole
lines too: HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE.
Packed table (KR_SLOT_LOG2W=0): one isolated 8-byte descriptor gather per probe (oracle count `probes`) moves one
64-byte sector as the counter says; the rest are lines.
"""
import json, subprocess, sys
prof, benchf, outf = sys.argv[1:4]
tag = prof.rstrip('/').split('prof_')[-1]
summ = json.load(open(f"{prof}/summary_{tag}.json"))
line = [l for l in open(benchf) if l.startswith('{"metric"')][-1]
b = json.loads(line)
pmc = {k: v for k, v in summ["pmc_max"].items() if "kr_scan" in k}  # max over dispatches = a full-size launch
(name, c), = pmc.items()
fetch_b, write_b = c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
n = b["roofline"]["reads_per_launch"]
per = b["roofline"]["oracle_counts_per_read"]
req = fetch_b / 64.0
slotted = "--packed" not in sys.argv  # scripts/profile.sh runs the default (slotted) layout; pass --packed for KR_SLOT_LOG2W=0 runs
small = 0.0 if slotted else min(req, n * per["probes"])  # isolated descriptor gathers: one request, 64 B each
lines = req - small
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krepp_amd import srcinfo
# the GPU box has no .git: build() recorded, where it had one, the commit that last touched the scan kernel's sources and a
# digest of them; the digest is recomputed here from the files that were profiled
bi = srcinfo.build_info()
commit = bi.get("scan_commit")
out = {
    "kernel": b["roofline"].get("kernel", "kr_scan_kernel"), "workload": b["config"]["workload"], "table": "slotted" if slotted else "packed",
    "profile": tag, "commit": commit, "head_at_build": bi.get("head"), "scan_src_sha": bi["scan_src_sha"],
    "scan_sources": list(srcinfo.SCAN_SOURCES), "scan_ms": b["kernel_ms"]["scan"],
    "reads_per_launch": n, "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
    "read_requests": req, "isolated_gather_requests": small, "line_requests": lines,
    "hbm_bytes_lower_bound_all_64B": fetch_b + write_b,
    "hbm_bytes_upper_bound_all_128B": 2 * fetch_b + write_b,
    "hbm_bytes_per_launch": lines * 128.0 + small * 64.0 + write_b,
    "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_read"] * n,
    "hbm_GBps_at_scan_ms": (lines * 128.0 + small * 64.0 + write_b) / (b["kernel_ms"]["scan"] * 1e-3) / 1e9,
    "method": __doc__.split("Method", 1)[1].strip(),
}
# ---- the other stages: every kernel's full-size launch (maximum over its dispatches), summed per stage.  Their reads are mostly
# isolated gathers (colour ids, se_to_pse pairs, (d, v) of a problem): a request moves a 64-byte sector or a 128-byte line, the
# counter does not say which, so both bounds are kept and the LOWER one (every request one sector) is what bench.py reports.
stages = {"accumulate": ("kr_acc_kernel",), "llh_select": ("kr_dedup", "kr_llh", "kr_select", "kr_rows")}
out["stages"] = {}
for st_, pats in stages.items():
    f_kb = w_kb = 0.0
    names = []
    for k_, c_ in summ["pmc_max"].items():
        if any(p_ in k_ for p_ in pats) and "FETCH_SIZE" in c_ and "WRITE_SIZE" in c_:
            f_kb += c_["FETCH_SIZE"]
            w_kb += c_["WRITE_SIZE"]
            names.append(k_)
    out["stages"][st_] = {"kernels": names, "FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb,
                          "hbm_bytes_lower_bound_all_64B": (f_kb + w_kb) * 1024.0, "hbm_bytes_upper_bound_all_128B": (2 * f_kb + w_kb) * 1024.0,
                          "hbm_bytes_per_launch": (f_kb + w_kb) * 1024.0, "ms": b["kernel_ms"]["accumulate" if st_ == "accumulate" else "llh_select"]}
out["stage_src_sha"] = srcinfo.stage_src_sha()
out["stage_sources"] = list(srcinfo.STAGE_SOURCES)
json.dump(out, open(outf, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("table", "hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "read_requests", "hbm_GBps_at_scan_ms")}))
