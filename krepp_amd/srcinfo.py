"""Which state of the scan kernel's sources a profile belongs to (used by bench.py, scripts/traffic.py and build()).

`roofline.traffic` in bench.py's line comes from PMC passes of an earlier rocprofv3 run (profiles/traffic_latest.json).
It is only meaningful while kr_scan_kernel is the kernel that was profiled, so the profile records a digest of the
kernel's sources and bench.py reports the traffic only when the digest still matches.  The GPU boxes have no .git:
the digest is computed from the files; the commit that last touched them is recorded by build() where .git exists
(krepp_amd/lib/build_info.json, which travels with the built library).
"""
from __future__ import annotations

import hashlib
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# everything kr_scan_kernel_t is compiled from (front end, probe list, slot scan, item stage) and nothing else
SCAN_SOURCES = ("krepp_amd/csrc/kr_dev_scan.inc", "krepp_amd/csrc/kr_dev_scan_pipe.inc", "krepp_amd/csrc/kr_dev_common.inc", "krepp_amd/csrc/kr_devutil.h")
# the kernels behind the other stages of a step (accumulate; de-duplication, likelihood, selection, row compaction): their traffic in
# profiles/traffic_latest.json ("stages") is reported only while THESE are the sources that were profiled
STAGE_SOURCES = ("krepp_amd/csrc/kr_dev_common.inc", "krepp_amd/csrc/kr_dev_expand.inc", "krepp_amd/csrc/kr_dev_accumulate.inc",
                 "krepp_amd/csrc/kr_dev_likelihood.inc", "krepp_amd/csrc/kr_devutil.h")
BUILD_INFO = os.path.join(ROOT, "krepp_amd", "lib", "build_info.json")


def stage_src_sha() -> str:
    h = hashlib.sha256()
    for rel in STAGE_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def scan_src_sha() -> str:
    h = hashlib.sha256()
    for rel in SCAN_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _git(*args):
    r = subprocess.run(["git", "-C", ROOT, *args], capture_output=True, text=True)
    return r.stdout.strip() if r.returncode == 0 else None


def write_build_info() -> dict | None:
    """Called by build() where the repository is a git checkout; a no-op elsewhere (the GPU box uses the file it was sent)."""
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        return None
    dirty = bool(_git("status", "--porcelain", "--", *SCAN_SOURCES))
    info = {"head": _git("rev-parse", "--short", "HEAD"),
            "scan_commit": None if dirty else _git("log", "-1", "--format=%h", "--", *SCAN_SOURCES),
            "scan_sources_dirty": dirty, "scan_src_sha": scan_src_sha(), "scan_sources": list(SCAN_SOURCES)}
    os.makedirs(os.path.dirname(BUILD_INFO), exist_ok=True)
    with open(BUILD_INFO, "w") as f:
        json.dump(info, f, indent=1)
    return info


def build_info() -> dict:
    """build_info.json if it describes the sources as they are now, else just the digest."""
    sha = scan_src_sha()
    try:
        info = json.load(open(BUILD_INFO))
        if info.get("scan_src_sha") == sha:
            return info
    except Exception:
        pass
    return {"head": None, "scan_commit": None, "scan_src_sha": sha}
