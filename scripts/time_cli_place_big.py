#!/usr/bin/env python3
"""`krepp place` end to end (reader thread -> GPU workers -> ordered writer) on the 1000-genome index with its own Yule tree as
backbone.  usage: time_cli_place_big.py [reads, default 2,000,000]"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from krepp_amd import capi, synth
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
work = tempfile.mkdtemp(prefix="krepp_clipl_")
nwk_text = synth.yule_newick(1000, 2)
genomes = synth.evolve_genomes(nwk_text, 100_000, seed=2)
open(work + "/y.nwk", "w").write(nwk_text)
tsv = synth.write_genomes(genomes, work + "/g")
idx = work + "/idx"
capi.build_index(tsv, idx, nwk=work + "/y.nwk", k=29, w=35, h=13, m=4, r=1, frac=True, num_threads=min(32, os.cpu_count() or 1))
fq = work + "/reads.fq"
with open(fq, "wb") as f:
    done = 0
    while done < n:
        m = min(200_000, n - done)
        b = np.concatenate([synth.sample_reads(genomes, min(100_000, m - o), seed=7000 + (done + o) // 100_000)[0] for o in range(0, m, 100_000)])
        r = b.reshape(m, 150)
        f.write(b"".join(b"@r%d\n" % (done + i) + r[i].tobytes() + b"\n+\n" + b"I" * 150 + b"\n" for i in range(m)))
        done += m
exe = os.path.join(root, "krepp_amd", "lib", "krepp")
cfgs = ((["--tabular"], {"KR_CLI_WORKERS_PER_GPU": "2"}), (["--tabular"], {"KR_CLI_WORKERS_PER_GPU": "3"}), (["--tabular"], {"KR_CLI_WORKERS_PER_GPU": "2", "KR_CLI_BATCH_READS": "262144"}),
        (["--tabular"], {"KR_CLI_WORKERS_PER_GPU": "3", "KR_CLI_BATCH_READS": "262144"}), (["--summarize"], {"KR_CLI_WORKERS_PER_GPU": "2"}), (["--summarize"], {"KR_CLI_WORKERS_PER_GPU": "3"}),
        ([], {"KR_CLI_WORKERS_PER_GPU": "2"}), ([], {"KR_CLI_WORKERS_PER_GPU": "3"}))
for extra, env in cfgs:
    t = time.time()
    r = subprocess.run([exe, "place", "-i", idx, "-q", fq, "-o", work + "/out.txt"] + extra, capture_output=True, text=True, env=dict(os.environ, KR_CLI_TIMING="1", **env))
    dt = time.time() - t
    print("place", " ".join(extra) or "(jplace)", env, "rc", r.returncode, "%.2f s whole process" % dt,
          [l for l in r.stderr.strip().splitlines() if "timing" in l or "elapsed" in l], "output MB %.1f" % (os.path.getsize(work + "/out.txt") / 1e6), flush=True)
