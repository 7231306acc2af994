#!/usr/bin/env python3
"""Fuzz the query alphabet: reads with lowercase, IUPAC codes, gaps, control and high bytes at random places; rows against the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from krepp_amd import capi, synth
import pyoracle as po
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
idx = os.path.join(root, "tests", "golden", "toy_index")
hx = capi.HostIndex(idx); dx = hx.upload(0); ox = po.Index(idx)
g = synth.evolve_genomes(open(os.path.join(root, "tests", "golden", "tree_toy.nwk")).read(), 20000, seed=7)
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    rng = np.random.default_rng(seed)
    gl = list(g.values())
    reads = []
    for i in range(1500):
        s = gl[int(rng.integers(len(gl)))]
        L = int(rng.integers(0, 400))
        p = int(rng.integers(0, len(s) - L))
        r = s[p:p + L].copy()
        mode = i % 6
        if L:
            if mode == 1:
                r = np.frombuffer(r.tobytes().lower(), np.uint8).copy()
            elif mode == 2:
                k = rng.random(L) < 0.03
                r[k] = rng.choice(np.frombuffer(b"NRYKMSWBDHVnryk-*.", np.uint8), int(k.sum()))
            elif mode == 3:
                k = rng.random(L) < 0.02
                r[k] = rng.integers(0, 256, int(k.sum())).astype(np.uint8)
            elif mode == 4:
                k = rng.random(L) < 0.5
                r[k] = np.frombuffer(r[k].tobytes().lower(), np.uint8)
            elif mode == 5:
                r = rng.integers(0, 256, L).astype(np.uint8)
        reads.append(r.tobytes())
    bases = np.frombuffer(b"".join(reads), np.uint8)
    offs = np.cumsum([0] + [len(x) for x in reads]).astype(np.uint64)
    rn = [f"r{i}" for i in range(len(reads))]
    ref = ox.dist(bases, offs, rn, po.params(collect=4, num_threads=8))
    st = dx.stream(max_reads=len(reads), max_bases=len(bases) + 1, max_records=len(reads) * 64)
    st.submit(bases, offs); st.collect()
    ok = st.format_dist(hx, rn) == ref["text"]
    bad += not ok
    print("seed", seed, "rows", ref["text"].count("\n"), "equal", ok, flush=True)
    st.close()
print("fuzz finished, mismatching batches:", bad)
