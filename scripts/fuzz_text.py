#!/usr/bin/env python3
"""Fuzz the device-side report text (kr_dev_text.inc) against the host formatter (kr_format_dist): random report modes
(--multi / --no-multi, --filter, --dist-max, --hdist-th), read ids of length 0..200 with any bytes but NUL, batches of 1..5,000
reads with reads that keep no reference, several batches per stream, text buffers sized exactly / one byte short.
usage: scripts/fuzz_text.py [seeds, default 8]   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from krepp_amd import capi, synth

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
idx = os.path.join(root, "tests", "golden", "toy_index")
hx = capi.HostIndex(idx)
dx = hx.upload(0)
g = synth.evolve_genomes(open(os.path.join(root, "tests", "golden", "tree_toy.nwk")).read(), 20000, seed=7)
bad = 0
ncase = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    rng = np.random.default_rng(1000 + seed)
    pk = {}
    if rng.random() < 0.4: pk["multi"] = 0
    if rng.random() < 0.4: pk["no_filter"] = 0
    if rng.random() < 0.3: pk["dist_max"] = float(rng.choice([0.01, 0.05, 0.2, 0.33]))
    if rng.random() < 0.3: pk["hdist_th"] = int(rng.integers(1, 7))
    nmax = 5000
    st = dx.stream(capi.default_params(**pk), max_reads=nmax, max_bases=nmax * 400)
    sh = dx.stream(capi.default_params(**pk), max_reads=nmax, max_bases=nmax * 400)
    st.text_enable(hx, 8 << 20, 2 << 20)
    for rep in range(6):
        n = int(rng.integers(1, nmax))
        bases, offs, _ = synth.sample_reads(g, n, seed=seed * 100 + rep, length=int(rng.choice([60, 100, 150, 151])))
        names = []
        for i in range(n):
            L = int(rng.integers(0, 12)) if rng.random() < 0.9 else int(rng.integers(0, 200))
            names.append(bytes(rng.integers(1, 256, L).astype(np.uint8)).decode("latin-1"))
        # capi encodes names as UTF-8: keep them latin-1-clean by restricting to ASCII for the host formatter's char* round trip
        names = ["".join(ch if 33 <= ord(ch) < 127 else "_" for ch in nm) for nm in names]
        st.submit_text(bases, offs, names)
        got = st.collect_text()
        sh.submit(bases, offs, capi.KR_ROWS_ONLY)
        sh.collect()
        want = sh.format_dist(hx, names).encode()
        ncase += 1
        if got != want:
            bad += 1
            print("MISMATCH seed", seed, "rep", rep, pk, len(got), len(want))
    # a buffer one byte short of the last batch's text: KR_ERR_CAPACITY; exactly its size: fits
    for cap, ok in ((len(want) - 1, False), (len(want), True)):
        if cap <= 0:
            continue
        s2 = dx.stream(capi.default_params(**pk), max_reads=nmax, max_bases=nmax * 400)
        s2.text_enable(hx, cap, 2 << 20)
        s2.submit_text(bases, offs, names)
        try:
            t = s2.collect_text()
            if not ok or t != want:
                bad += 1
                print("CAPACITY case wrong: fitted in", cap, "of", len(want))
        except capi.KrError as e:
            if ok or e.code != capi.KR_ERR_CAPACITY:
                bad += 1
                print("CAPACITY case wrong:", e)
        s2.close()
    st.close(); sh.close()
print(f"{ncase} batches, {bad} mismatches")
sys.exit(1 if bad else 0)
