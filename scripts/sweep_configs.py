#!/usr/bin/env python3
"""Parity sweep over index / query configurations the fixed tests do not enumerate: k, w, h, m, r, frac, hdist_th,
read length.  For each: build a small index (CPU builder), run `dist` on the GPU and in the oracle, compare rows."""
import os, sys, tempfile, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from krepp_amd import capi, synth
import pyoracle as po
nwk = "((a:0.02,b:0.02):0.02,(c:0.03,(d:0.01,e:0.01):0.02):0.01,(f:0.05,g:0.002):0.01);"
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 9
g = synth.evolve_genomes(nwk, 30000, seed=SEED)
work = tempfile.mkdtemp(prefix="krepp_sweep_")
tsv = synth.write_genomes(g, os.path.join(work, "g"))
open(os.path.join(work, "t.nwk"), "w").write(nwk)
cfgs = []
for k, h in ((19, 3), (21, 7), (24, 8), (26, 10), (29, 13), (31, 15), (31, 16) if False else (30, 14)):
    for m, r, frac in ((1, 0, True), (2, 1, False), (3, 1, True), (4, 3, True), (4, 0, False), (7, 2, True)):
        cfgs.append((k, k + ((k * 7 + m) % 9), h, m, r, frac))
rng = np.random.default_rng(SEED)
bad = 0
for ci, (k, w, h, m, r, frac) in enumerate(cfgs):
    idx = os.path.join(work, f"ix{ci}")
    try:
        capi.build_index(tsv, idx, nwk=os.path.join(work, "t.nwk"), k=k, w=w, h=h, m=m, r=r, frac=frac, num_threads=8, seed=ci + 1)
    except capi.KrError as e:
        print("cfg", (k, w, h, m, r, frac), "build refused:", e)
        continue
    hx = capi.HostIndex(idx); dx = hx.upload(0); ox = po.Index(idx)
    for th, L in ((4, 150), (1, 100), (6, 151), (9, 260), (0, 90), (4, int(rng.integers(29, 900))), (int(rng.integers(0, 12)), int(rng.integers(60, 400)))):
        if th > 16 or (k - h) < 1:
            continue
        bases, offs, rn = synth.sample_reads(g, 300, seed=int(rng.integers(1 << 30)), length=L)
        rn = rn if rn is not None else [f'r{i}' for i in range(300)]
        ref = ox.dist(bases, offs, rn, po.params(hdist_th=th, collect=4, num_threads=8))
        st = dx.stream(params=capi.default_params(hdist_th=th), max_reads=300, max_bases=len(bases), max_records=300 * 64)
        try:
            st.submit(bases, offs)
            res = st.collect()
        except capi.KrError as e:  # a degenerate LSH (h = 3: 64 rows): far more than 256 hits per read; fewer reads per batch
            st.close()
            nr = 20
            bases, offs, rn = bases[:int(offs[nr])], offs[:nr + 1], rn[:nr] if rn is not None else None
            ref = ox.dist(bases, offs, rn, po.params(hdist_th=th, collect=4, num_threads=8))
            st = dx.stream(params=capi.default_params(hdist_th=th), max_reads=300, max_bases=len(bases) + 1, max_records=300 * 64)
            try:
                st.submit(bases, offs)
                res = st.collect()
            except capi.KrError as e2:
                print("cfg", (k, w, h, m, r, frac), "th", th, "L", L, "capacity even for 20 reads:", str(e2)[:90])
                st.close()
                continue
        text = st.format_dist(hx, rn)
        ok = text == ref["text"]
        if not ok:
            bad += 1
            gl, rl = text.splitlines(), ref["text"].splitlines()
            nd = sum(a != b for a, b in zip(gl, rl)) + abs(len(gl) - len(rl))
            print("MISMATCH cfg", (k, w, h, m, r, frac), "th", th, "L", L, "rows", len(gl), len(rl), "differing", nd)
        st.close()
    print("cfg", (k, w, h, m, r, frac), "done", flush=True)
    dx.close()
    import shutil
    shutil.rmtree(idx, ignore_errors=True)  # h = 14, 15: the bucket-offset file alone is GBs
print("sweep finished, mismatching cases:", bad)
