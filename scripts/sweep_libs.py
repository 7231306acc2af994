#!/usr/bin/env python3
"""Parity sweep over index directories holding several partial libraries (the reference's -m/-r sharding)."""
import os, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from krepp_amd import capi, synth
import pyoracle as po
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(SEED)
nwk = "((a:0.02,b:0.02):0.02,(c:0.03,(d:0.01,e:0.01):0.02):0.01,(f:0.05,g:0.002):0.01);"
g = synth.evolve_genomes(nwk, 25000, seed=SEED)
work = tempfile.mkdtemp(prefix="krepp_lsweep_")
tsv = synth.write_genomes(g, os.path.join(work, "g"))
open(os.path.join(work, "t.nwk"), "w").write(nwk)
bad = n = 0
for k, w, h in ((21, 25, 7), (27, 35, 11), (29, 31, 13)):
    ppos = sorted(rng.choice(k, h, replace=False).tolist(), reverse=True)  # the libraries of one index share the LSH
    for libs in ([(4, 0, False), (4, 1, False), (4, 2, False), (4, 3, False)], [(4, 1, True), (4, 2, False)], [(3, 0, True), (3, 2, False)],
                 [(2, 0, False), (2, 1, False)], [(5, 2, True), (5, 4, False)], [(4, 3, False)]):
        idx = os.path.join(work, f"ix_{k}_{len(libs)}_{libs[0][0]}")
        shutil.rmtree(idx, ignore_errors=True)
        for m, r, frac in libs:
            capi.build_index(tsv, idx, nwk=os.path.join(work, "t.nwk"), k=k, w=w, h=h, m=m, r=r, frac=frac, num_threads=8, ppos=ppos)
        hx = capi.HostIndex(idx); dx = hx.upload(0); ox = po.Index(idx)
        assert hx.view.nlibs == len(libs)
        for th, L in ((4, 150), (2, 90), (7, 300)):
            bases, offs, _ = synth.sample_reads(g, 300, seed=int(rng.integers(1 << 30)), length=L)
            rn = [f"r{i}" for i in range(300)]
            ref = ox.dist(bases, offs, rn, po.params(hdist_th=th, collect=4, num_threads=8))
            st = dx.stream(params=capi.default_params(hdist_th=th), max_reads=300, max_bases=len(bases), max_records=300 * 64)
            st.submit(bases, offs); st.collect()
            n += 1
            if st.format_dist(hx, rn) != ref["text"]:
                bad += 1
                print("MISMATCH", (k, w, h), libs, "th", th, "L", L)
            st.close()
        dx.close()
        shutil.rmtree(idx, ignore_errors=True)
print("library sweep finished:", n, "cases, mismatching:", bad)
