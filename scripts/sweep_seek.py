#!/usr/bin/env python3
"""Parity sweep for `sketch` + `seek`: sketch parameters x hdist_th x read lengths; rows against the oracle's direct
restatement of src/seek.cpp."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from krepp_amd import capi, synth
import pyoracle as po
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(SEED)
work = tempfile.mkdtemp(prefix="krepp_ssweep_")
ACGT = np.frombuffer(b"ACGT", np.uint8)
bad = n_cases = 0
for ci, (k, h) in enumerate(((19, 4), (21, 7), (24, 8), (26, 10), (28, 12), (31, 15))):
    for m, r, frac in ((1, 0, True), (2, 1, False), (4, 1, True), (5, 3, True)):
        contigs = [rng.choice(ACGT, int(n)).tobytes() for n in (30000, 50, 12000)]
        fa = os.path.join(work, f"g{ci}_{m}.fa")
        with open(fa, "wb") as f:
            for i, c in enumerate(contigs):
                f.write(b">c%d\n" % i + c + b"\n")
        sk = os.path.join(work, f"s{ci}_{m}.skc")
        w = k + int(rng.integers(0, 9))
        capi.build_sketch(fa, sk, k=k, w=w, h=h, m=m, r=r, frac=frac, seed=SEED + ci)
        osk = po.Sketch(sk)
        hx = capi.HostIndex(sk, sketch=True); dx = hx.upload(0)
        for th in (4, 0, 2, 7):
            L = int(rng.integers(40, 500))
            reads, names = [], []
            g = np.frombuffer(contigs[0], np.uint8)
            for i in range(200):
                p = int(rng.integers(0, len(g) - L))
                s = g[p:p + L].copy()
                mut = rng.random(L) < (0.0, 0.02, 0.06, 0.15, 0.4)[i % 5]
                s[mut] = rng.choice(ACGT, int(mut.sum()))
                if i % 2:
                    s = synth.COMP[s[::-1]]
                if i % 13 == 0:
                    s[L // 3] = ord("N")
                reads.append(s.tobytes()); names.append(f"q{i}")
            bases = np.frombuffer(b"".join(reads), np.uint8)
            offs = np.cumsum([0] + [len(x) for x in reads]).astype(np.uint64)
            want = osk.seek(bases, offs, names, hdist_th=th)["text"]
            st = dx.stream(params=capi.default_params(hdist_th=th), max_reads=200, max_bases=len(bases), max_records=200 * 4)
            st.submit(bases, offs); st.collect()
            got = st.format_seek(hx, dx, names, hdist_th=th)
            st.close()
            n_cases += 1
            if got != want:
                bad += 1
                gl, wl = got.splitlines(), want.splitlines()
                print("MISMATCH", (k, w, h, m, r, frac), "th", th, "L", L, [(a, b) for a, b in zip(gl, wl) if a != b][:3])
        dx.close(); osk.close()
        os.remove(sk); os.remove(fa)
print("seek sweep finished:", n_cases, "cases, mismatching:", bad)
