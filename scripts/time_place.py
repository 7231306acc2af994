#!/usr/bin/env python3
"""Throughput of `place` through the C ABI (device front end + host tree aggregation + GPU likelihoods) on the toy25 index."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from krepp_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
work = tempfile.mkdtemp(prefix="krepp_pl_")
nwk = os.path.join(root, "tests", "golden", "tree_toy.nwk")
g = synth.evolve_genomes(open(nwk).read(), 400_000, seed=7)
tsv = synth.write_genomes(g, os.path.join(work, "g"))
idx = os.path.join(work, "idx")
capi.build_index(tsv, idx, nwk=nwk, k=27, w=35, h=11, m=4, r=1, frac=True, num_threads=8)
b, o, names = synth.sample_reads(g, n, seed=5)
names = names if names is not None else ["r%d" % i for i in range(n)]
hx = capi.HostIndex(idx)
for tab in (0, 1, 2):
    pl = capi.Placer(hx, None, 0, tabular=tab, max_reads=n, max_bases=len(b))
    pl.place(b, o, names)
    import ctypes as C
    t = time.time()
    pl.st.submit(np.ascontiguousarray(b), np.ascontiguousarray(o), capi.KR_TAP_ACCS)
    rv = capi.KrResultView()
    capi.check(pl.lib.kr_batch_collect(pl.st.h, C.byref(rv)))
    t1 = time.time()
    arr = (C.c_char_p * len(names))(*[x.encode() for x in names])
    t2 = time.time()
    txt, ln, pls, npl = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64()
    capi.check(pl.lib.kr_place_batch(hx.h, pl.dx.h, pl.pt, C.byref(rv), o.ctypes.data, arr, C.byref(pl.popts), tab, C.byref(pl.prev),
                                     C.byref(txt), C.byref(ln), C.byref(pls), C.byref(npl)))
    t3 = time.time()
    print(f"  submit+collect {t1 - t:.3f} s, names {t2 - t1:.3f} s, kr_place_batch {t3 - t2:.3f} s")
    arr = (C.c_char_p * len(names))(*[x.encode() for x in names])
    for host in (True, False):
        t = time.time()
        text, p = pl.place(b, o, names, host=host, c_names=arr, want_placements=(tab == 2))
        dt = time.time() - t
        print(f"  {'host' if host else 'device'} back end, submit to text: {n} reads in {dt:.3f} s = {n / dt / 1e6:.2f} M reads/s")
    tm = pl.st.timing()
    print(f"mode {tab}: {n} reads in {dt:.3f} s = {n / dt / 1e6:.2f} M reads/s; device front end {tm.ms_total:.1f} ms; placements {len(p)}; text MB {len(text) / 1e6:.1f}")
    pl.close()
