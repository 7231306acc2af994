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
    # two streams in turn (as the CLI's two workers do): the front end of batch i+1 runs on the GPU while the host writes batch i
    st2 = pl.dx.stream(params=pl.st.params, max_reads=n, max_bases=len(b), max_records=n * 128)
    sts = [pl.st, st2]
    ob = (len(b) + 7) & ~7  # reads and offsets in page-locked memory: no staging copy in submit
    pin = pl.lib.kr_host_alloc(ob + 8 * len(o))
    C.memmove(pin, np.ascontiguousarray(b).ctypes.data, len(b))
    C.memmove(pin + ob, np.ascontiguousarray(o).ctypes.data, 8 * len(o))
    bb = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=(len(b),))
    oo = np.ctypeslib.as_array(C.cast(pin + ob, C.POINTER(C.c_uint64)), shape=(len(o),))
    FL = capi.KR_TAP_ACCS | capi.KR_BASES_PINNED
    nb = 8
    def back_end(st):
        txt, ln, pls, npl = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64()
        capi.check(pl.lib.kr_place_stream(hx.h, pl.dx.h, pl.pt, st.h, n, oo.ctypes.data, arr, C.byref(pl.popts), tab, C.byref(pl.prev),
                                          C.byref(txt), C.byref(ln), C.byref(pls) if tab == 2 else None, C.byref(npl) if tab == 2 else None))
        pl.lib.kr_free(txt), pl.lib.kr_free(pls)
    for st in sts:  # warm-up: workspaces
        st.submit(bb, oo, FL)
        back_end(st)
    t = time.time()
    sts[0].submit(bb, oo, FL)
    for i in range(nb):
        if i + 1 < nb:
            sts[(i + 1) % 2].submit(bb, oo, FL)
        back_end(sts[i % 2])
    dtp = time.time() - t
    print(f"  two streams in turn, C ABI only (submit + kr_place_stream, text freed): {nb} x {n} reads in {dtp:.3f} s = {nb * n / dtp / 1e6:.2f} M reads/s")
    # two host threads, a stream each (ctypes releases the GIL): one thread's last phase runs while the other's batch is on the GPU
    import threading
    prevs = [C.c_int(0), C.c_int(0)]
    def worker(w):
        for _ in range(nb // 2):
            sts[w].submit(bb, oo, FL)
            txt, ln, pls, npl = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64()
            capi.check(pl.lib.kr_place_stream(hx.h, pl.dx.h, pl.pt, sts[w].h, n, oo.ctypes.data, arr, C.byref(pl.popts), tab, C.byref(prevs[w]),
                                              C.byref(txt), C.byref(ln), C.byref(pls) if tab == 2 else None, C.byref(npl) if tab == 2 else None))
            pl.lib.kr_free(txt), pl.lib.kr_free(pls)
    ths = [threading.Thread(target=worker, args=(w,)) for w in range(2)]
    t = time.time()
    for th in ths: th.start()
    for th in ths: th.join()
    dtp = time.time() - t
    print(f"  two host threads with a stream each, C ABI only: {nb} x {n} reads in {dtp:.3f} s = {nb * n / dtp / 1e6:.2f} M reads/s")
    st2.close()
    pl.lib.kr_host_free(pin)
    tm = pl.st.timing()
    print(f"mode {tab}: {n} reads in {dt:.3f} s = {n / dt / 1e6:.2f} M reads/s; device front end {tm.ms_total:.1f} ms; placements {len(p)}; text MB {len(text) / 1e6:.1f}")
    pl.close()
