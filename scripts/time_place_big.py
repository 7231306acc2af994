#!/usr/bin/env python3
"""`place` through the C ABI on the 1000-genome index (its own Yule tree as backbone): reads per second of kr_place_stream
(device back end: both launches of kr_place_kernel + kr_place_llh_kernel, last phase on the host), batches that fell back to
the host back end, reads that took the second (global-scratch) launch.  usage: time_place_big.py [reads per batch]"""
import ctypes as C, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from krepp_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
work = tempfile.mkdtemp(prefix="krepp_plb_")
nwk_text = synth.yule_newick(1000, 2)
genomes = synth.evolve_genomes(nwk_text, 100_000, seed=2)
open(work + "/y.nwk", "w").write(nwk_text)
tsv = synth.write_genomes(genomes, work + "/g")
idx = work + "/idx"
capi.build_index(tsv, idx, nwk=work + "/y.nwk", k=29, w=35, h=13, m=4, r=1, frac=True, num_threads=min(32, os.cpu_count() or 1))
chunks = [synth.sample_reads(genomes, min(100_000, n - o), seed=900 + c)[0] for c, o in enumerate(range(0, n, 100_000))]
b = np.concatenate(chunks)
o = np.arange(n + 1, dtype=np.uint64) * np.uint64(150)
names = [f"q{i}" for i in range(n)]
arr = (C.c_char_p * n)(*[x.encode() for x in names])
hx = capi.HostIndex(idx)
only = os.environ.get("KR_TIME_PLACE_ONLY")  # e.g. "jplace": that mode's single calls only (with KR_PLACE_TIMING=1: the phases of a call)
for tab, label in ((1, "--tabular"), (2, "--summarize"), (0, "jplace")):
    if only and label.lstrip("-") != only:
        continue
    pl = capi.Placer(hx, None, 0, tabular=tab, max_reads=n, max_bases=len(b))
    pl.place(b, o, names, c_names=arr, want_placements=(tab == 2))  # warm-up: workspaces
    d0, h0 = capi.place_counters()
    hv0 = capi.place_heavy_reads()
    best = best_py = 1e9
    for _ in range(3):
        pl.prev = C.c_int(0)
        t = time.time()
        text, p = pl.place(b, o, names, c_names=arr, want_placements=(tab == 2))
        best_py = min(best_py, time.time() - t)
    # Round 6: the LIBRARY's time -- the call above also copies the text into a Python string and decodes it (70 MB of jplace
    # text: longer than the library takes to make it) and, for --summarize, copies the placements into a numpy array; until round
    # 6 that was in the figure quoted as `a call`
    for _ in range(5):
        pl.prev = C.c_int(0)
        t = time.time()
        tl, npl_ = pl.place(b, o, names, c_names=arr, want_placements=(tab == 2), keep_text=False)
        best = min(best, time.time() - t)
    d1, h1 = capi.place_counters()
    print(f"{label}: {n} reads, submit to text {best * 1e3:.1f} ms = {n / best / 1e6:.2f} M reads/s (library: kr_batch_submit + kr_place_stream; "
          f"{best_py * 1e3:.1f} ms = {n / best_py / 1e6:.2f} M reads/s with the text copied into a Python string as rounds 3-5 timed it); "
          f"device batches {d1 - d0}, host fallbacks {h1 - h0}, heavy reads (upper bound) {capi.place_heavy_reads() - hv0}; text {len(text) / 1e6:.1f} MB")
    pl.close()

# NT host threads with a stream each (as the CLI's workers run; ctypes releases the GIL): one thread's last phase on the host while
# another's batch is on the device -- the throughput of kr_place_stream as a pipeline
import threading
for NT in (() if only else (2, 3, 4)):
  for tab, label in ((0, "jplace"), (1, "--tabular"), (2, "--summarize")):
    pl = capi.Placer(hx, None, 0, tabular=tab, max_reads=n, max_bases=len(b))
    sts = [pl.st] + [pl.dx.stream(params=pl.st.params, max_reads=n, max_bases=len(b), max_records=n * 128) for _ in range(NT - 1)]
    ob = (len(b) + 7) & ~7
    pin = pl.lib.kr_host_alloc(ob + 8 * len(o))
    C.memmove(pin, np.ascontiguousarray(b).ctypes.data, len(b))
    C.memmove(pin + ob, np.ascontiguousarray(o).ctypes.data, 8 * len(o))
    bb = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=(len(b),))
    oo = np.ctypeslib.as_array(C.cast(pin + ob, C.POINTER(C.c_uint64)), shape=(len(o),))
    FL = capi.KR_TAP_ACCS | capi.KR_BASES_PINNED
    prevs = [C.c_int(0) for _ in range(NT)]
    def one(w):
        sts[w].submit(bb, oo, FL)
        txt, ln, pls, npl = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64()
        capi.check(pl.lib.kr_place_stream(hx.h, pl.dx.h, pl.pt, sts[w].h, n, oo.ctypes.data, arr, C.byref(pl.popts), tab, C.byref(prevs[w]),
                                          C.byref(txt), C.byref(ln), C.byref(pls) if tab == 2 else None, C.byref(npl) if tab == 2 else None))
        pl.lib.kr_free(txt), pl.lib.kr_free(pls)
    for w in range(NT):
        one(w)  # warm-up: workspaces
    per = 6
    def worker(w):
        for _ in range(per):
            one(w)
    ths = [threading.Thread(target=worker, args=(w,)) for w in range(NT)]
    t = time.time()
    for th in ths: th.start()
    for th in ths: th.join()
    dtp = time.time() - t
    print(f"{label}: {NT} host threads with a stream each, C ABI only (submit + kr_place_stream): {NT * per} x {n} reads in {dtp:.3f} s = {NT * per * n / dtp / 1e6:.2f} M reads/s")
    for st_ in sts[1:]:
        st_.close()
    pl.lib.kr_host_free(pin)
    pl.close()
