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
for tab, label in ((1, "--tabular"), (2, "--summarize"), (0, "jplace")):
    pl = capi.Placer(hx, None, 0, tabular=tab, max_reads=n, max_bases=len(b))
    pl.place(b, o, names, c_names=arr, want_placements=(tab == 2))  # warm-up: workspaces
    d0, h0 = capi.place_counters()
    hv0 = capi.place_heavy_reads()
    best = 1e9
    for _ in range(3):
        pl.prev = C.c_int(0)
        t = time.time()
        text, p = pl.place(b, o, names, c_names=arr, want_placements=(tab == 2))
        best = min(best, time.time() - t)
    d1, h1 = capi.place_counters()
    print(f"{label}: {n} reads, submit to text {best * 1e3:.1f} ms = {n / best / 1e6:.2f} M reads/s; device batches {d1 - d0}, host fallbacks {h1 - h0}, "
          f"heavy reads (upper bound, 3 runs) {capi.place_heavy_reads() - hv0}; text {len(text) / 1e6:.1f} MB")
    pl.close()
