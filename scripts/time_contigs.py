#!/usr/bin/env python3
"""Long query sequences: the same batch of contigs through the C ABI as tiles across waves (default) and with one wave per
sequence (KR_NO_TILES=1), next to 150-bp reads of the same total length.  Index: 25 references of 400 kb (k27 / w35 / h11).
usage: time_contigs.py [contig length] [contigs]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from krepp_amd import capi, synth
L = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 8
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
work = tempfile.mkdtemp(prefix="krepp_ctg_")
nwk = os.path.join(root, "tests", "golden", "tree_toy.nwk")
g = synth.evolve_genomes(open(nwk).read(), 400_000, seed=7)
tsv = synth.write_genomes(g, os.path.join(work, "g"))
idx = os.path.join(work, "idx")
capi.build_index(tsv, idx, nwk=nwk, k=27, w=35, h=11, m=4, r=1, frac=True, num_threads=8)
rng = np.random.default_rng(3)
gl = list(g.values())
seqs = []
for i in range(nc):  # contigs: stretches of the references with 1 % substitutions
    o = int(rng.integers(0, 400_000 - L + 1))
    s = gl[i % len(gl)][o:o + L].copy()
    mut = rng.random(len(s)) < 0.01
    s[mut] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(mut.sum()))]
    seqs.append(s)
bases = np.concatenate(seqs)
offs = np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint64)
hx = capi.HostIndex(idx)
dx = hx.upload(0)
nreads_eq = len(bases) // 150
rb, ro, _ = synth.sample_reads(g, min(nreads_eq, 200_000), seed=4)
def timed(b, o, label, max_reads):
    st = dx.stream(max_reads=max_reads, max_bases=len(b) + 64, max_records=max_reads * 64)
    best = 1e9
    for _ in range(3):
        t = time.time()
        st.submit(b, o, capi.KR_ROWS_ONLY)
        rv = st.collect_view()
        best = min(best, time.time() - t)
    tm = st.timing()
    print(f"{label}: {len(o) - 1} sequences, {len(b) / 1e6:.2f} Mb: {best * 1e3:.1f} ms = {len(b) / best / 1e6:.0f} Mb/s "
          f"(scan {tm.ms_scan:.2f} ms, accumulate {tm.ms_acc:.2f} ms, likelihood {tm.ms_llh:.2f} ms); rows {rv.nrows}")
    st.close()
    return rv.nrows
vmax = int(len(bases) // 128 + len(seqs) + 1024)
a = timed(bases, offs, "contigs, tiles across waves", vmax)
os.environ["KR_NO_TILES"] = "1"
b = timed(bases, offs, "contigs, one wave per sequence", vmax)
os.environ.pop("KR_NO_TILES")
assert a == b, (a, b)
timed(rb, ro, "150-bp reads", len(ro))
