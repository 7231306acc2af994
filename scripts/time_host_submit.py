#!/usr/bin/env python3
"""PCIe-inclusive rate of the C ABI: kr_batch_submit with HOST buffers (staging copy + H2D + kernels + wait), toy25 index."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from krepp_amd import capi, synth
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
work = tempfile.mkdtemp(prefix="krepp_h2d_")
nwk = os.path.join(root, "tests", "golden", "tree_toy.nwk")
g = synth.evolve_genomes(open(nwk).read(), 400_000, seed=7)
tsv = synth.write_genomes(g, os.path.join(work, "g"))
idx = os.path.join(work, "idx")
capi.build_index(tsv, idx, nwk=nwk, k=27, w=35, h=11, m=4, r=1, frac=True, num_threads=8)
dx = capi.HostIndex(idx).upload(0)
bases, offsets = bench.make_reads(g, n, seed=1)
st = dx.stream(max_reads=n, max_bases=len(bases), max_records=n * 16)
for i in range(4):
    t = time.perf_counter()
    st.submit(bases, offsets)
    st.wait()
    dt = time.perf_counter() - t
    tm = st.timing()
    print(f"step {i}: {dt*1e3:.1f} ms wall = {n/dt/1e6:.1f} M reads/s; h2d {tm.ms_h2d:.2f} ms, kernels {tm.ms_total:.2f} ms")
