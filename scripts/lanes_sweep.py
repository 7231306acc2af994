#!/usr/bin/env python3
"""Experiment: ms per 1 M-read step on the 10 GB index against the number of lanes and the share of the chip the
persistent scan / accumulate kernels take (stream-creation knobs KR_LANES, KR_DEBUG_SCAN_BLOCKS_PER_CU,
KR_DEBUG_ACC_WAVES).  usage: lanes_sweep.py [reads_per_step]"""
import itertools
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from krepp_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
work = tempfile.mkdtemp(prefix="krepp_sweep_")
nwk_text = synth.yule_newick(1000, 2)
genomes = synth.evolve_genomes(nwk_text, 100_000, seed=2)
open(os.path.join(work, "y.nwk"), "w").write(nwk_text)
tsv = synth.write_genomes(genomes, os.path.join(work, "g"))
idx = os.path.join(work, "idx")
capi.build_index(tsv, idx, nwk=os.path.join(work, "y.nwk"), k=29, w=35, h=13, m=4, r=1, frac=True, num_threads=32)
hx = capi.HostIndex(idx)
dx, _ = synth.inflate_and_upload(torch, capi, hx, dev, 0, 10.0)
batches = []
for j in range(3):
    chunks = [synth.sample_reads(genomes, 100_000, seed=(1 + 1000 * j) * 1000 + c)[0] for c in range(n // 100_000)]
    b = np.concatenate(chunks)
    o = np.arange(n + 1, dtype=np.uint64) * np.uint64(150)
    batches.append((torch.from_numpy(b).to(dev), torch.from_numpy(o.view(np.int64)).to(dev)))


def run(env, steps=8):
    for k, v in env.items():
        os.environ[k] = str(v)
    st = dx.stream(max_reads=n, max_bases=n * 150, max_records=n * 64)
    for k in env:
        os.environ.pop(k)
    for i in range(2):
        st.submit_device(batches[i % 3][0].data_ptr(), batches[i % 3][1].data_ptr(), n)
        st.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        st.submit_device(batches[i % 3][0].data_ptr(), batches[i % 3][1].data_ptr(), n)
        st.wait()
    dt = (time.perf_counter() - t0) / steps * 1e3
    tm = st.timing()
    st.close()
    return dt, tm


configs = [dict(KR_LANES=1)]
for lanes in (2, 4, 8):
    configs.append(dict(KR_LANES=lanes))
for lanes, sb, aw in itertools.product((2, 4), (3, 2), (12, 8)):
    configs.append(dict(KR_LANES=lanes, KR_DEBUG_SCAN_BLOCKS_PER_CU=sb, KR_DEBUG_ACC_WAVES=aw))
configs.append(dict(KR_LANES=4, KR_DEBUG_SCAN_BLOCKS_PER_CU=3))
configs.append(dict(KR_LANES=4, KR_DEBUG_ACC_WAVES=12))
configs.append(dict(KR_LANES=4, KR_LANE_MIN_READS=32768))
for env in configs:
    dt, tm = run(env)
    print(json.dumps({"env": env, "ms_per_step": round(dt, 3), "lanes": tm.lanes, "sum_scan": round(tm.ms_scan, 2), "sum_acc": round(tm.ms_acc, 2),
                      "sum_llh": round(tm.ms_llh, 2), "span": round(tm.ms_total, 2)}), flush=True)
