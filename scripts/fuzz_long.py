#!/usr/bin/env python3
"""Fuzz the long-sequence path (kr_dev_tiles.inc): batches that mix reads, sequences around the tiling threshold and contigs of up to
40 kb, with substitutions, N runs and chimeric stretches, on streams with room for all, some or none of the tiles; report text,
per-(leaf, strand) histograms and hdist_filt against the oracle.  usage: fuzz_long.py [seeds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from krepp_amd import capi, synth
import pyoracle as po
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
idx = os.path.join(root, "tests", "golden", "toy_index")
hx = capi.HostIndex(idx); dx = hx.upload(0); ox = po.Index(idx)
g = synth.evolve_genomes(open(os.path.join(root, "tests", "golden", "tree_toy.nwk")).read(), 20000, seed=7)
gl = list(g.values())
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    rng = np.random.default_rng(1000 + seed)
    seqs = []
    for i in range(int(rng.integers(3, 60))):
        kind = int(rng.integers(0, 5))
        if kind == 0:
            L = int(rng.integers(0, 300))
        elif kind == 1:
            L = int(rng.integers(1030, 1060))  # around 1,024 k-mer positions (k = 21)
        elif kind == 2:
            L = int(np.exp(rng.uniform(np.log(300), np.log(40000))))
        elif kind == 3:
            L = int(rng.integers(1, 40)) * 128 + int(rng.integers(18, 24))  # around multiples of the tile length
        else:
            L = int(rng.integers(2000, 20000))
        parts, left = [], L
        while left > 0:  # stretches of different references, either strand
            s = gl[int(rng.integers(len(gl)))]
            n = min(left, int(rng.integers(1, 20000)))
            p = int(rng.integers(0, len(s) - n + 1))
            c = s[p:p + n].copy()
            if rng.integers(0, 2):
                c = synth.COMP[c[::-1]]
            parts.append(c)
            left -= n
        r = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
        if L:
            k = rng.random(L) < rng.choice([0.0, 0.01, 0.05])
            r[k] = rng.choice(np.frombuffer(b"ACGTN", np.uint8), int(k.sum()))
            for _ in range(int(rng.integers(0, 3))):  # N runs, some across tile boundaries
                a = int(rng.integers(0, L)); b = min(L, a + int(rng.integers(1, 200)))
                r[a:b] = ord("N")
        seqs.append(r.tobytes())
    bases = np.frombuffer(b"".join(seqs), np.uint8)
    offs = np.cumsum([0] + [len(x) for x in seqs]).astype(np.uint64)
    rn = [f"s{i}" for i in range(len(seqs))]
    ref = ox.dist(bases, offs, rn, po.params(collect=7, num_threads=8))
    acc = ref["accs"][ref["accs"]["passed"] == 1]
    want = sorted(zip(acc["read"].tolist(), ((acc["se"] << 1) | acc["strand"]).tolist(), [tuple(x[:5]) for x in acc["hist"].tolist()]))
    tiles_all = len(seqs) + len(bases) // 128 + 8
    ok = True
    for max_reads in (tiles_all, len(seqs) + int(rng.integers(0, max(1, len(bases) // 128))), len(seqs)):
        st = dx.stream(max_reads=max_reads, max_bases=len(bases) + 64, max_records=1 << 20)
        st.submit(bases, offs, capi.KR_TAP_ACCS)
        res = st.collect()
        got = sorted(zip(res.rec_read.tolist(), res.rec_key.tolist(), [tuple(x) for x in res.rec_hist.tolist()]))
        ok = ok and got == want and st.format_dist(hx, rn) == ref["text"] and st.readtaps(len(rn)).tolist() == ref["reads"]["hdist_filt"].tolist()
        st.submit(bases, offs)  # without the tap: packed words for the reads, planes for the long sequences
        st.collect()
        ok = ok and st.format_dist(hx, rn) == ref["text"]
        st.close()
    bad += not ok
    print("seed", seed, "sequences", len(seqs), "bases", len(bases), "rows", ref["text"].count("\n"), "equal", ok, flush=True)
print("fuzz finished, mismatching batches:", bad)
