#!/usr/bin/env python3
"""Parity sweep for `place`: random rooted trees (multifurcations, unlabelled and labelled internal nodes, missing branch
lengths), the backbone / a different -t tree / a lineage file as placement tree, option combinations; text and
placements against the oracle."""
import os, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from krepp_amd import capi, synth
import pyoracle as po

SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(SEED)


def random_tree(names, label_internal, with_lengths=True):
    items = [(n, 1) for n in names]
    k = 0
    while len(items) > 1:
        arity = 2 if rng.random() < 0.7 or len(items) < 3 else min(len(items), int(rng.integers(3, 5)))
        idx = sorted(rng.choice(len(items), arity, replace=False).tolist(), reverse=True)
        kids = [items.pop(i) for i in idx]
        def br(x):
            return f"{x}:{rng.uniform(0.001, 0.05):.5f}" if with_lengths and rng.random() < 0.95 else x
        lab = f"N{k}" if label_internal and rng.random() < 0.6 else ""
        k += 1
        items.append(("(" + ",".join(br(c[0]) for c in kids) + ")" + lab, sum(c[1] for c in kids)))
    return items[0][0] + ";"


bad = 0
work = tempfile.mkdtemp(prefix="krepp_psweep_")
for case in range(6):
    n = int(rng.integers(5, 30))
    names = [f"g{i:02d}" for i in range(n)]
    nwk = random_tree(names, label_internal=bool(case & 1))
    g = synth.evolve_genomes(nwk, 8000, seed=SEED * 100 + case)
    d = os.path.join(work, f"c{case}")
    os.makedirs(d)
    tsv = synth.write_genomes(g, os.path.join(d, "g"))
    open(os.path.join(d, "t.nwk"), "w").write(nwk)
    idx = os.path.join(d, "ix")
    capi.build_index(tsv, idx, nwk=os.path.join(d, "t.nwk"), k=25, w=31, h=9, m=3, r=1, frac=True, num_threads=8)
    hx = capi.HostIndex(idx)
    ox = po.Index(idx)
    bases, offs, _ = synth.sample_reads(g, 400, seed=case + 7, length=int(rng.integers(90, 320)))
    rn = [f"r{i}" for i in range(400)]
    # placement trees: backbone, another random tree over a subset (+ a foreign leaf), a random lineage file
    sub = [x for x in names if rng.random() < 0.8] + ["foreign"]
    qtree = random_tree(sub if len(sub) > 2 else names, label_internal=True, with_lengths=bool(case % 3))
    ranks = "kpcofgs"
    lin = ""
    for x in names:
        if rng.random() < 0.85:
            taxa = [f"{ranks[j]}__T{j}_{int(rng.integers(0, 2 + j))}" for j in range(int(rng.integers(1, 7)))]
            if rng.random() < 0.3:
                taxa.append("s__")
            lin += x + "\t" + "; ".join(taxa) + "\n"
    trees = [("backbone", None, None), ("-t", qtree, None)] + ([("-l", None, lin)] if lin else [])
    for tname, tq, tl in trees:
        try:
            if tl is not None:
                ox.set_lineage_tree(tl)
            else:
                ox.set_placement_tree(tq)
        except RuntimeError as e:
            print("case", case, tname, "oracle refuses:", str(e)[:80])
            continue
        for opts in (dict(), dict(multi=0), dict(no_filter=1), dict(tau=1, chisq=3.841), dict(multi=0, no_filter=1, tau=3)):
            okw = dict(no_filter=0); okw.update(opts)
            pk = dict(opts); nf = pk.pop("no_filter", None)
            for mode in (0, 1, 2):
                placer = capi.Placer(hx, tq, 0, tabular=mode, max_reads=400, max_bases=len(bases), lineage_text=tl, **pk)
                if nf:
                    placer.popts.no_filter = 1
                text, pl = placer.place(bases, offs, rn)
                if mode == 2:
                    ok = placer.summary() == ox.place_summarize(bases, offs, po.params(**okw))
                else:
                    want = ox.place(bases, offs, rn, po.params(**okw), tabular=bool(mode))
                    ok = text == want["text"] and placer.frame(1, "i", 400) == ox.place_frame(1, mode, "i", 400)
                placer.close()
                if not ok:
                    bad += 1
                    print("MISMATCH case", case, tname, opts, "mode", mode, "n", n)
    print("case", case, "leaves", n, "done", flush=True)
    shutil.rmtree(d, ignore_errors=True)
print("place sweep finished, mismatching cases:", bad)
