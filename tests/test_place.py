"""`krepp place`: oracle restatement vs the product (GPU front end + GPU
likelihoods + host tree aggregation)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def placements_key(pl):
    return sorted((int(r), int(e)) for r, e in zip(pl["read"], pl["edge"]))


def test_oracle_place_basics(po, toy_index_dir, toy_reads):
    names, bases, offs = toy_reads
    ox = po.Index(toy_index_dir)
    ox.set_placement_tree(None)
    r = ox.place(bases, offs, names, po.params(no_filter=0))
    pl = r["placements"]
    assert len(pl) > 300
    # LWRs of a read sum to 1; edge numbers are valid; a single placement has LWR 1
    for rd in np.unique(pl["read"]):
        s = pl[pl["read"] == rd]
        assert abs(s["lwr"].sum() - 1.0) < 1e-9
        assert (s["edge"] < 46).all()  # the root (edge 46) is never a placement (src/query.cpp:277)
    # jplace framing is valid JSON
    import json
    doc = json.loads(ox.place_frame(0) + r["text"] + ox.place_frame(1, False, "krepp place", len(names)))
    assert doc["version"] == 3 and len(doc["placements"]) == len(np.unique(pl["read"]))
    assert doc["tree"].endswith("{46};") and "{37}" in doc["tree"]
    # --no-multi: one placement per placed read
    r1 = ox.place(bases, offs, names, po.params(no_filter=0, multi=0))
    assert len(r1["placements"]) == len(np.unique(r1["placements"]["read"]))
    # tabular rows
    rt = ox.place(bases, offs, names, po.params(no_filter=0), tabular=True)
    assert rt["text"].count("\n") == len(pl) and ox.place_frame(0, True, "x").endswith("SEQ_ID\tDISTAL_NODE\tEDGE_NUM\tLWR\tDIST\n")


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [dict(), dict(multi=0), dict(no_filter=1), dict(tau=1, chisq=3.841), dict(hdist_th=3)])
def test_place_matches_oracle(capi, po, toy_index_dir, toy_reads, opts):
    names, bases, offs = toy_reads
    hx = capi.HostIndex(toy_index_dir)
    ox = po.Index(toy_index_dir)
    ox.set_placement_tree(None)
    okw = dict(no_filter=0)
    okw.update(opts)
    for tabular in (False, True):
        want = ox.place(bases, offs, names, po.params(**okw), tabular=tabular)
        pk = dict(opts)
        pk.pop("no_filter", None)
        placer = capi.Placer(hx, None, 0, tabular=tabular, max_reads=len(names), max_bases=len(bases), **pk)
        if opts.get("no_filter"):
            placer.popts.no_filter = 1
        text, pl = placer.place(bases, offs, names)
        assert placements_key(pl) == placements_key(want["placements"]), (opts, tabular)
        a = np.sort(pl, order=["read", "edge"])
        b = np.sort(want["placements"], order=["read", "edge"])
        for f in ("lwr", "d_llh", "pendant", "distal"):
            assert np.allclose(a[f], b[f], rtol=1e-6, atol=1e-9), f
        assert np.allclose(a["v_llh"], b["v_llh"], rtol=1e-9)
        assert text == want["text"], (opts, tabular)
        assert placer.frame(0, "inv") == ox.place_frame(0, tabular, "inv")
        assert placer.frame(1, "inv", len(names)) == ox.place_frame(1, tabular, "inv", len(names))
        placer.close()


@pytest.mark.gpu
def test_place_on_user_tree(capi, po, toy_index_dir, toy_reads):
    """-t: index leaves mapped onto another rooted tree (some leaves absent, one extra leaf)."""
    names, bases, offs = toy_reads
    nwk = open(os.path.join(GOLDEN, "tree_toy.nwk")).read()
    # drop one leaf (its sibling keeps the clade), add a leaf the index does not know
    q = nwk.replace("(G000735195:0.0276038,G000018865:0.0228997)N2640:0.160977", "(G000735195:0.03,NEWLEAF:0.02)N2640:0.160977")
    assert q != nwk
    hx = capi.HostIndex(toy_index_dir)
    ox = po.Index(toy_index_dir)
    ox.set_placement_tree(q)
    want = ox.place(bases, offs, names, po.params(no_filter=0))
    placer = capi.Placer(hx, q, 0, max_reads=len(names), max_bases=len(bases))
    text, pl = placer.place(bases, offs, names)
    assert placements_key(pl) == placements_key(want["placements"]) and len(pl) > 200
    assert text == want["text"]
    assert placer.frame(1, "i", 3) == ox.place_frame(1, False, "i", 3)


@pytest.mark.gpu
def test_cli_place_end_to_end(po, toy_index_dir, toy_reads, tmp_path):
    import json
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "krepp_amd", "lib", "krepp")
    fq = os.path.join(GOLDEN, "toy_reads.fq")
    names, bases, offs = toy_reads
    ox = po.Index(toy_index_dir)
    ox.set_placement_tree(None)
    r = subprocess.run([exe, "place", "-i", toy_index_dir, "-q", fq], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = ox.place(bases, offs, names, po.params(no_filter=0))
    inv = f"{exe} place -i {toy_index_dir} -q {fq}"
    assert r.stdout == ox.place_frame(0) + want["text"] + ox.place_frame(1, False, inv, len(names))
    doc = json.loads(r.stdout)
    assert doc["metadata"]["num_queries"] == "308" and len(doc["placements"]) > 250
    r = subprocess.run([exe, "place", "-i", toy_index_dir, "-q", fq, "--tabular", "--no-multi", "--tau", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    wt = ox.place(bases, offs, names, po.params(no_filter=0, multi=0, tau=1), tabular=True)
    inv = f"{exe} place -i {toy_index_dir} -q {fq} --tabular --no-multi --tau 1"
    assert r.stdout == ox.place_frame(0, True, inv) + wt["text"]
