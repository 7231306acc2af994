"""CPU tests of the oracle: golden vectors, independent closed forms, Brent's properties."""
import json
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN, rows_of_oracle
from helpers import closed_form, front_end_py, hd32, revcomp, row_of, write_index


def test_newick_postorder_numbering_matches_readme(po, toy_index_dir):
    # README.md:100-118 of the reference: edge numbers of test/tree_toy.nwk (edge = se - 1)
    ox = po.Index(toy_index_dir)
    by_name = {ox.name(se): se - 1 for se in range(1, ox.info.nnodes + 1)}
    assert ox.info.nnodes == 47 and ox.info.nleaves == 25
    assert by_name["G001610775"] == 37
    assert by_name["G000341695"] == 39
    assert by_name["N4337"] == 40
    assert by_name["N3634"] == 41
    # post-order: every parent has a larger se than its children, root is last
    for se in range(1, 47):
        assert ox.parent(se) > se
    assert ox.parent(47) == 0


def test_balanced_tree_from_reflist(po, tmp_path):
    # Node::generate_tree (src/phytree.cpp:217-253): second half first.
    # names a,b,c,d,e: halves [c,d,e] then [a,b]; [c,d,e] -> [d,e] then [c]; [d,e] -> [e],[d]
    names = list("abcde")
    d = str(tmp_path / "ix")
    write_index(d, 21, 7, 4, 1, True, [20, 19, 17, 13, 6, 4, 2], {}, [(0, 0)] * 10, [0.0] * 10, reflist=names)
    ox = po.Index(d)
    order = [ox.name(se) for se in range(1, ox.info.nnodes + 1)]
    assert order == ["e", "d", "2", "c", "4", "b", "a", "7", "8"]  # unlabelled nodes print se-1
    assert [ox.kind(se) for se in range(1, 10)] == [1, 1, 2, 1, 2, 1, 1, 2, 2]
    assert ox.info.wbackbone == 0


def test_front_end_closed_form(po, toy_index_dir, toy_reads):
    ox = po.Index(toy_index_dir)
    ppos, npos = ox.positions()
    assert list(ppos) == sorted(ppos, reverse=True) and list(npos) == sorted(npos)
    names, bases, offs = toy_reads
    checked = 0
    for r in list(range(0, 40)) + list(range(300, len(names))):
        seq = bytes(bases[int(offs[r]):int(offs[r + 1])]).decode()
        fe = ox.front_end(seq.encode())
        want = front_end_py(seq, 21, ppos, npos)
        assert len(fe["kpos"]) == len(want), names[r]
        for i, w in enumerate(want):
            assert (int(fe["kpos"][i]), int(fe["strand"][i]), int(fe["enc_bp"][i]), int(fe["enc_lr"][i]),
                    int(fe["rix"][i]), int(fe["enc32"][i])) == w
            assert bool(fe["pas"][i]) == (w[4] % 4 <= 1)  # m4 r1 frac: residues 0 and 1
            checked += 1
    assert checked > 5000


def test_revcomp_and_conversion(po):
    l = po.lib()
    rng = np.random.default_rng(1)
    for k in (19, 21, 27, 29, 31):
        for _ in range(50):
            s = "".join("ACGT"[i] for i in rng.integers(0, 4, k))
            bp, lr, _, _ = closed_form(s, [0, 1, 2], list(range(3, k)))
            rbp, rlr, _, _ = closed_form(revcomp(s), [0, 1, 2], list(range(3, k)))
            assert l.ko_revcomp_bp64(bp, k) == rbp          # src/common.hpp:177-186
            assert l.ko_conv_bp64_lr64(bp) == lr            # src/common.hpp:223
            assert l.ko_conv_bp64_lr64(rbp) == rlr
            assert l.ko_revcomp_bp64(rbp, k) == bp


def test_brent_properties(po):
    """Boost's minimiser is unpinned (absent submodule); check what any faithful Brent run
    must satisfy: it stops with the minimum bracketed within fract2 = 2*(2^-15*|x| + 2^-17) of x,
    so for these unimodal likelihoods f(x) is no worse than f three bracket-widths away."""
    rng = np.random.default_rng(3)
    for _ in range(300):
        k, h, th = 27, 11, 4
        hist = np.floor(rng.random(5) ** 2 * rng.integers(1, 60))
        if hist.sum() == 0:
            hist[0] = 1
        uc = float(rng.integers(0, 124))
        rho = float(rng.uniform(0.05, 0.6))
        d, v, ne = po.brent(k, h, th, hist, uc, rho)
        assert 1e-10 <= d <= 0.5 and 2 <= ne < 200
        assert v == po.llh(k, h, th, hist, uc, rho, d)
        fract2 = 2 * (2 ** -15 * d + 2 ** -17)
        for dn in (d - 3 * fract2, d + 3 * fract2):
            if not (1e-10 < dn < 0.5):
                continue  # the minimum sits on a bound: x stops within fract2 of it
            assert po.llh(k, h, th, hist, uc, rho, dn) >= v - 1e-12 * abs(v)


def test_perfect_match_distance_prints_00001(po):
    # an error-free read reports 0.00001 (termination at ~tol/4)
    d, v, ne = po.brent(27, 11, 4, [124, 0, 0, 0, 0], 0.0, 0.1)
    assert f"{d:.5f}" == "0.00001"


def test_oracle_matches_committed_expected(po, toy_index_dir, toy_reads):
    exp = json.load(open(os.path.join(GOLDEN, "toy_expected.json")))
    names, bases, offs = toy_reads
    ox = po.Index(toy_index_dir)
    for tag, p in [("default", po.params(collect=7)), ("filter", po.params(collect=5, no_filter=0)),
                   ("nomulti", po.params(collect=5, multi=0)), ("dmax", po.params(collect=5, dist_max=0.05)),
                   ("th2", po.params(collect=5, hdist_th=2))]:
        r = ox.dist(bases, offs, names, p)
        got = [(int(x["read"]), int(x["se"]), float(x["d_llh"]).hex()) for x in r["rows"]]
        assert got == [tuple(x) for x in exp[tag]["rows"]], tag
        assert r["text"] == exp[tag]["text"]
        assert r["counters"] == exp[tag]["counters"]
    assert "NA\tNaN" in exp["default"]["text"]


def test_oracle_threads_agree(po, toy_index_dir, toy_reads):
    names, bases, offs = toy_reads
    ox = po.Index(toy_index_dir)
    a = ox.dist(bases, offs, names, po.params(collect=5, num_threads=1))
    b = ox.dist(bases, offs, names, po.params(collect=5, num_threads=4))
    assert a["text"] == b["text"] and a["counters"] == b["counters"]


PPOS = [20, 19, 17, 13, 6, 4, 2]
NPOS = [p for p in range(21) if p not in PPOS]


def crafted(tmp_path, po):
    """Three leaves x,y,z under ((x,y)n1,z)root; se: x=1 y=2 n1=3 z=4 root=5; colours:
    6 = {x,z} = (1,4), 7 = {y} u {} = (2,0), 8 = (6,2) = {x,z,y}."""
    s = "ACGTTGCAAGGCTTAACGTCA"  # 21-mer
    kmers = {}
    def add(km, se):
        _, _, rix, enc = closed_form(km, PPOS, NPOS)
        row = row_of(rix, 4, 1, True)
        assert row is not None, "pick k-mers whose residue is served"
        kmers.setdefault(row, []).append((enc, se))
        return rix, enc
    return s, add, kmers


def test_per_position_min_and_colour_expansion(po, tmp_path):
    # Craft a probe with two table entries, hd 0 -> colour {x,z}, hd 1 -> colour {x,z,y}:
    # x and z must count hd 0 only (per-position minimum, src/query.hpp:153-176), y counts hd 1.
    rng = np.random.default_rng(7)
    for attempt in range(200):
        s = "".join("ACGT"[i] for i in rng.integers(0, 4, 21))
        f = closed_form(s, PPOS, NPOS)
        if row_of(f[2], 4, 1, True) is None:
            continue
        # mutate one non-LSH position -> same row, hd 1
        p = NPOS[3]
        i = 20 - p
        alt = s[:i] + ("A" if s[i] != "A" else "C") + s[i + 1:]
        g = closed_form(alt, PPOS, NPOS)
        assert g[2] == f[2] and hd32(g[3], f[3]) == 1
        rc = closed_form(revcomp(s), PPOS, NPOS)
        if row_of(rc[2], 4, 1, True) is not None:
            continue  # keep the reverse strand silent for a crisp expectation
        break
    row = row_of(f[2], 4, 1, True)
    rows = {row: [(f[3], 6), (g[3], 8)]}
    pse = [(0, 0), (0, 1), (0, 2), (1, 2), (0, 4), (3, 4), (1, 4), (2, 0), (6, 2)]
    rho = [0.0, 0.2, 0.2, 0.0, 0.2, 0.0]
    d = str(tmp_path / "ix")
    write_index(d, 21, 7, 4, 1, True, PPOS, rows, pse, rho, nwk="((x:0.1,y:0.1)n1:0.1,z:0.2);")
    ox = po.Index(d)
    seq = np.frombuffer(s.encode(), np.uint8)
    r = ox.dist(seq, np.array([0, 21], np.uint64), ["q"], po.params(collect=7))
    accs = {(int(a["se"]), int(a["strand"])): a["hist"][:5].tolist() for a in r["accs"]}
    assert accs == {(1, 0): [1, 0, 0, 0, 0], (4, 0): [1, 0, 0, 0, 0], (2, 0): [0, 1, 0, 0, 0]}
    assert len(r["hits"]) == 2 and r["reads"]["hdist_filt"][0].tolist() == [0, 0xFFFFFFFF]
    # hdist_filt = 0 -> limit 1: y (hdist_min 1) passes, all three get rows
    assert sorted(int(x) for x in r["rows"]["se"]) == [1, 2, 4]
    assert r["counters"]["pse_reads"] == 1 + 2  # colour 6 -> 1 read; colour 8 -> 8 and 6
