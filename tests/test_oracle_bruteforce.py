"""Differential test of the oracle against a second, independent formulation of the per-read path
(tests/brute_dist.py: set semantics, no BFS, no arrival order, closed-form front end, scipy minimiser).
The reference itself cannot be built in this image (parallel-hashmap, CLI11 and Boost are absent), so the
integer hot path of the oracle is pinned by two independent restatements agreeing, on the committed toy index,
on crafted tables and on a builder-made index of another shape."""
import numpy as np
import pytest

from brute_dist import BruteIndex
from helpers import closed_form, revcomp, row_of, write_index


def compare(po, index_dir, reads, th=4, check_llh=True):
    ox = po.Index(index_dir)
    bx = BruteIndex(index_dir)
    bases = np.frombuffer("".join(reads).encode(), np.uint8)
    offs = np.cumsum([0] + [len(x) for x in reads]).astype(np.uint64)
    ref = ox.dist(bases, offs, None, po.params(collect=3, hdist_th=th))
    hits_by_read, accs_by_read = {}, {}
    for h in ref["hits"]:
        hits_by_read.setdefault(int(h["read"]), set()).add((int(h["strand"]), int(h["kpos"]), int(h["cmer_index"]), int(h["hd"]), int(h["se"])))
    for a in ref["accs"]:
        accs_by_read.setdefault(int(a["read"]), {})[(int(a["strand"]), int(a["se"]))] = a
    nacc = nllh = 0
    for r, seq in enumerate(reads):
        b = bx.dist(seq, th)
        assert int(ref["reads"]["onmers"][r]) == b["onmers"], r
        assert ref["reads"]["hdist_filt"][r].tolist() == b["hdist_filt"], r
        assert hits_by_read.get(r, set()) == b["hits"], r
        oa = accs_by_read.get(r, {})
        assert set(oa) == set(b["accs"]), r
        for key, (hist, mc, hmin, passed) in b["accs"].items():
            a = oa[key]
            assert tuple(a["hist"][:th + 1].tolist()) == hist and int(a["match_count"]) == mc and int(a["hdist_min"]) == hmin, (r, key)
            assert bool(a["passed"]) == passed, (r, key)
            nacc += 1
            if passed and check_llh and nllh < 400:
                d, v, rho, uc = bx.minimise(hist, mc, b["onmers"], key[1], th)
                assert a["rho"] == pytest.approx(rho, rel=1e-15)
                # the objective at the oracle's minimiser, evaluated by the independent formula, equals the oracle's value ...
                assert bx.f(hist, uc, rho, float(a["d_llh"]), th) == pytest.approx(float(a["v_llh"]), rel=1e-9, abs=1e-9)
                # ... and is a minimum.  Interior minima: not above scipy's, to the flatness Brent's 16-bit tolerance
                # allows.  Minima at the lower bound (every match exact): Brent's stopping rule 2^-15 (|x| + 1/4) ends the
                # search about 1e-5 above the bound (the reference reports d = 1.26e-05 for such reads), scipy goes to 1e-10.
                if d > 1e-4:
                    assert float(a["v_llh"]) <= v + 1e-6 * max(1.0, abs(v))
                    assert float(a["d_llh"]) == pytest.approx(d, rel=2e-3)
                else:
                    assert float(a["d_llh"]) < 1e-4 and sum(hist[1:]) == 0
                nllh += 1
    return nacc, nllh


def test_oracle_vs_bruteforce_on_toy_index(po, toy_index_dir, toy_reads):
    names, bases, offs = toy_reads
    reads = [bytes(bases[int(offs[r]):int(offs[r + 1])]).decode() for r in range(len(names))]
    pick = reads[:150] + reads[300:]  # sampled reads of every divergence class + the edge-case reads at the end
    nacc, nllh = compare(po, toy_index_dir, pick)
    assert nacc > 300 and nllh > 100


@pytest.mark.parametrize("th", [0, 1, 4, 6, 16])
def test_oracle_vs_bruteforce_on_crafted_table(po, tmp_path, th):
    """same crafted table as tests/test_gpu_parity.py::test_crafted_min_rule_null_nodes_and_th: two entries reaching one
    leaf at different hd from the same position, an empty colour, hd above the threshold, a colour id out of range"""
    PPOS = [20, 19, 17, 13, 6, 4, 2]
    NPOS = [p for p in range(21) if p not in PPOS]
    rng = np.random.default_rng(17)
    while True:
        s = "".join("ACGT"[i] for i in rng.integers(0, 4, 60))
        rows = {}
        for i in range(0, 40):
            f = closed_form(s[i:i + 21], PPOS, NPOS)
            row = row_of(f[2], 4, 1, True)
            if row is None:
                continue
            ents = rows.setdefault(row, [])
            ents += [(f[3], 6), (f[3] ^ (1 << 2), 8), (f[3] ^ 0b11, 0), (f[3] ^ 0b11111100000, 1), (f[3] ^ (1 << 5), 99)]
        if all(len({e for e, _ in v}) == len(v) for v in rows.values()):
            break
    pse = [(0, 0), (0, 1), (0, 2), (1, 2), (0, 4), (3, 4), (1, 4), (2, 0), (6, 2)]
    rho = [0.0, 0.2, 0.25, 0.0, 0.3, 0.0]
    d = str(tmp_path / "ix")
    write_index(d, 21, 7, 4, 1, True, PPOS, rows, pse, rho, nwk="((x:0.1,y:0.1)n1:0.1,z:0.2);")
    reads = [s, revcomp(s), s[:21], s[5:50].lower(), s[:30] + "N" + s[31:], s[:20], ""]
    nacc, _ = compare(po, d, reads, th=th, check_llh=th <= 6)
    assert nacc > 0


def test_oracle_vs_bruteforce_on_a_no_frac_index(capi, po, synth, tmp_path):
    """another index shape, made by the builder: k 24, h 8, m 3, r 1, no_frac (one residue of three)"""
    nwk = "((a:0.02,b:0.03)n1:0.02,(c:0.04,(d:0.01,e:0.02)n3:0.02)n2:0.01);"
    g = synth.evolve_genomes(nwk, 6000, seed=31)
    tsv = synth.write_genomes(g, str(tmp_path / "g"))
    (tmp_path / "t.nwk").write_text(nwk)
    idx = str(tmp_path / "ix")
    capi.build_index(tsv, idx, nwk=str(tmp_path / "t.nwk"), k=24, w=31, h=8, m=3, r=1, frac=False, num_threads=2)
    bases, offs, _ = synth.sample_reads(g, 60, seed=5)
    reads = [bytes(bases[int(offs[r]):int(offs[r + 1])]).decode() for r in range(60)]
    nacc, nllh = compare(po, idx, reads)
    assert nacc > 100 and nllh > 50
