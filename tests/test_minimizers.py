"""Build-side minimizers: the CPU leaf stage against an independent pure-Python
simulation of RSeq::extract_mers' ring buffer (src/rqseq.cpp:51-144), and the HIP kernel against
the CPU leaf stage (bit-exact minimizer sets and HyperLogLog sums)."""
import numpy as np
import pytest

from helpers import closed_form, row_of

PPOS = [20, 19, 17, 13, 6, 4, 2]
NPOS = [p for p in range(21) if p not in PPOS]


def fmix(x):
    x ^= x >> 33
    x = (x * 0xff51afd7ed558ccd) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 33
    x = (x * 0xc4ceb9fe1a85ec53) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 33)


def ring_buffer_sim(contigs, k, w, m, r, frac, ppos, npos):
    """Literal simulation: returns (sorted unique keys, list of per-contig (c1 hashes, c2 hashes))."""
    keys, sketches = set(), []
    P, N = sorted(ppos), sorted(npos)
    for t in contigs:
        if len(t) < w:
            continue
        ldiff = w - k + 1
        win = [(0, 0)] * ldiff
        kix = l = 0
        c1, c2 = [], []
        for i in range(1, len(t) + 1):
            ch = t[i - 1]
            if ch not in "ACGTacgt":
                l = 0
                continue
            l += 1
            if l < k:
                continue
            f = closed_form(t[i - k:i], ppos, npos)
            win[kix % ldiff] = (f[0], fmix(f[0]))
            c1.append(win[kix % ldiff][1] & 0xFFFFFFFF)
            kix += 1
            if l < w and i != len(t):
                continue
            x, z = min(win, key=lambda e: e[1])
            c2.append(z & 0xFFFFFFFF)
            codes = [(x >> (2 * p)) & 3 for p in range(k)]
            rix = sum(codes[P[j]] << (2 * j) for j in range(len(P)))
            enc = sum(((codes[N[j]] & 1) << j) | ((codes[N[j]] >> 1) << (16 + j)) for j in range(len(N)))
            row = row_of(rix, m, r, frac)
            if row is not None:
                keys.add((row << 32) | enc)
        sketches.append((c1, c2))
    return np.array(sorted(keys), dtype=np.uint64), sketches


def hll12(hashes):
    reg = np.zeros(4096, np.int64)
    for h in hashes:
        ix, rest = h >> 20, (h << 12) & 0xFFFFFFFF
        lz = 32 - rest.bit_length() if rest else 32
        reg[ix] = max(reg[ix], min(20, lz) + 1)
    mreg = 4096.0
    est = (0.7213 / (1.0 + 1.079 / mreg)) * mreg * mreg / float(np.sum(1.0 / (1 << reg)))
    zeros = int((reg == 0).sum())
    if est <= 2.5 * mreg and zeros:
        est = mreg * np.log(mreg / zeros)
    return est


def make_contigs(rng):
    def rnd(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    c = [rnd(3000), rnd(26), rnd(27), rnd(28), rnd(40) + "N" + rnd(23), rnd(500).lower(),
         rnd(100) + "NNN" + rnd(22) + "N" + rnd(21) + "R" + rnd(24), rnd(30) + "N" + rnd(25),  # short final runs
         "N" * 40, rnd(21) + "N" + rnd(5),  # ends without a k-mer
         rnd(24) + "N" * 3, rnd(2100) + "N" + rnd(2050)]
    return c


@pytest.mark.parametrize("k,w,h", [(21, 27, 7), (21, 21, 7), (21, 60, 7)])
def test_cpu_leaf_stage_matches_ring_buffer_simulation(capi, k, w, h):
    rng = np.random.default_rng(5)
    contigs = make_contigs(rng)
    bases = np.frombuffer("".join(contigs).encode(), np.uint8)
    offs = np.cumsum([0] + [len(c) for c in contigs]).astype(np.uint64)
    keys, n1, n2 = capi.minimizers(bases, offs, k, w, h, PPOS)
    want, sk = ring_buffer_sim(contigs, k, w, 4, 1, True, PPOS, NPOS)
    assert keys.tolist() == want.tolist() and len(want) > 100
    assert n1 == sum(hll12(a) for a, _ in sk) and n2 == sum(hll12(b) for _, b in sk)


@pytest.mark.gpu
@pytest.mark.parametrize("k,w,h,m,r,frac", [(21, 27, 7, 4, 1, True), (29, 35, 13, 4, 1, True), (27, 27, 11, 8, 3, False),
                                            (31, 90, 15, 4, 0, True), (19, 255, 3, 2, 1, True)])
def test_hip_minimizers_bit_exact(capi, synth, k, w, h, m, r, frac):
    rng = np.random.default_rng(9)
    ppos = sorted(rng.choice(k, h, replace=False).tolist(), reverse=True)
    contigs = make_contigs(rng)
    g = synth.evolve_genomes("(a:0.1,b:0.1);", 300_000, seed=3)["a"].tobytes().decode()
    g = g[:100_000] + "N" * 7 + g[100_000:200_000] + "n" + g[200_000:]
    contigs += [g, g[:4095], g[:4096], g[:4097], g[5000:5000 + 2048 + w - 1]]
    bases = np.frombuffer("".join(contigs).encode(), np.uint8)
    offs = np.cumsum([0] + [len(c) for c in contigs]).astype(np.uint64)
    kc, n1c, n2c = capi.minimizers(bases, offs, k, w, h, ppos, m=m, r=r, frac=frac)
    kg, n1g, n2g = capi.minimizers(bases, offs, k, w, h, ppos, m=m, r=r, frac=frac, device=0)
    assert kg.tolist() == kc.tolist() and len(kc) > 1000
    assert (n1g, n2g) == (n1c, n2c)


@pytest.mark.gpu
@pytest.mark.parametrize("k,w,h", [(21, 27, 7), (21, 21, 7), (21, 60, 7)])
def test_hip_minimizers_match_ring_buffer_simulation_directly(capi, k, w, h):
    """The HIP kernel against the independent pure-Python ring-buffer simulation itself (not through the product's
    CPU leaf stage): minimizer key sets bit-exact, HyperLogLog sums equal — including the stale-slot emission at the
    end of a contig whose last run of valid bases is shorter than w (src/rqseq.cpp:108-116)."""
    rng = np.random.default_rng(5)
    contigs = make_contigs(rng)
    contigs.append("".join("ACGT"[i] for i in rng.integers(0, 4, 9000)))  # crosses several 2048-position tiles
    bases = np.frombuffer("".join(contigs).encode(), np.uint8)
    offs = np.cumsum([0] + [len(c) for c in contigs]).astype(np.uint64)
    keys, n1, n2 = capi.minimizers(bases, offs, k, w, h, PPOS, device=0)
    want, sk = ring_buffer_sim(contigs, k, w, 4, 1, True, PPOS, NPOS)
    assert keys.tolist() == want.tolist() and len(want) > 100
    assert n1 == sum(hll12(a) for a, _ in sk) and n2 == sum(hll12(b) for _, b in sk)


@pytest.mark.gpu
def test_index_built_with_gpu_leaf_stage_is_identical(capi, synth, tmp_path):
    import os, time
    nwk = "((a:0.03,b:0.03):0.02,(c:0.05,(d:0.01,e:0.01):0.02):0.01);"
    g = synth.evolve_genomes(nwk, 200_000, seed=8)
    tsv = synth.write_genomes(g, str(tmp_path / "g"), contigs=3)
    (tmp_path / "t.nwk").write_text(nwk)
    d1, d2 = str(tmp_path / "cpu"), str(tmp_path / "gpu")
    t0 = time.time()
    capi.build_index(tsv, d1, nwk=str(tmp_path / "t.nwk"), k=29, w=35, h=13, m=4, r=1, frac=True, seed=3)
    t1 = time.time()
    capi.build_index(tsv, d2, nwk=str(tmp_path / "t.nwk"), k=29, w=35, h=13, m=4, r=1, frac=True, seed=3, gpu_minimizers=True)
    t2 = time.time()
    for f in ("cmer", "inc", "crecord", "metadata", "tree", "reflist"):
        assert open(os.path.join(d1, f + "-m4r1-frac"), "rb").read() == open(os.path.join(d2, f + "-m4r1-frac"), "rb").read(), f
    print(f"build: cpu leaf stage {t1 - t0:.2f} s, gpu leaf stage {t2 - t1:.2f} s")
