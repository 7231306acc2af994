"""BASELINE.json configs[4] under pytest: `krepp place` against the 25-reference index of the metric's shape
(-k 27 -w 35 -h 11, m4r1-frac; 25 synthetic genomes of 400 kb evolved down tests/golden/tree_toy.nwk, seed 7: the
index of bench.py --workload toy25) with tree_toy.nwk as the backbone.

1. 20,000 reads against the oracle's restatement of IBatch::place_sequences / report_placement
   (src/query.cpp:198-333): jplace and --tabular text byte for byte, placements field by field;
2. ONE batch of 1,000,000 reads (the 10 M of the config are ten of these): the device back end (kr_place_kernel +
   kr_place_llh_kernel through kr_place_stream) against the host back end (kr_place_batch) bit for bit -- text and
   placement structs -- and NOT ONE batch may have fallen back to the host (kr_place_counters).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

N_ORACLE, N_FULL = 20_000, 1_000_000


@pytest.fixture(scope="module")
def k27(capi, synth, tmp_path_factory):
    work = tmp_path_factory.mktemp("place_k27")
    nwk = os.path.join(GOLDEN, "tree_toy.nwk")
    genomes = synth.evolve_genomes(open(nwk).read(), 400_000, seed=7)
    tsv = synth.write_genomes(genomes, str(work / "g"))
    idx = str(work / "idx")
    capi.build_index(tsv, idx, nwk=nwk, k=27, w=35, h=11, m=4, r=1, frac=True, num_threads=min(8, os.cpu_count() or 1))
    return idx, genomes


def placements_key(pl):
    return sorted((int(r), int(e)) for r, e in zip(pl["read"], pl["edge"]))


def test_place_k27_h11_20000_reads_match_the_oracle(capi, po, synth, k27):
    idx, genomes = k27
    bases, offs, names = synth.sample_reads(genomes, N_ORACLE, seed=1)
    hx = capi.HostIndex(idx)
    ox = po.Index(idx)
    ox.set_placement_tree(None)  # the index's own tree (TargetIndex::ensure_backbone, src/krepp.cpp:48-64)
    dev0, host0 = capi.place_counters()
    for tabular in (False, True):
        want = ox.place(bases, offs, names, po.params(no_filter=0, num_threads=min(16, os.cpu_count() or 1)), tabular=tabular)
        placer = capi.Placer(hx, None, 0, tabular=tabular, max_reads=N_ORACLE, max_bases=len(bases))
        text, pl = placer.place(bases, offs, names)
        assert len(pl) > N_ORACLE // 2
        assert placements_key(pl) == placements_key(want["placements"]), tabular
        a = np.sort(pl, order=["read", "edge"])
        b = np.sort(want["placements"], order=["read", "edge"])
        for f in ("lwr", "d_llh", "pendant", "distal"):
            assert np.allclose(a[f], b[f], rtol=1e-6, atol=1e-9), f  # the north star's DIST tolerance
        assert np.allclose(a["v_llh"], b["v_llh"], rtol=1e-9)
        assert text == want["text"], tabular
        assert placer.frame(0, "inv") == ox.place_frame(0, tabular, "inv")
        assert placer.frame(1, "inv", N_ORACLE) == ox.place_frame(1, tabular, "inv", N_ORACLE)
        placer.close()
    dev1, host1 = capi.place_counters()
    assert dev1 - dev0 == 2 and host1 == host0, "a batch fell back to the host back end"


def test_place_k27_h11_one_million_reads_device_equals_host_no_fallback(capi, synth, k27):
    idx, genomes = k27
    chunks = [synth.sample_reads(genomes, 100_000, seed=5000 + c)[0] for c in range(N_FULL // 100_000)]
    bases = np.concatenate(chunks)
    offs = np.arange(N_FULL + 1, dtype=np.uint64) * np.uint64(150)
    names = [f"r{i}" for i in range(N_FULL)]
    import ctypes as C

    c_names = (C.c_char_p * N_FULL)(*[n.encode() for n in names])
    hx = capi.HostIndex(idx)
    placer = capi.Placer(hx, None, 0, tabular=False, max_reads=N_FULL, max_bases=len(bases))
    dev0, host0 = capi.place_counters()
    text_d, pl_d = placer.place(bases, offs, names, host=False, c_names=c_names)
    dev1, host1 = capi.place_counters()
    assert (dev1 - dev0, host1 - host0) == (1, 0), "the 1 M-read batch did not run its back end on the device"
    placer.prev = C.c_int(0)  # (a second batch of one run would start with the jplace separator)
    text_h, pl_h = placer.place(bases, offs, names, host=True, c_names=c_names)
    assert len(pl_d) > N_FULL // 2 and len(np.unique(pl_d["read"])) > N_FULL // 2
    assert pl_d.tobytes() == pl_h.tobytes(), "device and host back ends differ in a placement"
    assert text_d == text_h, "device and host back ends differ in the jplace text"
    # every placed read's LWRs sum to 1 (src/query.cpp:284-296)
    sums = np.bincount(pl_d["read"].astype(np.int64), weights=pl_d["lwr"], minlength=N_FULL)
    placed = np.bincount(pl_d["read"].astype(np.int64), minlength=N_FULL) > 0
    assert np.allclose(sums[placed], 1.0, atol=1e-9)
    placer.close()
