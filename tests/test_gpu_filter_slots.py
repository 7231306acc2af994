"""Filter slots (slot format 9; krepp_amd/csrc/kr_dev_scan_filt.inc): the bucket scan of IMers::add_matching_mer
(src/query.cpp:352-368) through 128-byte lines of 24-bit codes -- 12 of the 16 non-LSH positions -- whose candidates the accumulate
kernel verifies beside its colour gather.  Everything the reference computes must come out bit for bit as through the other
table layouts: table hits (src/index.cpp:160-168, src/common.hpp:175), hdist_filt (src/query.cpp:366-368), histograms
(src/query.hpp:153-176), rows; on sparse tables (every bucket in the first line), dense ones (buckets in the second line, bucket
tails beyond 80 entries in the packed arrays), reads of one / two / many segments, tiled long sequences, other thresholds
(including one the candidates' 3-bit hd12 cannot carry, which must fall back to the packed scan), export / import."""
import os

import numpy as np
import pytest

from conftest import assert_rows_close, rows_of_oracle

pytestmark = pytest.mark.gpu


def hits_key(h, with_se=True):
    cols = [h["read"].astype(np.uint64), h["strand"].astype(np.uint64), h["kpos"].astype(np.uint64), h["cmer_index"].astype(np.uint64),
            h["hd"].astype(np.uint64)]
    if with_se:
        cols.append(h["se"].astype(np.uint64))
    return np.sort(np.rec.fromarrays(cols))


def accs_want(ref, np_=5):
    acc = ref["accs"][ref["accs"]["passed"] == 1]
    return sorted(zip(acc["read"].tolist(), ((acc["se"] << 1) | acc["strand"]).tolist(), [tuple(x[:np_]) for x in acc["hist"].tolist()]))


def accs_got(res):
    return sorted(zip(res.rec_read.tolist(), res.rec_key.tolist(), [tuple(x) for x in res.rec_hist.tolist()]))


def check_against_oracle(capi, po, dx, ox, bases, offs, names, th=4, expect_fk=True):
    ref = ox.dist(bases, offs, names, po.params(collect=7, hdist_th=th))
    st = dx.stream(params=capi.default_params(hdist_th=th), max_reads=len(offs) - 1, max_bases=max(1, len(bases)), max_records=(len(offs) - 1) * 512)
    st.submit(bases, offs, capi.KR_TAP_ACCS | capi.KR_TAP_HITS)
    res = st.collect()
    gh, rh = st.hits(), ref["hits"]
    assert len(gh) == len(rh), (len(gh), len(rh))
    assert (hits_key(gh) == hits_key(rh)).all(), "table hits differ from the oracle"
    assert accs_got(res) == accs_want(ref, th + 1), "histograms differ from the oracle"
    assert_rows_close(res.rows(), rows_of_oracle(ref))
    # per-read taps: hdist_filt of both strands (src/query.cpp:366-368) and the valid k-mer count
    assert st.readtaps(len(offs) - 1).tolist() == ref["reads"]["hdist_filt"].tolist(), "hdist_filt differs from the oracle"
    assert res.read_onmers.tolist() == ref["reads"]["onmers"].tolist()
    st.close()
    # the same batch without taps: the packed problem word instead of planes, no hit tap in the scan kernel
    st = dx.stream(params=capi.default_params(hdist_th=th), max_reads=len(offs) - 1, max_bases=max(1, len(bases)), max_records=(len(offs) - 1) * 512)
    st.submit(bases, offs)
    assert_rows_close(st.collect().rows(), rows_of_oracle(ref))
    st.close()
    return ref


def test_sparse_table_every_bucket_in_the_first_line(capi, po, toy_index_dir, toy_reads, monkeypatch):
    monkeypatch.setenv("KR_SLOT_LOG2W", "9")
    hx = capi.HostIndex(toy_index_dir)
    dx = hx.upload(0)
    assert dx.slot_format == 9
    ox = po.Index(toy_index_dir)
    names, bases, offs = toy_reads
    check_against_oracle(capi, po, dx, ox, bases, offs, names)
    dx.close(), hx.close(), ox.close()


@pytest.fixture(scope="module")
def dense(capi, synth, tmp_path_factory):
    """A dense table whose buckets straddle both limits: k = 21, h = 7 (8,192 rows), 120 unrelated genomes of 25 kb with a skewed
    base composition (40 % A): ~45 entries per bucket on average, from a handful in the rows whose LSH positions avoid A to
    hundreds in the A-rich rows -- first line only, second line, and tails beyond 80 entries all occur."""
    work = tmp_path_factory.mktemp("dense")
    n = 120
    names = [f"s{i}" for i in range(n)]
    nwk = "(" + ",".join(f"{x}:0.6" for x in names) + ");"
    rng = np.random.default_rng(31)
    g = {x: np.frombuffer(b"ACGT", np.uint8)[rng.choice(4, size=25_000, p=[0.4, 0.2, 0.2, 0.2])] for x in names}
    tsv = synth.write_genomes(g, str(work / "g"))
    (work / "t.nwk").write_text(nwk)
    idx = str(work / "ix")
    capi.build_index(tsv, idx, nwk=str(work / "t.nwk"), k=21, w=27, h=7, m=4, r=1, frac=True, num_threads=8)
    return idx, g


@pytest.mark.parametrize("th", [4, 3, 6, 7])
def test_dense_table_second_line_and_tails(capi, po, synth, dense, monkeypatch, th):
    idx, g = dense
    monkeypatch.setenv("KR_SLOT_LOG2W", "9")
    hx = capi.HostIndex(idx)
    la = hx.lib_arrays(0)
    lens = np.diff(np.concatenate([[0], la["inc"]]).astype(np.int64))
    assert (lens <= 40).sum() > 100 and ((lens > 40) & (lens <= 80)).sum() > 100 and (lens > 80).sum() > 10, "the fixture must cover all three bucket classes"
    dx = hx.upload(0)
    assert dx.slot_format == 9
    ox = po.Index(idx)
    # 150-bp reads: 130 k-mer positions at k = 21 = two segments (the two-segment instantiation); 120-bp: one; 400-bp: four (merge)
    for length, nreads, seed in ((150, 600, 3), (120, 600, 4), (400, 200, 5)):
        bases, offs, rn = synth.sample_reads(g, nreads, seed=seed, length=length)
        check_against_oracle(capi, po, dx, ox, bases, offs, rn, th=th)
    dx.close(), hx.close(), ox.close()


def test_same_rows_as_the_256_byte_slots_and_the_packed_table(capi, po, synth, dense, monkeypatch):
    idx, g = dense
    bases, offs, rn = synth.sample_reads(g, 3000, seed=11, length=150)
    out = {}
    for fmt in ("9", "6", "0"):
        monkeypatch.setenv("KR_SLOT_LOG2W", fmt)
        hx = capi.HostIndex(idx)
        dx = hx.upload(0)
        assert dx.slot_format == int(fmt)
        st = dx.stream(max_reads=len(offs) - 1, max_bases=len(bases), max_records=(len(offs) - 1) * 512)
        st.submit(bases, offs, capi.KR_TAP_ACCS)
        res = st.collect()
        out[fmt] = (accs_got(res), res.rows())
        st.close(), dx.close(), hx.close()
    assert out["9"] == out["6"] == out["0"] and len(out["9"][1]) > 3000


def test_tiled_long_sequences_on_filter_slots(capi, po, synth, dense, monkeypatch):
    idx, g = dense
    monkeypatch.setenv("KR_SLOT_LOG2W", "9")
    rng = np.random.default_rng(8)
    names_g = list(g)
    seqs = []
    for L in (150, 5000, 1044, 1045, 150, 12000, 20, 2600):
        s = bytearray(g[names_g[int(rng.integers(0, len(names_g)))]][:L].tobytes())
        for _ in range(L // 50):
            s[int(rng.integers(0, L))] = b"ACGTN"[int(rng.integers(0, 5))]
        seqs.append(bytes(s))
    bases = np.frombuffer(b"".join(seqs), np.uint8)
    offs = np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint64)
    names = [f"q{i}" for i in range(len(seqs))]
    hx = capi.HostIndex(idx)
    dx = hx.upload(0)
    ox = po.Index(idx)
    ref = ox.dist(bases, offs, names, po.params(collect=7))
    for no_tiles in (False, True):
        if no_tiles:
            monkeypatch.setenv("KR_NO_TILES", "1")
        st = dx.stream(max_reads=4096, max_bases=len(bases) + 64, max_records=1 << 20)
        st.submit(bases, offs, capi.KR_TAP_ACCS)
        res = st.collect()
        assert accs_got(res) == accs_want(ref), f"histograms differ (KR_NO_TILES={no_tiles})"
        assert_rows_close(res.rows(), rows_of_oracle(ref))
        assert st.readtaps(len(names)).tolist() == ref["reads"]["hdist_filt"].tolist()  # the sequence's minima, not a tile's
        st.close()
    dx.close(), hx.close(), ox.close()


def test_export_import_of_filter_slots(capi, po, synth, dense, monkeypatch):
    import torch

    idx, g = dense
    monkeypatch.setenv("KR_SLOT_LOG2W", "9")
    hx = capi.HostIndex(idx)
    dx = hx.upload(0)
    desc, bufs = dx.export()
    dx2, bufs2 = capi.DeviceIndex.import_empty(desc, 0)
    assert [b for _, b in bufs] == [b for _, b in bufs2] and dx2.device_bytes == dx.device_bytes and dx2.slot_format == 9

    class DevPtr:
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

    for (p1, nb), (p2, _) in zip(bufs, bufs2):
        if nb:
            torch.as_tensor(DevPtr(p2, nb), device="cuda:0").copy_(torch.as_tensor(DevPtr(p1, nb), device="cuda:0"))
    torch.cuda.synchronize()
    (rep,) = dx.broadcast([0])  # the one-rank RCCL path carries the slot-ordered colours too
    bases, offs, rn = synth.sample_reads(g, 500, seed=4, length=150)
    want = rows_of_oracle(po.Index(idx).dist(bases, offs, rn, po.params(collect=1)))
    for d in (dx, dx2, rep):
        st = d.stream(max_reads=len(offs) - 1, max_bases=len(bases), max_records=(len(offs) - 1) * 512)
        st.submit(bases, offs)
        assert_rows_close(st.collect().rows(), want)
        st.close()
    for d in (rep, dx2, dx):
        d.close()
    hx.close()
