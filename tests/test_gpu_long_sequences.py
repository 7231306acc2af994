"""Long query sequences across waves (krepp_amd/csrc/kr_dev_tiles.inc): a host batch that holds sequences of more than 1,024
k-mer positions is submitted as tiles of 128 positions; the tiles' histograms are added per (leaf, strand), the hdist_filt test
(src/query.cpp:101-106,119) is applied with the SEQUENCE's minimum, and the results must be those of the reference's serial scan
(search_mers, src/query.cpp:40-94) -- the oracle's -- and of the untiled device path (KR_NO_TILES), bit for bit."""
import os

import numpy as np
import pytest

from conftest import assert_rows_close, rows_of_oracle

pytestmark = pytest.mark.gpu


def make_batch(g, seed):
    rng = np.random.default_rng(seed)
    names_g = list(g)
    seqs = []
    # reads, sequences just below / at / above the tiling threshold (k = 21: 1,024 positions = 1,044 bases), contigs, a whole genome
    for L in (150, 149, 1043, 1044, 1045, 1172, 1173, 150, 3000, 20, 5000, 12345, 20000, 150, 0, 2500):
        name = names_g[int(rng.integers(0, len(names_g)))]
        o = int(rng.integers(0, 20000 - L + 1))
        s = bytearray(g[name][o:o + L].tobytes())
        for _ in range(L // 40):  # substitutions and the odd N
            s[int(rng.integers(0, L))] = b"ACGTN"[int(rng.integers(0, 5))]
        if L == 12345:  # a run of N across a tile boundary, and a long stretch from another genome (chimeric contig)
            s[1270:1300] = b"N" * 30
            other = g[names_g[(names_g.index(name) + 7) % len(names_g)]]
            s[6000:9000] = other[100:3100].tobytes()
        if L == 2500 and rng.integers(0, 2):  # reverse strand
            comp = bytes.maketrans(b"ACGT", b"TGCA")
            s = bytearray(bytes(s).translate(comp)[::-1])
        seqs.append(bytes(s))
    bases = np.frombuffer(b"".join(seqs), np.uint8)
    offs = np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint64)
    return bases, offs, [f"q{i}_L{len(s)}" for i, s in enumerate(seqs)]


@pytest.fixture(scope="module")
def toy(capi, po, toy_index_dir):
    hx = capi.HostIndex(toy_index_dir)
    return hx, hx.upload(0), po.Index(toy_index_dir)


def run(capi, dx, bases, offs, flags=0, **pkw):
    st = dx.stream(params=capi.default_params(**pkw), max_reads=4096, max_bases=len(bases) + 64, max_records=1 << 20)
    st.submit(bases, offs, flags)
    return st, st.collect()


def accs_of(res):
    return sorted(zip(res.rec_read.tolist(), res.rec_key.tolist(), [tuple(x) for x in res.rec_hist.tolist()]))


def test_tiled_sequences_equal_the_oracle_and_the_untiled_path(capi, po, toy, toy_genomes, monkeypatch):
    hx, dx, ox = toy
    bases, offs, names = make_batch(toy_genomes, 5)
    ref = ox.dist(bases, offs, names, po.params(collect=7))
    acc = ref["accs"][ref["accs"]["passed"] == 1]
    want = sorted(zip(acc["read"].tolist(), ((acc["se"] << 1) | acc["strand"]).tolist(), [tuple(x[:5]) for x in acc["hist"].tolist()]))
    st, res = run(capi, dx, bases, offs, capi.KR_TAP_ACCS)
    assert res.read_onmers.tolist() == ref["reads"]["onmers"].tolist()
    assert st.readtaps(len(names)).tolist() == ref["reads"]["hdist_filt"].tolist()  # the sequence's minima, not a tile's
    assert accs_of(res) == want, "histograms of tiled sequences differ from the oracle"
    assert max(h[2][0] for h in accs_of(res)) > 255  # a contig's counts: more than the packed word of a record holds
    assert_rows_close(res.rows(), rows_of_oracle(ref))
    assert sorted(st.format_dist(hx, names).splitlines()) == sorted(ref["text"].splitlines())
    # the same batch, one wave per sequence
    monkeypatch.setenv("KR_NO_TILES", "1")
    st1, res1 = run(capi, dx, bases, offs, capi.KR_TAP_ACCS)
    monkeypatch.delenv("KR_NO_TILES")
    assert accs_of(res1) == accs_of(res) and res1.rows() == res.rows()
    # report modes without taps (records as key + packed word for the reads, planes for the long sequences), rows only
    for pkw, okw in ((dict(), dict()), (dict(no_filter=0), dict(no_filter=0)), (dict(multi=0), dict(multi=0)), (dict(hdist_th=3), dict(hdist_th=3)),
                     (dict(dist_max=0.05), dict(dist_max=0.05))):
        want_rows = rows_of_oracle(ox.dist(bases, offs, names, po.params(collect=0, **okw)))
        for fl in (0, capi.KR_ROWS_ONLY):
            _, r0 = run(capi, dx, bases, offs, fl, **pkw)
            assert_rows_close(r0.rows(), want_rows)


def test_tiled_batch_through_the_device_view_and_place(capi, po, toy, toy_genomes, toy_index_dir):
    import torch

    hx, dx, ox = toy
    bases, offs, names = make_batch(toy_genomes, 9)
    st, res = run(capi, dx, bases, offs)
    # device view: the per-read arrays are those of the real reads
    st.submit(bases, offs)
    rv = st.collect_device()
    assert rv.nreads == len(names)

    class DevPtr:
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

    import ctypes as C

    def dev(ptr, n, dt):
        return torch.as_tensor(DevPtr(C.cast(ptr, C.c_void_p).value, n * np.dtype(dt).itemsize), device="cuda:0").cpu().numpy().view(dt)

    n = rv.nreads
    off, cnt = dev(rv.read_off, n, np.uint32), dev(rv.read_cnt, n, np.uint32)
    k_, d_, s_ = dev(rv.rec_key, rv.nrecs, np.uint32), dev(rv.rec_d, rv.nrecs, np.float64), dev(rv.rec_sel, rv.nrecs, np.uint8)
    got = sorted((r, int(k_[i]) >> 1, float(d_[i])) for r in range(n) for i in range(off[r], off[r] + cnt[r]) if s_[i])
    assert got == res.rows()
    # place: the records of the long sequences (histogram planes at their first tile) through the device back end
    ox.set_placement_tree(None)
    want = ox.place(bases, offs, names, po.params(no_filter=0))
    placer = capi.Placer(hx, None, 0, max_reads=4096, max_bases=len(bases) + 64)
    text, pl = placer.place(bases, offs, names)
    assert text == want["text"]
    placer.close()


def test_a_tiled_batch_that_overflows_is_run_again_untiled(capi, po, toy, toy_genomes, monkeypatch):
    """The tiles' own records need room in the device buffers; when there is none the library runs the batch again with one wave
    per sequence instead of failing it (the caller's buffers stay valid until wait / collect returns).  The overflow is
    simulated (KR_DEBUG_TILE_OVERFLOW): the stream's slack for partly used record chunks alone holds a small batch's tiles."""
    hx, dx, ox = toy
    bases, offs, names = make_batch(toy_genomes, 5)
    ref = ox.dist(bases, offs, names, po.params(collect=0))
    monkeypatch.setenv("KR_DEBUG_TILE_OVERFLOW", "1")
    st, res = run(capi, dx, bases, offs)
    assert_rows_close(res.rows(), rows_of_oracle(ref))
    st.submit(bases, offs)  # and the stream is as good as new
    assert_rows_close(st.collect().rows(), rows_of_oracle(ref))


def test_max_records_bounds_a_tiled_batch_by_what_survives(capi, po, toy, toy_genomes):
    """a batch that produces more records than the stream's max_records fails with KR_ERR_CAPACITY.  For a tiled batch
    the count is of the records that SURVIVE -- the sequences' merged records and the ordinary reads' -- not of the tiles' own records,
    which become holes (kr_tile_merge_kernel counts them): with room for the survivors the batch succeeds, with less it fails."""
    hx, dx, ox = toy
    bases, offs, names = make_batch(toy_genomes, 5)
    nrec = len(ox.dist(bases, offs, names, po.params(collect=1))["accs"])  # every (read, strand, leaf) accumulator
    ref = ox.dist(bases, offs, names, po.params(collect=7))
    surviving = int((ref["accs"]["passed"] == 1).sum())
    assert surviving > 40
    st = dx.stream(max_reads=4096, max_bases=len(bases) + 64, max_records=surviving + 8)
    st.submit(bases, offs)
    assert_rows_close(st.collect().rows(), rows_of_oracle(ref))
    st.close()
    st = dx.stream(max_reads=4096, max_bases=len(bases) + 64, max_records=surviving // 2)
    st.submit(bases, offs)
    with pytest.raises(capi.KrError) as e:
        st.collect()
    assert e.value.code == capi.KR_ERR_CAPACITY
    st.close()


def test_a_stream_with_room_for_some_of_the_tiles_tiles_what_fits(capi, po, toy, toy_genomes):
    """A long sequence of nt tiles takes nt - 1 reads more than the batch has; a stream created for fewer reads than the tiles of
    all the batch's sequences tiles those that fit (in order) and leaves the others to one wave each: same results."""
    hx, dx, ox = toy
    rng = np.random.default_rng(21)
    names_g = list(toy_genomes)
    seqs = []
    for L in (5000, 150, 5000, 3000, 150, 5000):  # 39, 39, 24 and 39 tiles: 38, 38, 23 and 38 reads more than the batch has
        gname = names_g[int(rng.integers(0, len(names_g)))]
        o = int(rng.integers(0, 20000 - L + 1))
        seqs.append(toy_genomes[gname][o:o + L].tobytes())
    bases = np.frombuffer(b"".join(seqs), np.uint8)
    offs = np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint64)
    names = [f"s{i}" for i in range(len(seqs))]
    ref = ox.dist(bases, offs, names, po.params(collect=7))
    acc = ref["accs"][ref["accs"]["passed"] == 1]
    want = sorted(zip(acc["read"].tolist(), ((acc["se"] << 1) | acc["strand"]).tolist(), [tuple(x[:5]) for x in acc["hist"].tolist()]))
    for max_reads in (70, 48, 43, 6):  # room for: the first and the 3 kb one; the first; the 3 kb one alone; none
        st = dx.stream(params=capi.default_params(), max_reads=max_reads, max_bases=len(bases) + 64, max_records=1 << 18)
        st.submit(bases, offs, capi.KR_TAP_ACCS)
        res = st.collect()
        assert accs_of(res) == want, max_reads
        assert_rows_close(res.rows(), rows_of_oracle(ref))
        assert st.readtaps(len(names)).tolist() == ref["reads"]["hdist_filt"].tolist()


def test_fuzzed_mixes_of_reads_and_contigs(capi):
    """scripts/fuzz_long.py: random batches of reads, sequences around the tiling threshold and around multiples of the tile length,
    chimeric contigs on either strand with substitutions and N runs, on streams with room for all, some or none of the tiles:
    histograms, hdist_filt and report text against the oracle, with and without the histogram tap."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_long.py"), "8"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "mismatching batches: 0" in out.stdout, out.stdout[-2000:]
