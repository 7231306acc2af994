"""The report rows of `krepp dist` as text written by the GPU (csrc/kr_dev_text.inc; kr_stream_text_enable /
kr_batch_submit_text / kr_batch_collect_text) against IBatch::report_distances (src/query.cpp:158-196): byte-identical to the
oracle's text and to the host formatter's (kr_format_dist) in every report mode, with reads that keep no reference
(`SEQ_ID\\tNA\\tNaN`), ids of different lengths, several batches on one stream, the capacity and fall-back paths."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def toy(capi, po, toy_index_dir):
    hx = capi.HostIndex(toy_index_dir)
    dx = hx.upload(0)
    ox = po.Index(toy_index_dir)
    yield hx, dx, ox
    dx.close()
    hx.close()


def host_text(capi, hx, dx, bases, offs, names, **pkw):
    st = dx.stream(capi.default_params(**pkw), max_reads=len(names), max_bases=len(bases) + 64)
    st.submit(bases, offs, capi.KR_ROWS_ONLY)
    st.collect()
    t = st.format_dist(hx, names)
    st.close()
    return t


@pytest.mark.parametrize("pkw,okw", [(dict(), dict()), (dict(no_filter=0), dict(no_filter=0)), (dict(multi=0), dict(multi=0)),
                                     (dict(dist_max=0.05), dict(dist_max=0.05)), (dict(hdist_th=2), dict(hdist_th=2))])
def test_device_text_equals_oracle_text_in_every_report_mode(capi, po, toy, toy_reads, pkw, okw):
    hx, dx, ox = toy
    names, bases, offs = toy_reads
    want = ox.dist(bases, offs, names, po.params(collect=4, **okw))["text"]
    st = dx.stream(capi.default_params(**pkw), max_reads=len(names), max_bases=len(bases) + 64)
    st.text_enable(hx, 1 << 22, 1 << 16)
    st.submit_text(bases, offs, names)
    got = st.collect_text().decode()
    # the oracle's rows of one read come in hash-map order: compare the reads' row SETS in read order
    def by_read(t):
        out, cur = [], None
        for line in t.splitlines():
            sid = line.split("\t", 1)[0]
            if sid != cur:
                out.append([])
                cur = sid
            out[-1].append(line)
        return [sorted(g) for g in out]
    assert by_read(got) == by_read(want)
    assert got == host_text(capi, hx, dx, bases, offs, names, **pkw)  # byte for byte, row order included
    assert "\tNA\tNaN\n" in got or pkw.get("hdist_th") is None
    # the same stream again, with ids of other lengths, then an ordinary submit: nothing sticks
    names2 = [("q" * (1 + i % 37)) + str(i) for i in range(len(names))]
    st.submit_text(bases, offs, names2)
    got2 = st.collect_text().decode()
    assert got2 == host_text(capi, hx, dx, bases, offs, names2, **pkw)
    st.submit(bases, offs, capi.KR_ROWS_ONLY)
    st.collect()
    assert st.format_dist(hx, names) == got
    with pytest.raises(capi.KrError) as e:
        st.collect_text()
    assert e.value.code == capi.KR_ERR_STATE
    st.close()


def test_device_text_on_200000_reads_and_its_error_paths(capi, synth, toy_genomes, tmp_path):
    """25 references (-k 27 -w 35 -h 11), 200,000 reads -- one in a hundred with an N, one in ten unrelated (NA rows) -- in one
    batch and in three: the device's bytes equal kr_format_dist's; a text buffer that is too small is KR_ERR_CAPACITY (and the
    halves fit); a batch with a long sequence (tiled on the device) is KR_ERR_UNSUPPORTED and still served by kr_batch_collect."""
    nwk = os.path.join(GOLDEN, "tree_toy.nwk")
    genomes = synth.evolve_genomes(open(nwk).read(), 400_000, seed=7)
    tsv = synth.write_genomes(genomes, str(tmp_path / "g"))
    idx = str(tmp_path / "idx")
    capi.build_index(tsv, idx, nwk=nwk, k=27, w=35, h=11, m=4, r=1, frac=True, num_threads=min(8, os.cpu_count() or 1))
    hx = capi.HostIndex(idx)
    dx = hx.upload(0)
    n = 200_000
    bases = np.concatenate([synth.sample_reads(genomes, 100_000, seed=300 + c)[0] for c in range(2)])
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(150)
    names = [f"read_{i}" if i % 3 else f"r{i}/1" for i in range(n)]
    want = host_text(capi, hx, dx, bases, offs, names)
    assert want.count("\n") > n and "\tNA\tNaN\n" in want
    st = dx.stream(max_reads=n, max_bases=len(bases) + 64)
    st.text_enable(hx, len(want) + 4096, 16 * n)
    st.submit_text(bases, offs, names)
    assert st.collect_text().decode() == want
    parts = []
    for lo, hi in ((0, 70_001), (70_001, 70_002), (70_002, n)):
        st.submit_text(bases[lo * 150: hi * 150], offs[lo: hi + 1] - offs[lo], names[lo:hi])
        parts.append(st.collect_text().decode())
    assert "".join(parts) == want
    st.close()
    # too little room for the text: KR_ERR_CAPACITY, nothing written past the buffer, and the halves fit
    small = dx.stream(max_reads=n, max_bases=len(bases) + 64)
    small.text_enable(hx, len(want) * 3 // 5, 16 * n)
    small.submit_text(bases, offs, names)
    with pytest.raises(capi.KrError) as e:
        small.collect_text()
    assert e.value.code == capi.KR_ERR_CAPACITY
    half = n // 2
    small.submit_text(bases[: half * 150], offs[: half + 1], names[:half])
    a = small.collect_text().decode()
    small.submit_text(bases[half * 150:], offs[half:] - offs[half], names[half:])
    assert a + small.collect_text().decode() == want
    # ids that do not fit: refused at submit
    with pytest.raises(capi.KrError) as e:
        small.submit_text(bases, offs, ["x" * 40] * n)
    assert e.value.code == capi.KR_ERR_CAPACITY
    small.close()
    # a long sequence in the batch: it runs as tiles, the rows stay record slots -> UNSUPPORTED, kr_batch_collect serves it
    contig = np.ascontiguousarray(next(iter(genomes.values()))[:5000], dtype=np.uint8)
    b2 = np.concatenate([bases[:1500], contig])
    o2 = np.concatenate([offs[:11], [np.uint64(1500 + len(contig))]]).astype(np.uint64)
    n2 = names[:10] + ["contig"]
    lt = dx.stream(max_reads=64, max_bases=len(b2) + 64)
    lt.text_enable(hx, 1 << 20, 1 << 12)
    lt.submit_text(b2, o2, n2)
    with pytest.raises(capi.KrError) as e:
        lt.collect_text()
    assert e.value.code == capi.KR_ERR_UNSUPPORTED
    lt.collect()
    t2 = lt.format_dist(hx, n2)
    assert t2.startswith(want[:50]) and "contig\t" in t2
    # ... and when the tiles' records "do not fit" (KR_DEBUG_TILE_OVERFLOW: kr_batch_wait runs the batch again, one wave per
    # sequence) the rerun is still a text batch: the device's bytes, equal to the host formatter's
    os.environ["KR_DEBUG_TILE_OVERFLOW"] = "1"
    try:
        lt.submit_text(b2, o2, n2)
        assert lt.collect_text().decode() == t2
    finally:
        del os.environ["KR_DEBUG_TILE_OVERFLOW"]
    lt.close()
    dx.close()
    hx.close()
