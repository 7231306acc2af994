"""GPU parity tests: the HIP path, called through the C ABI, against the oracle.
Bar: bit-exact for rix/enc32/table hits/histograms/row sets; DIST within 1e-6 relative
(north star) — in practice ~1e-11 (device vs glibc pow/log)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, assert_rows_close, rows_of_oracle
from helpers import closed_form, hd32, revcomp, row_of, write_index

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def toy(capi, po, toy_index_dir):
    hx = capi.HostIndex(toy_index_dir)
    dx = hx.upload(0)
    return hx, dx, po.Index(toy_index_dir)


def gpu_dist(capi, dx, bases, offs, flags=0, **pkw):
    st = dx.stream(params=capi.default_params(**pkw), max_reads=len(offs) - 1, max_bases=max(1, len(bases)))
    st.submit(bases, offs, flags)
    res = st.collect()
    return st, res


@pytest.mark.parametrize("planes", ["0", "1", "2"], ids=["windows_pext", "bit_planes", "byte_tables"])
def test_front_end_bit_exact(capi, toy, toy_reads, monkeypatch, planes):
    """rix / enc32 / validity of every k-mer and strand (src/common.hpp:177-243, src/lshf.cpp:39-69) through both front ends: per-lane
    windows + software PEXT, the bit-plane form (kr_dev_common.inc: front_end_planes) and the byte tables in LDS that the slotted scan
    kernel uses (front_end_tab)."""
    monkeypatch.setenv("KR_DEBUG_FE_PLANES", planes)
    hx, dx, ox = toy
    names, bases, offs = toy_reads
    lens = np.diff(offs).astype(int)
    stride = int(lens.max()) - 21 + 1
    rix, enc, valid, pas = dx.front_end(bases, offs, stride)
    for r in range(len(names)):
        fe = ox.front_end(bytes(bases[int(offs[r]):int(offs[r + 1])]))
        seen = np.zeros((stride, 2), bool)
        for i in range(len(fe["kpos"])):
            kp, s = int(fe["kpos"][i]), int(fe["strand"][i])
            assert valid[r, kp, s] == 1, (names[r], kp, s)
            assert rix[r, kp, s] == fe["rix"][i] and enc[r, kp, s] == fe["enc32"][i] and pas[r, kp, s] == fe["pas"][i]
            seen[kp, s] = True
        assert not valid[r][~seen].any(), names[r]  # windows with N / past the end are invalid


def test_hits_accumulators_rows_match_committed_golden(capi, toy, toy_reads):
    hx, dx, ox = toy
    names, bases, offs = toy_reads
    exp = json.load(open(os.path.join(GOLDEN, "toy_expected.json")))["default"]
    st, res = gpu_dist(capi, dx, bases, offs, capi.KR_TAP_ACCS | capi.KR_TAP_HITS)
    assert len(st.hits()) == exp["nhits"]
    assert res.read_onmers.tolist() == exp["onmers"]
    assert st.readtaps(len(names)).tolist() == exp["hdist_filt"]
    want_acc = sorted((a[0], (a[1] << 1) | a[2], tuple(a[4])) for a in exp["accs"] if a[3])
    got_acc = sorted(zip(res.rec_read.tolist(), res.rec_key.tolist(), [tuple(x) for x in res.rec_hist.tolist()]))
    assert got_acc == want_acc
    want = sorted((r, s, float.fromhex(d)) for r, s, d in exp["rows"] if s != 0)
    assert_rows_close(res.rows(), want)
    assert sorted(st.format_dist(hx, names).splitlines()) == sorted(exp["text"].splitlines())


@pytest.mark.parametrize("tag,pkw", [("filter", dict(no_filter=0)), ("nomulti", dict(multi=0)),
                                     ("dmax", dict(dist_max=0.05)), ("th2", dict(hdist_th=2))])
def test_report_modes_match_golden(capi, toy, toy_reads, tag, pkw):
    hx, dx, ox = toy
    names, bases, offs = toy_reads
    exp = json.load(open(os.path.join(GOLDEN, "toy_expected.json")))[tag]
    st, res = gpu_dist(capi, dx, bases, offs, 0, **pkw)
    want = sorted((r, s, float.fromhex(d)) for r, s, d in exp["rows"] if s != 0)
    assert_rows_close(res.rows(), want)
    assert sorted(st.format_dist(hx, names).splitlines()) == sorted(exp["text"].splitlines())


def test_hits_bit_exact_vs_oracle(capi, po, toy, toy_genomes, synth):
    hx, dx, ox = toy
    bases, offs, names = synth.sample_reads(toy_genomes, 3000, seed=11)
    ref = ox.dist(bases, offs, names, po.params(collect=7))
    st, res = gpu_dist(capi, dx, bases, offs, capi.KR_TAP_ACCS | capi.KR_TAP_HITS)
    gh, rh = st.hits(), ref["hits"]
    key = lambda h: sorted(zip(h["read"].tolist(), h["strand"].tolist(), h["kpos"].tolist(), h["cmer_index"].tolist(),
                               h["hd"].tolist(), h["se"].tolist()))
    assert key(gh) == key(rh)
    acc = ref["accs"][ref["accs"]["passed"] == 1]
    want = sorted(zip(acc["read"].tolist(), ((acc["se"] << 1) | acc["strand"]).tolist(), [tuple(x[:5]) for x in acc["hist"].tolist()]))
    got = sorted(zip(res.rec_read.tolist(), res.rec_key.tolist(), [tuple(x) for x in res.rec_hist.tolist()]))
    assert got == want
    assert_rows_close(res.rows(), rows_of_oracle(ref))
    assert sorted(st.format_dist(hx, names).splitlines()) == sorted(ref["text"].splitlines())
    # determinism: a second run returns identical bytes per read
    st2, res2 = gpu_dist(capi, dx, bases, offs, capi.KR_TAP_ACCS)
    a = sorted(zip(res.rec_read.tolist(), res.rec_key.tolist(), res.rec_d.tolist(), res.rec_sel.tolist()))
    b = sorted(zip(res2.rec_read.tolist(), res2.rec_key.tolist(), res2.rec_d.tolist(), res2.rec_sel.tolist()))
    assert a == b


@pytest.mark.parametrize("k,w,h,m,r,frac", [(27, 35, 11, 4, 1, True), (29, 35, 13, 64, 3, False), (19, 19, 3, 2, 0, True),
                                            (31, 40, 15, 8, 5, True)])
def test_other_index_shapes(capi, po, synth, tmp_path, k, w, h, m, r, frac):
    nwk_path = os.path.join(GOLDEN, "tree_toy.nwk")
    g = synth.evolve_genomes(open(nwk_path).read(), 30000, seed=5)
    tsv = synth.write_genomes(g, str(tmp_path / "g"))
    idx = str(tmp_path / "ix")
    capi.build_index(tsv, idx, nwk=nwk_path, k=k, w=w, h=h, m=m, r=r, frac=frac, num_threads=4)
    hx = capi.HostIndex(idx)
    dx = hx.upload(0)
    ox = po.Index(idx)
    bases, offs, names = synth.sample_reads(g, 1500, seed=2)
    ref = ox.dist(bases, offs, names, po.params(collect=7))
    st, res = gpu_dist(capi, dx, bases, offs, capi.KR_TAP_ACCS | capi.KR_TAP_HITS)
    assert len(st.hits()) == len(ref["hits"])
    assert res.read_onmers.tolist() == ref["reads"]["onmers"].tolist()
    assert_rows_close(res.rows(), rows_of_oracle(ref))
    assert sorted(st.format_dist(hx, names).splitlines()) == sorted(ref["text"].splitlines())


def test_reflist_only_index_and_two_libraries(capi, po, synth, tmp_path):
    g = synth.evolve_genomes("((a:0.02,b:0.02):0.02,(c:0.03,(d:0.01,e:0.01):0.02):0.01);", 20000, seed=9)
    tsv = synth.write_genomes(g, str(tmp_path / "g"))
    idx = str(tmp_path / "ix")
    pp = [20, 19, 17, 13, 6, 4, 2]
    # two partial libraries (no_frac r=0 and r=2), no tree: reflist + generated balanced tree
    capi.build_index(tsv, idx, k=21, w=27, h=7, m=4, r=0, frac=False, ppos=pp)
    capi.build_index(tsv, idx, k=21, w=27, h=7, m=4, r=2, frac=False, ppos=pp)
    hx = capi.HostIndex(idx)
    assert hx.view.nlibs == 2 and hx.view.wbackbone == 0
    dx = hx.upload(0)
    ox = po.Index(idx)
    bases, offs, names = synth.sample_reads(g, 1000, seed=4)
    ref = ox.dist(bases, offs, names, po.params(collect=7))
    st, res = gpu_dist(capi, dx, bases, offs, capi.KR_TAP_HITS)
    gh, rh = st.hits(), ref["hits"]
    key = lambda h_: sorted(zip(h_["read"].tolist(), h_["strand"].tolist(), h_["kpos"].tolist(), h_["lib"].tolist(),
                                h_["cmer_index"].tolist(), h_["hd"].tolist()))
    assert key(gh) == key(rh)
    assert_rows_close(res.rows(), rows_of_oracle(ref))
    assert sorted(st.format_dist(hx, names).splitlines()) == sorted(ref["text"].splitlines())


def test_crafted_min_rule_null_nodes_and_th(capi, po, tmp_path):
    """Crafted table: two entries in one bucket reaching the same leaf with different hd (the
    per-position minimum must win), an entry whose colour is the empty set, hd above the
    threshold, and a colour id beyond nsubsets."""
    PPOS = [20, 19, 17, 13, 6, 4, 2]
    NPOS = [p for p in range(21) if p not in PPOS]
    rng = np.random.default_rng(17)
    while True:
        s = "".join("ACGT"[i] for i in rng.integers(0, 4, 60))
        rows, ok = {}, True
        for i in range(0, 40):
            km = s[i:i + 21]
            f = closed_form(km, PPOS, NPOS)
            row = row_of(f[2], 4, 1, True)
            if row is None:
                continue
            ents = rows.setdefault(row, [])
            ents.append((f[3], 6))                       # hd 0 -> {x,z}
            ents.append((f[3] ^ (1 << 2), 8))            # hd 1 -> {x,z,y}
            ents.append((f[3] ^ 0b11, 0))                # hd 2 -> empty colour
            ents.append((f[3] ^ 0b11111100000, 1))       # hd 6 > th
            ents.append((f[3] ^ (1 << 5), 99))           # hd 1, colour id out of range -> dropped
        if all(len({e for e, _ in v}) == len(v) for v in rows.values()):
            break
    pse = [(0, 0), (0, 1), (0, 2), (1, 2), (0, 4), (3, 4), (1, 4), (2, 0), (6, 2)]
    rho = [0.0, 0.2, 0.25, 0.0, 0.3, 0.0]
    d = str(tmp_path / "ix")
    write_index(d, 21, 7, 4, 1, True, PPOS, rows, pse, rho, nwk="((x:0.1,y:0.1)n1:0.1,z:0.2);")
    hx = capi.HostIndex(d)
    dx = hx.upload(0)
    ox = po.Index(d)
    reads = [s, revcomp(s), s[:21], s[5:50].lower(), s[:30] + "N" + s[31:]]
    bases = np.frombuffer("".join(reads).encode(), np.uint8)
    offs = np.cumsum([0] + [len(x) for x in reads]).astype(np.uint64)
    for th in (0, 1, 4, 6, 16):
        ref = ox.dist(bases, offs, None, po.params(collect=7, hdist_th=th))
        st, res = gpu_dist(capi, dx, bases, offs, capi.KR_TAP_ACCS | capi.KR_TAP_HITS, hdist_th=th)
        assert len(st.hits()) == len(ref["hits"]) > 0
        acc = ref["accs"][ref["accs"]["passed"] == 1]
        want = sorted(zip(acc["read"].tolist(), ((acc["se"] << 1) | acc["strand"]).tolist(), [tuple(x[:th + 1]) for x in acc["hist"].tolist()]))
        got = sorted(zip(res.rec_read.tolist(), res.rec_key.tolist(), [tuple(x) for x in res.rec_hist.tolist()]))
        assert got == want, th
        assert_rows_close(res.rows(), rows_of_oracle(ref))
        assert st.readtaps(len(reads)).tolist() == ref["reads"]["hdist_filt"].tolist()


def test_long_reads_and_ragged_batches(capi, po, toy, toy_genomes):
    """Multi-segment reads (> 128 k-mer positions), empty reads, reads shorter than k."""
    hx, dx, ox = toy
    g = toy_genomes
    rng = np.random.default_rng(23)
    seqs = []
    for L in (0, 1, 20, 21, 22, 148, 149, 150, 151, 170, 300, 1000, 5000, 19999):
        name = list(g)[int(rng.integers(0, 25))]
        o = int(rng.integers(0, 20000 - L + 1))
        s = bytearray(g[name][o:o + L].tobytes())
        for _ in range(L // 50):  # a few substitutions and the odd N
            s[int(rng.integers(0, L))] = b"ACGTN"[int(rng.integers(0, 5))]
        seqs.append(bytes(s))
    bases = np.frombuffer(b"".join(seqs), np.uint8)
    offs = np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint64)
    names = [f"L{len(s)}" for s in seqs]
    ref = ox.dist(bases, offs, names, po.params(collect=7))
    st, res = gpu_dist(capi, dx, bases, offs, capi.KR_TAP_ACCS | capi.KR_TAP_HITS)
    assert res.read_onmers.tolist() == ref["reads"]["onmers"].tolist()
    gh, rh = st.hits(), ref["hits"]
    key = lambda h_: sorted(zip(h_["read"].tolist(), h_["strand"].tolist(), h_["kpos"].tolist(), h_["cmer_index"].tolist(), h_["hd"].tolist()))
    assert key(gh) == key(rh)
    acc = ref["accs"][ref["accs"]["passed"] == 1]
    want = sorted(zip(acc["read"].tolist(), ((acc["se"] << 1) | acc["strand"]).tolist(), [tuple(x[:5]) for x in acc["hist"].tolist()]))
    got = sorted(zip(res.rec_read.tolist(), res.rec_key.tolist(), [tuple(x) for x in res.rec_hist.tolist()]))
    assert got == want
    assert_rows_close(res.rows(), rows_of_oracle(ref))
    assert sorted(st.format_dist(hx, names).splitlines()) == sorted(ref["text"].splitlines())
    # without KR_TAP_ACCS a record is its key and the packed problem word only (no histogram planes, no read index): the
    # likelihood and selection kernels decode the word -- default, --filter (chi-square from the closest leaf's word),
    # --no-multi, another threshold (no word: planes are written), and the same through two lanes
    for pkw, okw in ((dict(), dict()), (dict(no_filter=0), dict(no_filter=0)), (dict(multi=0), dict(multi=0)), (dict(hdist_th=3), dict(hdist_th=3))):
        want_rows = rows_of_oracle(ox.dist(bases, offs, names, po.params(collect=0, **okw)))
        _, r0 = gpu_dist(capi, dx, bases, offs, 0, **pkw)
        assert_rows_close(r0.rows(), want_rows)
        assert r0.rec_hist is None
    # (the two longest sequences are submitted as tiles above: tests/test_gpu_long_sequences.py; a tiled batch is one lane)
    os.environ["KR_LANE_MIN_READS"], os.environ["KR_LANES"], os.environ["KR_NO_TILES"] = "4", "2", "1"
    try:
        stl, rl = gpu_dist(capi, dx, bases, offs, 0, no_filter=0)
        assert stl.timing().lanes == 2
        assert_rows_close(rl.rows(), rows_of_oracle(ox.dist(bases, offs, names, po.params(collect=0, no_filter=0))))
    finally:
        os.environ.pop("KR_LANE_MIN_READS"), os.environ.pop("KR_LANES"), os.environ.pop("KR_NO_TILES")


@pytest.mark.parametrize("slot_log2w,dbg", [("0", "8192"), ("5", "8192"), ("6", "8192"), ("6", "0"), ("8", "0")])
def test_overflow_path_many_leaves(capi, po, synth, tmp_path, monkeypatch, slot_log2w, dbg):
    """150-bp reads with k = 21 have 130 k-mer positions = two segments.  With debug bit 8192 they take the
    plane tables, and reads that reach more (strand, leaf) pairs than the LDS table holds go on to the
    global-memory accumulators; without it every segment runs in event mode and the segments are merged through
    the global count table.  Results must not change.  The table is dense (18 entries per bucket): scanned through
    the packed arrays ("0"), 128-byte slots (30 entries: some buckets continue in the packed array) and 256-byte
    slots."""
    monkeypatch.setenv("KR_SLOT_LOG2W", slot_log2w)
    monkeypatch.setenv("KR_DEBUG_SKIP", dbg)
    n = 96
    names = [f"s{i}" for i in range(n)]
    # star-ish tree of close relatives: every read matches nearly every genome
    nwk = "(" + ",".join(f"{x}:0.004" for x in names) + ");"
    g = synth.evolve_genomes(nwk, 6000, seed=13)
    tsv = synth.write_genomes(g, str(tmp_path / "g"))
    (tmp_path / "t.nwk").write_text(nwk)
    idx = str(tmp_path / "ix")
    capi.build_index(tsv, idx, nwk=str(tmp_path / "t.nwk"), k=21, w=27, h=7, m=4, r=1, frac=True, num_threads=4,
                     ppos=[20, 19, 17, 13, 6, 4, 2])
    hx = capi.HostIndex(idx)
    dx = hx.upload(0)
    ox = po.Index(idx)
    bases, offs, rn = synth.sample_reads(g, 400, seed=6)
    ref = ox.dist(bases, offs, rn, po.params(collect=7))
    st = dx.stream(max_reads=400, max_bases=len(bases), max_records=400 * 2 * n)
    st.submit(bases, offs, capi.KR_TAP_ACCS)
    res = st.collect()
    assert (st.timing().overflow_reads > 50) == (dbg == "8192")
    acc = ref["accs"][ref["accs"]["passed"] == 1]
    want = sorted(zip(acc["read"].tolist(), ((acc["se"] << 1) | acc["strand"]).tolist(), [tuple(x[:5]) for x in acc["hist"].tolist()]))
    got = sorted(zip(res.rec_read.tolist(), res.rec_key.tolist(), [tuple(x) for x in res.rec_hist.tolist()]))
    assert got == want
    assert_rows_close(res.rows(), rows_of_oracle(ref))
    # default record capacity is too small here: the library must say so, not truncate
    st2 = dx.stream(max_reads=400, max_bases=len(bases), max_records=1000)
    st2.submit(bases, offs)
    with pytest.raises(capi.KrError) as e:
        st2.collect()
    assert e.value.code == capi.KR_ERR_CAPACITY
    # ... and keeps saying so: a second collect / wait of the same batch is the same error, never truncated records
    for again in (st2.collect, st2.wait, st2.collect_device):
        with pytest.raises(capi.KrError) as e:
            again()
        assert e.value.code == capi.KR_ERR_CAPACITY
    # a submit that is rejected for its arguments leaves the stream's previous batch untouched
    st3 = dx.stream(max_reads=400, max_bases=len(bases), max_records=400 * 2 * n)
    st3.submit(bases, offs)
    first = st3.collect().rows()
    with pytest.raises(capi.KrError) as e:
        st3.submit(np.concatenate([bases, bases]), np.concatenate([offs, offs[1:] + offs[-1]])[: len(offs)] * 2)  # > max_bases
    assert e.value.code == capi.KR_ERR_ARG
    assert st3.collect().rows() == first


@pytest.mark.parametrize("dbg,n,length", [("0", 160, 150), ("2048", 160, 150), ("8", 160, 150), ("0", 160, 250), ("2048", 160, 250),
                                          ("0", 1200, 200)])
def test_single_segment_many_leaves_spill_paths(capi, po, synth, tmp_path, monkeypatch, dbg, n, length):
    """150-bp reads (one 128-position segment: event mode) against 160 close relatives: thousands of events and
    hundreds of (leaf, strand) keys per read -> events spill to global scratch, the epilogue runs several plane
    batches (or, forced by the debug bit, the single batch in global scratch; or, with event mode off, the
    level-1/level-2 plane tables), the passing-key table spills.  Histograms must be bit-exact in every mode.
    250-bp reads (222 positions = two segments) take the two-segment instantiation of the single-segment layout
    (8 position bits per event, 256-bit planes, 16-bit counters) through the same spill paths; against 1,200 relatives a
    200-bp read has 1,200 keys and ~50,000 events: events, key table and passing-key table all live in the wave's global scratch."""
    names = [f"s{i}" for i in range(n)]
    nwk = "(" + ",".join(f"{x}:0.003" for x in names) + ");"
    g = synth.evolve_genomes(nwk, 5000 if n == 160 else 2500, seed=17)
    tsv = synth.write_genomes(g, str(tmp_path / "g"))
    (tmp_path / "t.nwk").write_text(nwk)
    idx = str(tmp_path / "ix")
    capi.build_index(tsv, idx, nwk=str(tmp_path / "t.nwk"), k=29, w=31, h=13, m=2, r=0, frac=True, num_threads=4)
    hx = capi.HostIndex(idx)
    dx = hx.upload(0)
    ox = po.Index(idx)
    nreads = 300 if n == 160 else 120
    bases, offs, rn = synth.sample_reads(g, nreads, seed=9, length=length)
    ref = ox.dist(bases, offs, rn, po.params(collect=7))
    monkeypatch.setenv("KR_DEBUG_SKIP", dbg)
    st = dx.stream(max_reads=nreads, max_bases=len(bases), max_records=nreads * 2 * n)
    st.submit(bases, offs, capi.KR_TAP_ACCS)
    res = st.collect()
    acc = ref["accs"][ref["accs"]["passed"] == 1]
    assert len(acc) > nreads * (100 if n == 160 else 1000)  # the point of the test: many keys per read
    want = sorted(zip(acc["read"].tolist(), ((acc["se"] << 1) | acc["strand"]).tolist(), [tuple(x[:5]) for x in acc["hist"].tolist()]))
    got = sorted(zip(res.rec_read.tolist(), res.rec_key.tolist(), [tuple(x) for x in res.rec_hist.tolist()]))
    assert got == want
    assert_rows_close(res.rows(), rows_of_oracle(ref))


@pytest.mark.parametrize("n", [160, 330])
def test_report_modes_on_reads_of_many_records(capi, po, synth, tmp_path, n):
    """The selection (src/query.cpp:96-139,158-196) of reads with MANY records, in every report mode.  On the benchmark indexes one read
    in eight has more than 64 records and those hold two thirds of all records; the select kernel does them by the whole wave
    (select_big_read: up to 256 records in registers, more in two passes through memory), and the golden report-mode tests run on the
    25-reference index, where no read has more than 50.  160 / 330 close relatives: a read has ~n records (one strand of nearly every
    leaf) up to 2 n.  Default, --no-multi, --dist-max, both, --filter; plain batches and rows-only batches with DIST as an index."""
    names = [f"s{i}" for i in range(n)]
    nwk = "(" + ",".join(f"{x}:0.003" for x in names) + ");"
    g = synth.evolve_genomes(nwk, 4000 if n == 160 else 2500, seed=23)
    tsv = synth.write_genomes(g, str(tmp_path / "g"))
    (tmp_path / "t.nwk").write_text(nwk)
    idx = str(tmp_path / "ix")
    capi.build_index(tsv, idx, nwk=str(tmp_path / "t.nwk"), k=29, w=31, h=13, m=2, r=0, frac=True, num_threads=4)
    hx = capi.HostIndex(idx)
    dx = hx.upload(0)
    ox = po.Index(idx)
    nreads = 260
    bases, offs, rn = synth.sample_reads(g, nreads, seed=4)
    seen_mid = seen_big = False
    for pkw in (dict(), dict(multi=0), dict(dist_max=0.005), dict(multi=0, dist_max=0.005), dict(no_filter=0)):
        want = rows_of_oracle(ox.dist(bases, offs, rn, po.params(collect=0, **pkw)))
        for flags in (0, capi.KR_ROWS_ONLY | capi.KR_ROWS_INDEXED):
            if flags and "no_filter" in pkw:
                continue  # (rows as indices are for batches without --filter)
            st = dx.stream(params=capi.default_params(**pkw), max_reads=nreads, max_bases=len(bases), max_records=nreads * 2 * n)
            st.submit(bases, offs, flags)
            res = st.collect()
            got = res.rows() if not flags else sorted(zip(res.rec_read.tolist(), (res.rec_key >> 1).tolist(), res.rec_d.tolist()))  # (a rows-only batch holds rows only)
            assert_rows_close(got, want)
            if not flags:
                cnt = res.read_cnt
                seen_mid |= bool(((cnt > 64) & (cnt <= 256)).any())
                seen_big |= bool((cnt > 256).any())
            st.close()
    assert len(want) > 0
    assert seen_mid if n == 160 else seen_big  # the point of the test


def test_many_leaves_bitmap_spans_several_tiles(capi, po, synth, tmp_path):
    """2,300 references: 4,600 (leaf, strand) keys = 72 bitmap blocks, more than one 64-lane tile of the ordinal
    prefix, and level-2 / merge lists longer than a wave (the 10k-genome configuration in the small)."""
    n = 2300
    nwk = synth.yule_newick(n, 5)
    g = synth.evolve_genomes(nwk, 1500, seed=31)
    tsv = synth.write_genomes(g, str(tmp_path / "g"))
    (tmp_path / "t.nwk").write_text(nwk)
    idx = str(tmp_path / "ix")
    capi.build_index(tsv, idx, nwk=str(tmp_path / "t.nwk"), k=29, w=33, h=13, m=2, r=0, frac=True, num_threads=8)
    hx = capi.HostIndex(idx)
    dx = hx.upload(0)
    ox = po.Index(idx)
    for length, seed in ((150, 3), (260, 4)):  # one segment (event mode), two segments (merged through the count table)
        bases, offs, rn = synth.sample_reads(g, 400, seed=seed, length=length)
        ref = ox.dist(bases, offs, rn, po.params(collect=7))
        st = dx.stream(max_reads=400, max_bases=len(bases), max_records=400 * 2 * n)
        st.submit(bases, offs, capi.KR_TAP_ACCS)
        res = st.collect()
        acc = ref["accs"][ref["accs"]["passed"] == 1]
        want = sorted(zip(acc["read"].tolist(), ((acc["se"] << 1) | acc["strand"]).tolist(), [tuple(x[:5]) for x in acc["hist"].tolist()]))
        got = sorted(zip(res.rec_read.tolist(), res.rec_key.tolist(), [tuple(x) for x in res.rec_hist.tolist()]))
        assert got == want
        assert_rows_close(res.rows(), rows_of_oracle(ref))


def test_forty_thousand_leaves(capi, po, synth, tmp_path):
    """40,000 references (the reference's larger public indexes have 16-50 thousand): 80,000 key bits = 10 KB of LDS
    bitmap per wave, fewer resident accumulate waves, per-wave scratch bounded by running fewer waves."""
    n = 40000
    nwk = synth.yule_newick(n, 5)
    g = synth.evolve_genomes(nwk, 400, seed=31)
    tsv = synth.write_genomes(g, str(tmp_path / "g"))
    (tmp_path / "t.nwk").write_text(nwk)
    idx = str(tmp_path / "ix")
    capi.build_index(tsv, idx, nwk=str(tmp_path / "t.nwk"), k=29, w=33, h=13, m=2, r=0, frac=True, num_threads=8)
    hx = capi.HostIndex(idx)
    dx = hx.upload(0)
    ox = po.Index(idx)
    for length, seed in ((150, 3), (300, 4)):
        bases, offs, rn = synth.sample_reads(g, 600, seed=seed, length=length)
        ref = ox.dist(bases, offs, rn, po.params(collect=7, num_threads=8))
        acc = ref["accs"][ref["accs"]["passed"] == 1]
        want = sorted(zip(acc["read"].tolist(), ((acc["se"] << 1) | acc["strand"]).tolist(), [tuple(x[:5]) for x in acc["hist"].tolist()]))
        # default path, then the plane-table fallbacks forced (their per-wave tables are GBs at this tree size:
        # the very first batch of a fresh stream must already see them cleared)
        for dbg in ("0", "8", "8192"):
            os.environ["KR_DEBUG_SKIP"] = dbg
            try:
                st = dx.stream(max_reads=600, max_bases=len(bases), max_records=600 * 2048)
                st.submit(bases, offs, capi.KR_TAP_ACCS)
                res = st.collect()
            finally:
                del os.environ["KR_DEBUG_SKIP"]
            got = sorted(zip(res.rec_read.tolist(), res.rec_key.tolist(), [tuple(x) for x in res.rec_hist.tolist()]))
            assert got == want and len(got) > 2000, (length, dbg)
            assert_rows_close(res.rows(), rows_of_oracle(ref))
            st.close()
    # one leaf too many for the LDS bitmaps is refused at upload, not mis-handled
    assert hx.view.tree_nnodes < 2 * 65536


def test_large_clade_colours_spill_the_work_stack(capi, po, synth, tmp_path):
    """3,000 nearly identical genomes: almost every k-mer carries the colour of a clade of hundreds to thousands of
    leaves.  Walking such a colour through its parts fans out faster than the 64-wide pops consume it; the work stack
    (192 / 256 entries of LDS) moves its older half to global memory instead of giving up (before: KR_ERR_CAPACITY).  Since
    round 2 a clade is a run of leaf ranks (a "flat" colour) and is not walked at all: both forms of the index are checked."""
    n = 3000
    nwk = synth.yule_newick(n, 9, mean_blen=0.0003)
    g = synth.evolve_genomes(nwk, 500, seed=5)
    tsv = synth.write_genomes(g, str(tmp_path / "g"))
    (tmp_path / "t.nwk").write_text(nwk)
    idx = str(tmp_path / "ix")
    capi.build_index(tsv, idx, nwk=str(tmp_path / "t.nwk"), k=29, w=33, h=13, m=2, r=0, frac=True, num_threads=8)
    hx = capi.HostIndex(idx)
    dx = hx.upload(0)  # a clade is a "flat" colour (a run of leaf ranks): one gather, no walk, no fan-out
    os.environ["KR_FLAT_MAX"] = "0"  # ... and the same index with every colour walked through its parts: the stack spills
    try:
        dx_walk = hx.upload(0)
        os.environ["KR_FLAT_MAX"] = "4"  # ... and with lists of at most 4 leaves: walked colours whose parts are runs and lists
        dx_mix = hx.upload(0)
    finally:
        del os.environ["KR_FLAT_MAX"]
    assert dx.device_bytes > dx_mix.device_bytes > dx_walk.device_bytes  # (the leaf lists)
    ox = po.Index(idx)
    spills = 0
    for dx, length, dbg, nreads in ((dx, 150, "0", 120), (dx, 150, "8", 40), (dx, 300, "0", 40), (dx, 300, "8192", 40),
                                    (dx_walk, 150, "0", 120), (dx_walk, 150, "8", 40), (dx_walk, 300, "0", 40), (dx_walk, 300, "8192", 40),
                                    (dx_mix, 150, "0", 120), (dx_mix, 300, "0", 40)):
        bases, offs, rn = synth.sample_reads(g, nreads, seed=3 + length, length=length)
        ref = ox.dist(bases, offs, rn, po.params(collect=7, num_threads=8))
        acc = ref["accs"][ref["accs"]["passed"] == 1]
        want = sorted(zip(acc["read"].tolist(), ((acc["se"] << 1) | acc["strand"]).tolist(), [tuple(x[:5]) for x in acc["hist"].tolist()]))
        assert len(want) > nreads * 800  # most reads reach most of the tree
        os.environ["KR_DEBUG_SKIP"] = dbg
        try:
            st = dx.stream(max_reads=nreads, max_bases=len(bases), max_records=nreads * 2 * n + 4096)
            st.submit(bases, offs, capi.KR_TAP_ACCS)
            res = st.collect()
        finally:
            del os.environ["KR_DEBUG_SKIP"]
        got = sorted(zip(res.rec_read.tolist(), res.rec_key.tolist(), [tuple(x) for x in res.rec_hist.tolist()]))
        assert got == want, (length, dbg)
        assert_rows_close(res.rows(), rows_of_oracle(ref))
        if dx is dx_walk:
            spills += st.timing().stack_spills
        st.close()
    assert spills > 0


@pytest.mark.parametrize("cfg", [(19, 24, 3, 1, 0, True), (21, 21, 7, 2, 1, False), (24, 31, 8, 3, 1, True), (26, 32, 10, 4, 3, True),
                                 (29, 35, 13, 4, 0, False), (30, 33, 14, 7, 2, True), (31, 38, 15, 4, 1, True)])
def test_configuration_sweep(capi, po, synth, tmp_path, cfg):
    """k, w, h, m, r, frac x hdist_th x read length: index built by the CPU builder, rows byte-identical to the oracle's
    (scripts/sweep_configs.py runs the full grid: 42 index configurations x 5 query settings)."""
    k, w, h, m, r, frac = cfg
    nwk = "((a:0.02,b:0.02):0.02,(c:0.03,(d:0.01,e:0.01):0.02):0.01,(f:0.05,g:0.002):0.01);"
    g = synth.evolve_genomes(nwk, 20000, seed=9)
    tsv = synth.write_genomes(g, str(tmp_path / "g"))
    (tmp_path / "t.nwk").write_text(nwk)
    idx = str(tmp_path / "ix")
    capi.build_index(tsv, idx, nwk=str(tmp_path / "t.nwk"), k=k, w=w, h=h, m=m, r=r, frac=frac, num_threads=8, seed=k)
    hx = capi.HostIndex(idx)
    dx = hx.upload(0)
    ox = po.Index(idx)
    for th, L, nreads in ((4, 150, 200), (0, 90, 200), (6, 151, 200), (9, 260, 100)):
        if h == 3:
            nreads = 20  # 64 LSH rows: thousands of table hits per read
        bases, offs, _ = synth.sample_reads(g, nreads, seed=th + L, length=L)
        rn = [f"r{i}" for i in range(nreads)]
        ref = ox.dist(bases, offs, rn, po.params(hdist_th=th, collect=4, num_threads=8))
        st = dx.stream(params=capi.default_params(hdist_th=th), max_reads=nreads, max_bases=len(bases), max_records=nreads * 64)
        st.submit(bases, offs)
        st.collect()
        assert st.format_dist(hx, rn) == ref["text"], (cfg, th, L)
        st.close()


def test_device_brent_vs_oracle(capi, po, toy):
    hx, dx, ox = toy
    rng = np.random.default_rng(31)
    n = 20000
    hist = np.floor(rng.random((n, 5)) ** 3 * rng.integers(1, 125, (n, 1))).astype(np.uint32)
    hist[hist.sum(1) == 0, 0] = 1
    tot = hist.sum(1)
    onm = (tot + rng.integers(0, 60, n)).astype(np.uint32)
    rho = rng.uniform(0.02, 0.6, n)
    d, v = dx.brent(4, hist, onm, rho)
    bad = 0
    for i in range(n):
        od, ov, _ = po.brent(21, 7, 4, hist[i].astype(float), float(onm[i]) - float(tot[i]), float(rho[i]))
        assert abs(d[i] - od) <= 1e-6 * od, (i, d[i], od)
        bad += d[i] != od
    # identical Brent trajectories: differences are only pow/log rounding (~1e-11 relative)
    assert np.all(np.isfinite(v))


def test_large_batch_properties(capi, toy, toy_genomes, synth):
    """Size-independent properties on a batch far larger than the oracle is run on:
    (1) reverse-complementing every read leaves the (read, leaf, DIST) rows unchanged;
    (2) permuting the reads permutes the results; (3) splitting the batch changes nothing."""
    hx, dx, ox = toy
    bases, offs, _ = synth.sample_reads(toy_genomes, 200_000, seed=77)
    n = len(offs) - 1
    st, res = gpu_dist(capi, dx, bases, offs)
    rows = res.rows()
    assert len(rows) > n  # multi-hit reads dominate
    rc = synth.COMP[bases.reshape(n, 150)[:, ::-1]].reshape(-1)
    _, res_rc = gpu_dist(capi, dx, rc, offs)
    assert res_rc.rows() == rows
    perm = np.random.default_rng(1).permutation(n)
    pb = bases.reshape(n, 150)[perm].reshape(-1)
    _, res_p = gpu_dist(capi, dx, pb, offs)
    inv = {int(new): int(old) for new, old in enumerate(perm)}
    assert sorted((inv[r], s, d) for r, s, d in res_p.rows()) == rows
    half = n // 2
    _, r1 = gpu_dist(capi, dx, bases[: half * 150], offs[: half + 1])
    _, r2 = gpu_dist(capi, dx, bases[half * 150:], offs[half:] - offs[half])
    assert r1.rows() + [(r + half, s, d) for r, s, d in r2.rows()] == rows


def test_where_a_streams_buffers_lie_does_not_change_results(capi, toy, toy_genomes, synth, monkeypatch):
    """kr_debug_stream_move (the experiments on the scan's launch-time levels) gives one group of a stream's
    device buffers a new address, or its kernels a new HIP stream, between batches: same rows every time.  And nothing depends
    on what a fresh device buffer holds (KR_DEBUG_POISON fills every buffer of a new stream with 0xA5)."""
    hx, dx, ox = toy
    bases, offs, _ = synth.sample_reads(toy_genomes, 30_000, seed=78)
    st, res = gpu_dist(capi, dx, bases, offs)
    rows = res.rows()
    a0 = st.debug_addrs()
    for which in (0, 1, 2, 3, 4):
        st.debug_move(which)
        st.submit(bases, offs)
        assert st.collect().rows() == rows, which
    a1 = st.debug_addrs()
    assert all(a1[k] != a0[k] for k in ("items", "counters", "cursors", "rd_off", "rec_key", "dd"))
    with pytest.raises(capi.KrError):
        st.debug_move(5)
    st.close()
    monkeypatch.setenv("KR_DEBUG_POISON", "all")
    for flags in (0, capi.KR_TAP_ACCS):
        st2, res2 = gpu_dist(capi, dx, bases, offs, flags)
        assert res2.rows() == rows
        st2.close()


def test_indexed_rows_are_the_rows(capi, toy, toy_genomes, synth, monkeypatch):
    """KR_ROWS_INDEXED: a rows-only batch whose DIST column crosses PCIe as an index into the batch's distinct likelihood
    problems (8 bytes a row instead of 12, the distinct values once): the rows are bit for bit those of a plain rows-only
    batch, the host formatter prints the same text from them, fewer bytes come back; a batch that runs as several lanes (or
    filters) ignores the hint and returns rec_d as before (src/query.cpp:158-196 is what the rows are for)."""
    hx, dx, ox = toy
    bases, offs, names = synth.sample_reads(toy_genomes, 5003, seed=21)
    n = len(offs) - 1
    monkeypatch.setenv("KR_LANES", "1")
    st = dx.stream(max_reads=n, max_bases=len(bases))
    monkeypatch.delenv("KR_LANES")
    st.submit(bases, offs, capi.KR_ROWS_ONLY)
    plain = st.collect()
    text = st.format_dist(hx, names)
    b12 = st.last_d2h_bytes()
    rows = lambda r: sorted(zip(r.rec_read.tolist(), r.rec_key.tolist(), r.rec_d.view(np.uint64).tolist()))
    assert plain.rec_dix is None and plain.nrows == len(plain.rec_key) > n
    for _ in range(2):  # (twice: the page-locked arrays of the first indexed batch are reused by the second)
        st.submit(bases, offs, capi.KR_ROWS_ONLY | capi.KR_ROWS_INDEXED)
        ix = st.collect()
        # (the list is the extent of list positions the batch handed out, holes included: on a batch this small -- nearly every
        #  problem distinct -- it is as long as the rows; tests/test_gpu_syn1000.py has the batch where it is a twentieth)
        assert ix.rec_dix is not None and int(ix.rec_dix.max()) < len(ix.dist_list)
        assert rows(ix) == rows(plain) and ix.read_na.tolist() == plain.read_na.tolist()
        assert st.format_dist(hx, names) == text
        assert st.last_d2h_bytes() == 9 * n + 8 * len(ix.rec_key) + 8 * len(ix.dist_list) and b12 == 9 * n + 12 * len(ix.rec_key)
    # a plain batch after an indexed one on the same stream
    st.submit(bases, offs, capi.KR_ROWS_ONLY)
    again = st.collect()
    assert again.rec_dix is None and rows(again) == rows(plain)
    st.close()
    # several lanes: the hint is ignored
    monkeypatch.setenv("KR_LANE_MIN_READS", "500")
    monkeypatch.setenv("KR_LANES", "3")
    st3 = dx.stream(max_reads=n, max_bases=len(bases))
    st3.submit(bases, offs, capi.KR_ROWS_ONLY | capi.KR_ROWS_INDEXED)
    r3 = st3.collect()
    assert st3.timing().lanes == 3 and r3.rec_dix is None and rows(r3) == rows(plain)
    st3.close()
    # --filter: the hint is ignored (the select kernel writes the objective values the list would alias)
    pf = capi.default_params()
    pf.no_filter = 0
    monkeypatch.setenv("KR_LANES", "1")
    stf = dx.stream(pf, max_reads=n, max_bases=len(bases))
    stf.submit(bases, offs, capi.KR_ROWS_ONLY)
    want = stf.collect()
    stf.submit(bases, offs, capi.KR_ROWS_ONLY | capi.KR_ROWS_INDEXED)
    got = stf.collect()
    assert got.rec_dix is None and rows(got) == rows(want)
    stf.close()


def test_lanes_do_not_change_results(capi, toy, toy_genomes, synth, monkeypatch):
    """A batch cut into several lanes (own HIP stream, own H2D/D2H copies, slices of the result arrays) returns what one
    lane returns: records, histograms, rows, report text; host view, device view, KR_ROWS_ONLY and KR_BASES_PINNED."""
    import ctypes as C
    hx, dx, ox = toy
    bases, offs, names = synth.sample_reads(toy_genomes, 3001, seed=13)
    st1, one = gpu_dist(capi, dx, bases, offs, capi.KR_TAP_ACCS)
    assert st1.timing().lanes == 1
    text1 = st1.format_dist(hx, names)
    # (record slots are handed out to waves in chunks: the ORDER of the reads' record groups differs from run to run)
    key = lambda r: sorted(zip(r.rec_read.tolist(), r.rec_key.tolist(), r.rec_sel.tolist(), r.rec_d.tolist(),
                               r.rec_v.tolist()))
    hist = lambda r: sorted(zip(r.rec_read.tolist(), r.rec_key.tolist(), [tuple(x) for x in r.rec_hist.tolist()]))
    monkeypatch.setenv("KR_LANE_MIN_READS", "500")
    for lanes in ("2", "3", "5"):
        monkeypatch.setenv("KR_LANES", lanes)
        st, res = gpu_dist(capi, dx, bases, offs, capi.KR_TAP_ACCS)
        assert st.timing().lanes == int(lanes)
        assert key(res) == key(one) and hist(res) == hist(one)
        assert res.read_onmers.tolist() == one.read_onmers.tolist() and res.read_na.tolist() == one.read_na.tolist()
        assert st.format_dist(hx, names) == text1
        # the pipelined collect (no histograms: results leave lane by lane), twice in a row on one stream
        for _ in range(2):
            st.submit(bases, offs)
            r2 = st.collect()
            assert key(r2) == key(one)
        # rows only + page-locked input
        nb = len(bases)
        pin = capi.load().kr_host_alloc(nb)
        assert pin
        C.memmove(pin, bases.ctypes.data, nb)
        pb = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=(nb,))
        st.submit(pb, offs, capi.KR_BASES_PINNED | capi.KR_ROWS_ONLY)
        r3 = st.collect()
        # (rows-only batches leave the device as compact rows: the selected records and nothing else, every flag 1)
        assert sorted(zip(r3.rec_read.tolist(), r3.rec_key.tolist(), r3.rec_sel.tolist(), r3.rec_d.tolist())) == [t[:4] for t in key(one) if t[2] == 1]
        assert r3.nrows == len(r3.rec_key) == sum(t[2] for t in key(one)) and r3.rec_v is None
        assert st.format_dist(hx, names) == text1
        # ... and as record slots with flags when the compaction is switched off (KR_NO_ROW_COMPACTION: the path tiled batches take)
        monkeypatch.setenv("KR_NO_ROW_COMPACTION", "1")
        st.submit(pb, offs, capi.KR_BASES_PINNED | capi.KR_ROWS_ONLY)
        r4 = st.collect()
        monkeypatch.delenv("KR_NO_ROW_COMPACTION")
        assert sorted(zip(r4.rec_read.tolist(), r4.rec_key.tolist(), r4.rec_sel.tolist(), r4.rec_d.tolist())) == [t[:4] for t in key(one)]
        assert st.format_dist(hx, names) == text1
        # device view: offsets index the stream's arrays (lane slices)
        monkeypatch.setenv("KR_LANES_DEVICE", "1")
        std = dx.stream(max_reads=len(offs) - 1, max_bases=len(bases))
        monkeypatch.delenv("KR_LANES_DEVICE")
        import torch
        tb, to = torch.from_numpy(bases).cuda(), torch.from_numpy(offs.view(np.int64)).cuda()
        std.submit_device(tb.data_ptr(), to.data_ptr(), len(offs) - 1)
        rv = std.collect_device()
        assert std.timing().lanes == int(lanes)
        # without KR_TAP_ACCS the planes of most records are never written: the view must not expose them (include/krepp_amd.h)
        assert not rv.rec_hist and rv.rec_hist_stride == 0 and rv.rec_d and rv.rec_key
        std.submit_device(tb.data_ptr(), to.data_ptr(), len(offs) - 1, capi.KR_TAP_ACCS)
        rva = std.collect_device()
        assert rva.rec_hist and rva.rec_hist_stride > 0 and rva.rec_v
        std.submit_device(tb.data_ptr(), to.data_ptr(), len(offs) - 1)
        rv = std.collect_device()

        class DevPtr:
            def __init__(self, ptr, nbytes):
                self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

        def dev(ptr, n, dt):
            a = torch.as_tensor(DevPtr(C.cast(ptr, C.c_void_p).value, n * np.dtype(dt).itemsize), device="cuda:0").cpu().numpy()
            return a.view(dt)

        n = rv.nreads
        off, cnt = dev(rv.read_off, n, np.uint32), dev(rv.read_cnt, n, np.uint32)
        k_, d_, s_ = dev(rv.rec_key, rv.nrecs, np.uint32), dev(rv.rec_d, rv.nrecs, np.float64), dev(rv.rec_sel, rv.nrecs, np.uint8)
        got = sorted((r, int(k_[i]) >> 1, float(d_[i])) for r in range(n) for i in range(off[r], off[r] + cnt[r]) if s_[i])
        assert got == one.rows()
        capi.load().kr_host_free(pin)
        st.close()
        std.close()


def test_index_export_import_roundtrip(capi, toy, toy_reads):
    """The replication path used for multi-GPU: export flat buffers, import on a device,
    copy the bytes, query the replica."""
    import torch
    hx, dx, ox = toy
    names, bases, offs = toy_reads
    desc, bufs = dx.export()
    dx2, bufs2 = capi.DeviceIndex.import_empty(desc, 0)
    assert [b for _, b in bufs] == [b for _, b in bufs2]

    class DevPtr:
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

    for (p1, nb), (p2, _) in zip(bufs, bufs2):
        if nb:
            torch.as_tensor(DevPtr(p2, nb), device="cuda:0").copy_(torch.as_tensor(DevPtr(p1, nb), device="cuda:0"))
    torch.cuda.synchronize()
    _, a = gpu_dist(capi, dx, bases, offs)
    _, b = gpu_dist(capi, dx2, bases, offs)
    assert a.rows() == b.rows() and len(a.rows()) > 100


def test_index_broadcast_rccl(capi, toy, toy_reads):
    """kr_index_broadcast: the replica is filled by ncclBroadcast (RCCL, loaded on first use).  One GPU here, so the
    target is the root's own device (one-rank communicator); N>1 differs only in the communicator's size."""
    hx, dx, ox = toy
    names, bases, offs = toy_reads
    (rep,) = dx.broadcast([0])
    assert rep.device_bytes == dx.device_bytes
    _, a = gpu_dist(capi, dx, bases, offs)
    _, b = gpu_dist(capi, rep, bases, offs)
    assert a.rows() == b.rows() and len(a.rows()) > 100
    rep.close()
    _, c = gpu_dist(capi, dx, bases, offs)  # the root is unharmed by the replica's death
    assert c.rows() == a.rows()
    for bad in ([0, 0], [99], [-1]):
        with pytest.raises(capi.KrError) as e:
            dx.broadcast(bad)
        assert e.value.code == capi.KR_ERR_ARG
    assert dx.broadcast([]) == []


def test_export_import_of_a_slotted_index(capi, po, synth, tmp_path, monkeypatch):
    """A dense table carries a slotted copy of its bucket heads: the replica must receive it too."""
    import torch
    n = 40
    names = [f"s{i}" for i in range(n)]
    nwk = "(" + ",".join(f"{x}:0.01" for x in names) + ");"
    g = synth.evolve_genomes(nwk, 8000, seed=21)
    tsv = synth.write_genomes(g, str(tmp_path / "g"))
    (tmp_path / "t.nwk").write_text(nwk)
    idx = str(tmp_path / "ix")
    capi.build_index(tsv, idx, nwk=str(tmp_path / "t.nwk"), k=21, w=27, h=7, m=4, r=1, frac=True, num_threads=4)
    monkeypatch.setenv("KR_SLOT_LOG2W", "6")
    hx = capi.HostIndex(idx)
    dx = hx.upload(0)
    desc, bufs = dx.export()
    dx2, bufs2 = capi.DeviceIndex.import_empty(desc, 0)
    assert [b for _, b in bufs] == [b for _, b in bufs2]
    assert dx2.device_bytes == dx.device_bytes

    class DevPtr:
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

    for (p1, nb), (p2, _) in zip(bufs, bufs2):
        if nb:
            torch.as_tensor(DevPtr(p2, nb), device="cuda:0").copy_(torch.as_tensor(DevPtr(p1, nb), device="cuda:0"))
    torch.cuda.synchronize()
    bases, offs, rn = synth.sample_reads(g, 300, seed=4)
    ref = po.Index(idx).dist(bases, offs, rn, po.params(collect=1))
    _, a = gpu_dist(capi, dx, bases, offs)
    _, b = gpu_dist(capi, dx2, bases, offs)
    assert a.rows() == b.rows() and len(a.rows()) > 100
    assert_rows_close(b.rows(), rows_of_oracle(ref))


def test_occupancy_bitmap_of_a_sparse_table(capi, po, synth, tmp_path, monkeypatch):
    """A small index in a large table (2^21 rows, most of them empty) carries a one-bit-per-row occupancy bitmap that the
    scan consults before it fetches a bucket descriptor: same hits and rows with and without it, and the replica of
    such an index (export / import / RCCL broadcast) receives the bitmap."""
    nwk = "((a:0.03,b:0.03):0.02,(c:0.05,(d:0.01,e:0.01):0.02):0.01);"
    g = synth.evolve_genomes(nwk, 60_000, seed=12)
    tsv = synth.write_genomes(g, str(tmp_path / "g"))
    (tmp_path / "t.nwk").write_text(nwk)
    idx = str(tmp_path / "ix")
    capi.build_index(tsv, idx, nwk=str(tmp_path / "t.nwk"), k=27, w=35, h=11, m=4, r=1, frac=True, num_threads=4)
    bases, offs, rn = synth.sample_reads(g, 2000, seed=3)
    ref = po.Index(idx).dist(bases, offs, rn, po.params(collect=2))
    hx = capi.HostIndex(idx)
    out = {}
    for v in ("0", "1"):
        monkeypatch.setenv("KR_OCC_BITMAP", v)
        dx = hx.upload(0)
        st, res = gpu_dist(capi, dx, bases, offs, capi.KR_TAP_HITS)
        h = st.hits()
        out[v] = (sorted(zip(h["read"].tolist(), h["strand"].tolist(), h["kpos"].tolist(), h["cmer_index"].tolist(), h["hd"].tolist())), res.rows(), dx.device_bytes)
        if v == "1":
            (rep,) = dx.broadcast([0])
            assert rep.device_bytes == dx.device_bytes
            _, rr = gpu_dist(capi, rep, bases, offs)
            assert rr.rows() == res.rows()
            rep.close()
        st.close()
        dx.close()
    assert out["0"][0] == out["1"][0] and out["0"][1] == out["1"][1] and len(out["1"][0]) > 10_000
    assert out["1"][2] == out["0"][2] + (1 << 21) // 8  # the bitmap: one bit per row
    rh = ref["hits"]
    assert out["1"][0] == sorted(zip(rh["read"].tolist(), rh["strand"].tolist(), rh["kpos"].tolist(), rh["cmer_index"].tolist(), rh["hd"].tolist()))
    assert_rows_close(out["1"][1], rows_of_oracle(ref))


def test_cli_dist_end_to_end(capi, po, toy_index_dir, toy_reads, tmp_path):
    """BASELINE.json configs[0] plumbing: the `krepp dist` binary on an index directory and a FASTQ
    file; stdout must be the reference's header plus the oracle's rows (input order, 5 decimals)."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "krepp_amd", "lib", "krepp")
    fq = os.path.join(GOLDEN, "toy_reads.fq")
    names, bases, offs = toy_reads
    ox = po.Index(toy_index_dir)
    for extra, p in ([], po.params(collect=4)), (["--filter"], po.params(collect=4, no_filter=0)), \
                    (["--no-multi", "--hdist-th", "3"], po.params(collect=4, multi=0, hdist_th=3)), \
                    (["--dist-max", "0.05"], po.params(collect=4, dist_max=0.05)):
        r = subprocess.run([exe, "dist", "-i", toy_index_dir, "-q", fq] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.split("\n")
        assert lines[0].startswith("# software: krepp\tversion: v0.8.3\tinvocation :") and "dist -i" in lines[0]
        assert lines[1] == "SEQ_ID\tREFERENCE_NAME\tDIST"
        want = ox.dist(bases, offs, names, p)["text"]
        assert "\n".join(lines[2:]) == want, extra
        assert "Total number of sequences queried: 308" in r.stderr
    # gz input and -o
    import gzip, shutil
    gz = tmp_path / "reads.fq.gz"
    with open(fq, "rb") as fi, gzip.open(gz, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    out = tmp_path / "out.tsv"
    r = subprocess.run([exe, "--num-threads", "2", "dist", "-i", toy_index_dir, "-q", str(gz), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == ""
    assert out.read_text().split("\n", 2)[2] == ox.dist(bases, offs, names, po.params(collect=4))["text"]
    # a record buffer far too small for the batch: the CLI splits the batch and resubmits (KR_ERR_CAPACITY), same text
    env = dict(os.environ, KR_DEBUG_CLI_RECORDS="150")
    r = subprocess.run([exe, "dist", "-i", toy_index_dir, "-q", fq], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split("\n", 2)[2] == ox.dist(bases, offs, names, po.params(collect=4))["text"]


def test_cli_dist_summarize(po, toy_index_dir, toy_reads):
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "krepp_amd", "lib", "krepp")
    fq = os.path.join(GOLDEN, "toy_reads.fq")
    names, bases, offs = toy_reads
    r = subprocess.run([exe, "dist", "-i", toy_index_dir, "-q", fq, "--summarize"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split("\n")
    assert lines[1] == "REFERENCE_NAME\tWEIGHTED_COUNT\tSEQUENCE_ABUNDANCE"
    want = po.Index(toy_index_dir).summarize(bases, offs, po.params())
    got = "\n".join(lines[2:])
    # weighted counts are sums of 1/n in a different order: compare numerically, names exactly
    gw = [l.split("\t") for l in got.strip().split("\n")]
    ww = [l.split("\t") for l in want.strip().split("\n")]
    assert [g[0] for g in gw] == [w[0] for w in ww] and len(gw) >= 20
    for g, w in zip(gw, ww):
        assert abs(float(g[1]) - float(w[1])) <= 2e-5 and abs(float(g[2]) - float(w[2])) <= 2e-5
    assert abs(sum(float(g[2]) for g in gw) - 1.0) < 1e-3


def test_cli_contig_queries_multiline_fasta(po, synth, tmp_path):
    """Queries need not be reads: whole contigs (200 kb, wrapped FASTA lines) go through the same path -- thousands of
    segments per sequence in the accumulate kernel's merged mode -- and give the oracle's rows."""
    import subprocess
    nwk_path = os.path.join(GOLDEN, "tree_toy.nwk")
    g = synth.evolve_genomes(open(nwk_path).read(), 200_000, seed=7)
    tsv = synth.write_genomes(g, str(tmp_path / "g"))
    idx = str(tmp_path / "idx")
    from krepp_amd import capi
    capi.build_index(tsv, idx, nwk=nwk_path, k=27, w=35, h=11, m=4, r=1, frac=True, num_threads=8)
    rng = np.random.default_rng(3)
    names, seqs = [], []
    for nm, s in list(g.items())[:6]:
        s = s.copy()
        mut = rng.random(len(s)) < 0.02
        s[mut] = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(mut.sum()))
        for j, (a, b) in enumerate(((0, 200_000), (10_000, 70_000), (500, 3_500))):
            names.append(f"{nm}_c{j}")
            seqs.append(s[a:b].tobytes())
    fa = tmp_path / "q.fa"
    with open(fa, "wb") as f:
        for nm, s in zip(names, seqs):
            f.write(b">" + nm.encode() + b" description\n")
            for o in range(0, len(s), 80):
                f.write(s[o:o + 80] + b"\n")
    exe = os.path.join(ROOT, "krepp_amd", "lib", "krepp")
    r = subprocess.run([exe, "dist", "-i", idx, "-q", str(fa)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    bases = np.frombuffer(b"".join(seqs), np.uint8)
    offs = np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint64)
    ref = po.Index(idx).dist(bases, offs, names, po.params(collect=4, num_threads=8))
    assert r.stdout.split("\n", 2)[2] == ref["text"] and ref["text"].count("\n") > 50
