"""BASELINE.json configs[2] under pytest: 150-bp reads against the 1000-genome index (Yule tree, default
parameters -k 29 -w 35 -h 13, m4r1-frac: 2^25 rows) whose table is inflated to 10 GB resident in HBM — the
exact index bench.py measures (krepp_amd.synth.inflate_and_upload, same seeds).

Per table layout -- slot format 6, THE DEFAULT for this table and the one bench.py times (the slotted copy of the bucket
heads, W = 64 words = two 128-byte lines per probe); slot format 9, opt-in (KR_SLOT_FILTER / KR_SLOT_LOG2W=9: FILTER slots, one
128-byte line of 24-bit codes per probe, candidates verified by the accumulate kernel; kFilterByDefault = false in
kr_host_index.inc); and KR_SLOT_LOG2W=0, the packed table only -- 20,000 reads against the oracle holding the same table
(Index.replace_table) — table hits (src/query.cpp:352-368, src/index.cpp:160-168) and histograms (src/query.hpp:153-176)
bit-exact, DIST within the north star's 1e-6 relative — then a full 1,000,000-read batch through the size-independent
properties (reverse complement, permutation, split); on the default layout (and on the filter slots) the same three properties
once more at 8,000,000 reads per launch -- the size of bench.py's timed launches (4,000,000 on the filter slots) (item lists of hundreds of millions of
entries, every cursor range in use; the byte-table front end, finalize_events_fast, kr_select_lane_kernel and the row compaction
all on their default paths) -- compared through an order-independent checksum of the rows: of a rows-only batch (and its
KR_ROWS_INDEXED form) copied back, then of plain batches where their rows lie in HBM.
"""
import os

import numpy as np
import pytest

from conftest import assert_rows_close, rows_of_oracle

pytestmark = pytest.mark.gpu

N_GENOMES, GENOME_LEN, INDEX_GB = 1000, 100_000, 10.0
N_ORACLE, N_FULL, N_LAUNCH = 20_000, 1_000_000, 8_000_000  # N_LAUNCH = bench.py's reads per timed launch (format 6; the opt-in filter slots take half)


def rows_checksum(read, se, dbits):
    """Order-independent checksum of a multiset of (read, se, DIST bits) rows: (sum, xor) of a 64-bit mix of every row."""
    with np.errstate(over="ignore"):
        h = read.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + se.astype(np.uint64)
        h ^= h >> np.uint64(29)
        h *= np.uint64(0xBF58476D1CE4E5B9)
        h += dbits.astype(np.uint64)
        h ^= h >> np.uint64(32)
        h *= np.uint64(0x94D049BB133111EB)
        h ^= h >> np.uint64(31)
        return int(h.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(h)), int(len(h))


def rows_checksum_device(torch, dev, st, nreads, read_map=None, read_add=0, chunk=1 << 20):
    """(sum, count) of rows_checksum over the rows the stream's last launch left in its DEVICE arrays (kr_batch_collect_device), computed
    there: 265 M rows of an 8 M-read launch need not cross PCIe and numpy to be compared.  read_map: int64 tensor on the device."""
    from bench import dev_array

    rv = st.collect_device()
    off_all = dev_array(torch, dev, rv.read_off, nreads, torch.int32)
    cnt_all = dev_array(torch, dev, rv.read_cnt, nreads, torch.int32)
    key_all = dev_array(torch, dev, rv.rec_key, rv.nrecs, torch.int32)
    sel_all = dev_array(torch, dev, rv.rec_sel, rv.nrecs, torch.uint8)
    d_all = dev_array(torch, dev, rv.rec_d, rv.nrecs, torch.int64)  # the f64's bits
    M = (1 << 64) - 1
    lsr = lambda x, k: (x >> k) & ((1 << (64 - k)) - 1)
    i64 = lambda v: v - (1 << 64) if v >= (1 << 63) else v
    total = rows = 0
    for r0 in range(0, nreads, chunk):
        r1 = min(nreads, r0 + chunk)
        cnt = cnt_all[r0:r1].to(torch.int64)
        tot = int(cnt.sum().item())
        if not tot:
            continue
        first = torch.cumsum(cnt, 0) - cnt
        rd = torch.repeat_interleave(torch.arange(r1 - r0, device=dev), cnt)
        idx = off_all[r0:r1].to(torch.int64)[rd] + (torch.arange(tot, device=dev) - first[rd])
        key = key_all[idx].to(torch.int64) & 0xFFFFFFFF
        keep = (sel_all[idx] != 0) & (key != 0)
        rd, key, d = rd[keep] + r0, key[keep] >> 1, d_all[idx][keep]
        if read_map is not None:
            rd = read_map[rd]
        h = (rd + read_add) * i64(0x9E3779B97F4A7C15) + key  # (int64 arithmetic wraps like rows_checksum's uint64)
        h = h ^ lsr(h, 29)
        h = h * i64(0xBF58476D1CE4E5B9) + d
        h = h ^ lsr(h, 32)
        h = h * i64(0x94D049BB133111EB)
        h = h ^ lsr(h, 31)
        total = (total + int(h.sum().item())) & M
        rows += int(keep.sum().item())
    return total, rows


@pytest.fixture(scope="module")
def syn(capi, po, synth, tmp_path_factory):
    work = tmp_path_factory.mktemp("syn1000")
    nwk_text = synth.yule_newick(N_GENOMES, 2)
    genomes = synth.evolve_genomes(nwk_text, GENOME_LEN, seed=2)
    nwk = work / "yule.nwk"
    nwk.write_text(nwk_text)
    tsv = synth.write_genomes(genomes, str(work / "g"))
    idx = str(work / "idx")
    capi.build_index(tsv, idx, nwk=str(nwk), k=29, w=35, h=13, m=4, r=1, frac=True, num_threads=min(32, os.cpu_count() or 1))
    return idx, genomes


def make_reads(synth, genomes, n, seed):
    chunks = [synth.sample_reads(genomes, min(100_000, n - o), seed=seed * 1000 + c)[0] for c, o in enumerate(range(0, n, 100_000))]
    return np.concatenate(chunks), np.arange(n + 1, dtype=np.uint64) * np.uint64(150)


@pytest.mark.parametrize("slot_log2w", ["9", "6", "0"], ids=["filter_slots", "slotted_w64", "packed"])
def test_syn1000_10gb_index_vs_oracle_and_full_batch_properties(capi, po, synth, syn, monkeypatch, slot_log2w):
    import torch

    idx, genomes = syn
    monkeypatch.setenv("KR_SLOT_LOG2W", slot_log2w)
    dev = torch.device("cuda", 0)
    hx = capi.HostIndex(idx)
    dx, (inc, cmer) = synth.inflate_and_upload(torch, capi, hx, dev, 0, INDEX_GB)
    try:
        nk = cmer.size // 2
        assert len(inc) == 1 << 25 and nk * 8 >= 0.99 * INDEX_GB * 1e9
        # slotted: 2^25 rows x 256 B of slots on top of the packed table; filter slots: 2^25 x 640 B of slot-ordered colours more; packed: no slots
        assert dx.slot_format == int(slot_log2w)
        assert (dx.device_bytes > 38e9) == (slot_log2w == "9") and (dx.device_bytes > 18e9) == (slot_log2w != "0")
        ox = po.Index(idx)
        ox.replace_table(0, inc, cmer)
        del inc, cmer

        # ---- 20,000 reads: hits, histograms, rows against the oracle ----
        bases, offs = make_reads(synth, genomes, N_ORACLE, seed=1)
        ref = ox.dist(bases, offs, None, po.params(collect=3, num_threads=min(32, os.cpu_count() or 1)))
        st = dx.stream(max_reads=N_ORACLE, max_bases=len(bases), max_records=N_ORACLE * 256)
        st.submit(bases, offs, capi.KR_TAP_ACCS | capi.KR_TAP_HITS)
        res = st.collect()
        gh, rh = st.hits(), ref["hits"]
        assert len(gh) == len(rh) and len(rh) > 50 * N_ORACLE
        key = lambda h: np.sort(np.rec.fromarrays([h["read"].astype(np.uint64), h["strand"].astype(np.uint64), h["kpos"].astype(np.uint64),
                                                   h["cmer_index"].astype(np.uint64), h["hd"].astype(np.uint64), h["se"].astype(np.uint64)]))
        assert (key(gh) == key(rh)).all(), "table hits differ from the oracle"
        acc = ref["accs"][ref["accs"]["passed"] == 1]
        want = sorted(zip(acc["read"].tolist(), ((acc["se"] << 1) | acc["strand"]).tolist(), [tuple(x[:5]) for x in acc["hist"].tolist()]))
        got = sorted(zip(res.rec_read.tolist(), res.rec_key.tolist(), [tuple(x) for x in res.rec_hist.tolist()]))
        assert got == want, "histograms differ from the oracle"
        assert_rows_close(res.rows(), rows_of_oracle(ref), tol=1e-6)
        assert len(res.rows()) > 20 * N_ORACLE
        st.close()
        ox.close()

        # ---- 1,000,000 reads (one bench.py step): size-independent properties ----
        n = N_FULL
        bases, offs = make_reads(synth, genomes, n, seed=5)
        stf = dx.stream(max_reads=n, max_bases=len(bases), max_records=n * 64)

        def run(b, o):
            stf.submit(b, o)
            r = stf.collect()
            sel = r.rec_sel.astype(bool)
            # (read, se, DIST bits) of the output rows as one sortable array
            return np.stack([r.rec_read[sel].astype(np.uint64), (r.rec_key[sel] >> 1).astype(np.uint64), r.rec_d[sel].view(np.uint64)], axis=1)

        def canon(a):
            return a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]

        rows = canon(run(bases, offs))
        assert len(rows) > 20 * n
        rc = synth.COMP[bases.reshape(n, 150)[:, ::-1]].reshape(-1)
        assert (canon(run(rc, offs)) == rows).all(), "reverse-complemented reads give different rows"
        perm = np.random.default_rng(1).permutation(n)
        rp = run(bases.reshape(n, 150)[perm].reshape(-1), offs)
        rp[:, 0] = perm[rp[:, 0].astype(np.int64)].astype(np.uint64)
        assert (canon(rp) == rows).all(), "permuted reads give different rows"
        half = n // 2
        r1 = run(bases[: half * 150], offs[: half + 1])
        r2 = run(bases[half * 150:], offs[half:] - offs[half])
        r2[:, 0] += np.uint64(half)
        assert (canon(np.concatenate([r1, r2])) == rows).all(), "splitting the batch changes the rows"
        stf.close()
        # a stream given large one-lane batches tries other allocations of its item list during its first batches and keeps
        # the one the scan ran fastest on: same rows on every one of them
        monkeypatch.setenv("KR_LANES", "1")
        for trials in (("3", "0") if slot_log2w == "6" else ()):
            monkeypatch.setenv("KR_ITEM_PLACEMENT_TRIALS", trials)
            stf = dx.stream(max_reads=n, max_bases=len(bases), max_records=n * 64)
            for _ in range(9 if trials == "3" else 1):  # (the stream's own list is measured at its third launch, a trial list at its second)
                assert (canon(run(bases, offs)) == rows).all()
            ip = stf.item_placement()
            assert ip["tried"] == int(trials) and ip["kept"] <= ip["tried"] and (ip["scan_ns_per_read"] > 0) == (trials != "0")
            stf.close()
        monkeypatch.delenv("KR_LANES")
        monkeypatch.delenv("KR_ITEM_PLACEMENT_TRIALS", raising=False)

        # ---- 8,000,000 reads per launch on format 6 (= the default layout and the launch size bench.py times; 4,000,000 on
        #      format 9, the opt-in filter slots): the same properties at the size of bench.py's timed launches ----
        if slot_log2w in ("6", "9"):
            n = N_LAUNCH if slot_log2w == "6" else N_LAUNCH // 2
            bases, offs = make_reads(synth, genomes, n, seed=6)
            stl = dx.stream(max_reads=n, max_bases=len(bases), max_records=n * 64)

            def run_sum(b, o, read_map=None, read_add=0):
                stl.submit(b, o, capi.KR_ROWS_ONLY)
                r = stl.collect()
                sel = r.rec_sel.astype(bool)
                rd = r.rec_read[sel].astype(np.int64)
                if read_map is not None:
                    rd = read_map[rd]
                return rows_checksum(rd + read_add, r.rec_key[sel] >> 1, r.rec_d[sel].view(np.uint64))

            base_sum = run_sum(bases, offs)
            assert base_sum[2] > 20 * n
            plain_bytes = stl.last_d2h_bytes()
            if slot_log2w == "6":  # KR_ROWS_INDEXED at the size it is for: the same rows, DIST as an index into a list a twentieth their number
                monkeypatch.setenv("KR_LANES", "1")
                sti = dx.stream(max_reads=n, max_bases=len(bases), max_records=n * 64)
                monkeypatch.delenv("KR_LANES")
                sti.submit(bases, offs, capi.KR_ROWS_ONLY | capi.KR_ROWS_INDEXED)
                ri = sti.collect()
                assert ri.rec_dix is not None and 0 < len(ri.dist_list) < len(ri.rec_key) // 4
                assert rows_checksum(ri.rec_read.astype(np.int64), ri.rec_key >> 1, ri.rec_d.view(np.uint64)) == base_sum, "indexed rows differ"
                print(f"indexed rows: {sti.last_d2h_bytes() / 1e9:.2f} GB back against {plain_bytes / 1e9:.2f} GB, {len(ri.dist_list)} list entries for {ri.nrows} rows")
                assert sti.last_d2h_bytes() < 0.8 * plain_bytes
                del ri
                sti.close()
            # the properties on PLAIN batches (what bench.py's timed leg submits), their rows summed where they lie, in HBM: the rows-only
            # batch above crossed PCIe and numpy (3.2 GB, half a minute a launch); these do not
            def run_dev(b_, o_, read_map=None, read_add=0):
                stl.submit(b_, o_)
                stl.wait()
                return rows_checksum_device(torch, dev, stl, len(o_) - 1, read_map, read_add)

            base_dev = (base_sum[0], base_sum[2])
            assert run_dev(bases, offs) == base_dev, "full launch: a plain batch's rows differ from the rows-only batch's"
            assert run_dev(synth.COMP[bases.reshape(n, 150)[:, ::-1]].reshape(-1), offs) == base_dev, "full launch: reverse complement changes the rows"
            perm = np.random.default_rng(2).permutation(n)
            assert run_dev(bases.reshape(n, 150)[perm].reshape(-1), offs, read_map=torch.from_numpy(perm.astype(np.int64)).to(dev)) == base_dev, \
                "full launch: permutation changes the rows"
            half = n // 2
            a = run_dev(bases[: half * 150], offs[: half + 1])
            b = run_dev(bases[half * 150:], offs[half:] - offs[half], read_add=half)
            assert ((a[0] + b[0]) % (1 << 64), a[1] + b[1]) == base_dev, "full launch: splitting the batch changes the rows"
            stl.close()
    finally:
        dx.close()
        hx.close()
        torch.cuda.empty_cache()


def test_ten_thousand_genome_index_vs_oracle(capi, po, synth, tmp_path):
    """The per-GPU part of BASELINE.json configs[3] at full size: 10,000 genomes on a Yule tree (default parameters, 2^25
    rows), the table inflated to the config's 10 GB resident in HBM -- the index of `bench.py --workload syn10000`, same
    seeds; the 8-GPU run replicates this index per GPU and shards the reads, which changes nothing per GPU.  4,000 reads against the oracle: hits, histograms, rows; then 200,000 reads through the
    reverse-complement property.  20,000 key slots per wave: the accumulate kernel runs with fewer resident waves and its
    global-scratch paths are live."""
    import torch

    n_genomes = 10_000
    nwk_text = synth.yule_newick(n_genomes, 3)
    genomes = synth.evolve_genomes(nwk_text, 10_000, seed=3)
    (tmp_path / "y.nwk").write_text(nwk_text)
    tsv = synth.write_genomes(genomes, str(tmp_path / "g"))
    idx = str(tmp_path / "idx")
    capi.build_index(tsv, idx, nwk=str(tmp_path / "y.nwk"), k=29, w=35, h=13, m=4, r=1, frac=True, num_threads=min(32, os.cpu_count() or 1))
    hx = capi.HostIndex(idx)
    dx, (inc, cmer) = synth.inflate_and_upload(torch, capi, hx, torch.device("cuda", 0), 0, INDEX_GB)
    try:
        ox = po.Index(idx)
        assert ox.info.nleaves == n_genomes
        assert len(inc) == 1 << 25 and cmer.size // 2 * 8 >= 0.99 * INDEX_GB * 1e9 and dx.device_bytes > 18e9  # 10 GB table + slots
        ox.replace_table(0, inc, cmer)
        del inc, cmer
        n = 4000
        bases, offs = make_reads(synth, genomes, n, seed=9)
        ref = ox.dist(bases, offs, None, po.params(collect=3, num_threads=min(32, os.cpu_count() or 1)))
        st = dx.stream(max_reads=n, max_bases=len(bases), max_records=n * 1024)
        st.submit(bases, offs, capi.KR_TAP_ACCS | capi.KR_TAP_HITS)
        res = st.collect()
        gh, rh = st.hits(), ref["hits"]
        key = lambda h: np.sort(np.rec.fromarrays([h["read"].astype(np.uint64), h["strand"].astype(np.uint64), h["kpos"].astype(np.uint64),
                                                   h["cmer_index"].astype(np.uint64), h["hd"].astype(np.uint64), h["se"].astype(np.uint64)]))
        assert len(gh) == len(rh) > 10 * n and (key(gh) == key(rh)).all()
        acc = ref["accs"][ref["accs"]["passed"] == 1]
        want = sorted(zip(acc["read"].tolist(), ((acc["se"] << 1) | acc["strand"]).tolist(), [tuple(x[:5]) for x in acc["hist"].tolist()]))
        got = sorted(zip(res.rec_read.tolist(), res.rec_key.tolist(), [tuple(x) for x in res.rec_hist.tolist()]))
        assert got == want
        assert_rows_close(res.rows(), rows_of_oracle(ref), tol=1e-6)
        st.close()
        ox.close()
        n = 200_000
        bases, offs = make_reads(synth, genomes, n, seed=10)
        stf = dx.stream(max_reads=n, max_bases=len(bases), max_records=n * 256)

        def run(b):
            stf.submit(b, offs)
            r = stf.collect()
            sel = r.rec_sel.astype(bool)
            a = np.stack([r.rec_read[sel].astype(np.uint64), (r.rec_key[sel] >> 1).astype(np.uint64), r.rec_d[sel].view(np.uint64)], axis=1)
            return a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]

        rows = run(bases)
        assert len(rows) > 10 * n
        assert (run(synth.COMP[bases.reshape(n, 150)[:, ::-1]].reshape(-1)) == rows).all()
        stf.close()
    finally:
        dx.close()
        hx.close()
        torch.cuda.empty_cache()


def test_place_on_the_1000_genome_tree_never_leaves_the_device(capi, synth, syn):
    """`krepp place` on the 1000-genome index with its own Yule tree as backbone (the index as built, 8.3 M entries): reads
    there reach dozens of leaves on average and hundreds in the tail -- more than kr_place_kernel's LDS arrays hold (256
    leaves / 1024 distinct ancestors); those reads take the kernel's second launch (arrays in global scratch) and no batch
    falls back to the host back end.  200,000 reads: text and placements of the device back end equal the host back end's
    (kr_place_batch) bit for bit (src/query.cpp:218-333)."""
    import ctypes as C

    idx, genomes = syn
    n = 200_000
    bases, offs = make_reads(synth, genomes, n, seed=77)
    names = [f"q{i}" for i in range(n)]
    c_names = (C.c_char_p * n)(*[x.encode() for x in names])
    hx = capi.HostIndex(idx)
    placer = capi.Placer(hx, None, 0, tabular=True, max_reads=n, max_bases=len(bases))
    try:
        d0, h0 = capi.place_counters()
        hv0 = capi.place_heavy_reads()
        text_d, pl_d = placer.place(bases, offs, names, host=False, c_names=c_names)
        d1, h1 = capi.place_counters()
        assert (d1 - d0, h1 - h0) == (1, 0), "the batch fell back to the host back end"
        heavy = capi.place_heavy_reads() - hv0
        text_h, pl_h = placer.place(bases, offs, names, host=True, c_names=c_names)
        assert len(pl_d) > n // 2
        assert pl_d.tobytes() == pl_h.tobytes() and text_d == text_h
        print(f"heavy reads (upper bound): {heavy} of {n}")
        # round 6: the rows written on the device (the caller wants text, no placement records): the same bytes, also from the reads
        # that keep hundreds of candidates here (the toy trees of tests/test_place.py keep at most a few dozen); jplace and tabular
        t0 = capi.place_text_counters()
        placer.prev = C.c_int(0)
        text_dt, _ = placer.place(bases, offs, names, host=False, c_names=c_names, want_placements=False)
        t1 = capi.place_text_counters()
        assert (t1[0] - t0[0], t1[1] - t0[1]) == (1, 0), "the rows were formatted by the host"
        assert text_dt == text_h
        pj = capi.Placer(hx, None, 0, tabular=False, max_reads=n, max_bases=len(bases))
        try:
            want_j, _ = pj.place(bases, offs, names, host=True, c_names=c_names)
            pj.prev = C.c_int(0)
            got_j, _ = pj.place(bases, offs, names, host=False, c_names=c_names, want_placements=False)
            assert got_j == want_j and capi.place_text_counters()[0] == t1[0] + 1
        finally:
            pj.close()
    finally:
        placer.close()
        hx.close()


def test_place_on_the_1000_genome_tree_matches_the_oracle(capi, po, synth, syn):
    """`krepp place` on the 1000-genome Yule tree against the ORACLE's restatement of IBatch::place_sequences /
    report_placement (src/query.cpp:198-333; ancestors' fractional histograms, candidate rule, Brent on internal nodes,
    chi-square against the closest leaf, LWR), not only against this repo's own host back end: 2,000 reads, jplace and
    --tabular -- placements field by field (the north star's 1e-6 on the fp64 fields), the text byte for byte, and no batch
    on the host back end.  Reads here reach dozens of leaves and hundreds of ancestors: the kernel's global-scratch launch
    is live (the 25-leaf trees of tests/test_place.py never reach it)."""
    idx, genomes = syn
    n = 2000
    bases, offs, names = synth.sample_reads(genomes, n, seed=4242)
    hx = capi.HostIndex(idx)
    ox = po.Index(idx)
    ox.set_placement_tree(None)  # the index's own tree as backbone (TargetIndex::ensure_backbone, src/krepp.cpp:48-64)
    dev0, host0 = capi.place_counters()
    try:
        for tabular in (False, True):
            want = ox.place(bases, offs, names, po.params(no_filter=0, num_threads=min(16, os.cpu_count() or 1)), tabular=tabular)
            placer = capi.Placer(hx, None, 0, tabular=tabular, max_reads=n, max_bases=len(bases))
            text, pl = placer.place(bases, offs, names)
            placer.close()
            key = lambda p: sorted((int(r), int(e)) for r, e in zip(p["read"], p["edge"]))
            assert len(pl) > n // 2 and len(np.unique(pl["read"])) > n // 2
            assert key(pl) == key(want["placements"]), tabular
            a = np.sort(pl, order=["read", "edge"])
            b = np.sort(want["placements"], order=["read", "edge"])
            for f in ("lwr", "d_llh", "pendant", "distal"):
                assert np.allclose(a[f], b[f], rtol=1e-6, atol=1e-9), f
            assert np.allclose(a["v_llh"], b["v_llh"], rtol=1e-9)
            assert text == want["text"], tabular
        dev1, host1 = capi.place_counters()
        assert dev1 - dev0 == 2 and host1 == host0, "a batch fell back to the host back end"
    finally:
        ox.close()
        hx.close()
