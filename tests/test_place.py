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


# ---- lineage trees (-l) and place --summarize ------------------------------------------------

LINEAGES_SMALL = (
    "G1\tk__B; p__P1; c__C1; s__\n"
    "G2\tk__B; p__P1; c__C2; s__X y\textra column\n"
    "G3\tk__B;p__P2\n"
    "G4\tk__Arch; p__P3\n"
    "G5\tk__B; p__P1; c__C1; s__Z\n")
# Tree::parse_lineages by hand (src/phytree.cpp:320-369): taxa keep their first parent, children in attachment
# order, post-order edge numbers, no branch lengths, a node "root" on top
LINEAGES_SMALL_NWK = "((((G1{0},(G5{1})Z{2})C1{3},((G2{4})X y{5})C2{6})P1{7},(G3{8})P2{9})B{10},((G4{11})P3{12})Arch{13})root{14};"


def test_oracle_lineage_tree(po, toy_index_dir, toy_reads):
    ox = po.Index(toy_index_dir)
    ox.set_lineage_tree(LINEAGES_SMALL)
    assert ox.place_frame(1, False, "i", 0).split('"tree" : "')[1].split('"')[0] == LINEAGES_SMALL_NWK
    for bad in ("G1\n", "G1\t\n", "\n", "G1\tk__A\nG1\tk__A\n"):
        with pytest.raises(RuntimeError):
            ox.set_lineage_tree(bad)
    # the reference's toy lineages: 18 of the 25 references, one kingdom
    lin = open(os.path.join(GOLDEN, "lineages_toy.txt")).read()
    ox.set_lineage_tree(lin)
    tree = ox.place_frame(1, False, "i", 0).split('"tree" : "')[1].split('"')[0]
    assert tree.endswith(")Bacteria{%d})root{%d};" % (tree.count("{") - 2, tree.count("{") - 1))
    assert tree.count("G0") == 18 and "k__" not in tree and "Chloroflexus aurantiacus" in tree
    names, bases, offs = toy_reads
    r = ox.place(bases, offs, names, po.params(no_filter=0))
    pl = r["placements"]
    assert len(pl) > 100
    for rd in np.unique(pl["read"]):
        assert abs(pl[pl["read"] == rd]["lwr"].sum() - 1.0) < 1e-9
    assert (pl["distal"] == 0).all()  # no branch lengths in a taxonomy
    # --summarize: every placed read counts once
    txt = ox.place_summarize(bases, offs, po.params(no_filter=0))
    rows = [l.split("\t") for l in txt.splitlines()]
    assert abs(sum(float(x[2]) for x in rows) - len(np.unique(pl["read"]))) < 1e-3
    assert abs(sum(float(x[3]) for x in rows) - 1.0) < 1e-3
    assert ox.place_frame(0, 2, "inv").endswith("DISTAL_NODE\tEDGE_NUM\tWEIGHTED_COUNT\tSEQUENCE_ABUNDANCE\n")


# a taxon whose name comes back under another parent keeps its first parent; "D" is then left without children
# (the reference's traversal is undefined there): it ends its path like a leaf
LINEAGES_REUSED = "G1\tk__A; p__B\nG2\tk__C; p__D; p__B\n"
LINEAGES_REUSED_NWK = "(((G1{0},G2{1})B{2})A{3},(D{4})C{5})root{6};"


def test_host_lineage_tree_matches_oracle(capi, po, toy_index_dir):
    """kr_place_tree_create_lineage / kr_place_frame are host-only: no GPU needed."""
    import ctypes as C
    lib = capi.load()
    hx = capi.HostIndex(toy_index_dir)
    ox = po.Index(toy_index_dir)
    ox.set_lineage_tree(LINEAGES_REUSED)
    assert ox.place_frame(1, False, "i", 0).split('"tree" : "')[1].split('"')[0] == LINEAGES_REUSED_NWK
    for lin in (LINEAGES_SMALL, LINEAGES_REUSED, open(os.path.join(GOLDEN, "lineages_toy.txt")).read()):
        ox.set_lineage_tree(lin)
        pt = C.c_void_p()
        capi.check(lib.kr_place_tree_create_lineage(hx.h, lin.encode(), C.byref(pt)))
        for which, mode in ((0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2)):
            txt, ln = C.c_void_p(), C.c_uint64()
            capi.check(lib.kr_place_frame(pt, which, mode, b"inv", 7, C.byref(txt), C.byref(ln)))
            assert C.string_at(txt, ln.value).decode() == ox.place_frame(which, mode, "inv", 7), (which, mode)
            lib.kr_free(txt)
        lib.kr_place_tree_free(pt)
    pt = C.c_void_p()
    for bad in (b"G1\n", b"G1\t\n", b"", b"G1\tk__A\nG1\tk__A\n"):
        assert lib.kr_place_tree_create_lineage(hx.h, bad, C.byref(pt)) != 0
    assert "more than once" in lib.kr_last_error().decode()


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [dict(), dict(multi=0), dict(no_filter=1)])
def test_place_on_lineages_matches_oracle(capi, po, toy_index_dir, toy_reads, opts):
    names, bases, offs = toy_reads
    lin = open(os.path.join(GOLDEN, "lineages_toy.txt")).read()
    hx = capi.HostIndex(toy_index_dir)
    ox = po.Index(toy_index_dir)
    ox.set_lineage_tree(lin)
    okw = dict(no_filter=0)
    okw.update(opts)
    pk = dict(opts)
    pk.pop("no_filter", None)
    for tabular in (0, 1, 2):
        placer = capi.Placer(hx, None, 0, tabular=tabular, max_reads=len(names), max_bases=len(bases), lineage_text=lin, **pk)
        if opts.get("no_filter"):
            placer.popts.no_filter = 1
        text, pl = placer.place(bases, offs, names)
        if tabular == 2:
            assert text == "" and placer.summary() == ox.place_summarize(bases, offs, po.params(**okw))
            assert len(placer.summary().splitlines()) > 5
        else:
            want = ox.place(bases, offs, names, po.params(**okw), tabular=bool(tabular))
            assert placements_key(pl) == placements_key(want["placements"]) and len(pl) > 100
            assert text == want["text"], (opts, tabular)
        placer.close()


@pytest.mark.gpu
def test_place_summarize_on_backbone_matches_oracle(capi, po, toy_index_dir, toy_reads):
    names, bases, offs = toy_reads
    hx = capi.HostIndex(toy_index_dir)
    ox = po.Index(toy_index_dir)
    ox.set_placement_tree(None)
    placer = capi.Placer(hx, None, 0, tabular=2, max_reads=100, max_bases=len(bases))
    for lo in range(0, len(names), 100):  # several submits: the sums carry over
        hi = min(len(names), lo + 100)
        placer.place(bases[int(offs[lo]):int(offs[hi])], offs[lo:hi + 1] - offs[lo], names[lo:hi])
    # the oracle sums in 512-read batches, this run in 100-read submits: equal to the printed precision
    got = [l.split("\t") for l in placer.summary().splitlines()]
    want = [l.split("\t") for l in ox.place_summarize(bases, offs, po.params(no_filter=0)).splitlines()]
    assert [g[:2] for g in got] == [w[:2] for w in want] and len(got) > 10
    for g, w in zip(got, want):
        assert abs(float(g[2]) - float(w[2])) < 2e-5 and abs(float(g[3]) - float(w[3])) < 2e-5


@pytest.mark.gpu
def test_cli_place_lineages_and_summarize(po, toy_index_dir, toy_reads, tmp_path):
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "krepp_amd", "lib", "krepp")
    fq = os.path.join(GOLDEN, "toy_reads.fq")
    lf = os.path.join(GOLDEN, "lineages_toy.txt")
    names, bases, offs = toy_reads
    ox = po.Index(toy_index_dir)
    ox.set_lineage_tree(open(lf).read())
    r = subprocess.run([exe, "place", "-i", toy_index_dir, "-q", fq, "-l", lf], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "taxonomic lineage" in r.stderr
    want = ox.place(bases, offs, names, po.params(no_filter=0))
    inv = f"{exe} place -i {toy_index_dir} -q {fq} -l {lf}"
    assert r.stdout == ox.place_frame(0) + want["text"] + ox.place_frame(1, False, inv, len(names))
    r = subprocess.run([exe, "place", "-i", toy_index_dir, "-q", fq, "-l", lf, "--summarize"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    inv += " --summarize"
    assert r.stdout == ox.place_frame(0, 2, inv) + ox.place_summarize(bases, offs, po.params(no_filter=0))
    # on the backbone, with the device buffers forced to overflow so that batches are split and retried
    ox.set_placement_tree(None)
    env = dict(os.environ, KR_DEBUG_CLI_RECORDS="150")
    r = subprocess.run([exe, "place", "-i", toy_index_dir, "-q", fq, "--summarize"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    inv = f"{exe} place -i {toy_index_dir} -q {fq} --summarize"
    assert r.stdout == ox.place_frame(0, 2, inv) + ox.place_summarize(bases, offs, po.params(no_filter=0))


@pytest.mark.gpu
@pytest.mark.parametrize("tree", ["backbone", "user", "lineages"])
def test_device_back_end_equals_host_back_end(capi, po, toy_index_dir, toy_reads, toy_genomes, synth, tree):
    """kr_place_stream (ancestor accumulation, candidates, Brent and chi-square in kr_place_kernel) against
    kr_place_batch (the same on host threads + kr_llh_batch): identical text and placements, bit for bit, on the
    backbone, on a user tree that lacks some references, and on a lineage tree; all option sets and output modes."""
    names, bases, offs = toy_reads
    b2, o2, n2 = synth.sample_reads(toy_genomes, 5000, seed=23)
    hx = capi.HostIndex(toy_index_dir)
    kw = {}
    if tree == "user":
        nwk = open(os.path.join(GOLDEN, "tree_toy.nwk")).read()
        kw["nwk_text"] = nwk.replace("(G000735195:0.0276038,G000018865:0.0228997)N2640:0.160977", "(G000735195:0.03,NEWLEAF:0.02)N2640:0.160977")
        assert kw["nwk_text"] != nwk
    elif tree == "lineages":
        kw["lineage_text"] = open(os.path.join(GOLDEN, "lineages_toy.txt")).read()
    for opts in (dict(), dict(multi=0), dict(tau=1, chisq=3.841), dict(hdist_th=3)):
        for tabular in (0, 1, 2):
            for rb, ro, rn in ((bases, offs, names), (b2, o2, n2)):
                out = []
                for host in (False, True):
                    pl = capi.Placer(hx, kw.get("nwk_text"), 0, tabular=tabular, max_reads=len(rn), max_bases=len(rb),
                                     lineage_text=kw.get("lineage_text"), **opts)
                    text, p = pl.place(rb, ro, rn, host=host)
                    out.append((text, p.tobytes(), pl.summary() if tabular == 2 else ""))
                    pl.close()
                assert out[0] == out[1], (tree, opts, tabular)
                assert len(out[0][1]) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("tree", ["backbone", "user", "lineages"])
def test_rows_written_on_the_device_equal_the_host_s(capi, po, toy_index_dir, toy_reads, toy_genomes, synth, tree, monkeypatch):
    """Round 6: when the caller wants text and no placement records (the CLI's jplace and tabular modes), kr_place_stream has the rows
    written on the DEVICE (kr_place_text_* kernels: chi-square filter, LWR, Jukes-Cantor, "%.5f" digits, jplace framing) and copies
    bytes back.  Byte for byte the text of the host's last phase (kr_place_batch, which the oracle comparisons above pin): every
    tree, option set and text mode; two calls in a row (the ",\n" between the reads of a jplace file); several ranges; and with the
    device text switched off (KR_PLACE_HOST_TEXT) the same again."""
    names, bases, offs = toy_reads
    b2, o2, n2 = synth.sample_reads(toy_genomes, 6000, seed=31)
    hx = capi.HostIndex(toy_index_dir)
    kw = {}
    if tree == "user":
        nwk = open(os.path.join(GOLDEN, "tree_toy.nwk")).read()
        kw["nwk_text"] = nwk.replace("(G000735195:0.0276038,G000018865:0.0228997)N2640:0.160977", "(G000735195:0.03,NEWLEAF:0.02)N2640:0.160977")
    elif tree == "lineages":
        kw["lineage_text"] = open(os.path.join(GOLDEN, "lineages_toy.txt")).read()
    lib = capi.load()

    def texts(pl, host, want_pl):
        out = []
        for rb, ro, rn in ((bases, offs, names), (b2, o2, n2)):  # two calls on one Placer: `prev` carries over
            t, _ = pl.place(rb, ro, rn, host=host, want_placements=want_pl)
            out.append(t)
        return out

    for opts in (dict(), dict(multi=0), dict(tau=1, chisq=3.841), dict(hdist_th=3)):
        for tabular in (0, 1):
            mk = lambda: capi.Placer(hx, kw.get("nwk_text"), 0, tabular=tabular, max_reads=len(n2), max_bases=len(b2), lineage_text=kw.get("lineage_text"), **opts)
            pl = mk()
            want = texts(pl, True, True)  # the host back end
            pl.close()
            assert sum(len(t) for t in want) > 10000
            for ranges in (None, "3"):
                if ranges:
                    monkeypatch.setenv("KR_PLACE_RANGES", ranges)
                pl = mk()
                d0 = capi.place_text_counters()
                got = texts(pl, False, False)  # device back end, text from the device
                d1 = capi.place_text_counters()
                pl.close()
                assert got == want, (tree, opts, tabular, ranges)
                assert d1[0] - d0[0] == (2 if not ranges else 6) and d1[1] == d0[1], "the rows were not written on the device"
                if ranges:
                    monkeypatch.delenv("KR_PLACE_RANGES")
            monkeypatch.setenv("KR_PLACE_HOST_TEXT", "1")
            pl = mk()
            d0 = capi.place_text_counters()
            assert texts(pl, False, False) == want
            assert capi.place_text_counters() == d0
            pl.close()
            monkeypatch.delenv("KR_PLACE_HOST_TEXT")


@pytest.mark.gpu
@pytest.mark.parametrize("ranges", ["1", "2", "3", "16"])
def test_ranges_of_a_batch_give_the_batch_s_output(capi, po, toy_index_dir, toy_reads, toy_genomes, synth, monkeypatch, ranges):
    """kr_place_stream works through a batch in ranges of reads (the host's last phase of one range beside the place kernels of the
    next, round 5): whatever the number of ranges (KR_PLACE_RANGES; 2 by default from 131,072 reads), text -- jplace separators
    included --, placements and summary equal the host back end's, also when every read of a range exceeds the LDS limits and
    when the candidate slots run out and a range is run again."""
    b2, o2, n2 = synth.sample_reads(toy_genomes, 20_000, seed=29)
    hx = capi.HostIndex(toy_index_dir)
    for env in (dict(), dict(KR_DEBUG_PLACE_LDS="3,8")):
        for k_, v in env.items():
            monkeypatch.setenv(k_, v)
        for tabular in (0, 1, 2):
            ref = None
            for host in (True, False):
                monkeypatch.setenv("KR_PLACE_RANGES", ranges)
                pl = capi.Placer(hx, None, 0, tabular=tabular, max_reads=len(n2), max_bases=len(b2))
                text, p = pl.place(b2, o2, n2, host=host)
                got = (text, p.tobytes(), pl.summary() if tabular == 2 else "")
                pl.close()
                if ref is None:
                    ref = got
                assert got == ref, (ranges, tabular, env)
            assert len(ref[1]) > 0
        for k_ in env:
            monkeypatch.delenv(k_)


@pytest.mark.gpu
def test_large_batch_goes_through_the_thread_pool(capi, po, toy_index_dir, toy_reads):
    """Batches of more than 8,192 reads are cut into ranges handled by several host threads (kr::parallel_for):
    aggregation, candidate lists, text pieces and the jplace separators must come out as for one thread."""
    names, bases, offs = toy_reads
    reps = 70
    n1 = len(names)
    big_names = [f"{nm}_{r}" for r in range(reps) for nm in names]
    big_bases = np.tile(bases, reps)
    lens = np.diff(offs)
    big_offs = np.concatenate([[0], np.cumsum(np.tile(lens, reps))]).astype(np.uint64)
    assert len(big_names) == n1 * reps > 20000
    hx = capi.HostIndex(toy_index_dir)
    ox = po.Index(toy_index_dir)
    ox.set_placement_tree(None)
    for tabular in (0, 1, 2):
        placer = capi.Placer(hx, None, 0, tabular=tabular, max_reads=len(big_names), max_bases=len(big_bases))
        text, pl = placer.place(big_bases, big_offs, big_names)
        if tabular == 2:
            assert placer.summary() == ox.place_summarize(big_bases, big_offs, po.params(no_filter=0, num_threads=8))
        else:
            want = ox.place(big_bases, big_offs, big_names, po.params(no_filter=0, num_threads=8), tabular=bool(tabular))
            assert text == want["text"]
            assert placements_key(pl) == placements_key(want["placements"])
        placer.close()
    # dist rows through the same pool
    dx = hx.upload(0)
    st = dx.stream(max_reads=len(big_names), max_bases=len(big_bases), max_records=len(big_names) * 32)
    st.submit(big_bases, big_offs)
    st.collect()
    assert st.format_dist(hx, big_names) == ox.dist(big_bases, big_offs, big_names, po.params(collect=4, num_threads=8))["text"]


@pytest.mark.gpu
def test_cli_dist_on_a_file_large_enough_for_the_parallel_reader(po, toy_index_dir, toy_reads, tmp_path):
    """> 32 MB of plain FASTQ: the default configuration parses it in chunks on the thread pool, runs several
    65,536-read batches on two workers and writes them in input order."""
    import subprocess
    from conftest import ROOT
    names, bases, offs = toy_reads
    reps = 440
    fq = tmp_path / "big.fq"
    recs = []
    for i, nm in enumerate(names):
        t = bytes(bases[int(offs[i]):int(offs[i + 1])])
        recs.append((nm.encode(), t))
    with open(fq, "wb") as f:
        for r in range(reps):
            f.write(b"".join(b"@%s_%d some comment\n%s\n+\n%s\n" % (nm, r, t, b"I" * len(t)) for nm, t in recs))
    assert os.path.getsize(fq) > (32 << 20)
    exe = os.path.join(ROOT, "krepp_amd", "lib", "krepp")
    r = subprocess.run([exe, "dist", "-i", toy_index_dir, "-q", str(fq)], capture_output=True, text=True,
                       env=dict(os.environ, KR_CLI_TIMING="1"))
    assert r.returncode == 0, r.stderr
    ox = po.Index(toy_index_dir)
    one = ox.dist(bases, offs, names, po.params(collect=4))["text"].splitlines()
    got = r.stdout.splitlines()[2:]
    assert len(got) == len(one) * reps
    # every repetition is the toy batch with renamed reads
    for rep in (0, 1, reps // 2, reps - 1):
        chunk = got[rep * len(one):(rep + 1) * len(one)]
        want = [l.split("\t", 1)[0] + f"_{rep}\t" + l.split("\t", 1)[1] for l in one]
        assert chunk == want, rep
    assert len(set(l.split("\t", 1)[0] for l in got)) == len(names) * reps
    # device buffers far too small for a 65,536-read submit: the workers settle at a smaller piece size after the
    # first overflow instead of failing every full-size submit again; same output
    r2 = subprocess.run([exe, "dist", "-i", toy_index_dir, "-q", str(fq)], capture_output=True, text=True,
                        env=dict(os.environ, KR_DEBUG_CLI_RECORDS="20000"))
    assert r2.returncode == 0, r2.stderr
    assert r2.stdout.splitlines()[2:] == got
    # (round 5) the rows above were written as text by the GPU (kr_batch_submit_text / kr_batch_collect_text); the host formatter
    # (KR_CLI_HOST_TEXT=1: kr_format_dist on the box's CPUs, as before) and a text buffer too small for a batch (32 bytes per read:
    # KR_ERR_CAPACITY, the workers settle at a smaller piece) give the same bytes
    for env in (dict(KR_CLI_HOST_TEXT="1"), dict(KR_CLI_TEXT_PER_READ="32")):
        r3 = subprocess.run([exe, "dist", "-i", toy_index_dir, "-q", str(fq)], capture_output=True, text=True, env=dict(os.environ, **env))
        assert r3.returncode == 0, r3.stderr
        assert r3.stdout.splitlines()[2:] == got, env


@pytest.mark.gpu
@pytest.mark.parametrize("limits", ["3,8", "1,1", "6,1024", "3,8/global", "3,8/second-launch", "3,8/second-launch/global"])
def test_heavy_reads_stay_on_the_device(capi, po, toy_index_dir, toy_reads, toy_genomes, synth, monkeypatch, limits):
    """Reads with more leaves / distinct ancestors than kr_place_kernel's LDS arrays hold (256 / 1024; lowered here so
    that the 25-leaf tree has such reads) are done by the kernel's second launch, whose arrays hold the whole tree -- in its
    dynamic LDS where the tree is small enough, in global scratch otherwise ("/global": forced here):
    no batch goes to the host back end, and text, placements and summary equal the host back end's bit for bit and
    the oracle's (src/query.cpp:248-281, Minfo::add src/query.hpp:139-152)."""
    if limits.endswith("/global"):
        limits = limits.rsplit("/", 1)[0]
        monkeypatch.setenv("KR_PLACE_HEAVY_GLOBAL", "1")
    if limits.endswith("/second-launch"):  # every over-limit read through the list and the second launch, as before round 4
        limits = limits.split("/")[0]
        monkeypatch.setenv("KR_PLACE_BIG_FIRST", "0")
    names, bases, offs = toy_reads
    b2, o2, n2 = synth.sample_reads(toy_genomes, 5000, seed=29)
    hx = capi.HostIndex(toy_index_dir)
    ox = po.Index(toy_index_dir)
    ox.set_placement_tree(None)
    monkeypatch.setenv("KR_DEBUG_PLACE_LDS", limits)
    for rb, ro, rn in ((bases, offs, names), (b2, o2, n2)):
        for tabular in (0, 1, 2):
            out = []
            for host in (False, True):
                pl = capi.Placer(hx, None, 0, tabular=tabular, max_reads=len(rn), max_bases=len(rb))
                d0, h0 = capi.place_counters()
                hv0 = capi.place_heavy_reads()
                text, p = pl.place(rb, ro, rn, host=host)
                d1, h1 = capi.place_counters()
                if not host:
                    assert (d1 - d0, h1 - h0) == (1, 0), "the batch left the device"
                    # (reads done in global scratch: by the second launch -- counted in list chunks of 8 -- or, since round 4, by the
                    #  first launch itself, counted one by one: "6,1024" leaves a dozen of the 308 toy reads over the limit)
                    assert capi.place_heavy_reads() - hv0 > (len(rn) // 10 if limits != "6,1024" else 0), "no read exceeded the LDS arrays"
                out.append((text, p.tobytes(), pl.summary() if tabular == 2 else ""))
                pl.close()
            assert out[0] == out[1], (limits, tabular)
            if tabular != 2:
                want = ox.place(rb, ro, rn, po.params(no_filter=0), tabular=bool(tabular))
                assert out[0][0] == want["text"]
