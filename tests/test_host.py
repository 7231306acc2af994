"""CPU tests of the product's host side: C-ABI surface, index reader, FASTX reader, builder."""
import ctypes as C
import gzip
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from helpers import closed_form, front_end_py, revcomp, row_of, write_index


def test_library_exports_every_declared_symbol(capi):
    lib = capi.load()
    hdr = open(os.path.join(ROOT, "include", "krepp_amd.h")).read()
    import re
    declared = set(re.findall(r"KR_API\s+[\w\s\*]+?\b(kr_\w+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for s in declared:
        assert hasattr(lib, s), s
    assert b"krepp" in lib.kr_version()


def test_struct_sizes_match_header(capi):
    # the ctypes mirrors must match the C layout (spot-check through a tiny C probe would need a
    # compiler at test time; sizes are asserted against the known layout instead)
    assert C.sizeof(capi.KrLibView) == 4 * 8 + 8 + 6 * 4
    assert C.sizeof(capi.KrParams) == 32
    assert C.sizeof(capi.KrHit) == 32
    assert C.sizeof(capi.KrResultView) == 8 + 10 * 8 + 8 + 8 + 3 * 8
    assert C.sizeof(capi.KrTiming) == 32


def test_host_index_matches_oracle_loader(capi, po, toy_index_dir):
    hx = capi.HostIndex(toy_index_dir)
    ox = po.Index(toy_index_dir)
    assert (hx.k, hx.hh, hx.m, hx.nnodes) == (ox.info.k, ox.info.h, ox.info.m, ox.info.nnodes) == (21, 7, 4, 47)
    kinds = hx.kinds()
    for se in range(1, hx.nnodes + 1):
        assert hx.name(se) == ox.name(se)
        assert hx.parent(se) == ox.parent(se)
        assert kinds[se] == ox.kind(se)
        a, b = hx.blen(se), ox.blen(se)
        assert (np.isnan(a) and np.isnan(b)) or a == b
    pp, npo = hx.positions()
    op, on = ox.positions()
    assert pp.tolist() == op.tolist() and npo.tolist() == on.tolist()
    la = hx.lib_arrays(0)
    # on-disk layout: inc has nrows u64, cmer pairs sorted+unique per bucket
    raw = open(os.path.join(toy_index_dir, "inc-m4r1-frac"), "rb").read()
    assert np.frombuffer(raw[:4], "<u4")[0] == len(la["inc"]) == 8192
    assert la["inc"][-1] == len(la["cmer"]) == ox.info.nkmers
    start = 0
    for end in la["inc"][:2000]:
        e = la["cmer"][start:int(end), 0]
        assert np.all(e[1:] > e[:-1])
        start = int(end)
    # rho is halved at load for m4r1-frac (src/index.cpp:188-201)
    disk = open(os.path.join(toy_index_dir, "crecord-m4r1-frac"), "rb").read()
    nn, ns = np.frombuffer(disk[:8], "<u4")
    rho_disk = np.frombuffer(disk[8 + 8 * ns:], "<f8")
    assert np.allclose(la["rho"], rho_disk * 0.5) and nn == 48


def test_index_errors_are_reported(capi, tmp_path):
    with pytest.raises(capi.KrError) as e:
        capi.HostIndex(str(tmp_path / "nope"))
    assert e.value.code == capi.KR_ERR_IO
    d = tmp_path / "broken"
    d.mkdir()
    (d / "cmer-m4r1-frac").write_bytes(b"\0" * 8)
    (d / "metadata-m4r1-frac").write_bytes(b"\0" * 16)
    with pytest.raises(capi.KrError) as e:
        capi.HostIndex(str(d))
    assert e.value.code == capi.KR_ERR_FORMAT and "missing file" in str(e.value)


def test_incompatible_partial_libraries_rejected(capi, tmp_path):
    d = str(tmp_path / "ix")
    pse = [(0, 0)] * 4
    write_index(d, 21, 7, 4, 0, False, [20, 19, 17, 13, 6, 4, 2], {}, pse, [0.0] * 4, nwk="(a:1,b:1);")
    write_index(d, 21, 7, 4, 1, False, [20, 19, 17, 13, 6, 4, 3], {}, pse, [0.0] * 4, nwk="(a:1,b:1);")
    with pytest.raises(capi.KrError) as e:
        capi.HostIndex(d)
    assert "incompatible hash functions" in str(e.value)


def test_fastx_reader_matches_reference_kseq(capi, tmp_path):
    ks = json.load(open(os.path.join(GOLDEN, "kseq_ref.json")))
    names, bases, offs = capi.read_fastx(os.path.join(GOLDEN, "query_toy.fq"))
    seqs = [bytes(bases[int(offs[i]):int(offs[i + 1])]).decode() for i in range(len(names))]
    q = ks["query_toy"]
    assert len(names) == q["n"] == 100
    assert hashlib.sha256("\n".join(names).encode()).hexdigest() == q["sha_names"]
    assert hashlib.sha256("\n".join(seqs).encode()).hexdigest() == q["sha_seqs"]
    for key, text_key, gz in (("edge", "edge_text", False), ("trunc", "trunc_text", False), ("edge_gz", "edge_text", True)):
        p = tmp_path / (key + (".gz" if gz else ".fx"))
        if gz:
            with gzip.open(p, "wt") as f:
                f.write(ks[text_key])
        else:
            p.write_text(ks[text_key])
        for mb in (1, 76800):  # record-at-a-time and reference batch size
            n2, b2, o2 = capi.read_fastx(str(p), min_bases=mb)
            s2 = [bytes(b2[int(o2[i]):int(o2[i + 1])]).decode() for i in range(len(n2))]
            assert n2 == ks[key]["names"] and s2 == ks[key]["seqs"], key
    # kseq counts quality by length, so a short quality line swallows the next header (2 records)
    assert ks["trunc"]["n"] == 2
    # a quality string cut off by EOF ends reading like EOF (src/rqseq.cpp:189, src/kseq.h:215: -2)
    p = tmp_path / "eof.fq"
    p.write_text("@a\nACGT\n+\nIIII\n@b\nACGTAC\n+\nIII")
    n3, b3, o3 = capi.read_fastx(str(p))
    assert n3 == ["a"] and bytes(b3) == b"ACGT"
    import pyoracle
    ref = pyoracle.ref()
    if ref is not None:
        nb, sb, last = C.create_string_buffer(1024), C.create_string_buffer(1024), C.c_int()
        assert ref.ref_kseq_parse(str(p).encode(), nb, 1024, sb, 1024, C.byref(last)) == 1 and last.value == -2


def _read_both_ways(capi, path, min_bases):
    """the same file through the sequential reader and through the parallel chunk parser"""
    out = []
    for threads in ("0", "3"):
        os.environ["KR_FASTX_THREADS"], os.environ["KR_FASTX_PAR_MIN"] = threads, "0"
        try:
            st = {}
            n, b, o = capi.read_fastx(str(path), min_bases=min_bases, stats=st)
        finally:
            del os.environ["KR_FASTX_THREADS"], os.environ["KR_FASTX_PAR_MIN"]
        out.append((n, [bytes(b[int(o[i]):int(o[i + 1])]) for i in range(len(n))], st["parallel_chunks"]))
    return out


def test_parallel_fastq_reader_equals_sequential(capi, tmp_path):
    """Plain files are cut into chunks at guessed record starts and parsed by a thread pool; whatever the
    input, the records must be those of the sequential kseq-rule reader (which the golden vectors pin)."""
    rng = np.random.default_rng(5)
    qual_alphabet = np.frombuffer(bytes(range(33, 127)), np.uint8)

    def record(i, n=None):
        n = int(rng.integers(1, 200)) if n is None else n
        seq = rng.choice(np.frombuffer(b"ACGTN", np.uint8), n).tobytes()
        qual = rng.choice(qual_alphabet, n).tobytes()
        if i % 7 == 0:
            qual = b"@" + qual[1:]  # a quality line that looks like a header
        if i % 11 == 0:
            qual = b"+" + qual[1:]
        return b"@r%d some comment\n" % i + seq + b"\n+\n" + qual + b"\n"

    clean = [record(i) for i in range(4000)]
    ks = json.load(open(os.path.join(GOLDEN, "kseq_ref.json")))
    cases = {
        "clean": b"".join(clean),
        "no_final_newline": b"".join(clean)[:-1],
        "wrapped_in_the_middle": b"".join(clean[:1500]) + b"@w x\nACGT\nACGT\n+\nIIII\nIIII\n" + b"".join(clean[1500:]),
        "fasta_tail": b"".join(clean[:2500]) + b">f1\nACGTACGT\nACGT\n>f2\nAC\n",
        "crlf": b"".join(clean[:800]) + b"".join(clean[800:900]).replace(b"\n", b"\r\n") + b"".join(clean[900:1200]),
        "leading_junk": b"junk\n" + b"".join(clean[:300]),
        "edge_vectors_after_clean": b"".join(clean[:2000]) + ks["edge_text"].encode(),
        "truncated_after_clean": b"".join(clean[:2000]) + ks["trunc_text"].encode(),
        "fasta_only": b"".join(b">s%d\n" % i + rng.choice(np.frombuffer(b"ACGT", np.uint8), 300).tobytes() + b"\n" for i in range(300)),
        "empty": b"",
    }
    for key, data in cases.items():
        p = tmp_path / (key + ".fq")
        p.write_bytes(data)
        for mb in (1, 5000):  # 4 KB and 10 KB chunks: hundreds of chunk boundaries per file
            seq_res, par_res = _read_both_ways(capi, p, mb)
            assert par_res[0] == seq_res[0], (key, mb)
            assert par_res[1] == seq_res[1], (key, mb)
            assert seq_res[2] == 0
            if key in ("clean", "no_final_newline", "wrapped_in_the_middle", "fasta_tail", "crlf"):
                assert par_res[2] > 10, (key, mb, par_res[2])  # the pool did parse most of the file
            if key in ("leading_junk", "fasta_only", "empty"):
                assert par_res[2] == 0
        if key in ("clean", "fasta_tail", "empty", "leading_junk"):  # batches held by the caller while the reader goes on (kr_fastx_detach / release)
            for threads in ("0", "3"):
                os.environ["KR_FASTX_THREADS"], os.environ["KR_FASTX_PAR_MIN"] = threads, "0"
                try:
                    n, b, o = capi.read_fastx(str(p), min_bases=1, detach=3)
                    n5, b5, o5 = capi.read_fastx(str(p), min_bases=5000, detach=1)
                finally:
                    del os.environ["KR_FASTX_THREADS"], os.environ["KR_FASTX_PAR_MIN"]
                assert n == seq_res[0] == n5 and [bytes(b[int(o[i]):int(o[i + 1])]) for i in range(len(n))] == seq_res[1]
                assert [bytes(b5[int(o5[i]):int(o5[i + 1])]) for i in range(len(n5))] == seq_res[1]
        # a batch put together from several chunks (what a million-read batch is: chunks are capped at 96 MB)
        os.environ["KR_FASTX_CHUNK_MAX"] = "4096"
        try:
            seq_m, par_m = _read_both_ways(capi, p, 30_000)
        finally:
            del os.environ["KR_FASTX_CHUNK_MAX"]
        assert par_m[0] == seq_res[0] and par_m[1] == seq_res[1], key
        if key in ("clean", "no_final_newline", "crlf"):
            assert par_m[2] > 10
        if key == "clean":
            assert seq_res[0] == ["r%d" % i for i in range(4000)]
            assert seq_res[1] == [r.split(b"\n")[1] for r in clean]
        if key == "edge_vectors_after_clean":
            assert seq_res[0][2000:] == ks["edge"]["names"]


def _bgzf_bytes(data, block=60000, level=1):
    """block-gzipped form of `data`: independent gzip members with a BC extra field, plus the empty EOF member"""
    import struct
    import zlib
    out = bytearray()
    chunks = [data[i:i + block] for i in range(0, len(data), block)] + [b""]
    for c in chunks:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(c) + co.flush()
        bsize = 12 + 6 + len(comp) + 8 - 1
        out += b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
        out += comp + struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c))
    return bytes(out)


def test_steady_batches_of_the_pool_reader_are_one_chunk_each(capi, tmp_path):
    """A batch of ordinary FASTQ is ONE parsed chunk, handed over as it is (the CLI's zero-copy path, kr_fastx_detach, needs a batch
    that fits its job): a chunk of 2 * min_bases bytes holds about 0.94 * min_bases bases, which is a batch -- asking for all of
    min_bases took a second chunk for every steady batch, appended by copy (37,736-read batches for a request of 20,000 reads)."""
    rng = np.random.default_rng(3)
    n, L = 200_000, 150
    seqs = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(n, L))
    with open(tmp_path / "r.fq", "wb") as f:
        for i in range(n):
            f.write(b"@read%07d\n" % i + seqs[i].tobytes() + b"\n+\n" + b"I" * L + b"\n")
    lib = capi.load()
    h = C.c_void_p()
    capi.check(lib.kr_fastx_open(os.fsencode(str(tmp_path / "r.fq")), C.byref(h)))
    sizes, total = [], 0
    try:
        while True:
            b = capi.KrFastxBatch()
            capi.check(lib.kr_fastx_next(h, 20_000 * L, C.byref(b)))
            sizes.append(int(b.nreads))
            total += int(b.nreads)
            if not b.more:
                break
        lib.kr_fastx_parallel_chunks.restype = C.c_uint64
        lib.kr_fastx_parallel_chunks.argtypes = [C.c_void_p]
        chunks = int(lib.kr_fastx_parallel_chunks(h))
    finally:
        lib.kr_fastx_close(h)
    assert total == n
    steady = sizes[-6:-2]  # (the first chunks are short on purpose -- a ramp of growing chunks --, the last batches are what is left)
    assert steady and max(sizes) <= 20_000 and min(steady) >= 17_500, sizes
    assert chunks >= len(sizes) - 1 and chunks <= len(sizes) + 1, (chunks, len(sizes))


def test_block_gzipped_input_is_inflated_in_parallel(capi, tmp_path):
    """BGZF files (bgzip, many sequencing pipelines): the members are inflated by several threads; the records are
    those of the plain file, standard tools still read the file, a damaged member is an error, not a silent stop."""
    rng = np.random.default_rng(9)
    recs = []
    for i in range(9000):
        n = int(rng.integers(1, 260))
        recs.append(b"@r%d c\n" % i + rng.choice(np.frombuffer(b"ACGTN", np.uint8), n).tobytes() + b"\n+\n" + b"I" * n + b"\n")
    data = b"".join(recs) + b">tail\nACGT\nAC\n"
    plain, bg = tmp_path / "a.fq", tmp_path / "a.fq.gz"
    plain.write_bytes(data)
    bg.write_bytes(_bgzf_bytes(data))
    assert gzip.open(bg).read() == data  # a valid multi-member gzip file
    os.environ["KR_FASTX_PAR_MIN"] = "0"
    try:
        want = capi.read_fastx(str(plain), min_bases=50000)
        for threads in ("3", "1"):
            os.environ["KR_FASTX_THREADS"] = threads
            st = {}
            got = capi.read_fastx(str(bg), min_bases=50000, stats=st)
            assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
            assert st["gzip_chunks"]["parsed"] >= 1, st  # the threads that inflate a run of members parse its records too (kr_pgz.inc: Stitch)
        os.environ["KR_FASTX_THREADS"] = "0"  # zlib stream reader on the same file
        got = capi.read_fastx(str(bg), min_bases=50000)
        assert got[0] == want[0] and np.array_equal(got[1], want[1])
        os.environ["KR_FASTX_THREADS"] = "3"
        raw = bytearray(bg.read_bytes())
        raw[len(raw) // 2] ^= 0x55  # flip bits in the middle of some member
        bad = tmp_path / "bad.fq.gz"
        bad.write_bytes(bytes(raw))
        with pytest.raises(capi.KrError):
            capi.read_fastx(str(bad), min_bases=50000)
        # block-gzipped members followed by an ordinary gzip member (cat a.bgz b.gz): zlib takes over where they end
        extra = b"@x1\nACGTACGT\n+\nIIIIIIII\n>x2\nAC\n"
        mixed = tmp_path / "mixed.fq.gz"
        mixed.write_bytes(_bgzf_bytes(data[: len(data) - 14])[:-28] + gzip.compress(extra))  # without the EOF member
        plain2 = tmp_path / "b.fq"
        plain2.write_bytes(data[: len(data) - 14] + extra)
        w2 = capi.read_fastx(str(plain2), min_bases=50000)
        g2 = capi.read_fastx(str(mixed), min_bases=50000)
        assert g2[0] == w2[0] and np.array_equal(g2[1], w2[1]) and g2[0][-2:] == ["x1", "x2"]
        trunc = tmp_path / "trunc.fq.gz"
        trunc.write_bytes(bg.read_bytes()[: len(raw) // 2])
        with pytest.raises(capi.KrError):
            capi.read_fastx(str(trunc), min_bases=50000)
    finally:
        os.environ.pop("KR_FASTX_PAR_MIN", None)
        os.environ.pop("KR_FASTX_THREADS", None)


def test_block_gzipped_edge_cases(capi, tmp_path):
    """(1) the 1 MB run cut lands exactly before the empty EOF member, so a fresh run holds nothing but it: a valid
    file, not a damaged one; (2) a member whose extra field claims more bytes than the member has, and (3) a trailer
    that claims more than 64 KB, are errors (no over-read, no giant allocation)."""
    import struct
    rng = np.random.default_rng(3)
    recs, size = [], 0
    while size < 18 * 60000:
        n = int(rng.integers(100, 200))
        rec = b"@q%d\n" % len(recs) + rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes() + b"\n+\n" + b"I" * n + b"\n"
        recs.append(rec)
        size += len(rec)
    data = b"".join(recs)
    # stored (level 0) members of 60,000 bytes: 17 of them stay below 1 MB, the 18th crosses it and ends the run
    data = data[: 18 * 60000 - 200] + b"@last\n" + b"A" * 90 + b"\n+\n" + b"I" * 90 + b"\n"
    data = data + b"\n" * (18 * 60000 - len(data))
    assert len(data) == 18 * 60000
    bg = tmp_path / "edge.fq.gz"
    raw = _bgzf_bytes(data, block=60000, level=0)
    assert len(raw) - 28 >= (1 << 20) and len(raw) - 28 - (len(raw) - 28) // 18 < (1 << 20)
    bg.write_bytes(raw)
    plain = tmp_path / "edge.fq"
    plain.write_bytes(data)
    os.environ["KR_FASTX_PAR_MIN"] = "0"
    try:
        want = capi.read_fastx(str(plain), min_bases=50000)
        for threads in ("16", "2"):
            os.environ["KR_FASTX_THREADS"] = threads
            got = capi.read_fastx(str(bg), min_bases=50000)
            assert got[0] == want[0] and np.array_equal(got[1], want[1]) and got[0][-1] == "last"
        os.environ["KR_FASTX_THREADS"] = "2"
        good = _bgzf_bytes(b"@a\nACGT\n+\nIIII\n")
        # (2) xlen = 28 in a 40-byte member: 12 + 28 + 8 > 40
        evil = b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 28) + b"BC" + struct.pack("<HH", 2, 39) + b"XX" + struct.pack("<H", 18) + b"\x00" * 18
        assert len(evil) == 40
        f2 = tmp_path / "evil_xlen.fq.gz"
        f2.write_bytes(good[:-28] + evil + good[-28:])
        with pytest.raises(capi.KrError):
            capi.read_fastx(str(f2), min_bases=50000)
        # (3) a trailer that claims 16 MB
        m = bytearray(good[:-28])
        m[-4:] = struct.pack("<I", 1 << 24)
        f3 = tmp_path / "evil_isize.fq.gz"
        f3.write_bytes(bytes(m) + good[-28:])
        with pytest.raises(capi.KrError):
            capi.read_fastx(str(f3), min_bases=50000)
    finally:
        os.environ.pop("KR_FASTX_PAR_MIN", None)
        os.environ.pop("KR_FASTX_THREADS", None)


def test_builder_output_is_consistent_with_brute_force(capi, po, synth, tmp_path):
    """Independent check of the CPU builder: recompute, in pure Python, the minimizers of tiny
    genomes (src/rqseq.cpp:51-144 semantics) and the genome set of every indexed k-mer, and
    compare with what the colour DAG of the built index expands to."""
    nwk = "((a:0.05,b:0.05)n1:0.05,(c:0.1,d:0.1,e:0.1)n2:0.02);"
    g = synth.evolve_genomes(nwk, 3000, seed=3)
    tsv = synth.write_genomes(g, str(tmp_path / "g"), contigs=2)
    (tmp_path / "t.nwk").write_text(nwk)
    idx = str(tmp_path / "ix")
    k, w, h, m, r = 21, 27, 7, 4, 1
    capi.build_index(tsv, idx, nwk=str(tmp_path / "t.nwk"), k=k, w=w, h=h, m=m, r=r, frac=True, ppos=[20, 19, 17, 13, 6, 4, 2])
    hx = capi.HostIndex(idx)
    ppos, npos = hx.positions()
    la = hx.lib_arrays(0)
    kinds = hx.kinds()

    def fmix(x):
        x ^= x >> 33
        x = (x * 0xff51afd7ed558ccd) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 33
        x = (x * 0xc4ceb9fe1a85ec53) & 0xFFFFFFFFFFFFFFFF
        return x ^ (x >> 33)

    want = {}  # (row, enc32) -> set of leaf names
    for name, seq in g.items():
        s = seq.tobytes().decode()
        n = len(s)
        for c in range(2):  # two contigs, as written
            t = s[n * c // 2: n * (c + 1) // 2]
            ldiff = w - k + 1
            win = [(0, 0)] * ldiff
            kix = 0
            for i in range(k, len(t) + 1):
                f = closed_form(t[i - k:i], ppos, npos)
                win[kix % ldiff] = (f[0], fmix(f[0]))
                kix += 1
                l = i  # no N in these genomes: run length == i
                if l < w and i != len(t):
                    continue
                mn = min(win, key=lambda e: e[1])  # first minimum in slot order
                km_bp = mn[0]
                # decode enc_bp back to rix / enc32 through the closed form on the codes
                codes = [(km_bp >> (2 * p)) & 3 for p in range(k)]
                P, N = sorted(ppos.tolist()), sorted(npos.tolist())
                rix = sum(codes[P[j]] << (2 * j) for j in range(len(P)))
                enc = sum(((codes[N[j]] & 1) << j) | ((codes[N[j]] >> 1) << (16 + j)) for j in range(len(N)))
                row = row_of(rix, m, r, True)
                if row is not None:
                    want.setdefault((row, enc), set()).add(name)

    def expand(se):
        out, st = set(), [int(se)]
        while st:
            x = st.pop()
            if x == 0:
                continue
            if x <= hx.nnodes and kinds[x] == 1:
                out.add(hx.name(x))
                continue
            a, b = la["pse"][x]
            st += [int(a), int(b)]
        return out

    got = {}
    start = 0
    for row, end in enumerate(la["inc"]):
        for enc, se in la["cmer"][start:int(end)]:
            got[(row, int(enc))] = expand(se)
        start = int(end)
    assert got == want
    assert len(got) > 500 and max(len(v) for v in got.values()) >= 3
    # rho ~ 2/(w-k+2), halved at load
    leaf_rho = [la["rho"][se] for se in range(1, hx.nnodes + 1) if kinds[se] == 1]
    assert all(0.08 < x < 0.2 for x in leaf_rho)


def test_lsh_position_draw_matches_reference_default_seed(capi, synth, tmp_path):
    # without --seed, `index -k 27 -h 11` draws these positions (libstdc++)
    g = synth.evolve_genomes("(a:0.1,b:0.1);", 2000, seed=1)
    tsv = synth.write_genomes(g, str(tmp_path / "g"))
    idx = str(tmp_path / "ix")
    capi.build_index(tsv, idx, k=27, w=35, h=11, m=4, r=1, frac=True)
    pp, _ = capi.HostIndex(idx).positions()
    assert pp.tolist() == [26, 24, 22, 21, 17, 14, 8, 7, 5, 3, 2]
    assert os.path.exists(os.path.join(idx, "reflist-m4r1-frac")) and not os.path.exists(os.path.join(idx, "tree-m4r1-frac"))
    assert os.path.getsize(os.path.join(idx, "metadata-m4r1-frac")) == 16 + 27
    assert os.path.getsize(os.path.join(idx, "inc-m4r1-frac")) == 4 + 8 * 2 ** 21


def test_device_entry_points_fail_loudly_without_gpu(capi, toy_index_dir):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    hx = capi.HostIndex(toy_index_dir)
    with pytest.raises(capi.KrError) as e:
        hx.upload(0)
    assert e.value.code == capi.KR_ERR_NO_DEVICE


def test_round2_entry_points_reject_bad_arguments_without_gpu(capi):
    """kr_index_broadcast / kr_place_stream / kr_batch_* check their arguments before anything touches a device."""
    import ctypes as C
    lib = capi.load()
    outs = (C.c_void_p * 1)()
    devs = (C.c_int * 1)(0)
    assert lib.kr_index_broadcast(None, 1, devs, outs) == capi.KR_ERR_ARG
    assert b"kr_index_broadcast" in lib.kr_last_error()
    txt, ln = C.c_void_p(), C.c_uint64()
    prev = C.c_int(0)
    p = capi.default_params()
    assert lib.kr_place_stream(None, None, None, None, 1, None, None, C.byref(p), 0, C.byref(prev), C.byref(txt), C.byref(ln), None, None) == capi.KR_ERR_ARG
    assert lib.kr_batch_wait(None) == capi.KR_ERR_STATE
    assert lib.kr_batch_submit(None, None, None, 1, 0) == capi.KR_ERR_ARG
    lib.kr_host_free(None)  # a no-op


def test_device_formatter_rounding_is_printfs(capi):
    """"%.5f" as the device formatter prints it (fixed5_exact / fixed5_digits, csrc/kr_common.h: round-half-even of the exact binary
    value through one fma) against C's own conversion (Python's % operator calls it: correctly rounded): random values over the
    minimiser's range and beyond, every exact tie j / 2^q with its two neighbours, the decimal ties (i + 1/2) / 10^5 as doubles."""
    import ctypes as C
    import random
    lib = capi.load()
    buf = C.create_string_buffer(64)
    def chk(v):
        n = lib.kr_debug_fixed5(v, buf)
        assert n and buf.value.decode() == "%.5f" % v, (v.hex(), buf.value, "%.5f" % v)
    rng = random.Random(5)
    for _ in range(200_000):
        u = rng.random()
        chk(u * 0.5), chk(u * u * u * 1e-3), chk(u * 999.99)
    for q in range(1, 27):
        for j in range(0, 1 << min(q, 11)):
            v = j / float(1 << q)
            chk(v), chk(np.nextafter(v, 1.0)), chk(np.nextafter(v, -1.0) if v > 0 else 0.0)
    for i in range(0, 60_000, 7):
        v = (i + 0.5) / 1e5
        chk(v), chk(float(np.nextafter(v, 1.0))), chk(float(np.nextafter(v, 0.0)))
    for v in (0.0, 1e-10, 0.5, 0.499995, 9.999995, 99.999995, 999.99999):
        chk(v)
    for v in (-1e-9, 1000.0, 999.999995, 999.9999999, float(np.nextafter(1000.0, 0.0)), float("nan"), float("inf")):
        assert lib.kr_debug_fixed5(v, buf) == 0  # (values that round to "1000.00000" are out of range too: ten bytes)
    chk(999.99999), chk(999.999994)
    # the text entry points check their arguments before anything touches a device
    txt, ln = C.c_void_p(), C.c_uint64()
    assert lib.kr_stream_text_enable(None, None, 1, 1) == capi.KR_ERR_ARG
    assert lib.kr_batch_submit_text(None, None, None, 1, 0, None, None, 1) == capi.KR_ERR_ARG
    assert lib.kr_batch_collect_text(None, C.byref(txt), C.byref(ln)) == capi.KR_ERR_ARG


def test_place_rows_print_numbers_as_printf_does(capi):
    """The numbers of a `krepp place` row (pendant and distal lengths, -v_llh, LWR, d: std::fixed with 5 decimals, src/query.hpp:202-206)
    as the host's last phase prints them since round 6 (place_num, kr_place.cpp: the scaled value's own fraction away from a tie, the
    exact routine next to one, printf beyond 1000): equal to printf's "%.5f" on random values of both signs, on every j / 2^q tie and
    its neighbours, around the range's ends, NaN."""
    lib = capi.load()
    lib.kr_debug_place_fixed5.restype = C.c_int
    lib.kr_debug_place_fixed5.argtypes = [C.c_double, C.c_char_p]
    buf = C.create_string_buffer(96)

    def chk(v):
        n = lib.kr_debug_place_fixed5(float(v), buf)
        got = buf.value.decode()
        want = "%.5f" % v if v == v else ("-nan" if np.signbit(v) else "nan")
        assert n == len(got) and got == want, (v, got, want)

    rng = np.random.default_rng(11)
    for v in rng.random(20000) * 0.5:
        chk(v), chk(-v)
    for v in rng.random(20000) * 1000.0:
        chk(v), chk(-v)
    for v in np.exp(rng.normal(0, 6, 5000)):
        chk(v), chk(-v)
    for q in range(1, 22):
        for j in range(0, 1 << min(q, 9)):
            v = j / float(1 << q)
            for w in (v, float(np.nextafter(v, 1.0)), float(np.nextafter(v, -1.0)), v + 417.0):
                chk(w), chk(-w)
    for i in range(0, 60_000, 11):
        v = (i + 0.5) / 1e5
        chk(v), chk(float(np.nextafter(v, 1.0))), chk(float(np.nextafter(v, 0.0))), chk(-v)
    for v in (0.0, -0.0, 1e-10, -1e-10, 999.99999, 999.999994, 999.999995, 999.9999999, 1000.0, 1000.000004, 12345.678915, -98765.4321, 1e15, -1e15,
              float("nan"), -float("nan"), float("inf"), -float("inf")):
        chk(v)


def test_cli_usage_errors(capi):
    exe = os.path.join(ROOT, "krepp_amd", "lib", "krepp")
    r = subprocess.run([exe, "seek"], capture_output=True, text=True)
    assert r.returncode == 1 and "[ERROR]" in r.stderr and "krepp version: v0.8.3" in r.stderr
    r = subprocess.run([exe, "dist", "-i", "/nonexistent", "-q", "/nonexistent"], capture_output=True, text=True)
    assert r.returncode == 1 and "[ERROR]" in r.stderr


def test_cli_index_matches_library_builder(capi, synth, tmp_path):
    g = synth.evolve_genomes("((a:0.03,b:0.03):0.02,(c:0.05,d:0.05):0.01);", 8000, seed=4)
    tsv = synth.write_genomes(g, str(tmp_path / "g"))
    nwk = tmp_path / "t.nwk"
    nwk.write_text("((a:0.03,b:0.03):0.02,(c:0.05,d:0.05):0.01);")
    exe = os.path.join(ROOT, "krepp_amd", "lib", "krepp")
    d1, d2 = str(tmp_path / "cli"), str(tmp_path / "lib")
    r = subprocess.run([exe, "index", "-i", tsv, "-o", d1, "-t", str(nwk), "-k", "21", "-w", "27", "-h", "7", "--num-threads", "2"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    capi.build_index(tsv, d2, nwk=str(nwk), k=21, w=27, h=7)
    for f in ("cmer", "inc", "crecord", "metadata", "tree", "reflist"):
        a = open(os.path.join(d1, f + "-m4r1-frac"), "rb").read()
        b = open(os.path.join(d2, f + "-m4r1-frac"), "rb").read()
        assert a == b, f
    # default -h is k-16 and -w is k+6 when -w is not given (src/krepp.cpp:579-582)
    d3 = str(tmp_path / "def")
    r = subprocess.run([exe, "index", "-i", tsv, "-o", d3, "-k", "23", "-m", "64", "-r", "0", "--no-frac"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    md = open(os.path.join(d3, "metadata-m64r0-no_frac"), "rb").read()
    assert (md[0], md[1], md[2]) == (23, 29, 7)


def test_cli_help_and_usage_errors():
    """No GPU needed: --help lists the sub-commands and their options; misuse ends like the reference's error_exit
    (`[ERROR] ...` on stderr, status 1; src/common.cpp:20-24)."""
    exe = os.path.join(ROOT, "krepp_amd", "lib", "krepp")
    r = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert r.returncode == 0
    for word in ("dist", "place", "seek", "index", "sketch", "--hdist-th", "--lineage-file", "--summarize", "--tabular", "--kmer-len"):
        assert word in r.stdout, word
    r = subprocess.run([exe, "place", "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--tau" in r.stdout and "krepp index -i MAP.tsv" not in r.stdout
    for args in (["dist"], ["bogus"], [], ["dist", "-i", "/nonexistent", "-q", "/nonexistent"], ["sketch", "-i", "x"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True)
        assert r.returncode == 1 and "[ERROR]" in r.stderr, (args, r.stderr)
    r = subprocess.run([exe, "dist", "-i", os.path.join(GOLDEN, "toy_index"), "-q", os.path.join(GOLDEN, "toy_reads.fq"), "--dist-max", "0.5"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "--dist-max" in r.stderr


def _random_colour_table(rng, nleaves, nextra):
    """A random binary tree in post-order numbering (se 1..N, leaves and internal nodes, a few of them null) followed by
    `nextra` colours that are unions of two earlier ids (the way the index builder makes them, src/record.cpp), some of
    them naming ids the table does not define."""
    kind, pse, stack = [0], [(0, 0)], []
    made = 0
    while made < nleaves or len(stack) > 1:
        if made < nleaves and (len(stack) < 2 or rng.random() < 0.5):
            kind.append(1 if rng.random() > 0.03 else 0)  # a null leaf now and then
            pse.append((0, len(kind) - 1))  # (leaf se -> (0, se))
            stack.append(len(kind) - 1)
            made += 1
        else:
            b, a = stack.pop(), stack.pop()
            kind.append(2 if rng.random() > 0.02 else 0)  # a null internal node now and then
            pse.append((a, b))
            stack.append(len(kind) - 1)
    nnodes = len(kind) - 1
    for _ in range(nextra):
        n = len(pse)
        a = int(rng.integers(0, n + 3)) if rng.random() < 0.05 else int(rng.integers(1, n))  # sometimes undefined ids
        b = int(rng.integers(1, n))
        if rng.random() < 0.5 and n > nnodes + 2:  # chains: a colour that grew by one part at a time
            a = n - 1
        pse.append((a, b))
    return np.array(kind, np.uint8), np.array(pse, np.uint32), nnodes


def test_flat_colours_list_exactly_the_leaves_the_walk_reaches(capi):
    """kr_index_upload's host pass (colour_classes): a colour tagged flat must name -- as a run of ranks or as a list --
    exactly the set of leaves that the breadth-first walk of src/query.cpp:369-387 reaches from it; leaves carry their
    rank; null nodes and undefined ids are dropped; the parts of a walked colour carry the tags of those colours."""
    import sys
    rng = np.random.default_rng(20261003)
    for nleaves, nextra in ((2, 3), (9, 40), (60, 400), (300, 3000)):
        kind, pse, nnodes = _random_colour_table(rng, nleaves, nextra)
        n = len(pse)
        rank_of = {}
        for se in range(1, nnodes + 1):
            if kind[se] == 1:
                rank_of[se] = len(rank_of)
        memo = {}
        sys.setrecursionlimit(100000)

        def walk(se):  # the set of leaf ranks the reference's expansion updates
            if se == 0 or se >= n:
                return frozenset()
            if se in memo:
                return memo[se]
            if se <= nnodes and kind[se] == 0:
                r = frozenset()
            elif se <= nnodes and kind[se] == 1:
                r = frozenset([rank_of[se]])
            else:
                r = walk(int(pse[se][0])) | walk(int(pse[se][1]))
            memo[se] = r
            return r

        cls, dev, lists = capi.colour_classes(pse, kind)
        nflat = nrun = nlist = 0
        for se in range(1, n):
            c, low = int(cls[se]) >> 30, int(cls[se]) & 0x3FFFFFFF
            if se <= nnodes and kind[se] == 0:
                assert cls[se] == 0
            elif se <= nnodes and kind[se] == 1:
                assert c == 1 and low == rank_of[se]
            else:
                assert c in (2, 3) and low == se  # the id survives (the hits tap reports it)
                if c == 3:
                    base, cnt = int(dev[se][0]), int(dev[se][1])
                    ranks = lists[(base & 0x7FFFFFFF): (base & 0x7FFFFFFF) + cnt].tolist() if base >> 31 else list(range(base, base + cnt))
                    assert ranks == sorted(walk(se)), se
                    nflat += 1
                    nlist += base >> 31
                    nrun += 1 - (base >> 31)
                else:
                    for part, tagged in zip(pse[se].tolist(), dev[se].tolist()):
                        assert tagged == (int(cls[part]) if part < n else 0)
                    # walked only when it has to be: a part that is walked itself (or defined later), or more than 64
                    # leaves counting a leaf reached through both parts twice
                    def size(part):
                        if part == 0 or part >= n or (int(cls[part]) >> 30) == 0:
                            return 0
                        return 1 if (int(cls[part]) >> 30) == 1 else (int(dev[part][1]) if (int(cls[part]) >> 30) == 3 else 1 << 30)
                    a_, b_ = pse[se].tolist()
                    assert a_ >= se or b_ >= se or size(a_) + size(b_) > 64, se
        if nleaves >= 60:
            assert nrun > 10 and nlist > 10 and nflat > 0.3 * nextra
