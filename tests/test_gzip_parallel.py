"""Ordinary gzip query files inflated and parsed in parallel (krepp_amd/csrc/kr_pgz.inc).

The reference reads queries with kseq over gzread (src/rqseq.cpp:161-197): whatever the parallel path does, the records
must be those the sequential kseq-rule reader produces from the same bytes (which tests/golden/kseq_ref.json pins), a damaged
file must be an error as it is with zlib, and nothing may depend on how the compressor cut the stream into blocks."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest


def _fastq(n, seed=3, name=b"SRR77.%d %d/1", lo=60, hi=200, long_every=0):
    rng = np.random.default_rng(seed)
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 300000)]
    quals = np.frombuffer(bytes(range(35, 75)), np.uint8)
    out = []
    for i in range(n):
        L = int(rng.integers(lo, hi))
        if long_every and i % long_every == 0:
            L = 30000
        p = int(rng.integers(0, len(genome) - L))
        q = quals[np.minimum(39, rng.integers(0, 60, L))].tobytes()
        if i % 13 == 0:
            q = b"@" + q[1:]  # quality lines that look like headers ...
        if i % 17 == 0:
            q = b"+" + q[1:]  # ... or like separators
        out.append(b"@" + name % (i, i) + b"\n" + genome[p:p + L].tobytes() + b"\n+\n" + q + b"\n")
    return out


def _deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0, flush_mode=zlib.Z_SYNC_FLUSH):
    """a gzip member made by hand (so that strategy and flush points can be chosen)"""
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    body = bytearray()
    if flush_every:
        for i in range(0, len(data), flush_every):
            body += co.compress(data[i:i + flush_every])
            body += co.flush(flush_mode)
    else:
        body += co.compress(data)
    body += co.flush()
    return b"\x1f\x8b\x08\x00" + b"\x00" * 4 + b"\x00\x03" + bytes(body) + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


@pytest.fixture()
def par_env():
    keep = {k: os.environ.get(k) for k in ("KR_FASTX_PAR_MIN", "KR_FASTX_THREADS", "KR_PGZ", "KR_PGZ_CHUNK")}
    os.environ["KR_FASTX_PAR_MIN"] = "0"
    os.environ["KR_FASTX_THREADS"] = "3"
    os.environ["KR_PGZ_CHUNK"] = "150000"
    yield os.environ
    for k, v in keep.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _same(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_records_equal_the_plain_files_whatever_the_compressor_did(capi, tmp_path, par_env):
    recs = _fastq(12000)
    data = b"".join(recs)
    plain = tmp_path / "a.fq"
    plain.write_bytes(data)
    want = capi.read_fastx(str(plain), min_bases=200000)
    assert len(want[0]) == 12000
    variants = {
        "gzip1": gzip.compress(data, 1), "gzip6": gzip.compress(data, 6), "gzip9": gzip.compress(data, 9),
        "fixed_huffman": _deflate(data, 6, zlib.Z_FIXED),
        "huffman_only": _deflate(data, 6, zlib.Z_HUFFMAN_ONLY),
        "rle": _deflate(data, 6, zlib.Z_RLE),
        "stored": _deflate(data, 0),
        "sync_flushes": _deflate(data, 6, flush_every=70001),                        # empty stored blocks in the stream (pigz)
        "full_flushes": _deflate(data, 6, flush_every=50021, flush_mode=zlib.Z_FULL_FLUSH),
        "two_members": gzip.compress(data[:900001], 6) + gzip.compress(data[900001:], 2),  # the cut is in the middle of a record
        "name_and_comment": b"\x1f\x8b\x08\x18" + b"\x00" * 4 + b"\x00\x03" + b"reads.fq\x00a comment\x00" + _deflate(data, 6)[10:],
    }
    for key, blob in variants.items():
        p = tmp_path / (key + ".fq.gz")
        p.write_bytes(blob)
        assert gzip.open(p).read() == data, key
        for chunk in ("150000", "65536", "1000000"):
            par_env["KR_PGZ_CHUNK"] = chunk
            st = {}
            got = capi.read_fastx(str(p), min_bases=200000, stats=st)
            assert _same(got, want), (key, chunk, st)
            if key in ("gzip1", "gzip6", "gzip9", "rle", "two_members") and chunk != "1000000":
                # the pool did the work: nearly every chunk was found, accepted and handed out with its records parsed
                n_chunks = -(-len(blob) // int(chunk))
                assert st["gzip_chunks"]["parsed"] >= n_chunks - 3, (key, chunk, st, n_chunks)
        par_env["KR_PGZ"] = "0"  # zlib's gzread alone
        assert _same(capi.read_fastx(str(p), min_bases=200000), want), key
        par_env.pop("KR_PGZ")


def test_input_that_is_not_clean_four_line_fastq_goes_to_the_sequential_parser(capi, tmp_path, par_env):
    recs = _fastq(6000, seed=8)
    rng = np.random.default_rng(1)
    fasta = b"".join(b">s%d d\n" % i + rng.choice(np.frombuffer(b"ACGT", np.uint8), 400).tobytes() + b"\n" + b"ACGTT\n" for i in range(2500))
    cases = {
        "no_final_newline": b"".join(recs)[:-1],
        "wrapped_in_the_middle": b"".join(recs[:3000]) + b"@w x\nACGT\nACGT\n+\nIIII\nIIII\n" + b"".join(recs[3000:]),
        "fasta_tail": b"".join(recs[:4000]) + fasta,
        "fasta_only": fasta,
        "crlf_part": b"".join(recs[:2000]) + b"".join(recs[2000:2300]).replace(b"\n", b"\r\n") + b"".join(recs[2300:]),
        "leading_junk": b"junk\n" + b"".join(recs),
        "long_reads": b"".join(_fastq(300, seed=4, long_every=3)),
        "one_record": recs[0],
        "empty": b"",
    }
    for key, data in cases.items():
        plain, gz = tmp_path / (key + ".fq"), tmp_path / (key + ".fq.gz")
        plain.write_bytes(data)
        gz.write_bytes(gzip.compress(data, 6))
        par_env["KR_FASTX_THREADS"] = "0"
        want = capi.read_fastx(str(plain), min_bases=100000)
        par_env["KR_FASTX_THREADS"] = "3"
        for chunk in ("65536", "200000"):
            par_env["KR_PGZ_CHUNK"] = chunk
            st = {}
            got = capi.read_fastx(str(gz), min_bases=100000, stats=st)
            assert _same(got, want), (key, chunk, st)
        if key in ("no_final_newline", "wrapped_in_the_middle", "fasta_tail", "long_reads"):
            assert st["gzip_chunks"]["parsed"] >= 2, (key, st)


def test_damaged_gzip_files_are_errors(capi, tmp_path, par_env):
    data = b"".join(_fastq(9000, seed=11))
    blob = gzip.compress(data, 6)
    good = tmp_path / "good.fq.gz"
    good.write_bytes(blob)
    assert len(capi.read_fastx(str(good), min_bases=100000)[0]) == 9000

    def must_fail(name, raw):
        p = tmp_path / name
        p.write_bytes(raw)
        for pgz in ("1", "0"):  # the parallel path reports what zlib's gzread reports
            par_env["KR_PGZ"] = pgz
            with pytest.raises(capi.KrError):
                capi.read_fastx(str(p), min_bases=100000)
        par_env.pop("KR_PGZ")

    for frac in (0.1, 0.5, 0.93):
        raw = bytearray(blob)
        raw[int(len(raw) * frac)] ^= 0x10
        must_fail("flip_%d.fq.gz" % int(frac * 100), bytes(raw))
    must_fail("truncated.fq.gz", blob[: len(blob) // 2])
    must_fail("no_trailer.fq.gz", blob[:-8])
    raw = bytearray(blob)
    raw[-8] ^= 1  # CRC-32 in the trailer
    must_fail("bad_crc.fq.gz", bytes(raw))
    raw = bytearray(blob)
    raw[-2] ^= 1  # ISIZE
    must_fail("bad_len.fq.gz", bytes(raw))
    must_fail("member_header_cut_short.fq.gz", blob + blob[:12])
    # what gzread ignores is ignored: bytes behind a member that are not a member (zero padding, garbage)
    for name, raw in (("garbage_behind.fq.gz", blob + b"not a gzip member at all, but long enough to look like one"),
                      ("padding_then_member.fq.gz", blob + b"\x00" * 37 + gzip.compress(b"@z\nAC\n+\nII\n"))):
        p = tmp_path / name
        p.write_bytes(raw)
        got = capi.read_fastx(str(p), min_bases=100000)
        par_env["KR_PGZ"] = "0"
        assert _same(got, capi.read_fastx(str(p), min_bases=100000)) and len(got[0]) == 9000
        par_env.pop("KR_PGZ")


def test_decoder_against_zlib_on_random_streams(capi, tmp_path, par_env):
    """Many small files with different statistics (so that Huffman tables of every shape, long codes, short distance codes
    and stored blocks occur), each cut into chunks much smaller than usual: every byte goes through the hand-written
    decoder and must come out as zlib's (the reader checks CRC-32 and length of every member, and the records are compared)."""
    rng = np.random.default_rng(21)
    for trial in range(12):
        n = int(rng.integers(1500, 5000))
        alphabet = [b"ACGT", b"ACGTN", b"AC", b"ACGTRYKMSWBDHVN"][trial % 4]
        recs = []
        for i in range(n):
            L = int(rng.integers(20, 400)) if trial % 3 else 151
            s = rng.choice(np.frombuffer(alphabet, np.uint8), L).tobytes()
            if trial % 2:
                q = bytes([33 + int(x) for x in np.minimum(60, rng.geometric(0.1, L))])
            else:
                q = b"F" * L
            if q[0:1] in (b"@", b"+"):
                q = b"I" + q[1:]
            recs.append(b"@%s.%d\n" % ([b"r", b"ERR123456", b"A00123:45:HXXXXXXX:1:1101:1000"][trial % 3], i) + s + b"\n+\n" + q + b"\n")
        data = b"".join(recs)
        plain, gz = tmp_path / ("t%d.fq" % trial), tmp_path / ("t%d.fq.gz" % trial)
        plain.write_bytes(data)
        gz.write_bytes(gzip.compress(data, [1, 4, 6, 9][trial % 4]))
        par_env["KR_PGZ_CHUNK"] = "65536"
        want = capi.read_fastx(str(plain), min_bases=100000)
        st = {}
        got = capi.read_fastx(str(gz), min_bases=100000, stats=st)
        assert _same(got, want), (trial, st)
        assert st["gzip_chunks"]["parsed"] >= 1 or len(data) < 200000, (trial, st)


def test_block_gzipped_runs_come_with_their_records(capi, tmp_path, par_env):
    """BGZF: the threads that inflate a run of members (~1 MB of the file) parse its records too; the bytes between the runs'
    records go through the same exact-seam check as ordinary gzip chunks."""
    from test_host import _bgzf_bytes
    recs = _fastq(24000, seed=5)
    tail = b">f1 x\nACGTACGT\nACGT\n"
    for key, data in (("clean", b"".join(recs)), ("fasta_tail", b"".join(recs) + tail), ("no_final_newline", b"".join(recs)[:-1])):
        plain, bg = tmp_path / (key + ".fq"), tmp_path / (key + ".fq.gz")
        plain.write_bytes(data)
        bg.write_bytes(_bgzf_bytes(data, block=50000, level=0))  # stored members: ~3.6 MB of file = four runs
        par_env["KR_FASTX_THREADS"] = "0"
        want = capi.read_fastx(str(plain), min_bases=100000)
        for threads in ("3", "1"):
            par_env["KR_FASTX_THREADS"] = threads
            st = {}
            got = capi.read_fastx(str(bg), min_bases=100000, stats=st)
            assert _same(got, want), (key, threads, st)
            assert st["gzip_chunks"]["parsed"] >= 3, (key, st)


def test_a_stream_that_inflates_beyond_the_memory_bound_is_an_error_that_names_the_streaming_reader(capi, tmp_path, par_env):
    """deflate allows a thousandfold, and `depth` chunks are in flight: beyond 1 GB out of one 4 MB chunk the parallel reader gives
    up with a message that names KR_PGZ=0 (zlib's gzread, constant memory) instead of being killed for its memory.  Here with the
    bound lowered to 1 MB on an ordinary file."""
    import gzip
    rng = np.random.default_rng(3)
    recs = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, rng.choice(np.frombuffer(b"ACGT", np.uint8), 150).tobytes(), b"I" * 150) for i in range(40_000))
    p = tmp_path / "dense.fq.gz"
    p.write_bytes(gzip.compress(recs, 6))
    os.environ["KR_PGZ_MAX_OUT"] = "100000"
    try:
        with pytest.raises(capi.KrError) as e:
            capi.read_fastx(str(p), min_bases=1 << 20)
        assert e.value.code == capi.KR_ERR_IO and "KR_PGZ=0" in str(e.value)
        os.environ["KR_PGZ"] = "0"
        n, b, o = capi.read_fastx(str(p), min_bases=1 << 20)
        assert len(n) == 40_000 and n[-1] == "r39999"
    finally:
        os.environ.pop("KR_PGZ_MAX_OUT", None)
        os.environ.pop("KR_PGZ", None)


def test_header_crc_is_checked_like_zlib_checks_it(capi, tmp_path, par_env):
    """a member header with the FHCRC flag: right, the records come out of either reader; wrong, both refuse the file (zlib's
    inflate reports "header crc mismatch"; the parallel reader verified nothing there until round 5)"""
    import struct
    import zlib
    recs = b"".join(b"@r%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(5000))

    def member(data, good):
        hdr = b"\x1f\x8b\x08\x02" + b"\x00" * 4 + b"\x00\xff"
        c = (zlib.crc32(hdr) & 0xFFFF) ^ (0 if good else 1)
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        return hdr + struct.pack("<H", c) + co.compress(data) + co.flush() + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))

    os.environ["KR_PGZ_CHUNK"] = "65536"
    for good in (True, False):
        p = tmp_path / f"fh{int(good)}.fq.gz"
        p.write_bytes(member(recs, good))
        for pgz in ("1", "0"):
            os.environ["KR_PGZ"] = pgz
            if good:
                n, b, o = capi.read_fastx(str(p), min_bases=1 << 16)
                assert len(n) == 5000 and n[-1] == "r4999"
            else:
                with pytest.raises(capi.KrError) as e:
                    capi.read_fastx(str(p), min_bases=1 << 16)
                assert e.value.code == capi.KR_ERR_IO
