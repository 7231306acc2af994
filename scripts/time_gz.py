#!/usr/bin/env python3
"""Compressed query files: reader throughput (kr_fastx_* alone) and `krepp dist` end to end, on the 25-reference index.

usage: time_gz.py [reads]     (on the GPU box; writes nothing outside its temporary directory)
Files: plain FASTQ, gzip -1, gzip -6 (ordinary gzip: one dependent deflate stream), BGZF.  For each: the reader alone with the
parallel path off (zlib's gzread, what the reference does: src/rqseq.cpp:161-197) and on at 4 / 8 / 12 threads, then the CLI.
"""
import ctypes as C, os, struct, subprocess, sys, tempfile, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from krepp_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
work = tempfile.mkdtemp(prefix="krepp_gz_")
nwk = os.path.join(root, "tests", "golden", "tree_toy.nwk")
g = synth.evolve_genomes(open(nwk).read(), 400_000, seed=7)
tsv = synth.write_genomes(g, os.path.join(work, "g"))
idx = os.path.join(work, "idx")
capi.build_index(tsv, idx, nwk=nwk, k=27, w=35, h=11, m=4, r=1, frac=True, num_threads=8)
fq = os.path.join(work, "reads.fq")
rng = np.random.default_rng(1)
with open(fq, "wb") as f:
    done = 0
    while done < n:
        m = min(200_000, n - done)
        b, o, names = synth.sample_reads(g, m, seed=100 + done)
        r = b.reshape(m, 150)
        # Illumina-like qualities: mostly one value, the rest spread (so that the file compresses like sequencer output, ~3.3x)
        q = np.minimum(40, np.maximum(2, (37 + rng.normal(0, 3, (m, 150))).astype(int))).astype(np.uint8) + 33
        q[rng.random((m, 150)) < 0.7] = ord("F")
        rows = [b"@SRR0000.%d %d/1\n" % (done + i, done + i) + r[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n" for i in range(m)]
        f.write(b"".join(rows))
        done += m
t = time.time()
subprocess.run("gzip -1 -c %s > %s" % (fq, fq + ".1.gz"), shell=True, check=True)
t1 = time.time() - t
t = time.time()
subprocess.run("gzip -6 -c %s > %s" % (fq, fq + ".6.gz"), shell=True, check=True)
t6 = time.time() - t
raw = open(fq, "rb").read()
bg = bytearray()
for i in range(0, len(raw), 65280):
    c = raw[i:i + 65280]
    co = zlib.compressobj(4, zlib.DEFLATED, -15)
    comp = co.compress(c) + co.flush()
    bg += b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(comp) + 8 - 1) + comp + struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c))
bg += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
open(fq + ".bgzf.gz", "wb").write(bytes(bg))
del raw, bg
files = {"plain": fq, "gzip -1": fq + ".1.gz", "gzip -6": fq + ".6.gz", "BGZF": fq + ".bgzf.gz"}
print("reads", n, {k: "%.0f MB" % (os.path.getsize(v) / 1e6) for k, v in files.items()}, "gzip -1 took %.0f s, gzip -6 %.0f s" % (t1, t6), "usable cpus", len(os.sched_getaffinity(0)), flush=True)

lib = capi.load()
lib.kr_fastx_parallel_chunks.restype = C.c_uint64
lib.kr_fastx_parallel_chunks.argtypes = [C.c_void_p]


def reader(path):
    h = C.c_void_p()
    capi.check(lib.kr_fastx_open(os.fsencode(path), C.byref(h)))
    cnt, t0 = 0, time.time()
    while True:
        b = capi.KrFastxBatch()
        capi.check(lib.kr_fastx_next(h, 262144 * 150, C.byref(b)))
        cnt += b.nreads
        if not b.more:
            break
    dt = time.time() - t0
    lib.kr_fastx_close(h)
    assert cnt == n, (cnt, n)
    return dt


print("== the reader alone (kr_fastx_open / kr_fastx_next, batches of 262144 reads) ==")
for name, path in files.items():
    for env in ({"KR_FASTX_THREADS": "0"}, {"KR_FASTX_THREADS": "4"}, {}, {"KR_FASTX_THREADS": "12"}):
        keep = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        best = min(reader(path) for _ in range(2))
        for k, v in keep.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        print("%-8s %-28s %6.2f s  %6.2f M reads/s" % (name, env or "default (min(12, 3/4 of the usable CPUs))", best, n / best / 1e6), flush=True)

print("== krepp dist end to end ==")
exe = os.path.join(root, "krepp_amd", "lib", "krepp")
outs = {}
for name, path in files.items():
    for env in ({"KR_FASTX_THREADS": "0"}, {}):
        if name == "plain" and env:
            continue
        out = os.path.join(work, "out_%s_%d.txt" % (name.replace(" ", ""), len(env)))
        t = time.time()
        r = subprocess.run([exe, "dist", "-i", idx, "-q", path, "-o", out], capture_output=True, text=True, env=dict(os.environ, KR_CLI_TIMING="1", **env))
        dt = time.time() - t
        print("%-8s %-28s rc %d  %6.2f s  %6.2f M reads/s (whole process)" % (name, env or "default", r.returncode, dt, n / dt / 1e6),
              [l for l in r.stderr.strip().splitlines() if "timing" in l or "elapsed" in l], flush=True)
        outs[(name, len(env))] = out
ref = open(outs[("plain", 0)]).read().split("\n", 2)[2]
print("same rows as from the plain file:", {("%s/%s" % (k[0], "zlib" if k[1] else "parallel")): open(v).read().split("\n", 2)[2] == ref for k, v in outs.items()})
