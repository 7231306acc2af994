#!/usr/bin/env python3
"""End-to-end time of the `krepp dist` CLI (reader thread -> GPU worker -> ordered writer) on a synthetic FASTQ."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from krepp_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
work = tempfile.mkdtemp(prefix="krepp_cli_")
nwk = os.path.join(root, "tests", "golden", "tree_toy.nwk")
g = synth.evolve_genomes(open(nwk).read(), 400_000, seed=7)
tsv = synth.write_genomes(g, os.path.join(work, "g"))
idx = os.path.join(work, "idx")
capi.build_index(tsv, idx, nwk=nwk, k=27, w=35, h=11, m=4, r=1, frac=True, num_threads=8)
fq = os.path.join(work, "reads.fq")
with open(fq, "wb") as f:
    done = 0
    while done < n:
        m = min(200_000, n - done)
        b, o, names = synth.sample_reads(g, m, seed=100 + done)
        r = b.reshape(m, 150)
        q = b"I" * 150
        rows = [b"@r%d\n" % (done + i) + r[i].tobytes() + b"\n+\n" + q + b"\n" for i in range(m)]
        f.write(b"".join(rows))
        done += m
exe = os.path.join(root, "krepp_amd", "lib", "krepp")
for sub, extra, env in (("dist", [], {}), ("dist", [], {"KR_CLI_WORKERS_PER_GPU": "3"}), ("dist", [], {"KR_CLI_BATCH_READS": "262144"}),
                        ("dist", [], {"KR_CLI_BATCH_READS": "262144", "KR_CLI_WORKERS_PER_GPU": "3"}),
                        ("dist", [], {"KR_CLI_BATCH_READS": "1048576", "KR_CLI_WORKERS_PER_GPU": "3"}), ("dist", ["--summarize"], {}),
                        ("place", [], {}), ("place", ["--tabular"], {}), ("place", ["--summarize"], {})):
    t = time.time()
    r = subprocess.run([exe, sub, "-i", idx, "-q", fq, "-o", os.path.join(work, "out.txt")] + extra, capture_output=True, text=True,
                       env=dict(os.environ, KR_CLI_TIMING="1", **env))
    dt = time.time() - t
    print(sub, extra, env, [l for l in r.stderr.strip().splitlines() if "timing" in l or "elapsed" in l])
    print(sub, " ".join(extra), "rc", r.returncode, "reads", n, "seconds %.2f" % dt, "reads/s %.3g" % (n / dt), "output MB %.1f" % (os.path.getsize(os.path.join(work, "out.txt")) / 1e6))

# gzip input: inflated by one thread (zlib), parsed by the sequential reader
import shutil
if shutil.which("gzip"):
    ngz = min(n, 4_000_000)
    fqs = os.path.join(work, "sub.fq")
    with open(fq, "rb") as fi, open(fqs, "wb") as fo:
        for _ in range(ngz * 4):
            fo.write(fi.readline())
    import struct, zlib
    raw = open(fqs, "rb").read()
    bg = bytearray()
    for i in range(0, len(raw), 65280):
        c = raw[i:i + 65280]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        comp = co.compress(c) + co.flush()
        bg += b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(comp) + 8 - 1) + comp + struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c))
    bg += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    open(os.path.join(work, "sub_bgzf.fq.gz"), "wb").write(bytes(bg))
    del raw, bg
    subprocess.run(["gzip", "-1", "-f", fqs], check=True)
    t = time.time()
    r = subprocess.run([exe, "dist", "-i", idx, "-q", fqs + ".gz", "-o", os.path.join(work, "out.txt")], capture_output=True, text=True,
                       env=dict(os.environ, KR_CLI_TIMING="1"))
    print("dist on .gz", ngz, "reads", [l for l in r.stderr.strip().splitlines() if "timing" in l or "elapsed" in l])

    r = subprocess.run([exe, "dist", "-i", idx, "-q", os.path.join(work, "sub_bgzf.fq.gz"), "-o", os.path.join(work, "out2.txt")], capture_output=True, text=True,
                       env=dict(os.environ, KR_CLI_TIMING="1"))
    print("dist on BGZF", ngz, "reads", [l for l in r.stderr.strip().splitlines() if "timing" in l or "elapsed" in l])
    print("same output:", open(os.path.join(work, "out.txt")).read().split("\n", 2)[2] == open(os.path.join(work, "out2.txt")).read().split("\n", 2)[2])
