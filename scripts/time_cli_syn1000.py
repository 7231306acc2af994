#!/usr/bin/env python3
"""End-to-end time of the `krepp dist` CLI on the BENCHMARK index (BASELINE configs[2]: 1000-genome Yule index, -k 29 -w 35 -h 13,
2^25 rows, table inflated to 10 GB), written to disk in the reference's on-disk format and read back by the CLI like any index:
reader thread -> GPU workers -> ordered writer, plain FASTQ in, report text out (src/krepp.cpp:347-394).

usage: scripts/time_cli_syn1000.py [reads, default 8,000,000] [index GB, default 10]
Prints, per configuration, the CLI's own `elapsed` / `[timing]` lines (KR_CLI_TIMING=1) and reads/s from its elapsed time (query
phase only: the index load -- 10 GB from disk, re-layout on the device -- is printed separately as wall time minus elapsed)."""
import os
import struct
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from krepp_amd import capi, synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 8_000_000
index_gb = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
work = tempfile.mkdtemp(prefix="krepp_cli_syn_")
t0 = time.time()
nwk_text = synth.yule_newick(1000, 2)
genomes = synth.evolve_genomes(nwk_text, 100_000, seed=2)
nwk = os.path.join(work, "yule.nwk")
open(nwk, "w").write(nwk_text)
tsv = synth.write_genomes(genomes, os.path.join(work, "g"))
idx = os.path.join(work, "idx")
capi.build_index(tsv, idx, nwk=nwk, k=29, w=35, h=13, m=4, r=1, frac=True, num_threads=min(32, os.cpu_count() or 1))

# ---- the inflated table (same generator and seeds as bench.py), written over the built index's inc / cmer files
import torch

hx = capi.HostIndex(idx)
dx, (inc, cmer) = synth.inflate_and_upload(torch, capi, hx, torch.device("cuda", 0), 0, index_gb)
dx.close()
hx.close()
torch.cuda.empty_cache()
with open(os.path.join(idx, "inc-m4r1-frac"), "wb") as f:  # u32 nrows, u64 cumulative bucket ends (src/table.cpp:81-82)
    f.write(struct.pack("<I", len(inc)))
    f.write(inc.astype("<u8").tobytes())
with open(os.path.join(idx, "cmer-m4r1-frac"), "wb") as f:  # u64 nkmers, {u32 enc32, u32 se} (src/table.cpp:79-80)
    f.write(struct.pack("<Q", cmer.size // 2))
    cmer.astype("<u4").tofile(f)
nk = cmer.size // 2
del inc, cmer

# ---- reads: fixed-width four-line FASTQ records built as one byte matrix
fq = os.path.join(work, "reads.fq")
with open(fq, "wb") as f:
    done = 0
    while done < n:
        m = min(500_000, n - done)
        b = np.concatenate([synth.sample_reads(genomes, min(100_000, m - o), seed=4000 + (done + o) // 100_000)[0] for o in range(0, m, 100_000)])
        rec = np.empty((m, 2 + 8 + 1 + 150 + 3 + 150 + 1), np.uint8)
        rec[:, 0], rec[:, 1] = ord("@"), ord("r")
        ids = np.arange(done, done + m, dtype=np.int64)
        rec[:, 2:10] = ((ids[:, None] // 10 ** np.arange(7, -1, -1)) % 10 + 48).astype(np.uint8)
        rec[:, 10] = 10
        rec[:, 11:161] = b.reshape(m, 150)
        rec[:, 161], rec[:, 162], rec[:, 163] = 10, ord("+"), 10
        rec[:, 164:314] = ord("I")
        rec[:, 314] = 10
        f.write(rec.tobytes())
        done += m
print(f"set-up {time.time() - t0:.0f} s: index {nk * 8 / 1e9:.1f} GB in {idx}, {n} reads in {os.path.getsize(fq) / 1e9:.2f} GB of FASTQ", flush=True)

exe = os.path.join(root, "krepp_amd", "lib", "krepp")
# KR_TIME_CLI_OUT_DIR: where the report goes (default: the work directory, i.e. the box's disk; /dev/shm takes the disk out of the
# measurement -- 38 GB of rows for 50 M reads are written at the disk's 5 GB/s otherwise, profiles/round6_cli_syn1000_50m*.txt)
out_file = os.path.join(os.environ.get("KR_TIME_CLI_OUT_DIR", work), "krepp_cli_out.txt")
configs = [
    ("dist", [], {}, out_file),
    ("dist", [], {"KR_CLI_HOST_TEXT": "1"}, out_file),  # the host formatter of rounds 1-4, same box
    ("dist", [], {"KR_CLI_SERIAL_WRITE": "1"}, out_file),  # device text, batches written one after the other
    ("dist", [], {"KR_CLI_BATCH_READS": "65536"}, out_file),
    ("dist", [], {"KR_CLI_BATCH_READS": "1048576"}, out_file),
    ("dist", [], {"KR_CLI_WORKERS_PER_GPU": "3"}, out_file),
    ("dist", [], {}, "/dev/null"),
    ("dist", ["--summarize"], {}, out_file),
    ("dist", ["--no-multi"], {}, out_file),
    ("dist", ["--no-multi"], {"KR_CLI_WORKERS_PER_GPU": "3"}, out_file),  # 9
    ("dist", ["--summarize"], {"KR_CLI_WORKERS_PER_GPU": "3"}, out_file),  # 10
    ("dist", ["--no-multi"], {"KR_CLI_WORKERS_PER_GPU": "4"}, out_file),  # 11
    ("dist", ["--no-multi"], {"KR_CLI_BATCH_READS": "524288"}, out_file),  # 12
]
if os.environ.get("KR_TIME_CLI_CONFIGS"):  # e.g. "0,5,7"
    configs = [configs[int(i)] for i in os.environ["KR_TIME_CLI_CONFIGS"].split(",")]
for sub, extra, env, outp in configs:
    # a fresh output file every time: closing a file that was truncated and rewritten makes ext4 allocate its blocks at close()
    # (0.6 s for 6 GB, inside the CLI's elapsed time as inside the reference's) -- every configuration after the first paid that
    # in the earlier runs of this script (profiles/round5_cli_syn1000.txt: the gap between a worker's last batch and `elapsed`)
    if outp != "/dev/null" and os.path.exists(outp):
        os.remove(outp)
    t = time.time()
    r = subprocess.run([exe, sub, "-i", idx, "-q", fq, "-o", outp] + extra, capture_output=True, text=True, env=dict(os.environ, KR_CLI_TIMING="1", **env))
    dt = time.time() - t
    lines = [l for l in r.stderr.strip().splitlines() if "timing" in l or "elapsed" in l]
    el = [float(l.split("elapsed:")[1].split()[0]) for l in lines if "elapsed:" in l]
    size = os.path.getsize(outp) / 1e9 if outp != "/dev/null" and os.path.exists(outp) else 0.0
    print(f"{sub} {' '.join(extra)} {env} -> {outp if outp == '/dev/null' else 'file'}: rc {r.returncode}, wall {dt:.2f} s, query phase "
          f"{el[0] if el else float('nan'):.2f} s = {n / el[0] / 1e6 if el else float('nan'):.1f} M reads/s, index load + start-up {dt - (el[0] if el else 0):.1f} s, "
          f"output {size:.2f} GB ({size / el[0] if el else 0:.2f} GB/s of text)", flush=True)
    for l in lines:
        print("   ", l, flush=True)
    if r.returncode:
        print(r.stderr[-1500:])

# ---- KR_TIME_CLI_TRACE=1: the default configuration once more under rocprofv3 --kernel-trace (the binary itself after `--`): the
#      kernels of a CLI batch by name -- what the text kernels (kr_text_*) and the row compaction cost beside the rest
if os.environ.get("KR_TIME_CLI_TRACE"):
    import collections, csv, glob, re, statistics
    td = os.path.join(work, "trace")
    env = dict(os.environ, TMPDIR="/tmp", GPU_MAX_HW_QUEUES="8", KR_CLI_CLEAN_EXIT="1")  # (the tool writes its files when the process exits in order)
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", td, "--", exe, "dist", "-i", idx, "-q", fq, "-o", out_file],
                       capture_output=True, text=True, env=env)
    d = collections.defaultdict(list)
    for f in glob.glob(td + "/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(kr_\w+)(<[^>]*>)?", row["Kernel_Name"])
            if m:
                d[m.group(1)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    print(f"kernels of `krepp dist` (default configuration, {n} reads) under rocprofv3 --kernel-trace: name, launches, median ms, total ms", flush=True)
    for k_, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print(f"    {k_:32s} {len(v):5d} {statistics.median(v):9.3f} {sum(v):10.1f}", flush=True)
