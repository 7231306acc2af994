"""kr_index_broadcast in a process WITHOUT torch: a plain-C program over include/krepp_amd.h replicates an
uploaded index onto its own device through RCCL -- the library dlopen()s librccl.so.1 by itself, nobody has loaded torch's
private copy for it -- and answers queries from the replica; and `krepp dist --gpus 1` as the stand-alone CLI.  The load loop
this replaces for the GPUs after the first: src/krepp.cpp:92-106."""
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def test_plain_c_process_replicates_through_rccl_and_queries_the_replica(po, toy_index_dir, toy_reads, tmp_path):
    exe = str(tmp_path / "replica_driver")
    lib = os.path.join(ROOT, "krepp_amd", "lib")
    subprocess.run(["gcc", "-O1", "-std=c11", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_drivers", "replica_driver.c"),
                    "-o", exe, "-L", lib, "-lkrepp_amd", f"-Wl,-rpath,{lib}"], check=True)
    fq = os.path.join(GOLDEN, "toy_reads.fq")
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "PYTHONPATH", "KR_RCCL_LIB")}
    r = subprocess.run([exe, toy_index_dir, fq], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rccl = [l for l in r.stderr.splitlines() if l.startswith("rccl: ")]
    assert len(rccl) == 1 and "librccl" in rccl[0] and "torch" not in rccl[0], r.stderr  # the system RCCL, not torch's private copy
    names, bases, offs = toy_reads
    want = po.Index(toy_index_dir).dist(bases, offs, names, po.params(collect=4))["text"]
    # RCCL's version banner ("RCCL version : ...", "Librccl path : ...") goes to stdout when the first communicator is made;
    # the driver (the application, single-threaded at that point) points fd 1 at stderr around kr_index_broadcast -- the library
    # itself never touches file descriptors -- so stdout carries the rows and nothing else
    assert r.stdout == want
    banner = [l for l in r.stderr.splitlines() if l.startswith("Librccl path")]
    assert all("torch" not in l for l in banner), r.stderr


def test_cli_dist_gpus_1_without_python(po, toy_index_dir, toy_reads):
    exe = os.path.join(ROOT, "krepp_amd", "lib", "krepp")
    fq = os.path.join(GOLDEN, "toy_reads.fq")
    r = subprocess.run([exe, "dist", "-i", toy_index_dir, "-q", fq, "--gpus", "1", "--device", "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    names, bases, offs = toy_reads
    want = po.Index(toy_index_dir).dist(bases, offs, names, po.params(collect=4))["text"]
    body = r.stdout.split("\n", 2)[2]  # two header lines (src/krepp.cpp:311-319)
    assert body == want


def test_cli_on_the_references_own_query_file(po, toy_index_dir, tmp_path, synth, capi):
    """BASELINE.json configs[0]'s query file, test/query_toy.fq of the reference (committed as a fixture: data), through the
    stand-alone CLI and the HIP path.  Its 100 reads come from real genomes and share nothing with a synthetic index: against an
    index with the reference's toy parameters (-k 27 -w 35 -h 11, 25 synthetic references) every read gets the no-hit row
    `SEQ_ID\tNA\tNaN` (src/query.cpp:173-176) under the two header lines (src/krepp.cpp:311-319); against the committed k = 21
    index a few 21-mers match by chance, and the rows are the oracle's."""
    from conftest import read_fastq_simple

    exe = os.path.join(ROOT, "krepp_amd", "lib", "krepp")
    fq = os.path.join(GOLDEN, "query_toy.fq")
    names, bases, offs = read_fastq_simple(fq)
    ids = [n.split()[0] for n in names]
    assert len(ids) == 100
    # (1) the toy parameters of the reference's README / configs[0]
    nwk = os.path.join(GOLDEN, "tree_toy.nwk")
    g = synth.evolve_genomes(open(nwk).read(), 100_000, seed=7)
    tsv = synth.write_genomes(g, str(tmp_path / "g"))
    idx = str(tmp_path / "idx")
    capi.build_index(tsv, idx, nwk=nwk, k=27, w=35, h=11, m=4, r=1, frac=True, num_threads=4)
    r = subprocess.run([exe, "dist", "-i", idx, "-q", fq], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert lines[0].startswith("# software: krepp\tversion: ") and lines[1] == "SEQ_ID\tREFERENCE_NAME\tDIST"
    assert lines[2:] == [f"{i}\tNA\tNaN" for i in ids]
    assert "Total number of sequences queried: 100" in r.stderr
    assert "\n".join(lines[2:]) + "\n" == po.Index(idx).dist(bases, offs, ids, po.params(collect=4))["text"]
    # (2) the committed k = 21 index: whatever matches by chance, exactly as the oracle reports it
    r = subprocess.run([exe, "dist", "-i", toy_index_dir, "-q", fq], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split("\n", 2)[2] == po.Index(toy_index_dir).dist(bases, offs, ids, po.params(collect=4))["text"]
