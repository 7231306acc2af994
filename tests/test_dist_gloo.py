"""world_size-2 CPU test (gloo) of the multi-GPU plumbing: read sharding, descriptor and buffer
broadcast, and that sharded results concatenate to the unsharded result.  The per-rank "device"
work is done by the oracle here (no GPU in this container); on the GPU box the same helpers move
the real device buffers over RCCL (tests/test_gpu_parity.py::test_index_export_import_roundtrip
covers export/import on one GPU)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from krepp_amd import dist as kdist
    import pyoracle as po
    from conftest import read_fastq_simple

    names, bases, offs = read_fastq_simple(os.path.join(GOLDEN, "toy_reads.fq"))
    # 1. shards tile the batch exactly
    b, o, lo = kdist.shard_reads(bases, offs, rank, world)
    los = [None] * world
    dist.all_gather_object(los, (lo, len(o) - 1))
    assert sum(n for _, n in los) == len(names) and los[0][0] == 0 and los[1][0] == los[0][1]
    # 2. descriptor + flat buffers travel from rank 0 (CPU tensors stand in for device buffers)
    idx_files = sorted(f for f in os.listdir(os.path.join(GOLDEN, "toy_index")))
    blob = kdist.broadcast_blob(dist, ("desc:" + ",".join(idx_files)).encode() if rank == 0 else None)
    assert blob.decode().startswith("desc:cmer")
    tensors = []
    for f in idx_files:
        raw = np.fromfile(os.path.join(GOLDEN, "toy_index", f), dtype=np.uint8)
        t = torch.from_numpy(raw.copy()) if rank == 0 else torch.zeros(len(raw), dtype=torch.uint8)
        tensors.append(t)
    kdist.broadcast_buffers(dist, tensors)
    rep = os.path.join(tmpdir, f"replica{rank}")
    os.makedirs(rep, exist_ok=True)
    for f, t in zip(idx_files, tensors):
        t.numpy().tofile(os.path.join(rep, f))
    # 3. every rank queries ITS replica with ITS shard; rank 0 checks against the unsharded run
    ox = po.Index(rep)
    res = ox.dist(b, o, names[lo:lo + len(o) - 1], po.params(collect=4))
    texts = [None] * world
    dist.all_gather_object(texts, res["text"])
    if rank == 0:
        full = po.Index(os.path.join(GOLDEN, "toy_index")).dist(bases, offs, names, po.params(collect=4))
        assert "".join(texts) == full["text"]
    dist.barrier()
    dist.destroy_process_group()


def test_sharding_and_replication_world2(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)


def test_shard_bounds_cover_everything():
    from krepp_amd.dist import shard_bounds

    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_bench_gpus_n_launches_n_rank_processes():
    """`python bench.py --gpus 2` with no launcher around it must start two rank processes itself
    (the driver's N=1 command shape with N>1) and refuse a world size that contradicts --gpus."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == [0, 1] and out["distinct_processes"] == 2
    assert out["shards"][0][1] == out["shards"][1][0]
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                         env=dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=3" in bad.stderr


def test_bench_gpus_8_launches_eight_rank_processes():
    """The driver's 8-GPU command shape (`python bench.py --gpus 8`): eight rank processes, one rendezvous, eight contiguous
    shards that cover the job (the GPU half of this rehearsal is tests/test_gpu_bench.py::test_bench_eight_ranks_share_one_gpu)."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["ranks"] == list(range(8)) and out["distinct_processes"] == 8
    assert out["shards"][0][0] == 0 and all(out["shards"][i][1] == out["shards"][i + 1][0] for i in range(7))
