"""bench.py as the driver runs it, on the GPU box: the one-process default and the N-rank path.  One GPU here, so the
two ranks share it (--ranks-share-device) and the collective backend is gloo; everything else -- the launcher, rank 0
building and uploading the index, kr_index_export -> broadcast of every flat buffer -> kr_index_import on the other
rank, per-rank read shards, MAX-over-ranks timing -- is the code the 8-GPU run executes (there with RCCL)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_on_one_gpu():
    out = run_bench("--gpus", "2", "--backend", "gloo", "--ranks-share-device", "--workload", "toy25", "--reads-per-step", "100000",
                    "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--check-reads", "3000", "--distinct-batches", "2")
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert len(out["per_rank_reads_per_s"]) == 2 and min(out["per_rank_reads_per_s"]) > 0
    assert out["index_broadcast"]["bytes"] == out["config"]["index_device_bytes"] > 1e6
    assert out["check"]["rows_equal"] and out["check"]["max_rel_dist_err"] < 1e-6
    # whole-job value = reads of both ranks over the slowest rank's time
    assert out["value"] > 0.9 * min(out["per_rank_reads_per_s"]) and out["value"] <= 2.05 * max(out["per_rank_reads_per_s"])


def test_bench_default_line_small():
    out = run_bench("--workload", "toy25", "--reads-per-step", "200000", "--steps", "3", "--warmup", "1", "--cpu-seconds", "2",
                    "--check-reads", "3000", "--distinct-batches", "2")
    assert out["n_gpus"] == 1 and out["unit"] == "reads/s" and out["vs_baseline"] is None
    rl = out["roofline"]
    assert rl["bound"] == "hbm" and rl["unit"] == "GB/s" and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-9
    assert rl["traffic"] is None  # the committed PMC profile is of the 10 GB workload, not of this one
    # a fraction of a physical peak: the dominant kernel's OWN bytes over its own time, and every stage's
    assert 0 < rl["frac"] <= 1.0 and 0 < rl["step_frac"] <= 1.0
    assert [k["stage"] for k in rl["kernels"]] == ["scan", "accumulate", "llh_select"]
    for k in rl["kernels"]:
        assert 0 <= k["frac"] <= 1.0 and k["avg_launch_ms"] > 0 and abs(k["frac"] - k["achieved_GBps"] / rl["peak"]) < 1e-9
    assert rl["kernels"][0]["algorithmic_bytes_per_read"] == rl["algorithmic_bytes_per_read"] < rl["whole_path_algorithmic_bytes_per_read"]
    # the rows that were checked are those of the last TIMED launch
    ck = out["check"]
    assert ck["from_timed_launch"] and ck["reads_in_that_launch"] == 200000 and ck["rows_equal"] and ck["max_rel_dist_err"] < 1e-6
    assert ck["host_path_small_stream"]["rows_equal"]
    # ... and ALL rows of that launch are what a second, independent stream makes of the same batch
    wl = ck["whole_launch"]
    assert wl["reads"] == 200000 and wl["rows"] > 200000 and wl["equal_on_an_independent_stream"]
    assert "--workload syn" not in out["config"]["workload_choice"] and out["setup_parts_s"]["read_procs"] >= 1
    assert out["config"]["item_list_placement"]["finished_before_timing"]
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["cores"] >= 1 and out["cpu_baseline"]["value"] > 0
    assert out["check"]["rows_equal"]
    hi = out["value_host_inclusive"]
    assert hi["value"] > 0 and hi["streams"] == 3 and hi["steady_state"]["value"] > 0


def test_bench_one_rank_rccl_replicates_the_index():
    """`--force-replicate`: the N-rank replication path (kr_index_export -> DevPtr tensors over the library's own device
    buffers -> dist.broadcast per buffer) on a ONE-rank `nccl` (= RCCL) process group, on the GPU: torch, RCCL and raw
    device pointers meet on hardware here, before the driver's 8-GPU run (src/krepp.cpp:92-106 is the load loop it
    replaces).  The index that went through it still answers like the oracle."""
    out = run_bench("--force-replicate", "--backend", "nccl", "--workload", "toy25", "--reads-per-step", "100000", "--steps", "2",
                    "--warmup", "1", "--no-cpu-baseline", "--no-host-inclusive", "--check-reads", "3000", "--distinct-batches", "1")
    bc = out["index_broadcast"]
    assert bc["backend"] == "nccl" and bc["world"] == 1 and bc["bytes"] == out["config"]["index_device_bytes"] > 1e6
    assert bc["seconds"] > 0 and bc["GB_per_s"] > 0
    assert out["n_gpus"] == 1 and out["check"]["rows_equal"] and out["check"]["max_rel_dist_err"] < 1e-6


def test_bench_two_ranks_on_the_10000_genome_index():
    """BASELINE.json configs[3]'s code path walked on the one GPU at hand: `--workload syn10000` (10,000 genomes on a Yule tree,
    seed 3, the generator of the 8-GPU configuration) with a reduced table (1 GB instead of 10), two ranks sharing the device over
    gloo: rank 0 builds, inflates and uploads the index, every flat buffer is broadcast to rank 1 (kr_index_export ->
    kr_index_import), the ranks run their own read shards, rank 0 checks the rows of its last timed launch against the oracle.
    (The driver's SCALE run uses the same command with --backend nccl and one GPU per rank: BASELINE.md.)"""
    out = run_bench("--gpus", "2", "--backend", "gloo", "--ranks-share-device", "--workload", "syn10000", "--index-gb", "1", "--reads-per-step", "200000",
                    "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--check-reads", "2000", "--distinct-batches", "1")
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and "10000-genome" in out["config"]["workload"]
    assert len(out["per_rank_reads_per_s"]) == 2 and min(out["per_rank_reads_per_s"]) > 0
    assert out["index_broadcast"]["bytes"] == out["config"]["index_device_bytes"] > 1e9 and out["index_broadcast"]["world"] == 2
    assert out["check"]["from_timed_launch"] and out["check"]["rows_equal"] and out["check"]["max_rel_dist_err"] < 1e-6


def test_bench_eight_ranks_share_one_gpu():
    """The 8-rank start of BASELINE.json configs[3] rehearsed on one box: eight rank processes (launch_ranks), each generating
    its own read shard under the box's CPU quota while rank 0 builds the 10,000-genome index, ONE build + upload, SEVEN
    kr_index_import + broadcast receptions (gloo carrying the device buffers; the driver's run is this command with
    --backend nccl and a GPU per rank), eight read shards, MAX-over-ranks timing, and the rows of rank 0's last timed
    launch equal to the oracle's (src/krepp.cpp:92-106 is the load loop, :360-387 the batch loop this replaces)."""
    out = run_bench("--gpus", "8", "--backend", "gloo", "--ranks-share-device", "--workload", "syn10000", "--index-gb", "1",
                    "--reads-per-step", "100000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-host-inclusive",
                    "--check-reads", "2000", "--distinct-batches", "1")
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and "10000-genome" in out["config"]["workload"]
    assert len(out["per_rank_reads_per_s"]) == 8 and min(out["per_rank_reads_per_s"]) > 0
    assert out["index_broadcast"]["world"] == 8 and out["index_broadcast"]["bytes"] == out["config"]["index_device_bytes"] > 1e9
    assert out["check"]["from_timed_launch"] and out["check"]["rows_equal"] and out["check"]["max_rel_dist_err"] < 1e-6
    assert out["value"] <= 8.05 * max(out["per_rank_reads_per_s"])
    print(f"8 ranks on one GPU: setup_s {out['setup_s']:.1f}, broadcast {out['index_broadcast']['seconds']:.2f} s, "
          f"{out['value'] / 1e6:.1f} M reads/s whole job")


def test_bench_eight_ranks_at_the_full_table_size():
    """BASELINE.json configs[3]'s start at its real per-GPU size, on the one GPU at hand (round-5 review, item 1): eight ranks, the
    FULL 10 GB table -- eight replicas of the 19 GB index and eight 1 M-read streams resident together --, the index built and
    inflated once and replicated seven times, every rank generating its shard on `usable_cpus() // 8` worker processes; the set-up
    stays under two minutes, the rows of rank 0's timed launch equal the oracle's and, all of them, an independent stream's.
    (KR_ACC_SCRATCH_GB=2: a stream budgets its accumulate scratch by the device memory that is free when it is made -- a sixth of it,
    8..48 GB, 26 GB of which the 10,000-leaf tree uses at full residency -- and eight ranks that each see the whole GPU as theirs would
    not fit beside eight replicas in ONE GPU's 288 GB; a rank that owns its GPU does not need the cap.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["KR_ACC_SCRATCH_GB"] = "2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--ranks-share-device", "--workload", "syn10000",
                        "--reads-per-step", "1000000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-host-inclusive", "--check-reads", "2000"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 8 and "10000-genome" in out["config"]["workload"] and "10 GB" in out["config"]["workload"]
    assert out["index_broadcast"]["world"] == 8 and out["index_broadcast"]["bytes"] == out["config"]["index_device_bytes"] > 18e9
    assert out["setup_s"] < 120 and out["setup_parts_s"]["read_procs"] >= 1
    assert len(out["per_rank_reads_per_s"]) == 8 and min(out["per_rank_reads_per_s"]) > 0
    ck = out["check"]
    assert ck["from_timed_launch"] and ck["reads_in_that_launch"] == 1_000_000 and ck["rows_equal"] and ck["max_rel_dist_err"] < 1e-6
    assert ck["whole_launch"]["equal_on_an_independent_stream"] and ck["whole_launch"]["rows"] > 10_000_000
    print(f"8 ranks, full table: setup_s {out['setup_s']:.1f} ({out['setup_parts_s']}), {out['value'] / 1e6:.1f} M reads/s whole job")

