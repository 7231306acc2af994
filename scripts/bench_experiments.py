#!/usr/bin/env python3
"""Experiments around bench.py's workload that are NOT part of what the driver runs (they lived in bench.py until round 6).
Same set-up (bench.prepare: genomes, reads, index), then one of:

  --stream-variance N [--stream-variance-move 0,1,..] [--stream-variance-idle S]
        create N more streams one after another and print the scan kernel's launch time on each: does the placement of a
        stream's buffers in HBM matter (docs/design/03, the launch-time levels)?
  --churn-gb G
        allocate and free G GB three times and print the scan's launch times around it
  --host-leg-variant {no_d2h,no_h2d}
        the host-inclusive leg without the copy back / without the copy in: which direction of PCIe traffic slows the scan
  KR_BENCH_BALLAST_GB=G
        soak up G GB of device memory before the timed stream allocates

Usage: python scripts/bench_experiments.py [bench.py's options] [the options above]; results go to stderr / stdout as text.
"""
from __future__ import annotations

import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = bench.build_parser()
    ap.add_argument("--stream-variance", type=int, default=0)
    ap.add_argument("--stream-variance-move", default="",
                    help="comma-separated buffer groups (kr_debug_stream_move: 0 items, 1 per-read arrays, 2 counters, 3 records, 4 HIP "
                         "stream) moved one after the other, three launches after each move")
    ap.add_argument("--stream-variance-idle", type=float, default=0.0, help="run every stream three more times after this many seconds of idle GPU")
    ap.add_argument("--churn-gb", type=float, default=0.0)
    ap.add_argument("--host-leg-variant", default="full", choices=["full", "no_d2h", "no_h2d"])
    a = bench.finish_args(ap.parse_args())
    if a.gpus != 1:
        sys.exit("bench_experiments.py: one GPU, one process")
    c = bench.prepare(a)
    torch, n, nb_ = c.torch, c.n, c.nb
    ballast = None
    if os.environ.get("KR_BENCH_BALLAST_GB"):
        ballast = torch.empty(int(float(os.environ["KR_BENCH_BALLAST_GB"]) * (1 << 30)), dtype=torch.uint8, device=c.dev)
    bench.timed_leg(c)
    st = c.psts[0]
    print(json.dumps({"value": n * a.steps / c.dt, "ms_per_step": c.dt / a.steps * 1e3, "item_list_placement": c.item_placement,
                      "scan_ms": [round(t_[1], 2) for t_ in c.timed_tm], "acc_ms": [round(t_[2], 2) for t_ in c.timed_tm],
                      "llh_ms": [round(t_[3], 2) for t_ in c.timed_tm]}))
    if ballast is not None:
        del ballast
        torch.cuda.empty_cache()

    def launches(s_, k_):
        out_ = []
        for i in range(k_):
            db, do = c.d_batches[i % nb_]
            s_.submit_device(db.data_ptr(), do.data_ptr(), n)
            s_.wait()
            t_ = s_.timing()
            out_.append((round(t_.ms_scan, 2), round(t_.ms_acc, 2), round(t_.ms_llh, 2)))
        return out_

    if a.churn_gb > 0:  # does freeing a large buffer slow the scan launches that follow (the driver clears freed memory)?
        print(f"[churn] steady: {launches(st, 8)}", file=sys.stderr)
        for rep in range(3):
            big = torch.empty(int(a.churn_gb * (1 << 30)), dtype=torch.uint8, device=c.dev)
            torch.cuda.synchronize()
            print(f"[churn] after allocating {a.churn_gb} GB: {launches(st, 6)}", file=sys.stderr)
            del big
            torch.cuda.empty_cache()
            print(f"[churn] after freeing it: {launches(st, 14)}", file=sys.stderr)
    if a.stream_variance:
        extra = []
        names_ = ["item list", "per-read arrays", "counters + cursors", "records + dedup table", "HIP stream"]
        for q in range(a.stream_variance):
            sx = bench.new_stream(c)
            print(f"[stream-variance] stream {q}: (scan, accumulate, llh) ms = {launches(sx, 4)[1:]}", file=sys.stderr)
            print(f"[stream-variance] stream {q} addresses: " + " ".join(f"{k}={v:#x}" for k, v in sx.debug_addrs().items()), file=sys.stderr)
            for w_ in [int(x) for x in a.stream_variance_move.split(",") if x]:  # which of the stream's buffers (or its HIP stream) decides the level?
                sx.debug_move(w_)
                print(f"[stream-variance] stream {q} after moving its {names_[w_]}: {launches(sx, 3)}", file=sys.stderr)
                print(f"[stream-variance] stream {q} addresses: " + " ".join(f"{k}={v:#x}" for k, v in sx.debug_addrs().items()), file=sys.stderr)
            if a.stream_variance_idle > 0:  # the SAME stream (same buffers, same queue) again after the GPU sat idle
                for rep in range(3):
                    time.sleep(a.stream_variance_idle)
                    print(f"[stream-variance] stream {q} after {a.stream_variance_idle} s idle: {launches(sx, 3)}", file=sys.stderr)
            extra.append(sx)  # kept alive: the next stream's buffers land elsewhere
            if len(extra) > (0 if a.stream_variance_move else 2):  # (a stream holds ~60 GB of result buffers at 8 M reads per batch)
                extra.pop(0).close()
        for sx in extra:
            sx.close()
    if a.host_leg_variant != "full":
        print(json.dumps(bench.host_inclusive_leg(c, a.host_leg_variant)))
    for s_ in c.psts:
        s_.close()


if __name__ == "__main__":
    main()
