#!/usr/bin/env python3
"""Records per read of bench.py's workload (what the select kernels walk): histogram of kr_result_view.read_cnt over one launch.
Usage: python scripts/records_per_read_hist.py [bench.py's options]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    a = bench.finish_args(bench.build_parser().parse_args())
    c = bench.prepare(a)
    torch = c.torch
    st = bench.new_stream(c)
    db, do = c.d_batches[0]
    st.submit_device(db.data_ptr(), do.data_ptr(), c.n)
    st.wait()
    rv = st.collect_device()
    cnt = bench.dev_array(torch, c.dev, rv.read_cnt, c.n, torch.int32).to(torch.int64)
    tot = int(cnt.sum().item())
    print(f"{c.n} reads, {tot} records, {tot / c.n:.2f} per read, record slots handed out {rv.nrecs}")
    edges = [0, 1, 2, 3, 5, 9, 17, 33, 49, 65, 97, 129, 257, 513, 1025, 4097, 1 << 30]
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (cnt >= lo) & (cnt < hi)
        print(f"  n in [{lo:5d}, {hi - 1:10d}]: {int(m.sum().item()) / c.n * 100:6.2f} % of the reads, {int(cnt[m].sum().item()) / max(1, tot) * 100:6.2f} % of the records")
    # what a 64-record window that starts at a read's first record and takes whole reads only makes of it
    st.close()


if __name__ == "__main__":
    main()
