#!/usr/bin/env python3
"""How often do the likelihood problems of one batch come back in the next?  A problem is (leaf, the read's k-mer count, the five
Hamming-distance counts) -- what kr_dedup_kernel makes distinct WITHIN a batch; nothing is kept between batches.  Four batches of bench.py's
workload (KR_TAP_ACCS: the records' histograms come back), their sets of distinct problems, and how many of a batch's are already in the
union of the batches before it.  Usage: python scripts/llh_problem_repetition.py [bench.py's options] (use --reads-per-step 1000000)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from krepp_amd import capi  # noqa: E402


def main():
    a = bench.finish_args(bench.build_parser().parse_args())
    a.distinct_batches = max(a.distinct_batches, 4)
    c = bench.prepare(a)
    n = c.n
    st = c.dx.stream(max_reads=n, max_bases=len(c.bases), max_records=c.max_rec)
    seen = None
    for b in range(min(4, c.nb)):
        bases, offs = c.batches[b]
        st.submit(bases, offs, capi.KR_TAP_ACCS)
        r = st.collect()
        om = r.read_onmers[r.rec_read].astype(np.uint64)
        h = r.rec_hist.astype(np.uint64)
        direct = (h[:, :5].sum(axis=1) <= 3) & (h[:, :5].max(axis=1) <= 3)
        key = (r.rec_key.astype(np.uint64) >> np.uint64(1)) | (om << np.uint64(20))
        word = h[:, 0] | (h[:, 1] << np.uint64(8)) | (h[:, 2] << np.uint64(16)) | (h[:, 3] << np.uint64(24)) | (h[:, 4] << np.uint64(32))
        both = np.stack([key, word], axis=1)
        for name, m in (("all", np.ones(len(both), bool)), ("direct part", direct), ("table part", ~direct)):
            u = np.unique(both[m], axis=0)
            line = f"batch {b}: {name:11s} {int(m.sum()):>10d} records, {len(u):>9d} distinct problems"
            if seen is not None:
                prev = seen[name]
                v = np.concatenate([prev, u])
                uu, cnt = np.unique(v, axis=0, return_counts=True)
                again = int((cnt > 1).sum())
                line += f", {again:>9d} of them ({again / max(1, len(u)) * 100:5.1f} %) met in the batches before ({len(prev)} kept)"
                seen[name] = uu
            print(line, flush=True)
        if seen is None:
            seen = {}
            for name, m in (("all", np.ones(len(both), bool)), ("direct part", direct), ("table part", ~direct)):
                seen[name] = np.unique(both[m], axis=0)
    st.close()


if __name__ == "__main__":
    main()
