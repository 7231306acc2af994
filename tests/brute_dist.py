"""A second, independent formulation of `krepp dist` for single-library indexes — test infrastructure only.

Written from the on-disk format and the SET-LEVEL meaning of the reference's per-read path, not from its control
flow (the oracle, oracle/kr_oracle.cpp, restates the control flow):

  * front end: position-list closed form (tests/helpers.py), one k-mer string at a time — no rolling codes, no masks;
  * a colour is the SET of leaves it denotes, computed by memoised recursion over `se_to_pse` — no BFS queue
    (src/query.cpp:369-387 walks the same DAG breadth-first);
  * Minfo::update_match's arrival-order rule (src/query.hpp:153-176) as what it means: per (strand, leaf, read
    position) the MINIMUM Hamming distance over every table entry that matches at that position and whose colour
    contains the leaf; hist[x] = number of positions whose minimum is x;
  * hdist_filt (src/query.cpp:366-368,101-106,119): per strand the minimum hd over kept entries, whatever their colour;
  * likelihood (src/hdhistllh.hpp:71-89) written from the formula in SURVEY.md §8-a9 with math.comb / math.log, and
    minimised by scipy's bounded scalar minimiser (NOT Brent-with-16-bits: agreement of the minimum is checked to
    the minimiser's tolerance, the objective values much tighter).
"""
import math
import os
import struct

import numpy as np

from helpers import closed_form, hd32, revcomp, row_of


class BruteIndex:
    def __init__(self, index_dir):
        files = {f.split("-", 1)[0]: f for f in os.listdir(index_dir) if "." not in f}
        sfx = files["metadata"].split("-", 1)[1]
        rd = lambda t: open(os.path.join(index_dir, f"{t}-{sfx}"), "rb").read()
        md = rd("metadata")
        self.k, self.w, self.h, self.m, self.r, frac, self.nrows = struct.unpack_from("<BBBIIBI", md, 0)
        self.frac = bool(frac)
        self.ppos = list(md[16:16 + self.h])
        self.npos = list(md[16 + self.h:16 + self.k])
        inc = rd("inc")
        assert struct.unpack_from("<I", inc)[0] == self.nrows
        self.inc = np.frombuffer(inc, np.uint64, offset=4)
        cm = rd("cmer")
        nk = struct.unpack_from("<Q", cm)[0]
        self.cmer = np.frombuffer(cm, np.uint32, offset=8).reshape(nk, 2)
        cr = rd("crecord")
        self.nnodes, self.nsubsets = struct.unpack_from("<II", cr)
        self.pse = np.frombuffer(cr, np.uint32, count=2 * self.nsubsets, offset=8).reshape(-1, 2)
        self.rho = np.frombuffer(cr, np.float64, offset=8 + 8 * self.nsubsets, count=self.nnodes)
        # Index::make_rho_partial (src/index.cpp:188-201): residues served by the library / m
        self.rho_scale = ((self.r + 1) if self.frac else 1) / self.m
        self._leaves = {}

    def leaves(self, se):
        """the set of leaf colour ids a colour denotes"""
        if se in self._leaves:
            return self._leaves[se]
        if se == 0 or se >= self.nsubsets:
            out = frozenset()
        else:
            a, b = int(self.pse[se][0]), int(self.pse[se][1])
            if se < self.nnodes and a == 0 and b == se:
                out = frozenset([se])
            else:
                out = self.leaves(a) | self.leaves(b)
        self._leaves[se] = out
        return out

    def bucket(self, row):
        lo = int(self.inc[row - 1]) if row else 0
        return lo, int(self.inc[row])

    # ---- likelihood -----------------------------------------------------------------------------------------
    def f(self, hist, uc, rho, d, th):
        k, h = self.k, self.h
        ll = 0.0
        for x in range(th + 1):
            if hist[x]:
                ll += hist[x] * (k * math.log1p(-d) + x * (math.log(d) - math.log1p(-d)))
        M = 0.0
        for x in range(k + 1):
            c = math.comb(k, x) - (math.comb(k - h, x) if x <= th else 0)
            M += c * (1.0 - d) ** (k - x) * d ** x
        return -ll - uc * math.log(rho * M + 1.0 - rho)

    def dist(self, seq, th=4):
        """-> dict(onmers, hdist_filt[2], hits {(strand, kpos, cmer_index, hd, se)},
                   accs {(strand, leaf): (hist, match_count, hdist_min, passed)})"""
        k = self.k
        onmers = 0
        filt = [0xFFFFFFFF, 0xFFFFFFFF]
        hits = set()
        best = {}  # (strand, leaf) -> {kpos: min hd}
        for i in range(len(seq) - k + 1):
            km = seq[i:i + k]
            fwd = closed_form(km, self.ppos, self.npos)
            if fwd is None:
                continue
            onmers += 1
            for strand, f in ((0, fwd), (1, closed_form(revcomp(km), self.ppos, self.npos))):
                row = row_of(f[2], self.m, self.r, self.frac)
                if row is None:
                    continue
                lo, hi = self.bucket(row)
                for ci in range(lo, hi):
                    e, se = int(self.cmer[ci][0]), int(self.cmer[ci][1])
                    hd = hd32(e, f[3])
                    if hd > th:
                        continue
                    hits.add((strand, i, ci, hd, se))
                    filt[strand] = min(filt[strand], hd)
                    for leaf in self.leaves(se):
                        slot = best.setdefault((strand, leaf), {})
                        slot[i] = min(slot.get(i, 99), hd)
        accs = {}
        for (strand, leaf), posmin in best.items():
            hist = [0] * (th + 1)
            for hd in posmin.values():
                hist[hd] += 1
            hmin = min(posmin.values())
            lim = (2 * filt[strand] + 1) & 0xFFFFFFFF
            accs[(strand, leaf)] = (tuple(hist), len(posmin), hmin, hmin <= lim)
        return dict(onmers=onmers, hdist_filt=filt, hits=hits, accs=accs)

    def minimise(self, hist, match_count, onmers, leaf, th):
        from scipy.optimize import minimize_scalar

        rho = float(self.rho[leaf]) * self.rho_scale
        uc = onmers - match_count
        res = minimize_scalar(lambda d: self.f(hist, uc, rho, d, th), bounds=(1e-10, 0.5), method="bounded",
                              options={"xatol": 1e-12, "maxiter": 500})
        return float(res.x), float(res.fun), rho, uc
