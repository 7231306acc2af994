"""ctypes binding of oracle/liboracle.so and oracle/_ref/libkrepp_ref.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Never imported by krepp_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
ORACLE_SO = _HERE / "liboracle.so"
REF_SO = _HERE / "_ref" / "libkrepp_ref.so"


class KoInfo(C.Structure):
    _fields_ = [("k", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32), ("m", C.c_uint32), ("nlibs", C.c_uint32),
                ("nresidues", C.c_uint32), ("nnodes", C.c_uint32), ("nleaves", C.c_uint32), ("nkmers", C.c_uint64),
                ("nrows", C.c_uint64), ("wbackbone", C.c_uint32), ("pad", C.c_uint32)]


class KoParams(C.Structure):
    _fields_ = [("hdist_th", C.c_uint32), ("tau", C.c_uint32), ("chisq", C.c_double), ("dist_max", C.c_double),
                ("multi", C.c_uint32), ("no_filter", C.c_uint32), ("num_threads", C.c_uint32), ("collect", C.c_uint32)]


ACC_DT = np.dtype([("read", "<u4"), ("se", "<u4"), ("strand", "<u4"), ("match_count", "<u4"), ("hdist_min", "<u4"),
                   ("passed", "<u4"), ("rho", "<f8"), ("d_llh", "<f8"), ("v_llh", "<f8"), ("hist", "<u4", (17,)),
                   ("pad", "<u4")])
ROW_DT = np.dtype([("read", "<u4"), ("se", "<u4"), ("strand", "<u4"), ("match_count", "<u4"), ("d_llh", "<f8"),
                   ("v_llh", "<f8"), ("chisq", "<f8")])
HIT_DT = np.dtype([("read", "<u4"), ("strand", "<u4"), ("pos", "<u4"), ("kpos", "<u4"), ("lib", "<u4"), ("hd", "<u4"),
                   ("cmer_index", "<u8"), ("enc", "<u4"), ("se", "<u4")])
RI_DT = np.dtype([("onmers", "<u4"), ("hdist_filt", "<u4", (2,)), ("nrows", "<u4")])
COUNTER_NAMES = ["reads", "bases", "kmers_valid", "lsh_evals", "probes", "bucket_entries", "hits", "pse_reads",
                 "rho_reads", "accs", "brent_runs", "llh_evals", "rows"]


class KoCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_NAMES]


PLACE_DT = np.dtype([("read", "<u4"), ("edge", "<u4"), ("lwr", "<f8"), ("d_llh", "<f8"), ("v_llh", "<f8"),
                     ("pendant", "<f8"), ("distal", "<f8")])


class KoResult(C.Structure):
    _fields_ = [("nrows", C.c_uint64), ("naccs", C.c_uint64), ("nhits", C.c_uint64), ("rows", C.c_void_p),
                ("accs", C.c_void_p), ("hits", C.c_void_p), ("reads", C.c_void_p), ("placements", C.c_void_p),
                ("nplacements", C.c_uint64), ("text", C.c_void_p), ("text_len", C.c_uint64), ("counters", KoCounters)]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not ORACLE_SO.exists():
            raise ImportError(f"{ORACLE_SO} missing: run `make -C oracle`")
        l = C.CDLL(str(ORACLE_SO))
        vp = C.c_void_p
        l.ko_index_load.restype = vp
        l.ko_index_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        l.ko_index_free.argtypes = [vp]
        l.ko_index_replace_table.argtypes = [vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint64]
        l.ko_index_info.argtypes = [vp, C.POINTER(KoInfo)]
        l.ko_node_name.argtypes = [vp, C.c_uint32]
        l.ko_node_name.restype = C.c_char_p
        l.ko_node_kind.argtypes = [vp, C.c_uint32]
        l.ko_node_parent.argtypes = [vp, C.c_uint32]
        l.ko_node_parent.restype = C.c_uint32
        l.ko_node_blen.argtypes = [vp, C.c_uint32]
        l.ko_node_blen.restype = C.c_double
        l.ko_lsh_positions.argtypes = [vp, vp, vp]
        l.ko_front_end.argtypes = [vp, C.c_char_p, C.c_uint64] + [vp] * 7
        l.ko_front_end.restype = C.c_uint32
        l.ko_dist_batch.argtypes = [vp, vp, vp, vp, C.c_uint32, C.POINTER(KoParams), C.POINTER(KoResult)]
        l.ko_result_free.argtypes = [C.POINTER(KoResult)]
        l.ko_dist_summarize.argtypes = [vp, vp, vp, C.c_uint32, C.POINTER(KoParams), C.POINTER(KoResult)]
        l.ko_index_set_placement_tree.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int]
        l.ko_index_set_lineage_tree.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int]
        l.ko_place_summarize.argtypes = [vp, vp, vp, C.c_uint32, C.POINTER(KoParams), C.POINTER(KoResult)]
        l.ko_sketch_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        l.ko_sketch_load.restype = vp
        l.ko_sketch_free.argtypes = [vp]
        l.ko_sketch_free.restype = None
        l.ko_sketch_info.argtypes = [vp] + [vp] * 9
        l.ko_sketch_info.restype = None
        l.ko_sketch_codes.argtypes = [vp]
        l.ko_sketch_codes.restype = C.POINTER(C.c_uint32)
        l.ko_sketch_inc.argtypes = [vp]
        l.ko_sketch_inc.restype = C.POINTER(C.c_uint64)
        l.ko_sketch_positions.argtypes = [vp, vp, vp]
        l.ko_sketch_positions.restype = None
        l.ko_seek_batch.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.POINTER(KoResult)]
        l.ko_place_batch.argtypes = [vp, vp, vp, vp, C.c_uint32, C.POINTER(KoParams), C.c_int, C.POINTER(KoResult)]
        l.ko_place_frame.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.c_uint64]
        l.ko_place_frame.restype = vp
        l.ko_llh.argtypes = [C.c_uint32] * 3 + [vp, C.c_double, C.c_double, C.c_double]
        l.ko_llh.restype = C.c_double
        l.ko_brent.argtypes = [C.c_uint32] * 3 + [vp, C.c_double, C.c_double, vp, vp]
        l.ko_revcomp_bp64.argtypes = [C.c_uint64, C.c_uint32]
        l.ko_revcomp_bp64.restype = C.c_uint64
        l.ko_conv_bp64_lr64.argtypes = [C.c_uint64]
        l.ko_conv_bp64_lr64.restype = C.c_uint64
        l.ko_murmur3_x86_32.argtypes = [C.c_char_p, C.c_int, C.c_uint32]
        l.ko_murmur3_x86_32.restype = C.c_uint32
        l.ko_name_hash.argtypes = [C.c_char_p]
        l.ko_name_hash.restype = C.c_uint64
        l.ko_xur64.argtypes = [C.c_uint64]
        l.ko_xur64.restype = C.c_uint64
        _lib = l
    return _lib


def ref():
    """oracle/_ref/libkrepp_ref.so: the reference's own standalone sources, compiled."""
    global _ref
    if _ref is None:
        if not REF_SO.exists():
            return None
        r = C.CDLL(str(REF_SO))
        r.ref_llh.argtypes = [C.c_uint32] * 3 + [C.c_void_p, C.c_double, C.c_double, C.c_double]
        r.ref_llh.restype = C.c_double
        r.ref_murmur3_x86_32.argtypes = [C.c_char_p, C.c_int, C.c_uint32]
        r.ref_murmur3_x86_32.restype = C.c_uint32
        r.ref_kseq_parse.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
        r.ref_kseq_parse.restype = C.c_long
        r.ref_hll_estimate.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        r.ref_hll_estimate.restype = C.c_double
        _ref = r
    return _ref


def params(hdist_th=4, tau=2, chisq=2.706, dist_max=float("nan"), multi=1, no_filter=1, num_threads=1, collect=1):
    return KoParams(hdist_th, tau, chisq, dist_max, multi, no_filter, num_threads, collect)


class Index:
    def __init__(self, index_dir):
        self.l = lib()
        err = C.create_string_buffer(512)
        self.h = self.l.ko_index_load(os.fsencode(str(index_dir)), err, 512)
        if not self.h:
            raise RuntimeError("oracle: " + err.value.decode())
        self.info = KoInfo()
        self.l.ko_index_info(self.h, C.byref(self.info))

    def replace_table(self, lib_ix, inc, cmer):
        inc = np.ascontiguousarray(inc, dtype=np.uint64)
        cmer = np.ascontiguousarray(cmer, dtype=np.uint32)
        rc = self.l.ko_index_replace_table(self.h, lib_ix, inc.ctypes.data, len(inc), cmer.ctypes.data, cmer.size // 2)
        assert rc == 0
        self.l.ko_index_info(self.h, C.byref(self.info))

    def name(self, se):
        return self.l.ko_node_name(self.h, int(se)).decode()

    def kind(self, se):
        return self.l.ko_node_kind(self.h, int(se))

    def parent(self, se):
        return int(self.l.ko_node_parent(self.h, int(se)))

    def blen(self, se):
        return float(self.l.ko_node_blen(self.h, int(se)))

    def positions(self):
        p = np.zeros(self.info.h, np.uint8)
        n = np.zeros(self.info.k - self.info.h, np.uint8)
        self.l.ko_lsh_positions(self.h, p.ctypes.data, n.ctypes.data)
        return p, n

    def front_end(self, seq: bytes):
        n = max(1, 2 * len(seq))
        kpos = np.zeros(n, np.uint32)
        strand = np.zeros(n, np.uint8)
        bp = np.zeros(n, np.uint64)
        lr = np.zeros(n, np.uint64)
        rix = np.zeros(n, np.uint32)
        enc = np.zeros(n, np.uint32)
        pas = np.zeros(n, np.uint8)
        c = self.l.ko_front_end(self.h, seq, len(seq), kpos.ctypes.data, strand.ctypes.data, bp.ctypes.data,
                                lr.ctypes.data, rix.ctypes.data, enc.ctypes.data, pas.ctypes.data)
        return dict(kpos=kpos[:c], strand=strand[:c], enc_bp=bp[:c], enc_lr=lr[:c], rix=rix[:c], enc32=enc[:c], pas=pas[:c])

    def dist(self, bases, offsets, names=None, p=None):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        p = p or params()
        res = KoResult()
        arr = None
        if names is not None:
            arr = (C.c_char_p * n)(*[s.encode() for s in names])
        rc = self.l.ko_dist_batch(self.h, bases.ctypes.data, offsets.ctypes.data, arr, n, C.byref(p), C.byref(res))
        if rc:
            raise RuntimeError(f"oracle ko_dist_batch rc={rc}")

        def grab(ptr, cnt, dt):
            if cnt == 0:
                return np.zeros(0, dt)
            return np.frombuffer(C.string_at(ptr, cnt * dt.itemsize), dtype=dt).copy()

        out = dict(rows=grab(res.rows, res.nrows, ROW_DT), accs=grab(res.accs, res.naccs, ACC_DT),
                   hits=grab(res.hits, res.nhits, HIT_DT), reads=grab(res.reads, n, RI_DT),
                   text=C.string_at(res.text, res.text_len).decode() if res.text_len else "",
                   counters={k_: int(getattr(res.counters, k_)) for k_ in COUNTER_NAMES})
        self.l.ko_result_free(C.byref(res))
        return out

    def summarize(self, bases, offsets, p=None):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        p = p or params()
        res = KoResult()
        rc = self.l.ko_dist_summarize(self.h, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1, C.byref(p), C.byref(res))
        assert rc == 0
        txt = C.string_at(res.text, res.text_len).decode() if res.text_len else ""
        self.l.ko_result_free(C.byref(res))
        return txt

    def set_placement_tree(self, nwk_text=None):
        err = C.create_string_buffer(512)
        rc = self.l.ko_index_set_placement_tree(self.h, nwk_text.encode() if nwk_text is not None else None, err, 512)
        if rc:
            raise RuntimeError("oracle: " + err.value.decode())

    def set_lineage_tree(self, lineage_text):
        err = C.create_string_buffer(512)
        rc = self.l.ko_index_set_lineage_tree(self.h, lineage_text.encode(), err, 512)
        if rc:
            raise RuntimeError("oracle: " + err.value.decode())

    def place_summarize(self, bases, offsets, p=None):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        p = p or params(no_filter=0)
        res = KoResult()
        rc = self.l.ko_place_summarize(self.h, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1, C.byref(p), C.byref(res))
        if rc:
            raise RuntimeError(f"oracle ko_place_summarize rc={rc}")
        txt = C.string_at(res.text, res.text_len).decode() if res.text_len else ""
        self.l.ko_result_free(C.byref(res))
        return txt

    def place(self, bases, offsets, names=None, p=None, tabular=False):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        p = p or params(no_filter=0)
        res = KoResult()
        arr = (C.c_char_p * n)(*[s.encode() for s in names]) if names is not None else None
        rc = self.l.ko_place_batch(self.h, bases.ctypes.data, offsets.ctypes.data, arr, n, C.byref(p), int(tabular), C.byref(res))
        if rc:
            raise RuntimeError(f"oracle ko_place_batch rc={rc}")
        pl = (np.frombuffer(C.string_at(res.placements, res.nplacements * PLACE_DT.itemsize), dtype=PLACE_DT).copy()
              if res.nplacements else np.zeros(0, PLACE_DT))
        out = dict(placements=pl, text=C.string_at(res.text, res.text_len).decode() if res.text_len else "",
                   counters={k_: int(getattr(res.counters, k_)) for k_ in COUNTER_NAMES})
        self.l.ko_result_free(C.byref(res))
        return out

    def place_frame(self, which, tabular=False, invocation="", total=0):
        ptr = self.l.ko_place_frame(self.h, which, int(tabular), invocation.encode(), total)
        s = C.string_at(ptr).decode()
        C.CDLL(None).free(C.c_void_p(ptr))
        return s

    def close(self):
        if self.h:
            self.l.ko_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def llh(k, h, th, hist, uc, rho, d):
    a = np.ascontiguousarray(hist, dtype=np.float64)
    return lib().ko_llh(k, h, th, a.ctypes.data, uc, rho, d)


def brent(k, h, th, hist, uc, rho):
    a = np.ascontiguousarray(hist, dtype=np.float64)
    d = C.c_double()
    v = C.c_double()
    ne = lib().ko_brent(k, h, th, a.ctypes.data, uc, rho, C.addressof(d), C.addressof(v))
    return d.value, v.value, ne


def algorithmic_bytes(counters):
    """bytes the reference's algorithm must touch for these reads."""
    c = counters
    return (c["bases"] + 16 * c["probes"] + 8 * c["bucket_entries"] + 8 * c["pse_reads"] + 8 * c["rho_reads"]
            + 16 * c["rows"])


class Sketch:
    """A `krepp sketch` file read by the oracle (src/sketch.cpp:3-39) and `krepp seek` on it (src/seek.cpp)."""

    def __init__(self, path):
        self.l = lib()
        err = C.create_string_buffer(512)
        self.h = self.l.ko_sketch_load(str(path).encode(), err, 512)
        if not self.h:
            raise RuntimeError("oracle: " + err.value.decode())
        nk, nrows, k, w, h, m, r, frac = C.c_uint64(), *[C.c_uint32() for _ in range(7)]
        rho = C.c_double()
        self.l.ko_sketch_info(self.h, *[C.addressof(x) for x in (nk, nrows, k, w, h, m, r, frac, rho)])
        self.nkmers, self.nrows, self.k, self.w, self.hh, self.m, self.r, self.frac = (
            nk.value, nrows.value, k.value, w.value, h.value, m.value, r.value, frac.value)
        self.rho = rho.value
        self.codes = np.ctypeslib.as_array(self.l.ko_sketch_codes(self.h), shape=(self.nkmers,)).copy() if self.nkmers else np.zeros(0, np.uint32)
        self.inc = np.ctypeslib.as_array(self.l.ko_sketch_inc(self.h), shape=(self.nrows,)).copy()
        pp, npz = np.zeros(self.hh, np.uint8), np.zeros(self.k - self.hh, np.uint8)
        self.l.ko_sketch_positions(self.h, pp.ctypes.data, npz.ctypes.data)
        self.ppos, self.npos = pp, npz

    def seek(self, bases, offsets, names=None, hdist_th=4):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        res = KoResult()
        arr = (C.c_char_p * n)(*[s.encode() for s in names]) if names is not None else None
        rc = self.l.ko_seek_batch(self.h, bases.ctypes.data, offsets.ctypes.data, arr, n, hdist_th, C.byref(res))
        if rc:
            raise RuntimeError(f"oracle ko_seek_batch rc={rc}")
        rows = np.frombuffer(C.string_at(res.rows, res.nrows * ROW_DT.itemsize), dtype=ROW_DT).copy()
        text = C.string_at(res.text, res.text_len).decode() if res.text_len else ""
        self.l.ko_result_free(C.byref(res))
        return dict(rows=rows, text=text)

    def close(self):
        if self.h:
            self.l.ko_sketch_free(self.h)
            self.h = None
