"""ctypes binding of include/krepp_amd.h (the C ABI of the MI355X `krepp dist` path).

This module is plumbing only: it loads ``krepp_amd/lib/libkrepp_amd.so`` (built by
``__graft_entry__.build()`` / ``krepp_amd/csrc/Makefile``) and fails loudly if it is
missing.  There is no Python or CPU fallback for any device entry point.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "lib" / "libkrepp_amd.so"

KR_OK = 0
KR_ERR_ARG, KR_ERR_IO, KR_ERR_FORMAT, KR_ERR_NO_DEVICE = -1, -2, -3, -4
KR_ERR_NOMEM, KR_ERR_CAPACITY, KR_ERR_STATE, KR_ERR_UNSUPPORTED = -5, -6, -7, -8
KR_VIEW_HOST, KR_VIEW_DEVICE = 0, 1
KR_BASES_HOST, KR_BASES_DEVICE, KR_TAP_ACCS, KR_TAP_HITS, KR_BASES_PINNED, KR_ROWS_ONLY, KR_ROWS_INDEXED = 0, 1, 2, 4, 8, 16, 32

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)


class KrLibView(C.Structure):
    _fields_ = [("inc", u64p), ("cmer", u32p), ("pse", u32p), ("rho", f64p),
                ("nkmers", C.c_uint64), ("nrows", C.c_uint32), ("nsubsets", C.c_uint32),
                ("nnodes", C.c_uint32), ("r", C.c_uint32), ("frac", C.c_uint32), ("w", C.c_uint32)]


class KrIndexView(C.Structure):
    _fields_ = [("k", C.c_uint32), ("h", C.c_uint32), ("m", C.c_uint32),
                ("ppos", u8p), ("npos", u8p), ("nlibs", C.c_uint32), ("libs", C.POINTER(KrLibView)),
                ("tree_nnodes", C.c_uint32), ("node_kind", u8p), ("wbackbone", C.c_uint32)]


class KrIndexBuffer(C.Structure):
    _fields_ = [("dptr", C.c_void_p), ("bytes", C.c_uint64)]


class KrParams(C.Structure):
    _fields_ = [("hdist_th", C.c_uint32), ("tau", C.c_uint32), ("chisq", C.c_double),
                ("dist_max", C.c_double), ("multi", C.c_uint32), ("no_filter", C.c_uint32)]


class KrResultView(C.Structure):
    _fields_ = [("nreads", C.c_uint32), ("nrecs", C.c_uint32),
                ("read_off", u32p), ("read_cnt", u32p), ("read_onmers", u32p), ("read_na", u8p),
                ("rec_key", u32p), ("rec_sel", u8p), ("rec_d", f64p), ("rec_v", f64p),
                ("rec_chisq", f64p), ("rec_hist", u32p), ("rec_hist_stride", C.c_uint64), ("nrows", C.c_uint64),
                ("rec_dix", u32p), ("dist_list", f64p), ("ndist", C.c_uint64)]


class KrHit(C.Structure):
    _fields_ = [("read", C.c_uint32), ("kpos", C.c_uint32), ("strand", C.c_uint32), ("lib", C.c_uint32),
                ("cmer_index", C.c_uint64), ("hd", C.c_uint32), ("se", C.c_uint32)]


class KrTiming(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_scan", C.c_float), ("ms_acc", C.c_float), ("ms_llh", C.c_float),
                ("ms_h2d", C.c_float), ("overflow_reads", C.c_uint32), ("stack_spills", C.c_uint32), ("lanes", C.c_uint32)]


class KrFastxBatch(C.Structure):
    _fields_ = [("bases", u8p), ("offsets", u64p), ("names", C.POINTER(C.c_char_p)),
                ("nreads", C.c_uint32), ("more", C.c_uint32)]


class KrMinimizerResult(C.Structure):
    _fields_ = [("keys", u64p), ("nkeys", C.c_uint64), ("n1", C.c_double), ("n2", C.c_double)]


class KrBuildParams(C.Structure):
    _fields_ = [("k", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32), ("m", C.c_uint32), ("r", C.c_uint32),
                ("frac", C.c_uint32), ("num_threads", C.c_uint32), ("seed", C.c_uint32), ("ppos", u8p),
                ("gpu_minimizers", C.c_uint32), ("device", C.c_int32)]


# every symbol include/krepp_amd.h declares (tests check that the library exports them all)
EXPORTS = [
    "kr_host_index_load", "kr_host_sketch_load", "kr_format_seek", "kr_build_sketch", "kr_host_index_free", "kr_host_index_view", "kr_host_index_node_name",
    "kr_host_index_node_label", "kr_host_index_node_parent", "kr_host_index_node_blen",
    "kr_index_upload", "kr_index_free", "kr_index_export", "kr_index_import", "kr_index_device_bytes", "kr_index_slot_words", "kr_index_slot_format", "kr_index_broadcast",
    "kr_params_default", "kr_stream_create", "kr_stream_destroy", "kr_batch_submit", "kr_batch_wait",
    "kr_batch_collect", "kr_batch_collect_device", "kr_stream_text_enable", "kr_batch_submit_text", "kr_batch_collect_text", "kr_batch_hits", "kr_batch_readtaps",
    "kr_debug_front_end", "kr_debug_stream_move", "kr_debug_stream_addrs", "kr_debug_item_placement", "kr_debug_brent", "kr_debug_colour_classes", "kr_llh_batch", "kr_llh_eval_indexed", "kr_batch_timing",
    "kr_place_tree_create", "kr_place_tree_create_lineage", "kr_place_tree_nnodes", "kr_place_summary_add",
    "kr_place_summary_text", "kr_place_tree_free", "kr_place_tree_kinds", "kr_place_batch", "kr_place_stream", "kr_place_frame", "kr_place_counters",
    "kr_debug_last_d2h_bytes", "kr_debug_place_fixed5", "kr_place_text_counters", "kr_fastx_open", "kr_fastx_next", "kr_fastx_detach", "kr_fastx_release", "kr_fastx_close", "kr_fastx_parallel_chunks", "kr_fastx_pgz_stats", "kr_format_dist", "kr_debug_fixed5", "kr_free", "kr_host_alloc", "kr_host_free",
    "kr_build_index", "kr_minimizers_cpu", "kr_minimizers_device", "kr_minimizers_free", "kr_last_error", "kr_version",
]

_lib = None


class KrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"krepp_amd error {code}: {msg}")
        self.code = code


def load():
    """Load libkrepp_amd.so; raise if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the HIP extension is required; there is no CPU fallback)")
    lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    lib.kr_last_error.restype = C.c_char_p
    lib.kr_version.restype = C.c_char_p
    lib.kr_host_index_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.kr_host_index_free.argtypes = [vp]
    lib.kr_host_index_free.restype = None
    lib.kr_host_index_view.argtypes = [vp, C.POINTER(KrIndexView)]
    lib.kr_host_index_node_name.argtypes = [vp, C.c_uint32]
    lib.kr_host_index_node_name.restype = C.c_char_p
    lib.kr_host_index_node_label.argtypes = [vp, C.c_uint32]
    lib.kr_host_index_node_label.restype = C.c_char_p
    lib.kr_host_index_node_parent.argtypes = [vp, C.c_uint32]
    lib.kr_host_index_node_parent.restype = C.c_uint32
    lib.kr_host_index_node_blen.argtypes = [vp, C.c_uint32]
    lib.kr_host_index_node_blen.restype = C.c_double
    lib.kr_index_upload.argtypes = [C.POINTER(KrIndexView), C.c_int, C.c_uint32, C.POINTER(vp)]
    lib.kr_index_free.argtypes = [vp]
    lib.kr_index_free.restype = None
    lib.kr_index_export.argtypes = [vp, vp, u64p, C.POINTER(KrIndexBuffer), u32p]
    lib.kr_index_import.argtypes = [vp, C.c_uint64, C.c_int, C.POINTER(vp), C.POINTER(KrIndexBuffer), u32p]
    lib.kr_index_device_bytes.argtypes = [vp]
    lib.kr_index_broadcast.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]
    lib.kr_index_device_bytes.restype = C.c_uint64
    lib.kr_index_slot_words.argtypes = [vp]
    lib.kr_index_slot_format.argtypes = [vp]
    lib.kr_index_slot_format.restype = C.c_uint32
    lib.kr_debug_stream_move.argtypes = [vp, C.c_int]
    lib.kr_debug_stream_addrs.argtypes = [vp, u64p]
    lib.kr_debug_item_placement.argtypes = [vp, u32p, u32p, C.POINTER(C.c_double)]
    lib.kr_index_slot_words.restype = C.c_uint32
    lib.kr_params_default.argtypes = [C.POINTER(KrParams)]
    lib.kr_params_default.restype = None
    lib.kr_stream_create.argtypes = [vp, C.POINTER(KrParams), C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(vp)]
    lib.kr_stream_destroy.argtypes = [vp]
    lib.kr_stream_destroy.restype = None
    lib.kr_batch_submit.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32]
    lib.kr_debug_fixed5.argtypes = [C.c_double, C.c_char_p]
    lib.kr_debug_fixed5.restype = C.c_uint32
    lib.kr_stream_text_enable.argtypes = [vp, vp, C.c_uint64, C.c_uint64]
    lib.kr_batch_submit_text.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, vp, vp, C.c_uint32]
    lib.kr_batch_collect_text.argtypes = [vp, C.POINTER(vp), u64p]
    lib.kr_batch_wait.argtypes = [vp]
    lib.kr_batch_collect.argtypes = [vp, C.POINTER(KrResultView)]
    lib.kr_batch_collect_device.argtypes = [vp, C.POINTER(KrResultView)]
    lib.kr_batch_hits.argtypes = [vp, C.POINTER(C.POINTER(KrHit)), u64p]
    lib.kr_batch_readtaps.argtypes = [vp, C.POINTER(u32p)]
    lib.kr_debug_front_end.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
    lib.kr_debug_brent.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp]
    lib.kr_debug_colour_classes.argtypes = [vp, C.c_uint32, vp, C.c_uint32, vp, vp, vp, C.POINTER(C.c_uint64)]
    lib.kr_batch_timing.argtypes = [vp, C.POINTER(KrTiming)]
    lib.kr_llh_batch.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, vp, vp, vp, vp, vp, vp]
    lib.kr_place_tree_create.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    lib.kr_place_tree_create_lineage.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    lib.kr_place_tree_nnodes.argtypes = [vp]
    lib.kr_place_tree_nnodes.restype = C.c_uint32
    lib.kr_place_summary_add.argtypes = [vp, vp, C.c_uint64, vp, C.POINTER(C.c_double)]
    lib.kr_place_summary_text.argtypes = [vp, vp, C.c_double, C.POINTER(vp), u64p]
    lib.kr_place_tree_free.argtypes = [vp]
    lib.kr_place_tree_free.restype = None
    lib.kr_place_tree_kinds.argtypes = [vp]
    lib.kr_place_tree_kinds.restype = u8p
    lib.kr_place_batch.argtypes = [vp, vp, vp, C.POINTER(KrResultView), vp, C.POINTER(C.c_char_p), C.POINTER(KrParams), C.c_int,
                                   C.POINTER(C.c_int), C.POINTER(vp), u64p, C.POINTER(vp), u64p]
    lib.kr_place_stream.argtypes = [vp, vp, vp, vp, C.c_uint32, vp, C.POINTER(C.c_char_p), C.POINTER(KrParams), C.c_int,
                                    C.POINTER(C.c_int), C.POINTER(vp), u64p, C.POINTER(vp), u64p]
    lib.kr_place_frame.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.c_uint64, C.POINTER(vp), u64p]
    lib.kr_place_counters.argtypes = [u64p, u64p, u64p]
    lib.kr_place_counters.restype = None
    lib.kr_fastx_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.kr_fastx_next.argtypes = [vp, C.c_uint64, C.POINTER(KrFastxBatch)]
    lib.kr_fastx_detach.argtypes = [vp, C.POINTER(vp)]
    lib.kr_fastx_release.argtypes = [vp, vp]
    lib.kr_fastx_release.restype = None
    lib.kr_fastx_close.argtypes = [vp]
    lib.kr_fastx_close.restype = None
    lib.kr_format_dist.argtypes = [vp, C.POINTER(KrResultView), C.POINTER(C.c_char_p), C.POINTER(vp), u64p]
    lib.kr_free.argtypes = [vp]
    lib.kr_free.restype = None
    lib.kr_host_alloc.argtypes = [C.c_uint64]
    lib.kr_host_alloc.restype = vp
    lib.kr_host_free.argtypes = [vp]
    lib.kr_host_free.restype = None
    lib.kr_build_index.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(KrBuildParams)]
    lib.kr_minimizers_cpu.argtypes = [C.POINTER(KrBuildParams), vp, vp, C.c_uint32, C.POINTER(KrMinimizerResult)]
    lib.kr_minimizers_device.argtypes = [C.c_int, C.POINTER(KrBuildParams), vp, vp, C.c_uint32, C.POINTER(KrMinimizerResult)]
    lib.kr_minimizers_free.argtypes = [C.POINTER(KrMinimizerResult)]
    lib.kr_minimizers_free.restype = None
    _lib = lib
    return lib


def check(rc):
    if rc != KR_OK:
        raise KrError(rc, load().kr_last_error().decode(errors="replace"))


def _np(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).view(dtype).copy()


class HostIndex:
    """Index directory read into host memory (reference: TargetIndex::load_index, src/krepp.cpp:66-108)."""

    def __init__(self, index_dir, sketch=False):
        """sketch=True: `index_dir` is a `krepp sketch` file (kr_host_sketch_load), served as a one-leaf index"""
        self.lib = load()
        self.h = C.c_void_p()
        if sketch:
            self.lib.kr_host_sketch_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
            check(self.lib.kr_host_sketch_load(os.fsencode(str(index_dir)), C.byref(self.h)))
        else:
            check(self.lib.kr_host_index_load(os.fsencode(str(index_dir)), C.byref(self.h)))
        self.view = KrIndexView()
        check(self.lib.kr_host_index_view(self.h, C.byref(self.view)))

    k = property(lambda s: s.view.k)
    hh = property(lambda s: s.view.h)
    m = property(lambda s: s.view.m)
    nnodes = property(lambda s: s.view.tree_nnodes)

    def name(self, se):
        return self.lib.kr_host_index_node_name(self.h, int(se)).decode()

    def parent(self, se):
        return int(self.lib.kr_host_index_node_parent(self.h, int(se)))

    def blen(self, se):
        return float(self.lib.kr_host_index_node_blen(self.h, int(se)))

    def kinds(self):
        return _np(self.view.node_kind, self.view.tree_nnodes + 1, np.uint8)

    def positions(self):
        return _np(self.view.ppos, self.view.h, np.uint8), _np(self.view.npos, self.view.k - self.view.h, np.uint8)

    def lib_arrays(self, i=0):
        lv = self.view.libs[i]
        return dict(inc=_np(lv.inc, lv.nrows, np.uint64), cmer=_np(lv.cmer, 2 * lv.nkmers, np.uint32).reshape(-1, 2),
                    pse=_np(lv.pse, 2 * lv.nsubsets, np.uint32).reshape(-1, 2), rho=_np(lv.rho, lv.nnodes, np.float64),
                    r=lv.r, frac=lv.frac, w=lv.w, nnodes=lv.nnodes, nsubsets=lv.nsubsets)

    def upload(self, device=0):
        return DeviceIndex.from_view(self.view, device, KR_VIEW_HOST, keep=self)

    def close(self):
        if self.h:
            self.lib.kr_host_index_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_params(**kw):
    p = KrParams()
    load().kr_params_default(C.byref(p))
    for k_, v in kw.items():
        setattr(p, k_, v)
    return p


class DeviceIndex:
    """Index resident in one GPU's HBM (the read-only state IBatch borrows from Index)."""

    def __init__(self, handle, keep=None):
        self.lib = load()
        self.h = handle
        self._keep = keep

    @classmethod
    def from_view(cls, view, device=0, flags=KR_VIEW_HOST, keep=None):
        lib = load()
        h = C.c_void_p()
        check(lib.kr_index_upload(C.byref(view), int(device), int(flags), C.byref(h)))
        return cls(h, keep)

    def export(self):
        """(descriptor bytes, [(device pointer, bytes), ...]) of the flat buffers, for replication."""
        nb = C.c_uint64(0)
        n = C.c_uint32(0)
        check(self.lib.kr_index_export(self.h, None, C.byref(nb), None, C.byref(n)))
        desc = C.create_string_buffer(nb.value)
        bufs = (KrIndexBuffer * n.value)()
        check(self.lib.kr_index_export(self.h, desc, C.byref(nb), bufs, C.byref(n)))
        return desc.raw, [(b.dptr, b.bytes) for b in bufs]

    @classmethod
    def import_empty(cls, desc, device=0):
        """Allocate the buffers described by `desc` on `device`; the caller fills them."""
        lib = load()
        h = C.c_void_p()
        n = C.c_uint32(256)
        bufs = (KrIndexBuffer * 256)()
        check(lib.kr_index_import(desc, len(desc), int(device), C.byref(h), bufs, C.byref(n)))
        return cls(h), [(bufs[i].dptr, bufs[i].bytes) for i in range(n.value)]

    def broadcast(self, devices):
        """Replicas of this index on `devices`, filled by RCCL broadcast (kr_index_broadcast)."""
        n = len(devices)
        devs = (C.c_int * n)(*[int(d) for d in devices])
        outs = (C.c_void_p * n)()
        check(self.lib.kr_index_broadcast(self.h, n, devs, outs))
        return [DeviceIndex(C.c_void_p(outs[i])) for i in range(n)]

    @property
    def device_bytes(self):
        return int(self.lib.kr_index_device_bytes(self.h))

    @property
    def slot_words(self):
        return int(self.lib.kr_index_slot_words(self.h))

    @property
    def slot_format(self):
        return int(self.lib.kr_index_slot_format(self.h))

    def stream(self, params=None, max_reads=1 << 16, max_bases=None, max_records=0):
        return Stream(self, params or default_params(), max_reads, max_bases or max_reads * 160, max_records)

    def front_end(self, bases, offsets, stride):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        rix = np.zeros((n, stride, 2), np.uint32)
        enc = np.zeros((n, stride, 2), np.uint32)
        valid = np.zeros((n, stride, 2), np.uint8)
        pas = np.zeros((n, stride, 2), np.uint8)
        check(self.lib.kr_debug_front_end(self.h, bases.ctypes.data, offsets.ctypes.data, n, stride,
                                          rix.ctypes.data, enc.ctypes.data, valid.ctypes.data, pas.ctypes.data))
        return rix, enc, valid, pas

    def brent(self, th, hist, onmers, rho):
        hist = np.ascontiguousarray(hist, dtype=np.uint32)
        onmers = np.ascontiguousarray(onmers, dtype=np.uint32)
        rho = np.ascontiguousarray(rho, dtype=np.float64)
        n = len(onmers)
        d = np.zeros(n)
        v = np.zeros(n)
        check(self.lib.kr_debug_brent(self.h, th, n, hist.ctypes.data, onmers.ctypes.data, rho.ctypes.data,
                                      d.ctypes.data, v.ctypes.data))
        return d, v

    def close(self):
        if self.h:
            self.lib.kr_index_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Result:
    """Host copy of one batch's results (see kr_result_view)."""

    def __init__(self, rv, np_planes, copy_hist):
        n, c = rv.nreads, rv.nrecs
        self.nreads, self.nrecs, self.nrows = n, c, int(rv.nrows)
        self.read_off = _np(rv.read_off, n, np.uint32)
        self.read_cnt = _np(rv.read_cnt, n, np.uint32)
        self.read_onmers = _np(rv.read_onmers, n, np.uint32)
        self.read_na = _np(rv.read_na, n, np.uint8)
        key = _np(rv.rec_key, c, np.uint32)
        keep = key != 0  # record slots are handed out in per-wave chunks; key 0 marks an unused slot
        self.rec_key = key[keep]
        self.rec_sel = _np(rv.rec_sel, c, np.uint8)[keep]
        if rv.rec_dix:  # KR_ROWS_INDEXED: DIST as an index into the batch's distinct values
            self.rec_dix = _np(rv.rec_dix, c, np.uint32)[keep]
            self.dist_list = _np(rv.dist_list, int(rv.ndist), np.float64)
            self.rec_d = self.dist_list[self.rec_dix]
        else:
            self.rec_dix = None
            self.rec_d = _np(rv.rec_d, c, np.float64)[keep]
        self.rec_v = _np(rv.rec_v, c, np.float64)[keep] if rv.rec_v else None       # NULL with KR_ROWS_ONLY
        self.rec_chisq = _np(rv.rec_chisq, c, np.float64)[keep] if rv.rec_chisq else None
        self.rec_hist = (_np(rv.rec_hist, c * np_planes, np.uint32).reshape(np_planes, -1).T[keep]
                         if copy_hist and rv.rec_hist and c else (np.zeros((0, np_planes), np.uint32) if copy_hist else None))
        # read index of every record, and offsets into the compacted arrays
        # (vectorised: a record belongs to the read whose first record is the last start at or before it)
        rr = np.zeros(c, np.uint32)
        nzr = np.nonzero(self.read_cnt)[0]
        if c and len(nzr):
            start = np.zeros(c, np.int64)
            start[self.read_off[nzr]] = self.read_off[nzr].astype(np.int64) + 1
            at = np.zeros(c + 1, np.uint32)
            at[self.read_off[nzr].astype(np.int64) + 1] = nzr
            rr = at[np.maximum.accumulate(start)]
        self.rec_read = rr[keep]
        newpos = np.cumsum(keep) - 1
        nz = self.read_cnt > 0
        self.read_off = np.where(nz, newpos[np.minimum(self.read_off, max(c - 1, 0))] if c else 0, 0).astype(np.uint32)
        self.nrecs = int(keep.sum())

    def rows(self):
        """Sorted list of (read, se, d) output rows — `krepp dist` rows as a set."""
        sel = self.rec_sel.astype(bool)
        return sorted(zip(self.rec_read[sel].tolist(), (self.rec_key[sel] >> 1).tolist(), self.rec_d[sel].tolist()))


class Stream:
    def __init__(self, dindex, params, max_reads, max_bases, max_records=0):
        self.lib = load()
        self.ix = dindex
        self.params = params
        self.h = C.c_void_p()
        check(self.lib.kr_stream_create(dindex.h, C.byref(params), int(max_reads), int(max_bases), int(max_records), C.byref(self.h)))
        self._keep = None
        self._flags = 0

    def submit(self, bases, offsets, flags=0):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._keep = (bases, offsets)
        self._flags = flags
        check(self.lib.kr_batch_submit(self.h, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1, flags))

    def submit_device(self, bases_ptr, offsets_ptr, nreads, flags=0):
        self._flags = flags | KR_BASES_DEVICE
        check(self.lib.kr_batch_submit(self.h, int(bases_ptr), int(offsets_ptr), int(nreads), self._flags))

    def wait(self):
        check(self.lib.kr_batch_wait(self.h))

    def collect(self):
        rv = KrResultView()
        check(self.lib.kr_batch_collect(self.h, C.byref(rv)))
        self._rv = rv
        return Result(rv, self.params.hdist_th + 1, bool(self._flags & KR_TAP_ACCS))

    def collect_view(self):
        """kr_batch_collect without building numpy copies: the raw result view (pointers into the stream's pinned buffers)."""
        rv = KrResultView()
        check(self.lib.kr_batch_collect(self.h, C.byref(rv)))
        self._rv = rv
        return rv

    def last_d2h_bytes(self):
        b = C.c_uint64(0)
        self.lib.kr_debug_last_d2h_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        check(self.lib.kr_debug_last_d2h_bytes(self.h, C.byref(b)))
        return int(b.value)

    def collect_device(self):
        rv = KrResultView()
        check(self.lib.kr_batch_collect_device(self.h, C.byref(rv)))
        return rv

    def hits(self):
        p = C.POINTER(KrHit)()
        n = C.c_uint64()
        check(self.lib.kr_batch_hits(self.h, C.byref(p), C.byref(n)))
        dt = np.dtype([("read", "<u4"), ("kpos", "<u4"), ("strand", "<u4"), ("lib", "<u4"), ("cmer_index", "<u8"),
                       ("hd", "<u4"), ("se", "<u4")])
        if n.value == 0:
            return np.zeros(0, dt)
        raw = C.string_at(p, n.value * C.sizeof(KrHit))
        return np.frombuffer(raw, dtype=dt).copy()

    def readtaps(self, nreads):
        p = u32p()
        check(self.lib.kr_batch_readtaps(self.h, C.byref(p)))
        return _np(p, 2 * nreads, np.uint32).reshape(-1, 2)

    def debug_move(self, which):
        """experiments: one group of the stream's device buffers at a new address (kr_debug_stream_move)"""
        check(self.lib.kr_debug_stream_move(self.h, which))

    def item_placement(self):
        """(allocations of the item list tried, how many replaced the one before, scan ns per read on the one kept)"""
        a, b, c = C.c_uint32(0), C.c_uint32(0), C.c_double(0)
        check(self.lib.kr_debug_item_placement(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"tried": a.value, "kept": b.value, "scan_ns_per_read": c.value}

    def debug_addrs(self):
        a = (C.c_uint64 * 8)()
        check(self.lib.kr_debug_stream_addrs(self.h, a))
        return dict(zip(("items", "counters", "cursors", "rd_off", "rd_it_off", "rd_filt", "rec_key", "dd"), [int(x) for x in a]))

    def timing(self):
        t = KrTiming()
        check(self.lib.kr_batch_timing(self.h, C.byref(t)))
        return t

    def format_seek(self, host_index, device_index, names, hdist_th=4):
        """`krepp seek` rows from the collected batch (kr_format_seek)"""
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        txt = C.c_void_p()
        ln = C.c_uint64()
        self.lib.kr_format_seek.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(KrResultView), C.c_uint32, C.POINTER(C.c_char_p),
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        check(self.lib.kr_format_seek(host_index.h, device_index.h, C.byref(self._rv), hdist_th, arr, C.byref(txt), C.byref(ln)))
        s = C.string_at(txt, ln.value).decode()
        self.lib.kr_free(txt)
        return s

    def text_enable(self, host_index, max_text_bytes, max_id_bytes):
        """kr_stream_text_enable: this stream formats its rows-only batches on the device (csrc/kr_dev_text.inc)"""
        check(self.lib.kr_stream_text_enable(self.h, host_index.h, int(max_text_bytes), int(max_id_bytes)))

    def submit_text(self, bases, offsets, names, flags=0, sep=1):
        """kr_batch_submit_text with the reads' ids packed back to back, `sep` NUL bytes behind each"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        enc = [n.encode() for n in names]
        blob = (b"\0" * sep).join(enc) + b"\0" * sep
        lens = np.fromiter((len(e) + sep for e in enc), dtype=np.uint32, count=len(enc))
        id_off = np.zeros(len(enc) + 1, dtype=np.uint32)
        np.cumsum(lens, out=id_off[1:])
        ids = np.frombuffer(blob, dtype=np.uint8)
        self._keep = (bases, offsets, ids, id_off)
        self._flags = flags | KR_ROWS_ONLY
        check(self.lib.kr_batch_submit_text(self.h, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1, flags, ids.ctypes.data, id_off.ctypes.data, sep))

    def collect_text(self):
        """kr_batch_collect_text: the batch's report rows as the bytes the device wrote"""
        txt = C.c_void_p()
        ln = C.c_uint64()
        check(self.lib.kr_batch_collect_text(self.h, C.byref(txt), C.byref(ln)))
        return C.string_at(txt, ln.value) if ln.value else b""

    def format_dist(self, host_index, names):
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        txt = C.c_void_p()
        ln = C.c_uint64()
        check(self.lib.kr_format_dist(host_index.h, C.byref(self._rv), arr, C.byref(txt), C.byref(ln)))
        s = C.string_at(txt, ln.value).decode()
        self.lib.kr_free(txt)
        return s

    def close(self):
        if self.h:
            self.lib.kr_stream_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def colour_classes(pse, node_kind):
    """Host pass of kr_index_upload over a colour table (no device): (cls[nsubsets], pse_dev[nsubsets, 2], lists)."""
    lib = load()
    pse = np.ascontiguousarray(pse, dtype=np.uint32).reshape(-1, 2)
    node_kind = np.ascontiguousarray(node_kind, dtype=np.uint8)
    n = len(pse)
    cls = np.zeros(n, np.uint32)
    out = np.zeros((n, 2), np.uint32)
    nl = C.c_uint64(0)
    rc = lib.kr_debug_colour_classes(pse.ctypes.data, n, node_kind.ctypes.data, len(node_kind) - 1, cls.ctypes.data, out.ctypes.data, None,
                                     C.byref(nl))
    lists = np.zeros(max(1, nl.value), np.uint32)
    if rc != 0:  # the first call sized the list buffer
        check(lib.kr_debug_colour_classes(pse.ctypes.data, n, node_kind.ctypes.data, len(node_kind) - 1, cls.ctypes.data, out.ctypes.data,
                                          lists.ctypes.data, C.byref(nl)))
    return cls, out, lists[: nl.value]


def read_fastx(path, min_bases=76800, stats=None, detach=0):
    """All records of a FASTA/FASTQ(.gz) file via the library's reader: (names, bases, offsets).
    detach = K > 0: every batch is taken over (kr_fastx_detach), read only when K further batches have been parsed, then handed
    back (kr_fastx_release) -- the way the CLI's workers hold batches in flight."""
    lib = load()
    lib.kr_fastx_parallel_chunks.restype = C.c_uint64
    lib.kr_fastx_parallel_chunks.argtypes = [C.c_void_p]
    h = C.c_void_p()
    check(lib.kr_fastx_open(os.fsencode(str(path)), C.byref(h)))
    names, chunks, lens = [], [], []

    def take(b):
        if b.nreads:
            offs = np.ctypeslib.as_array(b.offsets, shape=(b.nreads + 1,)).copy()
            chunks.append(np.ctypeslib.as_array(b.bases, shape=(int(offs[-1]),)).copy() if offs[-1] else np.zeros(0, np.uint8))
            lens.extend(np.diff(offs).tolist())
            names.extend(b.names[i].decode() for i in range(b.nreads))

    held = []
    try:
        while True:
            b = KrFastxBatch()
            check(lib.kr_fastx_next(h, min_bases, C.byref(b)))
            if detach:
                hp = C.c_void_p()
                check(lib.kr_fastx_detach(h, C.byref(hp)))
                held.append((b, hp))
                while len(held) > detach:
                    ob, ohp = held.pop(0)
                    take(ob)
                    lib.kr_fastx_release(h, ohp)
            else:
                take(b)
            if not b.more:
                break
        for ob, ohp in held:
            take(ob)
            lib.kr_fastx_release(h, ohp)
    finally:
        if stats is not None:
            stats["parallel_chunks"] = int(lib.kr_fastx_parallel_chunks(h))
            a, b_, c_ = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
            lib.kr_fastx_pgz_stats.restype = None
            lib.kr_fastx_pgz_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
            lib.kr_fastx_pgz_stats(h, C.byref(a), C.byref(b_), C.byref(c_))
            stats["gzip_chunks"] = {"parsed": int(a.value), "discarded": int(b_.value), "gaps": int(c_.value)}
        lib.kr_fastx_close(h)
    bases = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    offsets = np.zeros(len(lens) + 1, np.uint64)
    offsets[1:] = np.cumsum(np.asarray(lens, dtype=np.uint64))
    return names, bases, offsets


def place_counters():
    """(batches whose `place` back end ran on the device, batches sent whole to the host back end) of this process."""
    a, b = C.c_uint64(0), C.c_uint64(0)
    load().kr_place_counters(C.byref(a), C.byref(b), None)
    return int(a.value), int(b.value)


def place_text_counters():
    """(ranges of reads whose `place` rows were written on the device, ranges the host formatted although device text was on)"""
    a, b = C.c_uint64(0), C.c_uint64(0)
    load().kr_place_text_counters.restype = None
    load().kr_place_text_counters(C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


def place_heavy_reads():
    """Reads (upper bound, in chunks of 8) that kr_place_kernel's second launch did with its arrays in global scratch."""
    c = C.c_uint64(0)
    load().kr_place_counters(None, None, C.byref(c))
    return int(c.value)


def build_index(input_tsv, out_dir, nwk=None, k=29, w=35, h=13, m=4, r=1, frac=True, num_threads=1, seed=0, ppos=None,
                gpu_minimizers=False, device=0):
    """`krepp index` on the CPU (reference: src/krepp.cpp:131-303)."""
    lib = load()
    p = KrBuildParams(k=k, w=w, h=h, m=m, r=r, frac=int(frac), num_threads=num_threads, seed=seed, ppos=None,
                      gpu_minimizers=int(gpu_minimizers), device=device)
    keep = None
    if ppos is not None:
        keep = (C.c_uint8 * len(ppos))(*ppos)
        p.ppos = C.cast(keep, u8p)
    check(lib.kr_build_index(os.fsencode(str(input_tsv)), os.fsencode(str(nwk)) if nwk else None,
                             os.fsencode(str(out_dir)), C.byref(p)))


def build_sketch(input_path, out_path, k=26, w=None, h=None, m=4, r=1, frac=True, seed=0, ppos=None):
    """`krepp sketch` (reference: src/krepp.cpp:110-129; defaults k 26, w k+6, h k-16)."""
    lib = load()
    w = k + 6 if w is None else w
    h = k - 16 if h is None else h
    p = KrBuildParams(k=k, w=w, h=h, m=m, r=r, frac=int(frac), num_threads=1, seed=seed, ppos=None, gpu_minimizers=0, device=0)
    keep = None
    if ppos is not None:
        keep = (C.c_uint8 * len(ppos))(*ppos)
        p.ppos = C.cast(keep, u8p)
    lib.kr_build_sketch.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(KrBuildParams)]
    check(lib.kr_build_sketch(os.fsencode(str(input_path)), os.fsencode(str(out_path)), C.byref(p)))


def minimizers(bases, offsets, k, w, h, ppos, m=4, r=1, frac=True, device=None):
    """(keys, n1, n2) of one genome: kr_minimizers_cpu (device=None) or kr_minimizers_device."""
    lib = load()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    keep = (C.c_uint8 * len(ppos))(*ppos)
    p = KrBuildParams(k=k, w=w, h=h, m=m, r=r, frac=int(frac), num_threads=1, seed=0, ppos=C.cast(keep, u8p),
                      gpu_minimizers=0, device=0)
    res = KrMinimizerResult()
    if device is None:
        check(lib.kr_minimizers_cpu(C.byref(p), bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1, C.byref(res)))
    else:
        check(lib.kr_minimizers_device(int(device), C.byref(p), bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                                       C.byref(res)))
    keys = _np(res.keys, res.nkeys, np.uint64)
    out = (keys, res.n1, res.n2)
    lib.kr_minimizers_free(C.byref(res))
    return out


PLACEMENT_DT = np.dtype([("read", "<u4"), ("edge", "<u4"), ("lwr", "<f8"), ("d_llh", "<f8"), ("v_llh", "<f8"),
                         ("pendant", "<f8"), ("distal", "<f8")])


class Placer:
    """`krepp place` on one GPU: placement tree + device index (uploaded with the tree's node kinds) + stream."""

    def __init__(self, host_index, nwk_text=None, device=0, tabular=False, max_reads=1 << 16, max_bases=None, lineage_text=None,
                 **place_opts):
        """tabular: False/0 jplace, True/1 --tabular, 2 --summarize; lineage_text: -l instead of a Newick tree"""
        self.lib = load()
        self.hx = host_index
        self.tabular = tabular
        self.pt = C.c_void_p()
        if lineage_text is not None:
            check(self.lib.kr_place_tree_create_lineage(host_index.h, lineage_text.encode(), C.byref(self.pt)))
        else:
            check(self.lib.kr_place_tree_create(host_index.h, nwk_text.encode() if nwk_text is not None else None, C.byref(self.pt)))
        self.wcount = np.zeros(self.lib.kr_place_tree_nnodes(self.pt) + 1, np.float64)
        self.twcount = C.c_double(0.0)
        view = KrIndexView()
        C.memmove(C.byref(view), C.byref(host_index.view), C.sizeof(view))
        view.node_kind = self.lib.kr_place_tree_kinds(self.pt)
        self.dx = DeviceIndex.from_view(view, device, KR_VIEW_HOST, keep=host_index)
        # options of `place` (filter defaults to on: src/krepp.cpp:593-630)
        self.popts = default_params(no_filter=0, **place_opts)
        front = default_params(hdist_th=self.popts.hdist_th)  # multi, no_filter, no dist-max: every chosen leaf
        self.st = self.dx.stream(params=front, max_reads=max_reads, max_bases=max_bases, max_records=max_reads * 128)
        self.prev = C.c_int(0)

    def frame(self, which, invocation="", total=0):
        txt, ln = C.c_void_p(), C.c_uint64()
        check(self.lib.kr_place_frame(self.pt, which, int(self.tabular), invocation.encode(), total, C.byref(txt), C.byref(ln)))
        s = C.string_at(txt, ln.value).decode()
        self.lib.kr_free(txt)
        return s

    def place(self, bases, offsets, names, host=False, c_names=None, want_placements=True, keep_text=True):
        """One batch.  host=False: kr_place_stream (tree aggregation and likelihoods on the device);
        host=True: kr_batch_collect + kr_place_batch (aggregation on the host).  Same output.
        keep_text=False (timing): the library's text and placements are not copied into Python objects -- copying and decoding 70 MB
        of jplace text takes Python longer than the library takes to make it; returns (bytes of text, number of placements)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.st.submit(bases, offsets, KR_TAP_ACCS)
        arr = c_names if c_names is not None else (C.c_char_p * len(names))(*[n.encode() for n in names])
        txt, ln, pls, npl = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64()
        if host:
            rv = KrResultView()
            check(self.lib.kr_batch_collect(self.st.h, C.byref(rv)))
            check(self.lib.kr_place_batch(self.hx.h, self.dx.h, self.pt, C.byref(rv), offsets.ctypes.data, arr, C.byref(self.popts),
                                          int(self.tabular), C.byref(self.prev), C.byref(txt), C.byref(ln),
                                          C.byref(pls) if want_placements else None, C.byref(npl) if want_placements else None))
        else:
            check(self.lib.kr_place_stream(self.hx.h, self.dx.h, self.pt, self.st.h, len(offsets) - 1, offsets.ctypes.data, arr,
                                           C.byref(self.popts), int(self.tabular), C.byref(self.prev), C.byref(txt), C.byref(ln),
                                           C.byref(pls) if want_placements else None, C.byref(npl) if want_placements else None))
        if not keep_text:
            if int(self.tabular) == 2:
                check(self.lib.kr_place_summary_add(self.pt, pls, npl.value, self.wcount.ctypes.data, C.byref(self.twcount)))
            self.lib.kr_free(txt)
            self.lib.kr_free(pls)
            return int(ln.value), int(npl.value)
        text = C.string_at(txt, ln.value).decode()
        pl = (np.frombuffer(C.string_at(pls, npl.value * PLACEMENT_DT.itemsize), dtype=PLACEMENT_DT).copy()
              if npl.value else np.zeros(0, PLACEMENT_DT))
        if int(self.tabular) == 2:
            check(self.lib.kr_place_summary_add(self.pt, pls, npl.value, self.wcount.ctypes.data, C.byref(self.twcount)))
        self.lib.kr_free(txt)
        self.lib.kr_free(pls)
        return text, pl

    def summary(self):
        txt, ln = C.c_void_p(), C.c_uint64()
        check(self.lib.kr_place_summary_text(self.pt, self.wcount.ctypes.data, self.twcount, C.byref(txt), C.byref(ln)))
        s = C.string_at(txt, ln.value).decode()
        self.lib.kr_free(txt)
        return s

    def close(self):
        if self.pt:
            self.st.close()
            self.dx.close()
            self.lib.kr_place_tree_free(self.pt)
            self.pt = C.c_void_p()
