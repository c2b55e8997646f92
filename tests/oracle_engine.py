r"""CPU-oracle engine for tests: same five operations as bblean_amd._engine.HipEngine,
backed by oracle/libbboracle.so.  TEST INFRASTRUCTURE - never imported by the product."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
_LIB = None
_W = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}


def oracle_lib() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    so = REPO / "oracle" / "libbboracle.so"
    src = REPO / "oracle" / "bb_oracle.c"
    if not so.is_file() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(REPO / "oracle"), "libbboracle.so"], check=True,
                       capture_output=True)
    lib = C.CDLL(str(so))
    vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    protos = {
        "bbo_popcount_rows": (None, [vp, i64, i64, vp]),
        "bbo_jt_arr_vec": (None, [vp, i64, i64, vp, vp, vp, vp, vp]),
        "bbo_unpack": (None, [vp, i64, i64, i64, vp]),
        "bbo_pack": (None, [vp, i64, i64, vp]),
        "bbo_centroid_from_sum": (None, [vp, i64, i64, C.c_int, vp]),
        "bbo_isim_from_sum": (f64, [vp, i64, i64]),
        "bbo_add_rows": (None, [vp, i64, i64, vp]),
        "bbo_most_dissimilar": (None, [vp, i64, i64, i64, C.POINTER(i64), C.POINTER(i64), vp, vp]),
        "bbo_isim_radius_compl_from_sum": (f64, [vp, i64, i64]),
        "bbo_merge_accept": (C.c_int, [C.c_int, f64, f64, vp, i64, vp, i64, vp, i64, i64, i64]),
        "bbo_tree_create": (vp, [i32, f64, i32, f64, vp, i64, i32]),
        "bbo_tree_destroy": (None, [vp]),
        "bbo_tree_set_merge": (None, [vp, i32, f64, vp, i64, f64, i32]),
        "bbo_tree_reset": (None, [vp]),
        "bbo_tree_fit_packed": (C.c_int, [vp, vp, i64, vp]),
        "bbo_tree_fit_buffers": (C.c_int, [vp, vp, i32, i64, vp]),
        "bbo_tree_leaf_count": (i64, [vp]),
        "bbo_tree_export_leaves": (None, [vp, vp, vp, vp, vp]),
        "bbo_tree_gather_buffers": (C.c_int, [vp, vp, i64, i32, vp]),
        "bbo_tree_stats": (None, [vp, vp]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


class OracleEngine:
    def __init__(self, branching_factor, threshold, criterion, tolerance, tol_table, n_features, device=0):
        self.lib = oracle_lib()
        self.n_features = int(n_features)
        self.nbytes = self.n_features // 8
        tab = np.ascontiguousarray(tol_table, dtype=np.float64)
        self._h = self.lib.bbo_tree_create(int(branching_factor), float(threshold), int(criterion),
                                           float(tolerance), tab.ctypes.data if tab.size else None,
                                           tab.size, self.n_features)
        if not self._h:
            raise RuntimeError("Only n_features divisible by 8 is supported")

    def set_merge(self, criterion, tolerance, tol_table, threshold, branching_factor):
        tab = np.ascontiguousarray(tol_table, dtype=np.float64)
        self.lib.bbo_tree_set_merge(self._h, int(criterion), float(tolerance),
                                    tab.ctypes.data if tab.size else None, tab.size,
                                    float(threshold), int(branching_factor))

    def reset(self):
        self.lib.bbo_tree_reset(self._h)

    def fit_packed(self, rows, stream=None):
        if hasattr(rows, "cpu"):
            rows = rows.cpu().numpy()
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        out = np.empty(rows.shape[0], dtype=np.uint32)
        self.lib.bbo_tree_fit_packed(self._h, rows.ctypes.data, rows.shape[0], out.ctypes.data)
        return out

    def fit_buffers(self, bufs, stream=None):
        bufs = np.ascontiguousarray(bufs)
        if bufs.dtype.kind != "u":
            bufs = bufs.astype(np.uint64)
        out = np.empty(bufs.shape[0], dtype=np.uint32)
        rc = self.lib.bbo_tree_fit_buffers(self._h, bufs.ctypes.data, bufs.dtype.itemsize,
                                           bufs.shape[0], out.ctypes.data)
        assert rc == 0
        return out

    def leaf_count(self):
        return int(self.lib.bbo_tree_leaf_count(self._h))

    def export_leaves(self, ls_width=None):
        k = self.leaf_count()
        ids = np.empty(k, dtype=np.uint32)
        ns = np.empty(k, dtype=np.uint64)
        cents = np.empty((k, self.nbytes), dtype=np.uint8)
        ls32 = np.empty((k, self.n_features), dtype=np.uint32) if ls_width else None
        self.lib.bbo_tree_export_leaves(self._h, ids.ctypes.data, ns.ctypes.data, cents.ctypes.data,
                                        ls32.ctypes.data if ls32 is not None else None)
        ls = ls32.astype(_W[ls_width]) if ls32 is not None else None
        return ids, ns, cents, ls

    def gather_buffers(self, positions, width):
        pos = np.ascontiguousarray(positions, dtype=np.int64)
        out = np.empty((pos.size, self.n_features + 1), dtype=_W[width])
        rc = self.lib.bbo_tree_gather_buffers(self._h, pos.ctypes.data, pos.size, int(width), out.ctypes.data)
        assert rc == 0, rc
        return out

    def stats(self):
        out = np.zeros(8, dtype=np.uint64)
        self.lib.bbo_tree_stats(self._h, out.ctypes.data)
        return out

    def close(self):
        if self._h:
            self.lib.bbo_tree_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
