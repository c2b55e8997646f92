r"""ctypes binding of libbbhip.so (the C ABI declared in include/bbhip.h).

There is NO fallback: if the shared library is missing or no gfx950 device is visible
the call raises.  (The reference silently falls back to NumPy when its extension is
absent, bblean/similarity.py:47-103; a GPU engine that silently ran on the CPU would be
a lie, so this one refuses instead.)
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_LIB: C.CDLL | None = None
_HERE = Path(__file__).resolve().parent

BBH_OK = 0
BBH_ERR_INVALID = 1
BBH_ERR_HIP = 2
BBH_ERR_NO_DEVICE = 3
BBH_ERR_CAPACITY = 4
BBH_ERR_STATE = 5

_vp = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64
_f64 = C.c_double
_int = C.c_int

_PROTOTYPES = {
    "bbh_last_error": (C.c_char_p, []),
    "bbh_device_count": (_int, []),
    "bbh_device_info": (_int, [_int, C.c_char_p, C.c_size_t]),
    "bbh_trim_cache": (_int, []),
    "bbh_set_memory_pressure_callback": (_int, [_vp]),
    "bbh_popcount_rows": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "bbh_jt_arr_vec": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bbh_jt_best_match": (_int, [_vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "bbh_unpack": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "bbh_pack": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "bbh_add_rows": (_int, [_vp, _i64, _i64, _int, _i64, _vp, _vp]),
    "bbh_centroid_from_sum": (_int, [_vp, _i32, _i64, _i64, _int, _vp, _vp]),
    "bbh_isim_from_sum": (_int, [_vp, _i32, _i64, _i64, C.POINTER(_f64), C.POINTER(_int), _vp]),
    "bbh_isim_rows": (_int, [_vp, _i64, _i64, _int, _i64, C.POINTER(_f64), C.POINTER(_int), _vp]),
    "bbh_isim_pair_min_gap": (_int, [_vp, _vp, _i64, _i64, C.POINTER(_f64), _vp]),
    "bbh_most_dissimilar": (
        _int,
        [_vp, _i64, _i64, _i64, C.POINTER(_i64), C.POINTER(_i64), _vp, _vp, _vp],
    ),
    "bbh_tree_create": (_int, [C.POINTER(_vp), _i32, _f64, _i32, _f64, _vp, _i64, _i32, _i32]),
    "bbh_tree_destroy": (_int, [_vp]),
    "bbh_tree_set_merge": (_int, [_vp, _i32, _f64, _vp, _i64, _f64, _i32]),
    "bbh_tree_reset": (_int, [_vp]),
    "bbh_tree_fit_packed": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "bbh_tree_fit_buffers": (_int, [_vp, _vp, _i32, _i64, _vp, _vp]),
    "bbh_trees_fit_packed": (_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "bbh_trees_fit_buffers": (_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "bbh_tree_leaf_count": (_int, [_vp, C.POINTER(_i64)]),
    "bbh_tree_export_leaves": (_int, [_vp, _vp, _vp, _vp, _vp, _i32]),
    "bbh_tree_gather_buffers": (_int, [_vp, _vp, _i64, _i32, _vp]),
    "bbh_tree_gather_centroids": (_int, [_vp, _vp, _i64, _vp]),
    "bbh_tree_stats": (_int, [_vp, _vp]),
    "bbh_tree_kernel_counts": (_int, [_vp, _vp]),
    "bbh_tree_sys_counts": (_int, [_vp, _vp]),
    "bbh_tree_memory": (_int, [_vp, _vp]),
    "bbh_tree_compact": (_int, [_vp, _i32]),
    "bbh_profile_enable": (_int, [_int]),
    "bbh_profile_reset": (_int, []),
    "bbh_profile_get": (_int, [C.c_char_p, C.POINTER(_i64), C.POINTER(_f64)]),
    "bbh_profile_units": (_int, [C.c_char_p, C.POINTER(_i64)]),
    "bbh_profile_longest": (_int, [C.c_char_p, C.POINTER(_f64), C.POINTER(_i64)]),
}

EXPORTED_SYMBOLS = tuple(_PROTOTYPES)


def library_path() -> Path:
    env = os.environ.get("BBHIP_LIBRARY")
    return Path(env) if env else _HERE / "libbbhip.so"


def load() -> C.CDLL:
    r"""Load libbbhip.so once and attach prototypes.  Raises if it is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not path.is_file():
        raise ImportError(
            f"{path} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
            "bblean_amd/csrc`). bblean_amd has no CPU fallback."
        )
    lib = C.CDLL(str(path))
    for name, (res, args) in _PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    _register_pressure_callback(lib)
    return lib


_PRESSURE_CB = None  # (kept alive: the library holds the raw function pointer)


def _register_pressure_callback(lib: C.CDLL) -> None:
    r"""When a device allocation of the library fails it asks the host side to let go of what it caches on the device:
    torch's caching allocator keeps freed tensors (round tables of gigabytes) that `hipMalloc` then cannot have."""
    global _PRESSURE_CB

    def _release() -> None:
        try:
            import sys

            torch = sys.modules.get("torch")
            if torch is not None and torch.cuda.is_initialized():
                torch.cuda.empty_cache()
        except Exception:
            pass

    _PRESSURE_CB = C.CFUNCTYPE(None)(_release)
    lib.bbh_set_memory_pressure_callback(C.cast(_PRESSURE_CB, C.c_void_p))


class BBHipError(RuntimeError):
    pass


def check(code: int) -> None:
    r"""Map a C status code to the exception type the reference raises."""
    if code == BBH_OK:
        return
    lib = load()
    msg = (lib.bbh_last_error() or b"").decode("utf-8", "replace")
    if code == BBH_ERR_INVALID:
        raise RuntimeError(msg)  # pybind11 turns std::runtime_error into RuntimeError
    if code == BBH_ERR_STATE:
        raise ValueError(msg)
    if code == BBH_ERR_NO_DEVICE:
        raise BBHipError(f"no usable gfx950 device: {msg}")
    if code == BBH_ERR_CAPACITY:
        raise MemoryError(msg)
    raise BBHipError(msg)


def ptr(a: object) -> int | None:
    r"""Address of a numpy array / torch tensor / raw int for the C ABI."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if isinstance(a, int):
        return a
    dp = getattr(a, "data_ptr", None)
    if dp is not None:
        return int(dp())
    raise TypeError(f"cannot take the address of {type(a)}")
