r"""`bblean.similarity` function surface on the MI355X kernels.

Same names, argument meaning and error behaviour as the reference module
(bblean/similarity.py:12-35) and the pybind11 functions it re-exports
(bblean/csrc/similarity.cpp:473-521).  Every function that the reference backs with
C++ runs here as a HIP kernel through the C ABI (include/bbhip.h); the thin composites
the reference writes in Python on top of them (`jt_isim_radius*`, `jt_sim_matrix_packed`,
`estimate_jt_std`, ...) are the same thin composites here.  There is no NumPy fallback.

Inputs may be NumPy arrays (staged to HBM for the call) or CUDA/HIP ``torch`` tensors
(used in place; outputs are then device tensors as well where that makes sense).
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np
from numpy.typing import NDArray

from bblean_amd import _lib
from bblean_amd.fingerprints import pack_fingerprints, unpack_fingerprints

__all__ = [
    "jt_isim_from_sum",
    "jt_isim",
    "jt_sim_packed",
    "jt_most_dissimilar_packed",
    "jt_isim_radius_from_sum",
    "jt_isim_radius_compl_from_sum",
    "jt_isim_diameter_from_sum",
    "jt_isim_radius",
    "jt_isim_radius_compl",
    "jt_isim_diameter",
    "centroid_from_sum",
    "centroid",
    "jt_isim_medoid",
    "jt_compl_isim",
    "jt_stratified_sampling",
    "jt_sim_matrix_packed",
    "jt_best_match_packed",
]


def _is_dev(x: object) -> bool:
    return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)


def _u8_2d(a: object, what: str = "Input array") -> tuple[object, int, int, int]:
    r"""(keepalive, n, nbytes, row_stride) of a 2-D uint8 array / device tensor."""
    if _is_dev(a):
        if a.dim() != 2:  # type: ignore[attr-defined]
            raise RuntimeError(f"{what} must be 2-dimensional")
        assert a.stride(1) == 1  # type: ignore[attr-defined]
        return a, int(a.shape[0]), int(a.shape[1]), int(a.stride(0))  # type: ignore[attr-defined]
    arr = np.asarray(a)
    if arr.ndim != 2:
        raise RuntimeError(f"{what} must be 2-dimensional")
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    return arr, arr.shape[0], arr.shape[1], arr.shape[1]


# ------------------------------------------------------------------ popcount -------
def _popcount_2d(a: object) -> NDArray[np.uint32]:
    r"""Row popcounts (similarity.cpp:99-141)."""
    lib = _lib.load()
    arr, n, nb, stride = _u8_2d(a)
    out = np.empty(n, dtype=np.uint32)
    _lib.check(lib.bbh_popcount_rows(_lib.ptr(arr), n, nb, stride, out.ctypes.data, None))
    return out


def _popcount_1d(a: NDArray[np.uint8]) -> int:
    r"""Popcount of one packed row (similarity.cpp:63-94)."""
    arr = np.asarray(a)
    if arr.ndim != 1:
        raise RuntimeError("Input array must be 1-dimensional")
    return int(_popcount_2d(arr.reshape(1, -1))[0])


# ------------------------------------------------------------ arr-vec Tanimoto -----
def _jt_sim_arr_vec_packed(arr: object, vec: object) -> NDArray[np.float64]:
    r"""Tanimoto of every packed row of ``arr`` against the packed row ``vec``
    (similarity.cpp:374-377).  float64, exact integer popcounts, one IEEE division."""
    lib = _lib.load()
    a, n, nb, stride = _u8_2d(arr, "arr")
    if _is_dev(vec):
        v = vec
        vdim, vlen = v.dim(), int(v.shape[-1])  # type: ignore[attr-defined]
    else:
        v = np.ascontiguousarray(vec, dtype=np.uint8)
        vdim, vlen = v.ndim, v.shape[-1] if v.ndim else 0
    if vdim != 1:
        raise RuntimeError("arr must be 2D, vec must be 1D")
    if vlen != nb:
        raise RuntimeError("Shapes should be (N, F) for arr and (F,) for vec")
    if _is_dev(a):
        import torch

        out_t = torch.empty(n, dtype=torch.float64, device=a.device)  # type: ignore[attr-defined]
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(
            lib.bbh_jt_arr_vec(_lib.ptr(a), n, nb, stride, _lib.ptr(v), None, _lib.ptr(out_t), None, None, st)
        )
        return out_t  # type: ignore[return-value]
    out = np.empty(n, dtype=np.float64)
    _lib.check(
        lib.bbh_jt_arr_vec(_lib.ptr(a), n, nb, stride, _lib.ptr(v), None, out.ctypes.data, None, None, None)
    )
    return out


def _jt_counts_arr_vec_packed(arr: object, vec: object) -> tuple[NDArray[np.uint32], NDArray[np.uint32]]:
    r"""Exact (intersection, union) popcounts of the same kernel (debug / parity)."""
    lib = _lib.load()
    a, n, nb, stride = _u8_2d(arr, "arr")
    v = np.ascontiguousarray(vec, dtype=np.uint8)
    inter = np.empty(n, dtype=np.uint32)
    union = np.empty(n, dtype=np.uint32)
    _lib.check(
        lib.bbh_jt_arr_vec(_lib.ptr(a), n, nb, stride, v.ctypes.data, None, None,
                           inter.ctypes.data, union.ctypes.data, None)
    )
    return inter, union


def jt_sim_packed(x: NDArray[np.uint8], y: NDArray[np.uint8]) -> NDArray[np.float64]:
    r"""General wrapper (similarity.py:218-236): two vectors, or a vector and an array."""
    xd = x.dim() if _is_dev(x) else np.ndim(x)  # type: ignore[attr-defined]
    yd = y.dim() if _is_dev(y) else np.ndim(y)  # type: ignore[attr-defined]
    if xd == 1 and yd == 1:
        return _jt_sim_arr_vec_packed(x.reshape(1, -1), y)[0]
    if xd == 2:
        return _jt_sim_arr_vec_packed(x, y)
    if yd == 2:
        return _jt_sim_arr_vec_packed(y, x)
    raise ValueError("Expected either two 1D vectors, or one 1D vector and one 2D array")


def jt_best_match_packed(
    queries: NDArray[np.uint8], centroids: NDArray[np.uint8], return_sims: bool = False
) -> tuple[NDArray[np.int32], NDArray[np.uint32], NDArray[np.uint32], NDArray[np.float64] | None]:
    r"""Batched descent step: first-argmax Tanimoto of each query against all centroid
    rows (`_jt_sim_arr_vec_packed` + `np.argmax`, bitbirch.py:317-320, for a whole
    batch of incoming fingerprints at once)."""
    lib = _lib.load()
    q, nq, nb, _ = _u8_2d(queries, "queries")
    c, nc, nb2, _ = _u8_2d(centroids, "centroids")
    if nb != nb2:
        raise RuntimeError("queries and centroids must have the same packed width")
    idx = np.empty(nq, dtype=np.int32)
    inter = np.empty(nq, dtype=np.uint32)
    union = np.empty(nq, dtype=np.uint32)
    sims = np.empty((nq, nc), dtype=np.float64) if return_sims else None
    _lib.check(
        lib.bbh_jt_best_match(_lib.ptr(q), nq, _lib.ptr(c), nc, nb, idx.ctypes.data,
                              inter.ctypes.data, union.ctypes.data,
                              sims.ctypes.data if sims is not None else None, None)
    )
    return idx, inter, union, sims


def jt_sim_matrix_packed(arr: NDArray[np.uint8]) -> NDArray[np.float64]:
    r"""All-pairs Tanimoto matrix (similarity.py:239-247): one batched kernel instead of
    N sequential arr-vec calls; the diagonal is 1 as in the reference."""
    a = np.ascontiguousarray(arr, dtype=np.uint8)
    _, _, _, sims = jt_best_match_packed(a, a, return_sims=True)
    assert sims is not None
    np.fill_diagonal(sims, 1.0)
    return sims


# ---------------------------------------------------------- pack / unpack ---------
def _unpack_fingerprints_hip(a: NDArray[np.uint8], n_features: int | None = None) -> NDArray[np.uint8]:
    r"""`_cpp_similarity.unpack_fingerprints` (similarity.cpp:204-214)."""
    lib = _lib.load()
    arr = np.ascontiguousarray(a, dtype=np.uint8)
    if arr.ndim not in (1, 2):
        raise RuntimeError("Input array must be 1- or 2-dimensional")
    two = arr.reshape(1, -1) if arr.ndim == 1 else arr
    n, nb = two.shape
    nf = nb * 8 if n_features is None else int(n_features)
    if nf % 8 != 0:
        raise RuntimeError("Only n_features divisible by 8 is supported")
    out = np.empty((n, nf), dtype=np.uint8)
    _lib.check(lib.bbh_unpack(two.ctypes.data, n, nb, nf, out.ctypes.data, None))
    return out[0] if arr.ndim == 1 else out


# --------------------------------------------------- linear sums and centroids ----
def _add_rows(arr: NDArray[np.uint8]) -> NDArray[np.uint64]:
    r"""Column sums (similarity.cpp:381-400)."""
    lib = _lib.load()
    a = np.asarray(arr)
    if a.ndim != 2:
        raise RuntimeError("Input array must be 2-dimensional")
    a = np.ascontiguousarray(a, dtype=np.uint8)
    out = np.empty(a.shape[1], dtype=np.uint64)
    _lib.check(lib.bbh_add_rows(a.ctypes.data, a.shape[0], a.shape[1], 0, a.shape[1], out.ctypes.data, None))
    return out


def _as_uint_ls(linear_sum: NDArray[np.integer]) -> NDArray[np.integer]:
    ls = np.ascontiguousarray(linear_sum)
    if ls.dtype.kind == "u" and ls.dtype.itemsize in (1, 2, 4, 8):
        return ls
    return ls.astype(np.uint64)  # pybind11 forcecast (similarity.cpp:48-50)


def centroid_from_sum(
    linear_sum: NDArray[np.integer], n_samples: int, *, pack: bool = True
) -> NDArray[np.uint8]:
    r"""Majority-vote centroid from a linear sum (_py_similarity.py:12-42,
    similarity.cpp:216-271): ``n<=1`` -> cast, else bit = ``ls >= n*0.5``."""
    lib = _lib.load()
    ls = _as_uint_ls(linear_sum)
    if ls.ndim != 1:
        raise RuntimeError("linear_sum must be 1-dimensional")
    nf = ls.shape[0]
    out = np.empty((nf + 7) // 8 if pack else nf, dtype=np.uint8)
    _lib.check(
        lib.bbh_centroid_from_sum(ls.ctypes.data, ls.dtype.itemsize, nf, int(n_samples), int(pack),
                                  out.ctypes.data, None)
    )
    return out


def centroid(
    fps: NDArray[np.uint8],
    input_is_packed: bool = True,
    n_features: int | None = None,
    *,
    pack: bool = True,
) -> NDArray[np.uint8]:
    r"""Majority-vote centroid of a set of fingerprints (_py_similarity.py:45-62)."""
    lib = _lib.load()
    a = np.ascontiguousarray(fps, dtype=np.uint8)
    nf = (a.shape[1] * 8 if n_features is None else n_features) if input_is_packed else a.shape[1]
    ls = np.empty(nf, dtype=np.uint64)
    _lib.check(lib.bbh_add_rows(a.ctypes.data, a.shape[0], a.shape[1], int(input_is_packed), nf, ls.ctypes.data, None))
    return centroid_from_sum(ls, len(a), pack=pack)


# ------------------------------------------------------------------- iSIM ---------
def jt_isim_from_sum(linear_sum: NDArray[np.integer], n_objects: int) -> float:
    r"""iSIM Tanimoto from a column sum (similarity.cpp:273-301): exact u64 moments,
    then ``a=(q-s)/2.0; a/((a+n*s)-q)`` in IEEE f64.  ``n_objects < 2`` warns and
    returns NaN (similarity.cpp:275-279)."""
    lib = _lib.load()
    ls = _as_uint_ls(linear_sum)
    if ls.ndim != 1:
        raise RuntimeError("linear_sum must be a 1D array")
    out = C.c_double(0.0)
    warn = C.c_int(0)
    _lib.check(
        lib.bbh_isim_from_sum(ls.ctypes.data, ls.dtype.itemsize, ls.shape[0], int(n_objects),
                              C.byref(out), C.byref(warn), None)
    )
    if warn.value:
        warnings.warn(
            f"Invalid n_objects = {n_objects} in isim. Expected n_objects >= 2",
            RuntimeWarning,
            stacklevel=2,
        )
    return float(out.value)


def _isim_rows(arr: NDArray[np.integer], packed: bool, n_features: int | None) -> float:
    lib = _lib.load()
    a = np.asarray(arr)
    if a.dtype != np.uint8:
        # the reference sums wider dtypes with NumPy first (similarity.py:67-90)
        u = unpack_fingerprints(a.astype(np.uint8), n_features) if packed else a
        return jt_isim_from_sum(np.sum(u, axis=0, dtype=np.uint64), len(a))
    a = np.ascontiguousarray(a)
    nf = (a.shape[1] * 8 if n_features is None else n_features) if packed else a.shape[1]
    out = C.c_double(0.0)
    warn = C.c_int(0)
    _lib.check(lib.bbh_isim_rows(a.ctypes.data, a.shape[0], a.shape[1], int(packed), nf,
                                 C.byref(out), C.byref(warn), None))
    if warn.value:
        warnings.warn(
            f"Invalid n_objects = {len(a)} in isim. Expected n_objects >= 2",
            RuntimeWarning,
            stacklevel=3,
        )
    return float(out.value)


def jt_isim_unpacked(arr: NDArray[np.integer]) -> float:
    return _isim_rows(arr, False, None)


def jt_isim_packed(arr: NDArray[np.integer], n_features: int | None = None) -> float:
    return _isim_rows(arr, True, n_features)


def jt_isim(fps: NDArray[np.integer], input_is_packed: bool = True, n_features: int | None = None) -> float:
    r"""Average Tanimoto of a set via iSIM (similarity.py:106-140)."""
    if input_is_packed:
        return jt_isim_packed(fps, n_features)
    return jt_isim_unpacked(fps)


def _sum_rows_u64(arr: NDArray[np.integer], input_is_packed: bool, n_features: int | None) -> NDArray[np.uint64]:
    lib = _lib.load()
    a = np.ascontiguousarray(arr, dtype=np.uint8)
    nf = (a.shape[1] * 8 if n_features is None else n_features) if input_is_packed else a.shape[1]
    ls = np.empty(nf, dtype=np.uint64)
    _lib.check(lib.bbh_add_rows(a.ctypes.data, a.shape[0], a.shape[1], int(input_is_packed), nf, ls.ctypes.data, None))
    return ls


def jt_isim_radius_compl_from_sum(ls: NDArray[np.integer], n: int) -> float:
    r"""1 - radius (similarity.py:192-202)."""
    cen = centroid_from_sum(ls, n, pack=False)
    ls_1 = np.add(ls, cen, dtype=np.uint64)
    jt = jt_isim_from_sum(ls, n)
    jt_1 = jt_isim_from_sum(ls_1, n + 1)
    return (jt_1 * (n + 1) - jt * (n - 1)) / 2


def jt_isim_radius_from_sum(ls: NDArray[np.integer], n: int) -> float:
    return 1 - jt_isim_radius_compl_from_sum(ls, n)


def jt_isim_diameter_from_sum(ls: NDArray[np.integer], n: int) -> float:
    return 1 - jt_isim_from_sum(ls, n)


def jt_isim_diameter(arr: NDArray[np.integer], input_is_packed: bool = True, n_features: int | None = None) -> float:
    return jt_isim_diameter_from_sum(_sum_rows_u64(arr, input_is_packed, n_features), len(arr))


def jt_isim_radius(arr: NDArray[np.integer], input_is_packed: bool = True, n_features: int | None = None) -> float:
    return jt_isim_radius_from_sum(_sum_rows_u64(arr, input_is_packed, n_features), len(arr))


def jt_isim_radius_compl(arr: NDArray[np.integer], input_is_packed: bool = True, n_features: int | None = None) -> float:
    return jt_isim_radius_compl_from_sum(_sum_rows_u64(arr, input_is_packed, n_features), len(arr))


# ------------------------------------------------------------ split primitive -----
def jt_most_dissimilar_packed(
    Y: NDArray[np.uint8], n_features: int | None = None
) -> tuple[int, int, NDArray[np.float64], NDArray[np.float64]]:
    r"""The node-split primitive (similarity.cpp:413-471): majority centroid of Y, the
    row least similar to it (fp_1), the row least similar to fp_1 (fp_2), and the
    similarities of all rows to both.  First index wins ties."""
    lib = _lib.load()
    y = np.asarray(Y)
    if y.ndim != 2:
        raise RuntimeError("Input array must be 2-dimensional")
    y = np.ascontiguousarray(y, dtype=np.uint8)
    n, nb = y.shape
    nf = nb * 8 if n_features is None else int(n_features)
    i1, i2 = C.c_int64(0), C.c_int64(0)
    s1 = np.empty(n, dtype=np.float64)
    s2 = np.empty(n, dtype=np.float64)
    _lib.check(lib.bbh_most_dissimilar(y.ctypes.data, n, nb, nf, C.byref(i1), C.byref(i2),
                                       s1.ctypes.data, s2.ctypes.data, None))
    return int(i1.value), int(i2.value), s1, s2


# ---------------------------------------------------- analysis-side composites ----
def jt_compl_isim(fps: NDArray[np.uint8], input_is_packed: bool = True, n_features: int | None = None) -> NDArray[np.float64]:
    r"""Complementary iSIM of every row (_py_similarity.py:65-83)."""
    if input_is_packed:
        fps = unpack_fingerprints(fps, n_features)
    n_objects = len(fps) - 1
    if n_objects < 2:
        warnings.warn("Invalid fps. len(fps) must be >= 3", RuntimeWarning, stacklevel=2)
        return np.full(len(fps), fill_value=np.nan, dtype=np.float64)
    total = _sum_rows_u64(fps, False, None)
    return np.array([jt_isim_from_sum(total - fp, n_objects) for fp in fps], dtype=np.float64)


def jt_isim_medoid(fps: NDArray[np.uint8], input_is_packed: bool = True, n_features: int | None = None, pack: bool = True) -> tuple[int, NDArray[np.uint8]]:
    r"""(_py_similarity.py:91-117)"""
    if not fps.size:
        raise ValueError("Size of fingerprints set must be > 0")
    if input_is_packed:
        fps = unpack_fingerprints(fps, n_features)
    idx = 0 if len(fps) < 3 else int(np.argmin(jt_compl_isim(fps, input_is_packed=False)))
    m = fps[idx]
    return (idx, pack_fingerprints(m)) if pack else (idx, m)


def jt_stratified_sampling(fps: NDArray[np.uint8], n_samples: int, input_is_packed: bool = True, n_features: int | None = None) -> NDArray[np.int64]:
    r"""(similarity.py:276-304)"""
    if n_samples == 0:
        return np.array([], dtype=np.int64)
    if n_samples > len(fps):
        raise ValueError("n_samples must be <= len(fps)")
    order = np.argsort(jt_compl_isim(fps, input_is_packed, n_features))
    return np.array([s[0] for s in np.array_split(order, n_samples)])


def estimate_jt_std(fps: NDArray[np.uint8], n_samples: int | None = None, input_is_packed: bool = True, n_features: int | None = None) -> float:
    r"""(similarity.py:250-273)"""
    num = len(fps)
    if n_samples is None:
        n_samples = max(num // 1000, 50)
    sample = fps[jt_stratified_sampling(fps, n_samples, input_is_packed, n_features)]
    if not input_is_packed:
        sample = pack_fingerprints(sample)
    m = jt_sim_matrix_packed(sample)
    iu = np.triu_indices(len(sample), k=1)
    return float(np.std(m[iu]))
