r"""Device tree engine handle: the Python face of the `bbh_tree_*` C ABI.

`BitBirch` (bitbirch.py) owns the host-side bookkeeping (molecule-index lists, dtype
groups, refine orchestration) and talks to the HBM-resident tree only through the five
operations below.  Tests exercise the same host logic against the CPU oracle by
injecting an object with the same methods (tests/oracle_engine.py); the product never
does that - `HipEngine` is the only engine the package constructs.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
from numpy.typing import NDArray

from bblean_amd import _lib

_WIDTH_DTYPES = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}


def _is_device_tensor(x: object) -> bool:
    return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)


def _launch_stream(x: object) -> int | None:
    r"""HIP stream the tree kernel must be ordered after for a device-resident input: torch's
    current stream (the one the tensor's producer ran on, by torch's own convention)."""
    if not _is_device_tensor(x):
        return None
    import torch

    return int(torch.cuda.current_stream(x.device).cuda_stream)  # type: ignore[attr-defined]


def _one_stream(have: "int | None", new: "int | None") -> "int | None":
    r"""The many-tree entry points make ONE launch on ONE stream: device inputs produced on different streams (or
    devices) would only be ordered behind one of their producers."""
    if have is not None and new is not None and have != new:
        raise RuntimeError("device inputs of one multi-tree launch must share one device and stream")
    return new if new is not None else have


class DevTable:
    r"""A BitFeature buffer table ``[k, n_features + 1]`` of `width`-byte unsigned integers that lives in
    HBM (multiround's round-* tables, reference multiround.py:132-143, without the trip through host
    memory).  `raw` is a 2-D ``torch.uint8`` tensor of shape (k, (n_features + 1) * width)."""

    __slots__ = ("raw", "width")

    def __init__(self, raw: object, width: int) -> None:
        self.raw = raw
        self.width = int(width)

    @property
    def shape(self) -> tuple[int, int]:
        return int(self.raw.shape[0]), int(self.raw.shape[1]) // self.width  # type: ignore[attr-defined]

    @property
    def dtype(self) -> np.dtype:
        return np.dtype(_WIDTH_DTYPES[self.width])

    @property
    def nbytes(self) -> int:
        return int(self.raw.numel())  # type: ignore[attr-defined]

    def __len__(self) -> int:
        return self.shape[0]

    def n_samples(self) -> NDArray[np.integer]:
        r"""The last column (n_samples of every BitFeature) on the host."""
        k, cols = self.shape
        col = self.raw.view(k, cols, self.width)[:, -1, :].contiguous().cpu().numpy()  # type: ignore[attr-defined]
        return col.view(self.dtype).reshape(k)

    def numpy(self) -> NDArray[np.integer]:
        k, cols = self.shape
        return self.raw.cpu().numpy().view(self.dtype).reshape(k, cols)  # type: ignore[attr-defined]

    def rows(self, lo: int, hi: int) -> "DevTable":
        return DevTable(self.raw[lo:hi], self.width)  # type: ignore[index]


class HipEngine:
    r"""One BitBIRCH tree resident in the HBM of one MI355X."""

    device_tables = True  # gather_buffers(device_out=True) / fit_buffers(DevTable) keep tables in HBM

    def __init__(
        self,
        branching_factor: int,
        threshold: float,
        criterion: int,
        tolerance: float,
        tol_table: NDArray[np.float64],
        n_features: int,
        device: int = 0,
    ) -> None:
        self._lib = _lib.load()
        self.device = int(device)
        self.n_features = int(n_features)
        self.nbytes = (self.n_features + 7) // 8
        self._h = C.c_void_p()
        tab = np.ascontiguousarray(tol_table, dtype=np.float64)
        _lib.check(
            self._lib.bbh_tree_create(
                C.byref(self._h),
                int(branching_factor),
                float(threshold),
                int(criterion),
                float(tolerance),
                tab.ctypes.data if tab.size else None,
                tab.size,
                self.n_features,
                int(device),
            )
        )

    # -- configuration ---------------------------------------------------------------
    def set_merge(
        self,
        criterion: int,
        tolerance: float,
        tol_table: NDArray[np.float64],
        threshold: float,
        branching_factor: int,
    ) -> None:
        tab = np.ascontiguousarray(tol_table, dtype=np.float64)
        _lib.check(
            self._lib.bbh_tree_set_merge(
                self._h,
                int(criterion),
                float(tolerance),
                tab.ctypes.data if tab.size else None,
                tab.size,
                float(threshold),
                int(branching_factor),
            )
        )

    def reset(self) -> None:
        _lib.check(self._lib.bbh_tree_reset(self._h))

    # -- insertion -------------------------------------------------------------------
    def fit_packed(self, rows: object, stream: int | None = None) -> NDArray[np.uint32]:
        r"""Insert packed fingerprints (numpy uint8 (n, nbytes) or a CUDA/HIP torch
        tensor of that shape, used in place).  Returns the leaf id of each element."""
        if _is_device_tensor(rows):
            if rows.stride(1) != 1:  # type: ignore[attr-defined]
                rows = rows.contiguous()  # type: ignore[attr-defined]
            n, nb = int(rows.shape[0]), int(rows.shape[1])  # type: ignore[attr-defined]
            stride = int(rows.stride(0))  # type: ignore[attr-defined]
            keep = rows
            if stream is None:
                stream = _launch_stream(rows)
        else:
            keep = np.ascontiguousarray(rows, dtype=np.uint8)
            n, nb = keep.shape
            stride = nb
        if nb != self.nbytes:
            raise RuntimeError(f"rows have {nb} bytes, tree expects {self.nbytes}")
        out = np.empty(n, dtype=np.uint32)
        _lib.check(
            self._lib.bbh_tree_fit_packed(
                self._h, _lib.ptr(keep), n, stride, out.ctypes.data, stream
            )
        )
        return out

    @staticmethod
    def fit_packed_many(engines: "list[HipEngine]", rows_list: list) -> list[NDArray[np.uint32]]:
        r"""Insert into several trees with ONE kernel launch (one workgroup per tree)."""
        lib = _lib.load()
        k = len(engines)
        keep, ns, strides, outs = [], [], [], []
        stream = None
        for eng, rows in zip(engines, rows_list):
            if _is_device_tensor(rows):
                if rows.stride(1) != 1:
                    rows = rows.contiguous()
                n, nb, st = int(rows.shape[0]), int(rows.shape[1]), int(rows.stride(0))
                keep.append(rows)
                stream = _one_stream(stream, _launch_stream(rows))
            else:
                arr = np.ascontiguousarray(rows, dtype=np.uint8)
                n, nb = arr.shape
                st = nb
                keep.append(arr)
            if nb != eng.nbytes:
                raise RuntimeError(f"rows have {nb} bytes, tree expects {eng.nbytes}")
            ns.append(n)
            strides.append(st)
            outs.append(np.empty(n, dtype=np.uint32))
        handles = (C.c_void_p * k)(*[e._h for e in engines])
        rows_p = (C.c_void_p * k)(*[_lib.ptr(r) if len(r) else None for r in keep])
        n_p = (C.c_int64 * k)(*ns)
        s_p = (C.c_int64 * k)(*strides)
        out_p = (C.c_void_p * k)(*[o.ctypes.data if o.size else None for o in outs])
        _lib.check(lib.bbh_trees_fit_packed(handles, k, rows_p, n_p, s_p, out_p, stream))
        return outs

    @staticmethod
    def fit_buffers_many(engines: "list[HipEngine]", bufs_list: list) -> list[NDArray[np.uint32]]:
        r"""`fit_buffers` for several trees with ONE kernel launch (one workgroup per tree)."""
        lib = _lib.load()
        k = len(engines)
        keep, ks, widths, outs, ptrs = [], [], [], [], []
        stream = None
        for eng, bufs in zip(engines, bufs_list):
            if isinstance(bufs, DevTable):  # already in HBM: used in place
                if bufs.shape[1] != eng.n_features + 1:
                    raise RuntimeError("buffers must have shape (k, n_features + 1)")
                raw = bufs.raw if bufs.raw.is_contiguous() else bufs.raw.contiguous()
                keep.append(raw)
                ks.append(bufs.shape[0])
                widths.append(bufs.width)
                ptrs.append(int(raw.data_ptr()) if bufs.shape[0] else None)
                stream = _one_stream(stream, _launch_stream(raw))
            else:
                b = np.ascontiguousarray(bufs)
                if b.ndim != 2 or b.shape[1] != eng.n_features + 1:
                    raise RuntimeError("buffers must have shape (k, n_features + 1)")
                if b.dtype.kind != "u":
                    b = b.astype(np.uint64)
                keep.append(b)
                ks.append(b.shape[0])
                widths.append(b.dtype.itemsize)
                ptrs.append(b.ctypes.data if b.size else None)
            outs.append(np.empty(ks[-1], dtype=np.uint32))
        handles = (C.c_void_p * k)(*[e._h for e in engines])
        bufs_p = (C.c_void_p * k)(*ptrs)
        w_p = (C.c_int32 * k)(*widths)
        k_p = (C.c_int64 * k)(*ks)
        out_p = (C.c_void_p * k)(*[o.ctypes.data if o.size else None for o in outs])
        _lib.check(lib.bbh_trees_fit_buffers(handles, k, bufs_p, w_p, k_p, out_p, stream))
        return outs

    def fit_buffers(self, bufs: "NDArray[np.integer] | DevTable", stream: int | None = None) -> NDArray[np.uint32]:
        r"""Insert BitFeature buffers, shape (k, n_features + 1), unsigned dtype (a host array, or a
        `DevTable` that is consumed where it lies in HBM)."""
        if isinstance(bufs, DevTable):
            if bufs.shape[1] != self.n_features + 1:
                raise RuntimeError("buffers must have shape (k, n_features + 1)")
            raw = bufs.raw if bufs.raw.is_contiguous() else bufs.raw.contiguous()
            k = bufs.shape[0]
            out = np.empty(k, dtype=np.uint32)
            if k:
                _lib.check(self._lib.bbh_tree_fit_buffers(
                    self._h, int(raw.data_ptr()), bufs.width, k, out.ctypes.data,
                    _launch_stream(raw) if stream is None else stream))
            return out
        bufs = np.ascontiguousarray(bufs)
        if bufs.ndim != 2 or bufs.shape[1] != self.n_features + 1:
            raise RuntimeError("buffers must have shape (k, n_features + 1)")
        if bufs.dtype.kind != "u":
            bufs = bufs.astype(np.uint64)
        k = bufs.shape[0]
        out = np.empty(k, dtype=np.uint32)
        _lib.check(
            self._lib.bbh_tree_fit_buffers(
                self._h, bufs.ctypes.data, bufs.dtype.itemsize, k, out.ctypes.data, stream
            )
        )
        return out

    # -- extraction ------------------------------------------------------------------
    def leaf_count(self) -> int:
        k = C.c_int64(0)
        _lib.check(self._lib.bbh_tree_leaf_count(self._h, C.byref(k)))
        return int(k.value)

    def export_leaves(
        self, ls_width: int | None = None
    ) -> tuple[NDArray[np.uint32], NDArray[np.uint64], NDArray[np.uint8], NDArray | None]:
        k = self.leaf_count()
        ids = np.empty(k, dtype=np.uint32)
        ns = np.empty(k, dtype=np.uint64)
        cents = np.empty((k, self.nbytes), dtype=np.uint8)
        ls = None
        if ls_width is not None:
            ls = np.empty((k, self.n_features), dtype=_WIDTH_DTYPES[ls_width])
        _lib.check(
            self._lib.bbh_tree_export_leaves(
                self._h,
                ids.ctypes.data,
                ns.ctypes.data,
                cents.ctypes.data,
                ls.ctypes.data if ls is not None else None,
                ls_width or 0,
            )
        )
        return ids, ns, cents, ls

    def gather_buffers(self, positions: NDArray[np.int64], width: int, device_out: bool = False) -> "NDArray[np.integer] | DevTable":
        r"""BitFeature buffer rows of the leaves at `positions` (chain order); `device_out`: the table
        is produced in HBM and stays there (`DevTable`)."""
        pos = np.ascontiguousarray(positions, dtype=np.int64)
        if device_out:
            import torch

            raw = torch.empty((pos.size, (self.n_features + 1) * width), dtype=torch.uint8,
                              device=torch.device("cuda", self.device))
            if pos.size:
                _lib.check(self._lib.bbh_tree_gather_buffers(self._h, pos.ctypes.data, pos.size, width, int(raw.data_ptr())))
            return DevTable(raw, width)
        out = np.empty((pos.size, self.n_features + 1), dtype=_WIDTH_DTYPES[width])
        if pos.size:
            _lib.check(
                self._lib.bbh_tree_gather_buffers(
                    self._h, pos.ctypes.data, pos.size, width, out.ctypes.data
                )
            )
        return out

    def stats(self) -> NDArray[np.uint64]:
        out = np.zeros(8, dtype=np.uint64)
        _lib.check(self._lib.bbh_tree_stats(self._h, out.ctypes.data))
        return out

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.bbh_tree_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass
