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
    memory).  `raw` is a 2-D ``torch.uint8`` tensor of shape (k_head, (n_features + 1) * width).

    **Singleton tail** (uint8 tables only): the rows behind `raw` that are BitFeatures of ONE fingerprint may
    be held in `tail`, a ``torch.uint8`` tensor (k_tail, n_features / 8) of PACKED rows - such a buffer row is
    the fingerprint's bits as bytes followed by n_samples = 1, i.e. 2049 bytes that say what 256 do.  The
    tables multiround produces are sorted by n_samples, largest first (bitbirch.py:1216-1222), so their
    singletons are exactly a tail - 99 % of the rows on sparse fingerprints that hardly merge.  The table the
    reference would hold is `raw` followed by the unpacked tail; `numpy()` returns precisely that, and
    inserting the head as buffers and the tail as fingerprints is the same sequence of BitFeatures
    (bitbirch.py:412-421 vs :422-448)."""

    __slots__ = ("raw", "width", "tail")

    def __init__(self, raw: object, width: int, tail: object = None) -> None:
        self.raw = raw
        self.width = int(width)
        self.tail = tail if tail is not None and int(tail.shape[0]) > 0 else None  # type: ignore[attr-defined]
        if self.tail is not None and self.width != 1:
            raise ValueError("only uint8 tables have a singleton tail")

    @property
    def n_head(self) -> int:
        return int(self.raw.shape[0])  # type: ignore[attr-defined]

    @property
    def n_tail(self) -> int:
        return 0 if self.tail is None else int(self.tail.shape[0])  # type: ignore[attr-defined]

    @property
    def shape(self) -> tuple[int, int]:
        return self.n_head + self.n_tail, int(self.raw.shape[1]) // self.width  # type: ignore[attr-defined]

    @property
    def dtype(self) -> np.dtype:
        return np.dtype(_WIDTH_DTYPES[self.width])

    @property
    def nbytes(self) -> int:
        r"""Bytes in HBM (what an exchange moves)."""
        return int(self.raw.numel()) + (0 if self.tail is None else int(self.tail.numel()))  # type: ignore[attr-defined]

    def __len__(self) -> int:
        return self.shape[0]

    def n_samples(self) -> NDArray[np.integer]:
        r"""The last column (n_samples of every BitFeature) on the host."""
        k, cols = self.n_head, self.shape[1]
        col = self.raw.view(k, cols, self.width)[:, -1, :].contiguous().cpu().numpy()  # type: ignore[attr-defined]
        col = col.view(self.dtype).reshape(k)
        if self.tail is not None:
            col = np.concatenate([col, np.ones(self.n_tail, dtype=self.dtype)])
        return col

    def numpy(self) -> NDArray[np.integer]:
        k, cols = self.n_head, self.shape[1]
        head = self.raw.cpu().numpy().view(self.dtype).reshape(k, cols)  # type: ignore[attr-defined]
        if self.tail is None:
            return head
        packed = self.tail.cpu().numpy()  # type: ignore[attr-defined]
        rows = np.ones((packed.shape[0], cols), dtype=np.uint8)
        rows[:, :-1] = np.unpackbits(packed, axis=1)[:, : cols - 1]
        return np.concatenate([head, rows])

    def rows(self, lo: int, hi: int) -> "DevTable":
        kh = self.n_head
        tail = None
        if self.tail is not None and hi > kh:
            tail = self.tail[max(lo - kh, 0):hi - kh]  # type: ignore[index]
        return DevTable(self.raw[min(lo, kh):min(hi, kh)], self.width, tail)  # type: ignore[index]


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
                ks.append(bufs.n_head)
                widths.append(bufs.width)
                ptrs.append(int(raw.data_ptr()) if bufs.n_head else None)
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
        # singleton tails: the same BitFeatures as packed fingerprints, behind their table's head
        tails = [(i, b.tail) for i, b in enumerate(bufs_list) if isinstance(b, DevTable) and b.tail is not None]
        if tails:
            t_out = HipEngine.fit_packed_many([engines[i] for i, _ in tails], [t for _, t in tails])
            for (i, _), o in zip(tails, t_out):
                outs[i] = np.concatenate([outs[i], o])
        return outs

    def fit_buffers(self, bufs: "NDArray[np.integer] | DevTable", stream: int | None = None) -> NDArray[np.uint32]:
        r"""Insert BitFeature buffers, shape (k, n_features + 1), unsigned dtype (a host array, or a
        `DevTable` that is consumed where it lies in HBM)."""
        if isinstance(bufs, DevTable):
            if bufs.shape[1] != self.n_features + 1:
                raise RuntimeError("buffers must have shape (k, n_features + 1)")
            raw = bufs.raw if bufs.raw.is_contiguous() else bufs.raw.contiguous()
            k = bufs.n_head
            out = np.empty(k, dtype=np.uint32)
            if k:
                _lib.check(self._lib.bbh_tree_fit_buffers(
                    self._h, int(raw.data_ptr()), bufs.width, k, out.ctypes.data,
                    _launch_stream(raw) if stream is None else stream))
            if bufs.tail is not None:  # the singleton tail: packed fingerprints (same BitFeatures, see DevTable)
                out = np.concatenate([out, self.fit_packed(bufs.tail, stream)])
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

    def gather_buffers(self, positions: NDArray[np.int64], width: int, device_out: bool = False,
                       n_tail: int = 0) -> "NDArray[np.integer] | DevTable":
        r"""BitFeature buffer rows of the leaves at `positions` (chain order); `device_out`: the table
        is produced in HBM and stays there (`DevTable`).  `n_tail` (device tables of width 1): the last n_tail
        positions are BitFeatures of one fingerprint and are kept as packed rows (`DevTable.tail`)."""
        pos = np.ascontiguousarray(positions, dtype=np.int64)
        if device_out:
            import torch

            dev = torch.device("cuda", self.device)
            n_tail = int(n_tail) if width == 1 else 0
            kh = pos.size - n_tail
            raw = torch.empty((kh, (self.n_features + 1) * width), dtype=torch.uint8, device=dev)
            if kh:
                _lib.check(self._lib.bbh_tree_gather_buffers(self._h, pos.ctypes.data, kh, width, int(raw.data_ptr())))
            tail = None
            if n_tail:
                tail = torch.empty((n_tail, self.nbytes), dtype=torch.uint8, device=dev)
                tpos = np.ascontiguousarray(pos[kh:])
                _lib.check(self._lib.bbh_tree_gather_centroids(self._h, tpos.ctypes.data, n_tail, int(tail.data_ptr())))
            return DevTable(raw, width, tail)
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

    def kernel_counts(self) -> NDArray[np.uint64]:
        r"""[0..2] elements inserted by the pipelined / steady-state / complete kernel, [3..5] their launches, [6] launches
        that ended with "tree shape not handled by the pipeline", [7] launches that ended on an exhausted pool."""
        out = np.zeros(8, dtype=np.uint64)
        _lib.check(self._lib.bbh_tree_kernel_counts(self._h, out.ctypes.data))
        return out

    def sys_counts(self) -> NDArray[np.uint64]:
        r"""The level-systolic kernel (one tree over many workgroups): [0] elements, [1] launches, [2] relaunches after a root
        split, [3] workgroups, [4] busy shader cycles of all its workgroups, [5] of workgroup 0, [6] of the busiest other
        workgroup, [7] launches it refused."""
        out = np.zeros(8, dtype=np.uint64)
        _lib.check(self._lib.bbh_tree_sys_counts(self._h, out.ctypes.data))
        return out

    def memory(self) -> NDArray[np.uint64]:
        r"""[0] node pools (bytes, capacity), [1] their used part, [2] cluster-feature pools, [3] peak of this tree's
        allocations, [4] compactions of the node pools, [5] / [6] nodes the last one sealed / left at full capacity,
        [7] sealed nodes thawed by an insertion."""
        out = np.zeros(8, dtype=np.uint64)
        _lib.check(self._lib.bbh_tree_memory(self._h, out.ctypes.data))
        return out

    def compact(self, seal: bool = True) -> None:
        r"""Compact the node pools now (tests; the engine does it on its own when large pools have to grow)."""
        _lib.check(self._lib.bbh_tree_compact(self._h, 1 if seal else 0))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.bbh_tree_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass
