r"""`BitBirch` with the reference's class surface, backed by the HBM-resident tree engine.

Drop-in for `bblean.bitbirch.BitBirch` (reference bitbirch.py:539-1473) on the hot path:
the per-fingerprint Python loop of `fit` / `_fit_buffers` (bitbirch.py:769-787,
:848-866) and everything it calls is ONE C-ABI call into libbbhip.so per batch; this
file only keeps what the reference keeps on the host *around* that loop:

  * molecule-index bookkeeping.  The reference stores a Python list on every
    `_BFSubcluster` and extends it on each merge (bitbirch.py:505, :524).  The engine
    instead returns, per inserted element, the id of the leaf BitFeature it ended in;
    member lists are rebuilt lazily as "elements grouped by leaf id, in insertion order",
    which is the same concatenation order (SURVEY.md section 8a rule 13).
  * result ordering: leaves in leaf-chain order, stably sorted by size (bitbirch.py:
    1216-1222), labels 1..K (bitbirch.py:1041-1042).
  * refine / recluster orchestration: dtype groups in first-seen order, exploded
    singletons appended to the uint8 group (bitbirch.py:1187-1308).
"""
from __future__ import annotations

import random
import typing as tp
import warnings
from pathlib import Path
from weakref import WeakSet

import numpy as np
from numpy.typing import NDArray

from bblean_amd._merges import BUILTIN_MERGES, MergeCriterion, get_merge_accept_fn
from bblean_amd.fingerprints import pack_fingerprints, unpack_fingerprints
from bblean_amd.utils import min_safe_uint

__all__ = ["BitBirch", "set_merge", "fit_concurrently"]

_Input = tp.Union[NDArray[np.integer], list]

_BITBIRCH_INSTANCES: "WeakSet[BitBirch]" = WeakSet()
_global_merge_accept: MergeCriterion | None = None


def set_merge(merge_criterion: str, tolerance: float = 0.05) -> None:
    r"""Legacy global merge setter (reference bitbirch.py:104-129).  Discouraged."""
    warnings.warn(
        "Use of the global `set_merge` function is highly discouraged,\n"
        " instead use either: "
        " bb_tree = BitBirch(...)\n"
        " bb_tree.set_merge(merge_criterion=..., tolerance=...)\n"
        " or directly: `bb_tree = BitBirch(..., merge_criterion=..., tolerance=...)`.",
        UserWarning,
    )
    global _global_merge_accept
    _global_merge_accept = get_merge_accept_fn(merge_criterion, tolerance)
    for tree in _BITBIRCH_INSTANCES:
        tree._merge_accept_fn = _global_merge_accept
        tree._push_merge_to_engine()


def _validate_n_features(X: _Input, input_is_packed: bool, n_features: int | None = None) -> int:
    r"""Same checks and messages as the reference (bitbirch.py:133-159)."""
    if len(X) == 0:
        raise ValueError("Input must have at least 1 fingerprint")
    width = len(X[0]) if isinstance(X, list) else X.shape[1]
    if input_is_packed:
        padded = width * 8
        if n_features is None:
            return padded
        if padded < n_features:
            raise ValueError("n_features is larger than the padded length, which is inconsistent")
        return n_features
    if n_features is not None and n_features != width:
        raise ValueError(
            "n_features is redundant for non-packed inputs"
            " if passed, it must be equal to X.shape[1] (or len(X[0]))."
            f" For passed X the inferred n_features was {width}."
            " If this value is not what you expected,"
            " make sure the passed X is actually unpacked."
        )
    return width


def _dtype_name_for(n_samples: int) -> str:
    r"""dtype group of a leaf BitFeature: always min_safe_uint(n_samples)
    (bitbirch.py:480, :493; SURVEY.md section 8a rule 11)."""
    return min_safe_uint(int(n_samples)).name


class _IndexLists:
    r"""CSR holder for per-element molecule-index lists (avoids millions of Python
    lists on the way into `_fit_buffers`)."""

    __slots__ = ("counts", "flat")

    def __init__(self, counts: NDArray[np.int64], flat: NDArray[np.int64]) -> None:
        self.counts = counts
        self.flat = flat

    @classmethod
    def from_sequences(cls, seqs: tp.Iterable[tp.Sequence[int]], k: int) -> "_IndexLists":
        counts = np.empty(k, dtype=np.int64)
        chunks: list[NDArray[np.int64]] = []
        i = 0
        for i, s in zip(range(k), seqs):
            counts[i] = len(s)
            chunks.append(np.asarray(s, dtype=np.int64).reshape(-1))
        got = len(chunks)
        counts = counts[:got]
        flat = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.int64)
        return cls(counts, flat)

    def __len__(self) -> int:
        return int(self.counts.size)

    def to_lists(self) -> list[list[int]]:
        offs = np.concatenate(([0], np.cumsum(self.counts))).tolist()  # (Python ints: indexing an array per cluster costs more)
        flat = self.flat.tolist()
        return [flat[a:b] for a, b in zip(offs[:-1], offs[1:])]


class _LeafBF:
    r"""Read-only view of one leaf BitFeature (what `_get_leaf_bfs` hands to
    `bblean.sklearn`, reference sklearn.py:90 / bitbirch.py:360-526)."""

    __slots__ = ("n_samples", "packed_centroid", "mol_indices", "n_features", "_tree", "_pos")

    def __init__(self, tree: "BitBirch", pos: int, n: int, cent: NDArray[np.uint8], mols: list[int]):
        self._tree = tree
        self._pos = pos
        self.n_samples = n
        self.packed_centroid = cent
        self.mol_indices = mols
        self.n_features = tree._n_features

    @property
    def dtype_name(self) -> str:
        return _dtype_name_for(self.n_samples)

    @property
    def unpacked_centroid(self) -> NDArray[np.uint8]:
        return unpack_fingerprints(self.packed_centroid, self.n_features)

    @property
    def _buffer(self) -> NDArray[np.integer]:
        width = min_safe_uint(self.n_samples).itemsize
        return self._tree._engine.gather_buffers(np.array([self._pos], dtype=np.int64), width)[0]

    @property
    def linear_sum(self) -> NDArray[np.integer]:
        return self._buffer[:-1]


class BitBirch:
    r"""BitBIRCH clustering on one MI355X; same constructor and methods as the reference
    class (bitbirch.py:596-643).  `_engine_factory` is a test hook (tests inject the CPU
    oracle to check this host logic without a GPU); the product default is the HIP
    engine and there is no CPU fallback."""

    def __init__(
        self,
        *,
        threshold: float = 0.65,
        branching_factor: int = 50,
        merge_criterion: str | MergeCriterion | None = None,
        tolerance: float | None = None,
        device: int = 0,
        _engine_factory: tp.Callable[..., tp.Any] | None = None,
    ):
        self.threshold = threshold
        self.branching_factor = branching_factor
        if _global_merge_accept is not None:
            if tolerance is not None:
                raise ValueError(
                    "tolerance can only be passed if "
                    "the *global* set_merge function has *not* been used"
                )
            if merge_criterion is not None:
                raise ValueError(
                    "merge_criterion can only be passed if "
                    "the *global* set_merge function has *not* been used"
                )
            self._merge_accept_fn = _global_merge_accept
        elif isinstance(merge_criterion, MergeCriterion):
            if tolerance is not None:
                raise ValueError("'tolerance' arg is disregarded for custom merge functions")
            self._merge_accept_fn = merge_criterion
        else:
            name = "diameter" if merge_criterion is None else merge_criterion
            self._merge_accept_fn = get_merge_accept_fn(name, 0.05 if tolerance is None else tolerance)

        self._device = device
        self._engine_factory = _engine_factory
        self._engine: tp.Any = None
        self._n_features: int = 0
        self._num_fitted_fps = 0
        self._is_init = False
        self._internal_released = False
        # one entry per fit call since the last reset: (leaf ids, counts | None, flat ids)
        self._log_leaf: list[NDArray[np.uint32]] = []
        self._log_counts: list[NDArray[np.int64] | None] = []
        self._log_ids: list[NDArray[np.int64]] = []
        self._cache: dict[str, tp.Any] = {}
        self._global_clustering_centroid_labels: NDArray[np.int64] | None = None
        self._n_global_clusters = 0
        _BITBIRCH_INSTANCES.add(self)

    # ---------------------------------------------------------------- properties ----
    @property
    def merge_criterion(self) -> str:
        return self._merge_accept_fn.name

    @merge_criterion.setter
    def merge_criterion(self, value: str) -> None:
        self.set_merge(criterion=value)

    @property
    def tolerance(self) -> float | None:
        return self._merge_accept_fn.tolerance

    @tolerance.setter
    def tolerance(self, value: float) -> None:
        self.set_merge(tolerance=value)

    @property
    def is_init(self) -> bool:
        return self._is_init

    @property
    def num_fitted_fps(self) -> int:
        return self._num_fitted_fps

    @property
    def _only_has_leaves(self) -> bool:
        return self._internal_released

    # ------------------------------------------------------------- configuration ----
    def set_merge(
        self,
        criterion: str | MergeCriterion | None = None,
        *,
        tolerance: float | None = None,
        threshold: float | None = None,
        branching_factor: int | None = None,
    ) -> None:
        r"""Same semantics as the reference (bitbirch.py:674-703), including that an
        unspecified tolerance resets it to 0.05 on criteria that have one."""
        if _global_merge_accept is not None:
            raise ValueError(
                "BitBirch.set_merge() can only called if "
                "the global set_merge() function has *not* been used"
            )
        _tolerance = 0.05 if tolerance is None else tolerance
        if isinstance(criterion, MergeCriterion):
            self._merge_accept_fn = criterion
        elif isinstance(criterion, str):
            self._merge_accept_fn = get_merge_accept_fn(criterion, _tolerance)
        if self._merge_accept_fn.has_tolerance:
            self._merge_accept_fn.tolerance = _tolerance
        elif tolerance is not None:
            raise ValueError(f"Can't set tolerance for {self._merge_accept_fn}")
        if threshold is not None:
            self.threshold = threshold
        if branching_factor is not None:
            self.branching_factor = branching_factor
        self._push_merge_to_engine()

    def _push_merge_to_engine(self) -> None:
        if self._engine is None:
            return
        fn = self._merge_accept_fn
        self._engine.set_merge(
            fn.code,
            0.0 if fn.tolerance is None else fn.tolerance,
            fn.tolerance_table(),
            self.threshold,
            self.branching_factor,
        )

    def _ensure_engine(self, n_features: int) -> None:
        if self._engine is not None and n_features != self._n_features:
            if self._is_init:
                raise ValueError(
                    f"tree was built with n_features={self._n_features}, got {n_features}"
                )
            # a reset (or never fitted) tree is re-initialised for the new width, like the
            # reference's _initialize_tree (bitbirch.py:880-884)
            close = getattr(self._engine, "close", None)
            if close is not None:
                close()
            self._engine = None
        if self._engine is not None:
            self._push_merge_to_engine()
            return
        fn = self._merge_accept_fn
        factory = self._engine_factory
        if factory is None:
            from bblean_amd._engine import HipEngine as factory  # no fallback: raises if absent
        self._engine = factory(
            self.branching_factor,
            self.threshold,
            fn.code,
            0.0 if fn.tolerance is None else fn.tolerance,
            fn.tolerance_table(),
            n_features,
            self._device,
        )
        self._n_features = n_features

    # ------------------------------------------------------------------ fitting ----
    def fit(
        self,
        X: _Input | Path | str | tp.Any,
        /,
        reinsert_indices: tp.Iterable[int] | None = None,
        input_is_packed: bool = True,
        n_features: int | None = None,
        max_fps: int | None = None,
    ) -> "BitBirch":
        r"""Insert fingerprints in order (reference bitbirch.py:705-788).

        ``X``: packed/unpacked array, list of rows, ``.npy`` path, or a device-resident
        ``torch.uint8`` tensor of packed rows (used in place, no PCIe copy).
        """
        rows, ids = self._prepare_fit(X, reinsert_indices, input_is_packed, n_features, max_fps)
        if len(ids):
            self._commit_fit(self._engine.fit_packed(rows), ids)
        return self

    def _prepare_fit(self, X, reinsert_indices, input_is_packed, n_features, max_fps):  # type: ignore[no-untyped-def]
        r"""Everything `fit` does before the hot loop: load / validate / pack the input, create the
        engine, resolve the molecule indices.  Returns (packed rows, indices)."""
        is_dev = hasattr(X, "data_ptr") and getattr(X, "is_cuda", False)
        if isinstance(X, (Path, str)):
            X = np.load(Path(X), mmap_mode="r")
            if max_fps is not None:
                X = X[:max_fps]
        elif max_fps is not None:
            X = X[:max_fps]
        if is_dev:
            if not input_is_packed:
                raise ValueError("device-resident input must be packed uint8")
            if X.shape[0] == 0:
                raise ValueError("Input must have at least 1 fingerprint")
            nf = X.shape[1] * 8 if n_features is None else n_features
            if X.shape[1] * 8 < nf:
                raise ValueError("n_features is larger than the padded length, which is inconsistent")
            rows: tp.Any = X
            n_rows = int(X.shape[0])
        else:
            nf = _validate_n_features(X, input_is_packed, n_features)
            arr = np.stack(X) if isinstance(X, list) else np.asarray(X)
            if input_is_packed:
                rows = np.ascontiguousarray(arr, dtype=np.uint8)
                if rows.shape[1] * 8 != nf:
                    # trailing pad bits beyond n_features are dropped by the reference's
                    # unpack(count=n_features); only byte-aligned n_features reach here
                    rows = pack_fingerprints(unpack_fingerprints(rows, nf))
            else:
                rows = pack_fingerprints(arr.astype(np.uint8, copy=False))
            n_rows = rows.shape[0]
        if nf % 8 != 0:
            raise RuntimeError("Only n_features divisible by 8 is supported")
        if self._only_has_leaves:
            raise ValueError("Internal nodes were released, call reset() before fit()")
        self._ensure_engine(nf)
        if reinsert_indices is None:
            ids = np.arange(self._num_fitted_fps, self._num_fitted_fps + n_rows, dtype=np.int64)
        elif isinstance(reinsert_indices, range):
            ids = np.arange(reinsert_indices.start, reinsert_indices.stop, reinsert_indices.step, dtype=np.int64)
        else:
            ids = np.fromiter(reinsert_indices, dtype=np.int64)
        if ids.size < n_rows:  # zip() in the reference stops at the shorter iterable
            n_rows = ids.size
            rows = rows[:n_rows]
        ids = ids[:n_rows]
        self._is_init = True
        return rows, ids

    def _commit_fit(self, leaf: NDArray[np.uint32], ids: NDArray[np.int64]) -> None:
        self._log_leaf.append(leaf)
        self._log_counts.append(None)
        self._log_ids.append(ids)
        self._num_fitted_fps += int(ids.size)
        self._cache.clear()

    def fit_reinsert(self, X, reinsert_indices, input_is_packed=True, n_features=None, max_fps=None):  # type: ignore[no-untyped-def]
        r""":meta private: (reference bitbirch.py:868-878)"""
        return self.fit(X, reinsert_indices, input_is_packed, n_features, max_fps)

    def _fit_buffers(
        self,
        X: _Input | Path | str,
        reinsert_index_seqs: tp.Iterable[tp.Sequence[int]] | tp.Literal["omit"] | _IndexLists = "omit",
    ) -> "BitBirch":
        r"""Insert BitFeature buffers ``[linear_sum | n_samples]`` in order
        (reference bitbirch.py:790-866)."""
        prepared = self._prepare_fit_buffers(X, reinsert_index_seqs)
        if prepared is not None:
            bufs, counts, flat = prepared
            self._commit_fit_buffers(self._engine.fit_buffers(bufs), counts, flat)
        return self

    def _prepare_fit_buffers(self, X, reinsert_index_seqs):  # type: ignore[no-untyped-def]
        r"""Everything `_fit_buffers` does before the hot loop.  Returns (buffers, member counts, member
        ids) or None when there is nothing to insert."""
        if isinstance(X, (Path, str)):
            X = np.load(Path(X), mmap_mode="r")
        is_dev = hasattr(X, "raw") and hasattr(X, "n_samples")  # _engine.DevTable: a table resident in HBM
        if is_dev:
            if len(X) == 0:
                raise ValueError("Input must have at least 1 fingerprint")
            nf = X.shape[1] - 1
            bufs = X
        else:
            nf = _validate_n_features(X, input_is_packed=False) - 1
        if is_dev:
            pass
        elif isinstance(X, list):
            first = np.asarray(X[0])
            bufs = np.stack([np.asarray(b).astype(first.dtype, copy=False) for b in X])
        else:
            bufs = np.asarray(X)
        if self._only_has_leaves:
            raise ValueError("Internal nodes were released, call reset() before fit()")
        self._ensure_engine(nf)
        k = bufs.shape[0]
        if isinstance(reinsert_index_seqs, str) and reinsert_index_seqs == "omit":
            # the reference zips against a generator of `num_fitted_fps` empty tuples
            k = min(k, self._num_fitted_fps)
            idx = _IndexLists(np.zeros(k, dtype=np.int64), np.zeros(0, dtype=np.int64))
            check = False
        elif isinstance(reinsert_index_seqs, _IndexLists):
            idx = reinsert_index_seqs
            check = True
        else:
            idx = _IndexLists.from_sequences(reinsert_index_seqs, k)
            check = True
        k = min(k, len(idx))
        bufs = bufs.rows(0, k) if is_dev else bufs[:k]
        counts = idx.counts[:k]
        flat = idx.flat[: int(counts.sum())]
        if check and k:
            n_col = (bufs.n_samples() if is_dev else bufs[:, -1]).astype(np.int64)
            bad = np.nonzero(counts != n_col)[0]
            if bad.size:
                i = int(bad[0])
                where = f" (buffer {i} of {k}" + (f", device table: {bufs.n_head} rows + {bufs.n_tail} packed)" if is_dev else ")")
                raise ValueError(
                    "Expected len(mol_indices) == buffer[-1],"
                    f" but found {int(counts[i])} != {int(n_col[i])}" + where
                )
        self._is_init = True
        if not k:
            return None
        return bufs, counts, flat

    def _commit_fit_buffers(self, leaf: NDArray[np.uint32], counts: NDArray, flat: NDArray) -> None:
        self._log_leaf.append(leaf)
        self._log_counts.append(counts.astype(np.int64))
        self._log_ids.append(flat.astype(np.int64))
        self._num_fitted_fps += int(counts.sum())
        self._cache.clear()

    # ----------------------------------------------------------- tree lifecycle ----
    def reset(self) -> None:
        r"""Drop the whole tree, keep merge parameters (reference bitbirch.py:1078)."""
        if self._engine is not None:
            self._engine.reset()
        self._num_fitted_fps = 0
        self._is_init = False
        self._internal_released = False
        self._log_leaf.clear()
        self._log_counts.clear()
        self._log_ids.clear()
        self._cache.clear()

    def delete_internal_nodes(self) -> None:
        r"""After this, only leaves may be read until `reset()` (bitbirch.py:1092-1104).
        Nothing is freed on the device (the pools are reused by the next fit); the flag
        reproduces the reference's state machine: it only trips once the root has split."""
        if self._engine is not None and int(self._engine.stats()[5]) > 1:
            self._internal_released = True

    # ----------------------------------------------------------------- results ----
    def _require_init(self) -> None:
        if not self._is_init:
            raise ValueError("The model has not been fitted yet.")

    def _leaves(self) -> dict[str, tp.Any]:
        r"""Leaf table in chain order + member lists, cached until the next fit."""
        self._require_init()
        if "leaves" in self._cache:
            return self._cache["leaves"]
        ids, ns, cents, _ = self._engine.export_leaves(None)
        k = ids.size
        # elements grouped by the leaf they ended in, insertion order inside a group
        if self._log_leaf:
            elem_leaf = np.concatenate(self._log_leaf)
            elem_cnt = np.concatenate(
                [np.ones(l.size, dtype=np.int64) if c is None else c
                 for l, c in zip(self._log_leaf, self._log_counts)]
            )
            flat_ids = np.concatenate(self._log_ids)
        else:
            elem_leaf = np.zeros(0, dtype=np.uint32)
            elem_cnt = np.zeros(0, dtype=np.int64)
            flat_ids = np.zeros(0, dtype=np.int64)
        elem_off = np.cumsum(elem_cnt) - elem_cnt
        order = np.argsort(elem_leaf, kind="stable")
        s_leaf = elem_leaf[order]
        s_cnt = elem_cnt[order]
        s_off = elem_off[order]
        total = int(s_cnt.sum())
        if total:
            dst_start = np.cumsum(s_cnt) - s_cnt
            gather = np.repeat(s_off - dst_start, s_cnt) + np.arange(total, dtype=np.int64)
            members = flat_ids[gather]
        else:
            members = np.zeros(0, dtype=np.int64)
        # per leaf id: [begin, end) into `members`
        max_id = int(max(ids.max(initial=0), s_leaf.max(initial=0))) + 1
        size_by_id = np.bincount(s_leaf, weights=s_cnt, minlength=max_id).astype(np.int64)
        end_by_id = np.cumsum(size_by_id)
        beg_by_id = end_by_id - size_by_id
        out = {
            "ids": ids,
            "n": ns,
            "cents": cents,
            "beg": beg_by_id[ids] if k else np.zeros(0, dtype=np.int64),
            "end": end_by_id[ids] if k else np.zeros(0, dtype=np.int64),
            "members": members,
        }
        self._cache["leaves"] = out
        return out

    def _leaf_order(self, sort: bool) -> NDArray[np.int64]:
        lv = self._leaves()
        k = lv["ids"].size
        if not sort:
            return np.arange(k, dtype=np.int64)
        # stable sort by n_samples, largest first (list.sort(reverse=True) is stable)
        return np.argsort(-lv["n"].astype(np.int64), kind="stable")

    def _members_of(self, positions: NDArray[np.int64]) -> list[list[int]]:
        lv = self._leaves()
        mem = lv["members"].tolist()
        beg = lv["beg"][positions].tolist()
        end = lv["end"][positions].tolist()
        return [mem[b:e] for b, e in zip(beg, end)]

    def get_cluster_mol_ids(self, sort: bool = True, global_clusters: bool = False) -> list[list[int]]:
        r"""Molecule indices of each cluster (reference bitbirch.py:969-988)."""
        lists = self._members_of(self._leaf_order(sort))
        if global_clusters:
            if self._global_clustering_centroid_labels is None:
                raise ValueError("Must perform global clustering before fetching global labels")
            labels = self._global_clustering_centroid_labels - 1
            return self._new_ids_from_labels(lists, labels, self._n_global_clusters)
        return lists

    @staticmethod
    def _new_ids_from_labels(members, labels, n_labels=None):  # type: ignore[no-untyped-def]
        if n_labels is None:
            n_labels = len(np.unique(labels))
        out: list[list[int]] = [[] for _ in range(n_labels)]
        for i, idxs in enumerate(members):
            out[labels[i]].extend(idxs)
        return out

    def get_centroids(self, sort: bool = True, packed: bool = True) -> list[NDArray[np.uint8]]:
        lv = self._leaves()
        cents = lv["cents"][self._leaf_order(sort)]
        if not packed:
            cents = unpack_fingerprints(cents, self._n_features)
        return list(cents)

    def get_centroids_mol_ids(self, sort: bool = True, packed: bool = True) -> dict[str, list]:
        r"""(reference bitbirch.py:895-907)"""
        return {
            "centroids": self.get_centroids(sort, packed),
            "mol_ids": self._members_of(self._leaf_order(sort)),
        }

    def get_assignments(
        self,
        n_mols: int | None = None,
        sort: bool = True,
        check_valid: bool = True,
        global_clusters: bool = False,
    ) -> NDArray[np.uint64]:
        r"""Cluster label (1..K) of every fitted fingerprint (bitbirch.py:1002-1047)."""
        if n_mols is not None:
            warnings.warn("The n_mols argument is redundant", DeprecationWarning)
        if n_mols is not None and n_mols != self.num_fitted_fps:
            raise ValueError(
                f"Provided n_mols {n_mols} is different"
                f" from the number of fitted fingerprints {self.num_fitted_fps}"
            )
        lv = self._leaves()
        order = self._leaf_order(sort)
        sizes = (lv["end"] - lv["beg"])[order]
        if global_clusters:
            if self._global_clustering_centroid_labels is None:
                raise ValueError("Must perform global clustering before fetching global labels")
            labels_per_leaf = np.asarray(self._global_clustering_centroid_labels, dtype=np.uint64)
        else:
            labels_per_leaf = np.arange(1, order.size + 1, dtype=np.uint64)
        assignments = np.zeros(self.num_fitted_fps, dtype=np.uint64)
        total = int(sizes.sum())
        if total:
            starts = lv["beg"][order]
            dst = np.cumsum(sizes) - sizes
            gather = np.repeat(starts - dst, sizes) + np.arange(total, dtype=np.int64)
            assignments[lv["members"][gather]] = np.repeat(labels_per_leaf[: order.size], sizes)
        if check_valid and (assignments == 0).any():
            raise ValueError("There are unasigned molecules")
        return assignments

    def dump_assignments(self, path, smiles=(), sort=True, global_clusters=False, check_valid=True):  # type: ignore[no-untyped-def]
        r"""(reference bitbirch.py:1049-1076)"""
        import pandas as pd

        if isinstance(smiles, str):
            smiles = [smiles]
        smiles = np.asarray(smiles, dtype=np.str_)
        a = self.get_assignments(sort=sort, check_valid=check_valid, global_clusters=global_clusters)
        if smiles.size and (len(a) != len(smiles)):
            raise ValueError(
                f"Len of the provided smiles {len(smiles)}"
                f" must match the number of fitted fingerprints {self.num_fitted_fps}"
            )
        df = pd.DataFrame({"assignments": a})
        if smiles.size:
            df["smiles"] = smiles
        df.to_csv(Path(path), index=False)

    def _get_leaf_bfs(self, sort: bool = True) -> list[_LeafBF]:
        lv = self._leaves()
        order = self._leaf_order(sort)
        lists = self._members_of(order)
        return [
            _LeafBF(self, int(p), int(lv["n"][p]), lv["cents"][p], m)
            for p, m in zip(order, lists)
        ]

    # medoids: analysis helpers on the host (reference bitbirch.py:909-967)
    def get_medoids_mol_ids(self, fps, sort=True, pack=True, global_clusters=False, input_is_packed=True, n_features=None):  # type: ignore[no-untyped-def]
        from bblean_amd.similarity import jt_isim_medoid

        members = self.get_cluster_mol_ids(sort=sort, global_clusters=global_clusters)
        if input_is_packed:
            fps = unpack_fingerprints(fps, n_features)
        medoids = np.zeros((len(members), fps.shape[1]), dtype=np.uint8)
        for i, m in enumerate(members):
            medoids[i] = jt_isim_medoid(fps[m], input_is_packed=False, pack=False)[1]
        if pack:
            medoids = pack_fingerprints(medoids)
        return {"medoids": medoids, "mol_ids": members}

    def get_medoids(self, fps, sort=True, pack=True, global_clusters=False, input_is_packed=True, n_features=None):  # type: ignore[no-untyped-def]
        return self.get_medoids_mol_ids(fps, sort, pack, global_clusters, input_is_packed, n_features)["medoids"]

    # -------------------------------------------------- BitFeature tables / refine ----
    def _group_positions(self, positions: NDArray[np.int64]) -> dict[str, NDArray[np.int64]]:
        r"""Split leaf positions into dtype groups in first-seen order
        (`_prepare_bf_to_buffer_dicts`, bitbirch.py:1298-1308)."""
        lv = self._leaves()
        n = lv["n"][positions]
        code = np.where(n <= 255, 0, np.where(n <= 65535, 1, np.where(n <= 4294967295, 2, 3)))
        names = ["uint8", "uint16", "uint32", "uint64"]
        groups: dict[str, NDArray[np.int64]] = {}
        if positions.size:
            _, first = np.unique(code, return_index=True)
            for c in code[np.sort(first)]:
                groups[names[int(c)]] = positions[code == c]
        return groups

    def _bf_tables(
        self, positions: NDArray[np.int64], device: bool = False
    ) -> tuple[dict[str, tp.Any], dict[str, _IndexLists]]:
        r"""Array form of `_bf_to_np`: per dtype group a (k, F+1) buffer table gathered
        on the device and the member lists as CSR.  `device`: the tables stay in HBM (`DevTable`)
        when the engine can do that (multiround hands them to the next round / the exchange)."""
        lv = self._leaves()
        bufs: dict[str, tp.Any] = {}
        mols: dict[str, _IndexLists] = {}
        dev_ok = device and getattr(self._engine, "device_tables", False)
        for name, pos in self._group_positions(positions).items():
            if dev_ok:
                n_tail = 0
                if name == "uint8":  # the run of one-fingerprint BitFeatures the table ends with stays packed (DevTable.tail)
                    ones = lv["n"][pos] == 1
                    n_tail = int(ones.size if ones.all() else np.argmax(~ones[::-1]))
                bufs[name] = self._engine.gather_buffers(pos, np.dtype(name).itemsize, device_out=True, n_tail=n_tail)
            else:
                bufs[name] = self._engine.gather_buffers(pos, np.dtype(name).itemsize)
            beg, end = lv["beg"][pos], lv["end"][pos]
            cnt = end - beg
            total = int(cnt.sum())
            dst = np.cumsum(cnt) - cnt
            gather = np.repeat(beg - dst, cnt) + np.arange(total, dtype=np.int64)
            mols[name] = _IndexLists(cnt.astype(np.int64), lv["members"][gather])
        return bufs, mols

    def _refine_tables(
        self,
        X: tp.Any,
        initial_mol: int = 0,
        input_is_packed: bool = True,
        n_largest: int = 1,
        device: bool = False,
    ) -> tuple[dict[str, NDArray[np.integer]], dict[str, _IndexLists]]:
        r"""`_bf_to_np_refine` (bitbirch.py:1224-1290) in array form: the `n_largest`
        biggest leaves are exploded into singleton uint8 buffers rebuilt from the
        original fingerprints and appended to the uint8 group, after the survivors.
        `device`: the tables stay in HBM (`DevTable`; the exploded fingerprints join the uint8 table's packed
        singleton tail) when the engine can do that and the input is packed."""
        order = self._leaf_order(True)
        dev_ok = device and input_is_packed and getattr(self._engine, "device_tables", False)
        if n_largest == 0:
            return self._bf_tables(order, device=dev_ok)
        if n_largest < 1:
            raise ValueError("n_largest must be >= 1")
        largest, rest = order[:n_largest], order[n_largest:]
        bufs, mols = self._bf_tables(rest, device=dev_ok)
        F = self._n_features
        big_rows: list[NDArray[np.uint8]] = []
        big_ids: list[NDArray[np.int64]] = []
        for members in self._members_of(largest):
            mol_idxs = np.asarray(members, dtype=np.int64)
            arr_idxs = mol_idxs - initial_mol
            if isinstance(X, (Path, str)):
                fps = np.load(X, mmap_mode="r")[arr_idxs]
            elif isinstance(X, (list, tuple)) and len(X) and isinstance(X[0], (Path, str)):
                # a sequence of files: members are re-read in ascending global index
                srt = np.argsort(arr_idxs, kind="stable")
                fps = _rows_from_file_seq([Path(p) for p in X], arr_idxs[srt])
                mol_idxs = mol_idxs[srt]
            elif isinstance(X, list):
                fps = np.stack([np.asarray(X[i]) for i in arr_idxs])
            elif hasattr(X, "data_ptr") and getattr(X, "is_cuda", False):  # device-resident shard
                import torch

                fps = X[torch.from_numpy(arr_idxs).to(X.device)]
                if not dev_ok:
                    fps = fps.cpu().numpy()
            else:
                fps = np.asarray(X)[arr_idxs]
            if dev_ok:
                # packed rows: they go into the table's singleton tail as they are - which they must be: uint8, F / 8 columns
                # (the host path normalises through unpack_fingerprints; here a wrong width would only surface as an engine
                # error about row bytes much later)
                if not hasattr(fps, "data_ptr"):
                    fps = np.ascontiguousarray(fps)
                    if fps.dtype != np.uint8:
                        fps = fps.astype(np.uint8)
                if fps.ndim != 2 or int(fps.shape[1]) != F // 8 or (hasattr(fps, "data_ptr") and str(fps.dtype) != "torch.uint8"):
                    raise ValueError(f"packed fingerprints of {F} features must be uint8 rows of {F // 8} bytes, got {tuple(fps.shape)} {fps.dtype}")
                big_rows.append(fps)
                big_ids.append(mol_idxs)
                continue
            fps = np.asarray(fps)
            if input_is_packed:
                fps = unpack_fingerprints(fps.astype(np.uint8, copy=False), F)
            big_rows.append(fps.astype(np.uint8))
            big_ids.append(mol_idxs)
        if big_rows and dev_ok:
            import torch

            from bblean_amd._engine import DevTable

            tdev = torch.device("cuda", self._engine.device)
            parts = [r if hasattr(r, "data_ptr") else torch.from_numpy(np.ascontiguousarray(r, dtype=np.uint8)).to(tdev) for r in big_rows]
            ids = np.concatenate(big_ids)
            ones = np.ones(ids.size, dtype=np.int64)
            if "uint8" in bufs:
                tab = bufs["uint8"]
                if tab.tail is not None:
                    parts = [tab.tail] + parts
                bufs["uint8"] = DevTable(tab.raw, 1, torch.cat(parts) if len(parts) > 1 else parts[0])
                old = mols["uint8"]
                mols["uint8"] = _IndexLists(np.concatenate([old.counts, ones]), np.concatenate([old.flat, ids]))
            else:
                empty = torch.empty((0, F + 1), dtype=torch.uint8, device=tdev)
                bufs["uint8"] = DevTable(empty, 1, torch.cat(parts) if len(parts) > 1 else parts[0])
                mols["uint8"] = _IndexLists(ones, ids)
        elif big_rows:
            rows = np.concatenate(big_rows)
            extra = np.empty((rows.shape[0], F + 1), dtype=np.uint8)
            extra[:, :-1] = rows
            extra[:, -1] = 1
            ids = np.concatenate(big_ids)
            ones = np.ones(ids.size, dtype=np.int64)
            if "uint8" in bufs:
                bufs["uint8"] = np.concatenate([bufs["uint8"], extra])
                old = mols["uint8"]
                mols["uint8"] = _IndexLists(
                    np.concatenate([old.counts, ones]), np.concatenate([old.flat, ids])
                )
            else:
                bufs["uint8"] = extra
                mols["uint8"] = _IndexLists(ones, ids)
        return bufs, mols

    def _bf_to_np(self):  # type: ignore[no-untyped-def]
        r"""dict dtype-name -> list of buffers, dict dtype-name -> list of index lists
        (reference bitbirch.py:1292-1296)."""
        bufs, mols = self._bf_tables(self._leaf_order(True))
        return {k: list(v) for k, v in bufs.items()}, {k: v.to_lists() for k, v in mols.items()}

    def _bf_to_np_refine(self, X, initial_mol=0, input_is_packed=True, n_largest=1):  # type: ignore[no-untyped-def]
        bufs, mols = self._refine_tables(X, initial_mol, input_is_packed, n_largest)
        return {k: list(v) for k, v in bufs.items()}, {k: v.to_lists() for k, v in mols.items()}

    def refine_inplace(
        self,
        X: tp.Any,
        initial_mol: int = 0,
        input_is_packed: bool = True,
        n_largest: int = 1,
    ) -> "BitBirch":
        r"""Break the largest cluster(s) into singletons and re-fit every BitFeature
        (reference bitbirch.py:1187-1214)."""
        self._require_init()
        self.delete_internal_nodes()
        # (the tables stay in HBM when the engine can do that: gathered there, re-inserted from there - next to the old tree's
        # pools, which reset() keeps; when HBM does not hold both, the tables take the reference's way through host memory)
        try:
            bufs, mols = self._refine_tables(X, initial_mol, input_is_packed, n_largest, device=True)
        except Exception as exc:
            if not _is_out_of_memory(exc):
                raise
            bufs, mols = self._refine_tables(X, initial_mol, input_is_packed, n_largest, device=False)
        self.reset()
        for name in bufs:
            self._fit_buffers(bufs[name], reinsert_index_seqs=mols[name])
        return self

    def recluster_inplace(
        self,
        iterations: int = 1,
        extra_threshold: float = 0.0,
        shuffle: bool = False,
        seed: int | None = None,
        verbose: bool = False,
        stop_early: bool = False,
    ) -> "BitBirch":
        r"""Re-insert all leaf BitFeatures, optionally raising the threshold
        (reference bitbirch.py:1110-1185)."""
        self._require_init()
        singletons_before = 0
        for _ in range(iterations):
            order = self._leaf_order(True)
            n = self._leaves()["n"][order]
            singletons = int((n == 1).sum())
            if stop_early and (singletons == 0 or singletons == singletons_before):
                break
            singletons_before = singletons
            if verbose:
                print(f"Current number of clusters: {order.size}")
                print(f"Current number of singletons: {singletons}")
            if shuffle:
                perm = list(range(order.size))
                random.seed(seed)
                random.shuffle(perm)
                order = order[np.asarray(perm, dtype=np.int64)]
            try:
                bufs, mols = self._bf_tables(order, device=True)
            except Exception as exc:
                if not _is_out_of_memory(exc):
                    raise
                bufs, mols = self._bf_tables(order, device=False)
            self.reset()
            self.threshold += extra_threshold
            for name in bufs:
                self._fit_buffers(bufs[name], reinsert_index_seqs=mols[name])
        if verbose:
            n = self._leaves()["n"]
            print(f"Final number of clusters: {n.size}")
            print(f"Final number of singletons: {int((n == 1).sum())}")
        return self

    def global_clustering(self, n_clusters: int, *, method: str = "kmeans", **method_kwargs: tp.Any) -> "BitBirch":
        r""":meta private: experimental in the reference (bitbirch.py:1355-1434)."""
        warnings.warn(
            "Global clustering is an experimental features"
            " it will be modified without warning, please do not use"
        )
        self._require_init()
        if method not in {"agglomerative", "kmeans", "kmeans-normalized"}:
            raise ValueError(f"Unknown method {method}")
        from sklearn.cluster import AgglomerativeClustering, KMeans
        from sklearn.exceptions import ConvergenceWarning

        centrals = np.vstack(self.get_centroids(packed=False))
        num = len(centrals)
        k = n_clusters
        if num < k:
            warnings.warn(
                f"Number of subclusters found ({num}) by BitBIRCH is less "
                "than ({n_clusters}). Decrease k or the threshold.",
                ConvergenceWarning,
                stacklevel=2,
            )
            k = num
        data = centrals
        if method == "kmeans-normalized":
            data = centrals / np.linalg.norm(centrals, axis=1, keepdims=True)
        if method in ("kmeans", "kmeans-normalized"):
            predictor = KMeans(n_clusters=k, **method_kwargs)
        else:
            predictor = AgglomerativeClustering(n_clusters=k, **method_kwargs)
        self._global_clustering_centroid_labels = predictor.fit_predict(data) + 1
        self._n_global_clusters = n_clusters if num > n_clusters else num
        return self

    def save(self, path: Path | str) -> None:
        raise NotImplementedError(
            "whole-tree pickling (reference bitbirch.py:1321-1353) is not provided; "
            "checkpoint with _bf_to_np() tables instead"
        )

    def __repr__(self) -> str:
        fn = self._merge_accept_fn
        parts = [
            f"threshold={self.threshold}",
            f"branching_factor={self.branching_factor}",
            f"merge_criterion='{fn.name if fn.name in BUILTIN_MERGES else fn}'",
        ]
        if self.tolerance is not None:
            parts.append(f"tolerance={self.tolerance}")
        return f"{self.__class__.__name__}({', '.join(parts)})"


def _is_out_of_memory(exc: BaseException) -> bool:
    r"""A device allocation that failed: torch's OutOfMemoryError, the engine's MemoryError (BBH_ERR_CAPACITY) or a HIP error
    that says so."""
    if isinstance(exc, MemoryError):
        return True
    name = type(exc).__name__
    return "OutOfMemory" in name or "out of memory" in str(exc).lower()


def fit_concurrently(
    trees: tp.Sequence[BitBirch],
    inputs: tp.Sequence[tp.Any],
    reinsert_indices: tp.Sequence[tp.Iterable[int] | None] | None = None,
    input_is_packed: bool = True,
    n_features: int | None = None,
    max_fps: int | None = None,
) -> None:
    r"""`tree.fit(X)` for several independent trees at once - the shards of multiround's first
    round (reference multiround.py:401-422 runs them in a process pool).  On the HIP engine all
    trees insert in ONE kernel launch, one workgroup per tree, concurrently on different compute
    units; results are identical to calling `fit` on each tree in turn."""
    if len(trees) != len(inputs):
        raise ValueError("need one input per tree")
    prepared = []
    for i, (t, X) in enumerate(zip(trees, inputs)):
        ri = None if reinsert_indices is None else reinsert_indices[i]
        prepared.append(t._prepare_fit(X, ri, input_is_packed, n_features, max_fps))
    engines = [t._engine for t in trees]
    many = getattr(type(engines[0]), "fit_packed_many", None) if engines else None
    # a single tree takes the single-tree entry point: host / file-backed rows are then streamed through
    # two HBM slabs instead of being staged whole
    if len(engines) > 1 and many is not None and all(type(e) is type(engines[0]) for e in engines):
        leaves = many(engines, [rows for rows, _ in prepared])
    else:
        leaves = [e.fit_packed(rows) for e, (rows, _) in zip(engines, prepared)]
    for t, leaf, (_, ids) in zip(trees, leaves, prepared):
        if len(ids):
            t._commit_fit(leaf, ids)


def fit_buffers_concurrently(
    trees: tp.Sequence[BitBirch],
    tables: tp.Sequence[tp.Sequence[tuple[tp.Any, tp.Any]]],
) -> None:
    r"""`tree._fit_buffers(bufs, idx)` for every ``(bufs, idx)`` of ``tables[i]``, in order, for
    several independent trees at once: the trees of one multiround merge round (reference
    multiround.py:240-264, one process per tree there).  Step s inserts the s-th table of every
    tree that has one in ONE kernel launch (one workgroup per tree); results are identical to
    looping over the trees."""
    if len(trees) != len(tables):
        raise ValueError("need one table list per tree")
    steps = max((len(t) for t in tables), default=0)
    for s in range(steps):
        live, prepared = [], []
        for tree, tabs in zip(trees, tables):
            if s < len(tabs):
                prep = tree._prepare_fit_buffers(tabs[s][0], tabs[s][1])
                if prep is not None:
                    live.append(tree)
                    prepared.append(prep)
        if not live:
            continue
        engines = [t._engine for t in live]
        many = getattr(type(engines[0]), "fit_buffers_many", None)
        if len(engines) > 1 and many is not None and all(type(e) is type(engines[0]) for e in engines):
            leaves = many(engines, [p[0] for p in prepared])
        else:  # a single tree takes the single-tree entry point (streamed input, singleton fast path)
            leaves = [e.fit_buffers(p[0]) for e, p in zip(engines, prepared)]
        for tree, leaf, (_, counts, flat) in zip(live, leaves, prepared):
            tree._commit_fit_buffers(leaf, counts, flat)


def _rows_from_file_seq(files: tp.Sequence[Path], idxs: NDArray[np.int64]) -> NDArray[np.uint8]:
    r"""Rows `idxs` (ascending, global numbering over the concatenated files) read with
    memory-mapped loads (the job of `_get_fingerprints_from_file_seq`,
    reference fingerprints.py:273-321)."""
    if np.any(np.diff(idxs) < 0):
        raise ValueError("idxs must be sorted")
    out: list[NDArray[np.uint8]] = []
    base = 0
    taken = 0
    for f in files:
        arr = np.load(f, mmap_mode="r")
        n = arr.shape[0]
        hi = int(np.searchsorted(idxs, base + n, side="left"))
        if hi > taken:
            out.append(np.asarray(arr[idxs[taken:hi] - base]).astype(np.uint8, copy=False))
            taken = hi
        base += n
    if taken != idxs.size:
        raise ValueError("idxs could not be extracted from files")
    return np.concatenate(out) if out else np.zeros((0, 0), dtype=np.uint8)
