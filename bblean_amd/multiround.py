r"""Multi-round BitBIRCH (reference: bblean/multiround.py) on the MI355X engine.

Two front ends over the same round logic:

* `run_multiround_bitbirch` - same signature, file names and file formats as the reference
  (`multiround.py:333-484`): shard s = input file s, round 1 fits every shard (+ optional
  refinement), later rounds re-insert the leaf BitFeature tables batch by batch, exchanging
  `round-{r}-bufs.label-{L}-uint{08,16,..}.npy` / `round-{r}-idxs....pkl` files
  (`multiround.py:132-143`).  One process, one GPU: the shards of round 1 and the batches of a merge
  round run CONCURRENTLY on the device, one workgroup per tree in shared kernel launches
  (`fit_concurrently` / `fit_buffers_concurrently`), where the reference uses a process pool.

* `run_multiround_distributed` - one process per GPU (`torch.distributed`, backend "nccl" =
  RCCL over xGMI; "gloo" in the CPU tests).  Round 1 is embarrassingly parallel (rank r owns
  shards r, r+W, ...).  Between rounds the leaf BitFeature tables are exchanged with a
  variable-size all-gather instead of files; the merge rounds keep the reference's batch
  composition and order (name-sorted pairs chunked by `bin_size`, uint16 tables before uint8
  inside a mid-round batch, plain name order in the final round), so the result is identical
  to the file-based run and to the reference for any number of ranks.
"""
from __future__ import annotations

import itertools
import os
import math
import pickle
import time
import typing as tp
from pathlib import Path

import numpy as np
from numpy.typing import NDArray

from bblean_amd.bitbirch import BitBirch, _IndexLists, fit_buffers_concurrently, fit_concurrently
from bblean_amd.utils import batched

__all__ = ["run_multiround_bitbirch", "run_multiround_distributed"]

# reference defaults (bblean/_config.py:23-33)
_DEF = dict(threshold=0.30, branching_factor=254, merge_criterion="diameter",
            refine_merge_criterion="tolerance-diameter", refine_threshold_change=0.0, tolerance=0.05)


class _Timer:
    def __init__(self) -> None:
        self.timings: dict[str, float] = {}
        self._t0: dict[str, float] = {}

    def init_timing(self, label: str) -> None:
        self._t0[label] = time.perf_counter()

    def end_timing(self, label: str) -> None:
        self.timings[label] = time.perf_counter() - self._t0.pop(label)


def _suffix(label: str, dtype_name: str) -> str:
    return f".label-{label}-{dtype_name.replace('8', '08')}"  # multiround.py:139


def _bits_of(name: str) -> int:
    return int(name.split("uint")[-1].split(".")[0])  # multiround.py:108


def _file_rows(path: tp.Any) -> int:
    if isinstance(path, ShardRows):
        return path.n
    if not isinstance(path, (Path, str)):
        return int(path.shape[0])  # an array / device tensor holding the shard
    with open(path, "rb") as f:
        major, minor = np.lib.format.read_magic(f)
        shape, _, _ = getattr(np.lib.format, f"read_array_header_{major}_{minor}")(f)
    return shape[0]


class ShardRows:
    r"""Placeholder for a shard another rank owns: only its row count is needed here (global molecule
    index ranges, reference multiround.py:316-327)."""

    def __init__(self, n: int) -> None:
        self.n = int(n)


def _files_range_tuples(files: tp.Sequence[tp.Any]) -> list[tuple[str, tp.Any, int, int]]:
    r"""(label, file, first global index, end) per shard (multiround.py:316-327)."""
    out, run = [], 0
    z = len(str(len(files)))
    for i, f in enumerate(files):
        n = _file_rows(f)
        out.append((str(i).zfill(z), Path(f) if isinstance(f, (Path, str)) else f, run, run + n))
        run += n
    return out


Tables = tuple[dict[str, NDArray[np.integer]], dict[str, _IndexLists]]


def _release_cached_hbm() -> None:
    r"""Between rounds: HBM that torch's allocator keeps cached after the tables of a round were freed goes back to the
    driver (the tree pools come from the library's own allocator, which cannot use it; at 100 M rows the two caches together
    decide whether the final tree fits next to the last round's tables)."""
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    except Exception:  # pragma: no cover - nothing to release without torch / a device
        pass


# host/device bytes of raw input handed to one concurrent launch group
_GROUP_BYTES = 24 << 30


def _groups(sizes: tp.Sequence[int], limit: int = _GROUP_BYTES, max_trees: int = 2048) -> list[range]:
    r"""Consecutive index ranges whose inputs stay under `limit` bytes (at least one per group)."""
    out, lo, acc = [], 0, 0
    for i, sz in enumerate(sizes):
        if i > lo and (acc + sz > limit or i - lo >= max_trees):
            out.append(range(lo, i))
            lo, acc = i, 0
        acc += sz
    if lo < len(sizes):
        out.append(range(lo, len(sizes)))
    return out


def _initial_rounds(
    infos: tp.Sequence[tuple[str, Path, int, int]], *, branching_factor: int, threshold: float, tolerance: float,
    merge_criterion: str, refinement: str, refine_merge_criterion: str, refine_threshold_change: float,
    n_features: int | None, input_is_packed: bool, max_fps: int | None, engine_factory: tp.Any, device: int,
    device_tables: bool = False,
) -> list[Tables]:
    r"""`_InitialRound.__call__` (multiround.py:175-216) for several shards at once, returning tables
    instead of files.  Every step that touches the device runs for all shards of a group in one
    launch; the per-shard results are those of the reference's one-process-per-shard pool."""
    out: list[Tables] = []
    sizes = [max(int(Path(f).stat().st_size) if isinstance(f, Path) else int(np.prod(f.shape)), 1) for _, f, _, _ in infos]
    for grp in _groups(sizes):
        part = [infos[i] for i in grp]
        trees = [BitBirch(branching_factor=branching_factor, threshold=threshold, merge_criterion=merge_criterion,
                          device=device, _engine_factory=engine_factory) for _ in part]
        fit_concurrently(trees, [f for _, f, _, _ in part], reinsert_indices=[range(s, e) for _, _, s, e in part],
                         input_is_packed=input_is_packed, n_features=n_features, max_fps=max_fps)
        for t in trees:
            t.delete_internal_nodes()
        if refinement == "none":
            out.extend(t._bf_tables(t._leaf_order(True), device=device_tables) for t in trees)
            continue
        tabs = [t._refine_tables(f, initial_mol=s, input_is_packed=input_is_packed, device=device_tables) for t, (_, f, s, _) in zip(trees, part)]
        if refinement == "full":
            for t in trees:
                t.reset()
                t.set_merge(refine_merge_criterion, tolerance=tolerance, threshold=threshold + refine_threshold_change)
            fit_buffers_concurrently(trees, [[(bufs[name], mols[name]) for name in bufs] for bufs, mols in tabs])
            for t in trees:
                t.delete_internal_nodes()
            tabs = [t._bf_tables(t._leaf_order(True), device=device_tables) for t in trees]
        out.extend(tabs)
        del trees, tabs
        _release_cached_hbm()
    return out


def _merge_rounds(
    batches: tp.Sequence[tp.Sequence[tuple[NDArray[np.integer], _IndexLists]]], *, branching_factor: int, threshold: float,
    tolerance: float, criterion: str, engine_factory: tp.Any, device: int,
) -> list[BitBirch]:
    r"""`_TreeMergingRound.__call__` (multiround.py:240-264) for several batches at once: one tree per
    batch, rebuilt from the batch's BitFeature tables in the given order, all trees in shared launches."""
    trees = [BitBirch(branching_factor=branching_factor, threshold=threshold, merge_criterion=criterion,
                      tolerance=tolerance, device=device, _engine_factory=engine_factory) for _ in batches]
    sizes = [sum(int(b.nbytes) for b, _ in batch) + 1 for batch in batches]
    for grp in _groups(sizes):
        fit_buffers_concurrently([trees[i] for i in grp], [list(batches[i]) for i in grp])
    for t in trees:
        t.delete_internal_nodes()
    return trees


def _merge_one_tree_streaming(
    pairs: tp.Sequence[tuple[Path, Path]], *, branching_factor: int, threshold: float, tolerance: float, criterion: str,
    engine_factory: tp.Any, device: int,
) -> BitBirch:
    r"""`_FinalTreeMergingRound` (multiround.py:284-312): one tree, the round files of the previous round in name order.
    The next file pair is read (un-pickling the member lists is most of it) by a helper thread while the device
    inserts the current one - the C calls release the GIL."""
    from concurrent.futures import ThreadPoolExecutor

    tree = BitBirch(branching_factor=branching_factor, threshold=threshold, merge_criterion=criterion,
                    tolerance=tolerance, device=device, _engine_factory=engine_factory)
    if pairs:
        with ThreadPoolExecutor(max_workers=1) as ex:
            fut = ex.submit(_load_pair, *pairs[0])
            for i in range(len(pairs)):
                bufs, idx = fut.result()
                if i + 1 < len(pairs):
                    fut = ex.submit(_load_pair, *pairs[i + 1])
                tree._fit_buffers(bufs, idx)
    tree.delete_internal_nodes()
    return tree


def _save_tables(out_dir: Path, bufs: dict, mols: dict, label: str, round_idx: int) -> None:
    r"""`_save_bufs_and_mol_idxs` (multiround.py:132-143): same names, same bytes."""
    for name, table in bufs.items():
        suf = _suffix(label, name)
        np.save(out_dir / f"round-{round_idx}-bufs{suf}.npy", np.ascontiguousarray(table))
        with open(out_dir / f"round-{round_idx}-idxs{suf}.pkl", "wb") as f:
            pickle.dump(mols[name].to_lists(), f)


def _load_pair(buf_path: Path, idx_path: Path) -> tuple[NDArray[np.integer], _IndexLists]:
    with open(idx_path, "rb") as f:
        lists = pickle.load(f)
    bufs = np.load(buf_path, mmap_mode="r")
    counts = np.fromiter(map(len, lists), dtype=np.int64, count=len(lists))
    flat = np.fromiter(itertools.chain.from_iterable(lists), dtype=np.int64, count=int(counts.sum()))
    return bufs, _IndexLists(counts, flat)


def run_multiround_bitbirch(
    input_files: tp.Sequence[Path],
    out_dir: Path,
    n_features: int | None = None,
    input_is_packed: bool = True,
    num_initial_processes: int = 10,
    num_midsection_processes: int | None = None,
    initial_merge_criterion: str = _DEF["merge_criterion"],
    branching_factor: int = _DEF["branching_factor"],
    threshold: float = _DEF["threshold"],
    midsection_threshold_change: float = _DEF["refine_threshold_change"],
    tolerance: float = _DEF["tolerance"],
    num_midsection_rounds: int = 1,
    bin_size: int = 10,
    max_tasks_per_process: int = 1,
    refinement_before_midsection: str = "full",
    split_largest_after_each_midsection_round: bool = False,
    midsection_merge_criterion: str = _DEF["refine_merge_criterion"],
    final_merge_criterion: str | None = None,
    mp_context: tp.Any = None,
    save_tree: bool = False,
    save_centroids: bool = True,
    max_fps: int | None = None,
    verbose: bool = False,
    cleanup: bool = True,
    device: int = 0,
    _engine_factory: tp.Any = None,
    _round1_engine_factory: tp.Any = "same",
) -> _Timer:
    r"""File-compatible multiround on one GPU.  The process-count arguments are accepted for
    signature compatibility; the result never depended on them (tests/test_multiround.py of the
    reference asserts that); shards and batches share kernel launches on the device instead."""
    if refinement_before_midsection not in ("full", "split", "none"):
        raise ValueError(f"Unknown refinement kind {refinement_before_midsection}")
    if num_midsection_processes is not None and num_midsection_processes > num_initial_processes:
        raise ValueError("Num. midsection procs. must be <= num. initial processes")
    if final_merge_criterion is None:
        final_merge_criterion = midsection_merge_criterion
    out_dir = Path(out_dir)
    input_files = [Path(f) if isinstance(f, (Path, str)) else f for f in input_files]  # arrays / tensors: shards in memory
    common = dict(branching_factor=branching_factor, tolerance=tolerance, engine_factory=_engine_factory, device=device)
    timer = _Timer()
    timer.init_timing("total")

    round_idx = 1
    timer.init_timing(f"round-{round_idx}")
    infos = _files_range_tuples(input_files)
    # (test hook: round 1 on another engine than the merge rounds - the round-* files are the wire format between them)
    r1 = common if _round1_engine_factory == "same" else dict(common, engine_factory=_round1_engine_factory)
    for info, (bufs, mols) in zip(infos, _initial_rounds(
            infos, threshold=threshold, merge_criterion=initial_merge_criterion,
            refinement=refinement_before_midsection, refine_merge_criterion=midsection_merge_criterion,
            refine_threshold_change=midsection_threshold_change, n_features=n_features,
            input_is_packed=input_is_packed, max_fps=max_fps, **r1)):
        _save_tables(out_dir, bufs, mols, info[0], 1)
    timer.end_timing(f"round-{round_idx}")

    def prev_pairs(r: int) -> list[tuple[Path, Path]]:
        return list(zip(sorted(out_dir.glob(f"round-{r - 1}-bufs*.npy")),
                        sorted(out_dir.glob(f"round-{r - 1}-idxs*.pkl"))))

    for _ in range(num_midsection_rounds):
        round_idx += 1
        timer.init_timing(f"round-{round_idx}")
        pairs = prev_pairs(round_idx)
        z = len(str(math.ceil(len(pairs) / bin_size)))
        batches = [sorted(batch, key=lambda p: _bits_of(p[0].name), reverse=True)  # multiround.py:104-111
                   for batch in batched(pairs, bin_size)]
        trees = _merge_rounds([[_load_pair(*p) for p in batch] for batch in batches],
                              threshold=threshold + midsection_threshold_change,
                              criterion=midsection_merge_criterion, **common)
        for i, tree in enumerate(trees):
            if split_largest_after_each_midsection_round:
                bufs, mols = tree._refine_tables(input_files)
            else:
                bufs, mols = tree._bf_tables(tree._leaf_order(True))
            _save_tables(out_dir, bufs, mols, str(i).zfill(z), round_idx)
        timer.end_timing(f"round-{round_idx}")

    round_idx += 1
    timer.init_timing(f"round-{round_idx}")
    tree = _merge_one_tree_streaming(prev_pairs(round_idx), threshold=threshold + midsection_threshold_change,
                                     criterion=final_merge_criterion, **common)
    if save_tree:
        raise NotImplementedError("whole-tree pickling is not provided (the reference's --save-tree is broken too)")
    _write_outputs(out_dir, tree, save_centroids)
    timer.end_timing(f"round-{round_idx}")
    if cleanup:
        for f in list(out_dir.glob("round-*.npy")) + list(out_dir.glob("round-*.pkl")):
            f.unlink()
    timer.end_timing("total")
    return timer


class _RowsAsList:
    r"""Pickles a 2-D array as ONE block that un-pickles (with nothing but NumPy installed) to ``list(array)``: the list
    of per-cluster centroid arrays the reference writes (multiround.py:308-309), without 360 k separate array reductions
    (1.4 s per million-row run; the block takes 30 ms)."""

    def __init__(self, rows: NDArray[np.uint8]) -> None:
        self.rows = rows

    def __reduce__(self) -> tuple[tp.Any, tuple[tp.Any]]:
        return (list, (np.ascontiguousarray(self.rows),))


def _write_outputs(out_dir: Path, tree: BitBirch, save_centroids: bool) -> None:
    r"""clusters.pkl / cluster-centroids-packed.pkl (multiround.py:303-312)."""
    if save_centroids:
        order = tree._leaf_order(True)
        with open(out_dir / "clusters.pkl", "wb") as f:
            pickle.dump(tree._members_of(order), f)
        with open(out_dir / "cluster-centroids-packed.pkl", "wb") as f:
            pickle.dump(_RowsAsList(tree._leaves()["cents"][order]), f)
    else:
        with open(out_dir / "clusters.pkl", "wb") as f:
            pickle.dump(tree.get_cluster_mol_ids(), f)


# ------------------------------------------------------------------------------------------
# one process per GPU
# ------------------------------------------------------------------------------------------
_CODE = {"uint8": 8, "uint16": 16, "uint32": 32, "uint64": 64}
_NAME = {v: k for k, v in _CODE.items()}

# one table on its way between rounds: (label, dtype name, table [k, F+1] (host array or DevTable), members)
Entry = tuple[str, str, tp.Any, _IndexLists]


def _entry_key(label: str, name: str) -> str:
    # the order sorted(glob("round-*-bufs*.npy")) gives: by file name = (label, uintNN)
    return f"label-{label}-{name.replace('8', '08')}"


def _tail_row_bytes(cols: int) -> int:
    r"""Bytes of one packed singleton row of a table of `cols` = n_features + 1 columns: what the sender ships
    (engine.nbytes = ceil(n_features / 8)) and what every receive buffer is sized with - ONE place for both."""
    return (cols - 1 + 7) // 8


class _Exchange:
    r"""Moves BitFeature tables between ranks for the next round: every table goes to the ONE rank that
    merges it (point-to-point, `batch_isend_irecv`), never to the others.

    1. descriptors: one fixed-size int64 all-gather (label, dtype bits, rows, columns, member ids) -
       no pickles;
    2. every rank derives the same global plan from them: entries in the reference's file order
       (`sorted(glob("round-*"))`, multiround.py:91-101), chunked into batches of `bin_size`, batch b
       merged by rank `owner(b)`;
    3. payloads: the table bytes, the member counts and the member ids (CSR) as three tensors per
       entry, sent straight from where they lie (HBM for `DevTable`s - with backend "nccl" that is
       RCCL over xGMI, GPU to GPU) and received into the buffer `fit_buffers` will read.
    """

    def __init__(self, dist: tp.Any, tdev: tp.Any, table_dev: tp.Any = None) -> None:
        # tdev: where the wire tensors live (cuda for "nccl", cpu for "gloo"); table_dev: the GPU received
        # tables are handed to the engine on (None: host arrays, the CPU oracle of the tests)
        self.dist, self.tdev, self.table_dev = dist, tdev, table_dev
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.bytes_sent = 0
        self.bytes_received = 0

    def _wire(self, a: tp.Any) -> tp.Any:
        import torch

        t = a.reshape(-1) if hasattr(a, "data_ptr") else torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1))
        return t if t.device == self.tdev else t.to(self.tdev)

    def _wire_table(self, table: tp.Any) -> list[tp.Any]:
        r"""The tensors a table travels as: its rows, and - a `DevTable` with a packed singleton tail - the tail."""
        if hasattr(table, "raw"):  # DevTable
            return [self._wire(table.raw)] + ([self._wire(table.tail)] if table.tail is not None else [])
        return [self._wire(table)]

    def _plan(self, mine: list[Entry], bin_size: int | None) -> list[list[tuple]]:
        r"""Steps 1 and 2: the descriptor all-gather and the batches every rank derives from it.  A plan entry is
        (key, src rank, src position, label, name, rows, columns, member ids, rows of the packed singleton tail)."""
        import torch

        dist, world = self.dist, self.world
        cnt = torch.tensor([len(mine)], dtype=torch.int64, device=self.tdev)
        cnts = [torch.zeros(1, dtype=torch.int64, device=self.tdev) for _ in range(world)]
        dist.all_gather(cnts, cnt)
        counts = [int(c.item()) for c in cnts]
        cap = max(max(counts), 1)
        desc = torch.zeros((cap, 7), dtype=torch.int64)
        for i, (lab, name, table, idx) in enumerate(mine):
            if not lab.isdigit():
                raise ValueError(f"table labels must be numeric, got {lab!r}")
            desc[i] = torch.tensor([int(lab), len(lab), _CODE[name], int(table.shape[0]), int(table.shape[1]),
                                    int(idx.flat.size), int(getattr(table, "n_tail", 0))], dtype=torch.int64)
        desc = desc.to(self.tdev)
        all_desc = [torch.zeros((cap, 7), dtype=torch.int64, device=self.tdev) for _ in range(world)]
        dist.all_gather(all_desc, desc)
        plan = []
        for r in range(world):
            rows = all_desc[r].cpu().numpy()
            for i in range(counts[r]):
                lab_i, lab_w, bits, k, cols, nids, ntail = (int(v) for v in rows[i])
                lab, name = str(lab_i).zfill(lab_w), _NAME[bits]
                plan.append((_entry_key(lab, name), r, i, lab, name, k, cols, nids, ntail))
        plan.sort(key=lambda e: e[0])
        return [plan] if bin_size is None else [list(c) for c in batched(plan, bin_size)]

    def run_streaming(self, mine: list[Entry], bin_size: int | None, owner: tp.Callable[[int], int], budget_bytes: int,
                      widest_first: bool, make_tree: tp.Callable[[], BitBirch]) -> tuple[list[tuple[int, BitBirch]], int]:
        r"""The exchange and the merge in one, with bounded memory on the merging rank (the reference's final round
        reads ONE ``(bufs, idxs)`` pair at a time from disk, multiround.py:284-312; holding every table of a batch next
        to the tree built from them does not fit 288 GB at 100 M rows that hardly merge).  Every batch's tables are cut,
        in insertion order, into chunks of rows of at most `budget_bytes` and packed into slabs of at most that size;
        slab s + 1 of every batch is on the wire (one grouped `batch_isend_irecv` per step, the same sequence of steps on
        all ranks) while slab s is inserted, so the merging rank holds two slabs, its trees and - on the host - the member lists of the tables whose
        chunks are still arriving (8 bytes per molecule, they travel with a table's first chunk and are dropped with its last).  `widest_first`: the merge rounds' order
        inside a batch, uint16 tables before uint8 (multiround.py:104-111).  Returns ([(batch, tree) for the batches
        this rank merges], number of batches); same trees as `run` + `_merge_rounds`, chunk for chunk the same stream."""
        import torch

        from bblean_amd._engine import DevTable

        dist, rank = self.dist, self.rank
        batches = self._plan(mine, bin_size)
        if widest_first:
            batches = [sorted(b, key=lambda e: _CODE[e[4]], reverse=True) for b in batches]
        # slabs[b] = [[(plan entry, first row, end row, is first chunk)], ...]
        slabs: list[list[list[tuple]]] = []
        for chunk in batches:
            cur: list[tuple] = []
            cur_bytes = 0
            out: list[list[tuple]] = []
            for ent in chunk:
                _, _, _, _, name, k, cols, _, ntail = ent
                kh = k - ntail  # rows [kh, k): the packed singleton tail (a chunk lies on one side of kh)
                a = 0
                while a < k or (k == 0 and a == 0):
                    row_bytes = cols * np.dtype(name).itemsize if a < kh or k == 0 else _tail_row_bytes(cols)
                    per = max(1, budget_bytes // max(row_bytes, 1))
                    b_ = min(kh if a < kh else k, a + per)
                    nb = (b_ - a) * row_bytes
                    if cur and cur_bytes + nb > budget_bytes:
                        out.append(cur)
                        cur, cur_bytes = [], 0
                    cur.append((ent, a, b_, a == 0))
                    cur_bytes += nb
                    a = b_
                    if k == 0:
                        break
            if cur:
                out.append(cur)
            slabs.append(out)
        owned = [b for b in range(len(batches)) if owner(b) == rank]
        trees = {b: make_tree() for b in owned}
        members: dict[tuple[int, str], _IndexLists] = {}   # (batch, key) -> the table's member lists, once its first chunk is here
        offsets: dict[tuple[int, str], NDArray[np.int64]] = {}
        n_steps = max((len(x) for x in slabs), default=0)
        cols_of: dict[tuple[int, int], int] = {}  # received chunks of a packed singleton tail -> columns of their table

        def post(step: int) -> tuple[list, dict, list]:
            ops, got, keep = [], {}, []  # keep: send buffers, alive until the step's requests are done
            cols_of.clear()
            for b, sl in enumerate(slabs):
                if step >= len(sl):
                    continue
                dst = owner(b)
                for pos, (ent, a, b_, first) in enumerate(sl[step]):
                    key, src, i, lab, name, k, cols, nids, ntail = ent
                    item = np.dtype(name).itemsize
                    in_tail = ntail > 0 and a >= k - ntail
                    if src == rank and dst == rank:
                        _, _, table, idx = mine[i]
                        part = table.rows(a, b_) if hasattr(table, "raw") else table[a:b_]
                        got[(b, pos)] = (key, name, part, idx if first else None, a, b_)
                    elif src == rank:
                        _, _, table, idx = mine[i]
                        parts = self._wire_table(table.rows(a, b_) if hasattr(table, "raw") else np.ascontiguousarray(table[a:b_]))
                        if first:
                            parts += [self._wire(idx.counts.astype(np.int64)), self._wire(idx.flat.astype(np.int64))]
                        for t in parts:
                            if t.numel():
                                ops.append(dist.P2POp(dist.isend, t, dst))
                                self.bytes_sent += int(t.numel())
                        keep.append(parts)
                    elif dst == rank:
                        tb = torch.empty((b_ - a, _tail_row_bytes(cols) if in_tail else cols * item), dtype=torch.uint8, device=self.tdev)
                        bufs = [tb]
                        if first:
                            bufs += [torch.empty(k * 8, dtype=torch.uint8, device=self.tdev),
                                     torch.empty(nids * 8, dtype=torch.uint8, device=self.tdev)]
                        for t in bufs:
                            if t.numel():
                                ops.append(dist.P2POp(dist.irecv, t.reshape(-1), src))
                                self.bytes_received += int(t.numel())
                        got[(b, pos)] = (key, name, bufs, None, a, b_)
                        cols_of[(b, pos)] = cols if in_tail else 0
            reqs = dist.batch_isend_irecv(ops) if ops else []
            return reqs, got, (keep, dict(cols_of))

        def finish(reqs: list, got: dict, keep_cols: tuple) -> dict[int, list[tuple[tp.Any, _IndexLists]]]:
            keep, recv_cols = keep_cols
            for req in reqs:
                req.wait()
            del keep
            if reqs and self.tdev.type == "cuda":
                torch.cuda.synchronize()
            per_batch: dict[int, list[tuple[int, tuple[tp.Any, _IndexLists]]]] = {}
            for (b, pos), val in got.items():
                key, name, payload, idx_full, a, b_ = val
                if isinstance(payload, list):  # received
                    tb = payload[0]
                    if len(payload) == 3:
                        idx_full = _IndexLists(payload[1].cpu().numpy().view(np.int64).copy(), payload[2].cpu().numpy().view(np.int64).copy())
                    if self.table_dev is not None:
                        tb = tb if tb.device == self.table_dev else tb.to(self.table_dev)
                        if recv_cols.get((b, pos), 0):  # a chunk of a packed singleton tail
                            part = DevTable(torch.empty((0, recv_cols[(b, pos)]), dtype=torch.uint8, device=tb.device), 1, tb)
                        else:
                            part = DevTable(tb, np.dtype(name).itemsize)
                    else:
                        assert not recv_cols.get((b, pos), 0), "a chunk of a packed singleton tail reached a rank that keeps its tables on the host"
                        part = tb.numpy().view(np.dtype(name)).reshape(b_ - a, -1)
                else:
                    part = payload
                if idx_full is not None:
                    members[(b, key)] = idx_full
                    offsets[(b, key)] = np.concatenate(([0], np.cumsum(idx_full.counts)))
                full, off = members[(b, key)], offsets[(b, key)]
                sub = _IndexLists(full.counts[a:b_], full.flat[int(off[a]):int(off[b_])])
                if b_ >= len(full.counts):  # the table's last chunk: its member lists are not needed any more
                    del members[(b, key)], offsets[(b, key)]
                per_batch.setdefault(b, []).append((pos, (part, sub)))
            return {b: [e for _, e in sorted(v, key=lambda pe: pe[0])] for b, v in per_batch.items()}

        pending = post(0) if n_steps else ([], {}, ([], {}))
        for step in range(n_steps):
            ready = finish(*pending)
            pending = post(step + 1) if step + 1 < n_steps else ([], {}, ([], {}))  # the next slab travels while this one is inserted
            live = [b for b in owned if b in ready]
            if live:
                fit_buffers_concurrently([trees[b] for b in live], [ready[b] for b in live])
            del ready
        for t in trees.values():
            t.delete_internal_nodes()
        return [(b, trees[b]) for b in owned], len(batches)

    def run(self, mine: list[Entry], bin_size: int | None, owner: tp.Callable[[int], int],
            ) -> tuple[list[tuple[int, list[Entry]]], int]:
        r"""Returns ([(batch index, entries of that batch in file order) for the batches this rank
        merges], total number of batches).  `bin_size=None`: one batch with everything."""
        import torch

        from bblean_amd._engine import DevTable

        dist, rank = self.dist, self.rank
        on_dev = self.tdev.type == "cuda"
        chunks = self._plan(mine, bin_size)  # 1. descriptors, 2. batches and their owners
        # 3. payloads
        ops, recv_into = [], {}
        keep_alive: list = []  # send buffers, until the requests are done
        mine_out: dict[int, list[tp.Any]] = {}
        for b, chunk in enumerate(chunks):
            dst = owner(b)
            for pos, (_, src, i, lab, name, k, cols, nids, ntail) in enumerate(chunk):
                item = np.dtype(name).itemsize
                if src == rank and dst == rank:
                    mine_out.setdefault(b, []).append((pos, mine[i]))
                elif src == rank:
                    _, _, table, idx = mine[i]
                    parts = self._wire_table(table) + [self._wire(idx.counts.astype(np.int64)), self._wire(idx.flat.astype(np.int64))]
                    keep_alive.append(parts)
                    for t in parts:
                        if t.numel():
                            ops.append(dist.P2POp(dist.isend, t, dst))
                            self.bytes_sent += int(t.numel())
                elif dst == rank:
                    tb = torch.empty((k - ntail, cols * item), dtype=torch.uint8, device=self.tdev)
                    tt = torch.empty((ntail, _tail_row_bytes(cols)), dtype=torch.uint8, device=self.tdev)  # the packed singleton tail
                    cb = torch.empty(k * 8, dtype=torch.uint8, device=self.tdev)
                    ib = torch.empty(nids * 8, dtype=torch.uint8, device=self.tdev)
                    for t in (tb, tt, cb, ib):
                        if t.numel():
                            ops.append(dist.P2POp(dist.irecv, t.reshape(-1), src))
                            self.bytes_received += int(t.numel())
                    recv_into[(b, pos)] = (lab, name, (tb, tt), cb, ib, k, cols)
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            if on_dev:
                torch.cuda.synchronize()
        del keep_alive
        for (b, pos), (lab, name, (tb, tt), cb, ib, k, cols) in recv_into.items():
            cnt_np = cb.cpu().numpy().view(np.int64).copy()
            ids_np = ib.cpu().numpy().view(np.int64).copy()
            table: tp.Any
            if self.table_dev is not None:  # stays a tensor end to end (gloo cannot carry device tensors: staged)
                table = DevTable(tb if tb.device == self.table_dev else tb.to(self.table_dev), np.dtype(name).itemsize,
                                 tt if tt.device == self.table_dev else tt.to(self.table_dev))
            else:
                # (a host table is the reference's full-width array: a sender only ships a packed tail from a device table, and
                # all ranks of a job share the choice)
                assert tt.numel() == 0, "a table with a packed singleton tail reached a rank that keeps its tables on the host"
                table = tb.numpy().view(np.dtype(name)).reshape(k, cols)
            mine_out.setdefault(b, []).append((pos, (lab, name, table, _IndexLists(cnt_np, ids_np))))
            recv_into[(b, pos)] = None  # (the staging tensors of the member lists are not kept next to the tables)
        out = [(b, [e for _, e in sorted(v, key=lambda pe: pe[0])]) for b, v in sorted(mine_out.items())]
        return out, len(chunks)


def run_multiround_distributed(
    input_files: tp.Sequence[tp.Any],
    out_dir: Path | None = None,
    *,
    n_features: int | None = None,
    input_is_packed: bool = True,
    initial_merge_criterion: str = _DEF["merge_criterion"],
    branching_factor: int = _DEF["branching_factor"],
    threshold: float = _DEF["threshold"],
    midsection_threshold_change: float = _DEF["refine_threshold_change"],
    tolerance: float = _DEF["tolerance"],
    num_midsection_rounds: int = 1,
    bin_size: int = 10,
    refinement_before_midsection: str = "full",
    midsection_merge_criterion: str = _DEF["refine_merge_criterion"],
    final_merge_criterion: str | None = None,
    save_centroids: bool = True,
    max_fps: int | None = None,
    device: int | None = None,
    return_tree: bool = False,
    recv_budget_mb: float | None = None,
    _engine_factory: tp.Any = None,
) -> tuple[tp.Any, _Timer]:
    r"""Multiround (reference multiround.py:333-484) with one rank per GPU (`torch.distributed` must be
    initialised; backend "nccl" = RCCL over xGMI, "gloo" in the CPU tests).

    `input_files[s]` is shard s: a ``.npy`` path, a packed array, a device-resident ``torch.uint8``
    tensor, or `ShardRows(n)` for a shard this rank does not own (rank r owns shards r, r+W, ...).
    Round 1 has no collective.  Between rounds the leaf BitFeature tables go, device to device, to the one
    rank that merges them (`_Exchange`); batch composition and order are the reference's (name-sorted
    tables chunked by `bin_size`, uint16 tables before uint8 inside a mid-round batch, plain name order
    in the final round), so the clusters equal the file-based run's and the reference's for any number of
    ranks.  Returns (clusters on rank 0 - or the final tree with `return_tree` - / None elsewhere,
    timings); writes clusters.pkl on rank 0 when `out_dir` is given.  `timer.exchange` holds the bytes
    this rank sent / received per round.

    `recv_budget_mb` (default: the environment variable ``BBHIP_RECV_BUDGET_MB``, unset = unbounded): a merging rank
    then never holds more than two slabs of that size of the tables it merges - they arrive in insertion order, slab
    s + 1 on the wire while slab s is inserted (`_Exchange.run_streaming`) - instead of every table of its batches.
    Same clusters; what makes 100 M rows that hardly merge fit one GPU's 288 GB on the rank of the final merge.
    (`_engine_factory` is a test hook: the CPU oracle engine.)
    """
    import torch
    import torch.distributed as dist

    if final_merge_criterion is None:
        final_merge_criterion = midsection_merge_criterion
    if recv_budget_mb is None and os.environ.get("BBHIP_RECV_BUDGET_MB"):
        recv_budget_mb = float(os.environ["BBHIP_RECV_BUDGET_MB"])
    budget = None if recv_budget_mb is None else max(1, int(recv_budget_mb * (1 << 20)))
    rank, world = dist.get_rank(), dist.get_world_size()
    on_gpu = dist.get_backend() == "nccl"
    dev_index = (torch.cuda.current_device() if device is None else device) if on_gpu else (device or 0)
    tdev = torch.device("cuda", dev_index) if on_gpu else torch.device("cpu")
    # BitFeature tables stay in HBM between rounds whenever the engine is the HIP engine
    dev_tables = _engine_factory is None and torch.cuda.is_available()
    table_dev = torch.device("cuda", dev_index) if dev_tables else None
    if dev_tables:
        torch.cuda.set_device(dev_index)
    common = dict(branching_factor=branching_factor, tolerance=tolerance, engine_factory=_engine_factory, device=dev_index)
    timer = _Timer()
    timer.exchange = {}  # type: ignore[attr-defined]
    timer.init_timing("total")

    # round 1: rank r owns shards r, r+W, ...  (no collective)
    timer.init_timing("round-1")
    infos = _files_range_tuples(list(input_files))
    mine: list[Entry] = []
    my_infos = [info for i, info in enumerate(infos) if i % world == rank]
    for info, (bufs, mols) in zip(my_infos, _initial_rounds(
            my_infos, threshold=threshold, merge_criterion=initial_merge_criterion,
            refinement=refinement_before_midsection, refine_merge_criterion=midsection_merge_criterion,
            refine_threshold_change=midsection_threshold_change, n_features=n_features,
            input_is_packed=input_is_packed, max_fps=max_fps, device_tables=dev_tables, **common)):
        for name in bufs:
            mine.append((info[0], name, bufs[name], mols[name]))
    timer.end_timing("round-1")

    round_idx = 1
    for _ in range(num_midsection_rounds):
        round_idx += 1
        timer.init_timing(f"round-{round_idx}")
        ex = _Exchange(dist, tdev, table_dev)
        if budget is not None:
            def mk_mid() -> BitBirch:
                return BitBirch(branching_factor=branching_factor, threshold=threshold + midsection_threshold_change,
                                merge_criterion=midsection_merge_criterion, tolerance=tolerance, device=dev_index,
                                _engine_factory=_engine_factory)

            owned, n_batches = ex.run_streaming(mine, bin_size, lambda b: b % world, budget, True, mk_mid)
            ordered = [(b, None) for b, _ in owned]
            trees = [t for _, t in owned]
        else:
            my_batches, n_batches = ex.run(mine, bin_size, lambda b: b % world)
            ordered = [(b, sorted(batch, key=lambda e: _CODE[e[1]], reverse=True)) for b, batch in my_batches]  # uint16 first
            trees = _merge_rounds([[(t, idx) for _, _, t, idx in batch] for _, batch in ordered],
                                  threshold=threshold + midsection_threshold_change,
                                  criterion=midsection_merge_criterion, **common)
        timer.exchange[f"round-{round_idx}"] = {"sent": ex.bytes_sent, "received": ex.bytes_received}  # type: ignore[attr-defined]
        z = len(str(n_batches))
        mine = []
        for (b, _), tree in zip(ordered, trees):
            bufs, mols = tree._bf_tables(tree._leaf_order(True), device=dev_tables)
            for name in bufs:
                mine.append((str(b).zfill(z), name, bufs[name], mols[name]))
        del trees
        _release_cached_hbm()
        timer.end_timing(f"round-{round_idx}")

    round_idx += 1
    timer.init_timing(f"round-{round_idx}")
    ex = _Exchange(dist, tdev, table_dev)
    tree: tp.Any = None
    if budget is not None:
        def mk_final() -> BitBirch:
            return BitBirch(branching_factor=branching_factor, threshold=threshold + midsection_threshold_change,
                            merge_criterion=final_merge_criterion, tolerance=tolerance, device=dev_index,
                            _engine_factory=_engine_factory)

        owned, _ = ex.run_streaming(mine, None, lambda b: 0, budget, False, mk_final)
        if rank == 0:
            tree = owned[0][1] if owned else mk_final()
    else:
        final, _ = ex.run(mine, None, lambda b: 0)
        if rank == 0:
            everything = final[0][1] if final else []
            tree = _merge_rounds([[(t, idx) for _, _, t, idx in everything]], threshold=threshold + midsection_threshold_change,
                                 criterion=final_merge_criterion, **common)[0]
    timer.exchange[f"round-{round_idx}"] = {"sent": ex.bytes_sent, "received": ex.bytes_received}  # type: ignore[attr-defined]
    del mine
    result: tp.Any = None
    if rank == 0:
        result = tree if return_tree else tree.get_cluster_mol_ids()
        if out_dir is not None:
            _write_outputs(Path(out_dir), tree, save_centroids)
    dist.barrier()
    timer.end_timing(f"round-{round_idx}")
    timer.end_timing("total")
    return result, timer
