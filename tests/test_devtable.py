r"""Host logic of the device-resident round tables (`bblean_amd._engine.DevTable`) and of the bounded exchange with packed
singleton tails - on CPU tensors, no GPU: what a table with a tail stands for (the reference's ``[k, F+1]`` uint8 array,
multiround.py:132-143: `numpy()` must return exactly that), row slicing across the head / tail boundary, and
`_Exchange.run_streaming` cutting such tables into slabs (chunks never straddle the boundary, member lists travel with
the first chunk, every row arrives once, in order, with its own member count)."""
import socket

import numpy as np
import pytest
import torch

from bblean_amd import multiround as mr
from bblean_amd._engine import DevTable
from bblean_amd.bitbirch import _IndexLists

F = 64


def _table(rng, kh, kt):
    ns = np.sort(rng.integers(2, 200, kh))[::-1].astype(np.uint8)
    head = rng.integers(0, 2, (kh, F + 1)).astype(np.uint8)
    head[:, -1] = ns
    tail = rng.integers(0, 256, (kt, F // 8)).astype(np.uint8)
    counts = np.concatenate([ns.astype(np.int64), np.ones(kt, dtype=np.int64)])
    flat = np.arange(int(counts.sum()), dtype=np.int64)
    want = np.concatenate([head, np.concatenate([np.unpackbits(tail, axis=1), np.ones((kt, 1), dtype=np.uint8)], axis=1)]) if kt else head
    return DevTable(torch.from_numpy(head.copy()), 1, torch.from_numpy(tail.copy()) if kt else None), _IndexLists(counts, flat), want


def test_devtable_with_tail_is_the_reference_table():
    rng = np.random.default_rng(1)
    tab, idx, want = _table(rng, 7, 11)
    assert tab.shape == (18, F + 1) and len(tab) == 18 and tab.n_head == 7 and tab.n_tail == 11
    assert tab.nbytes == 7 * (F + 1) + 11 * (F // 8)
    assert (tab.numpy() == want).all()
    assert (tab.n_samples().astype(np.int64) == idx.counts).all()
    for lo, hi in [(0, 18), (0, 7), (7, 18), (3, 12), (9, 15), (0, 0), (18, 18), (6, 8)]:
        part = tab.rows(lo, hi)
        assert len(part) == hi - lo
        assert (part.numpy() == want[lo:hi]).all(), (lo, hi)
    no_tail, _, want2 = _table(rng, 5, 0)
    assert no_tail.tail is None and (no_tail.numpy() == want2).all()
    with pytest.raises(ValueError):
        DevTable(torch.zeros((2, 2 * (F + 1)), dtype=torch.uint8), 2, torch.zeros((1, F // 8), dtype=torch.uint8))


def test_run_streaming_cuts_tables_with_tails_into_slabs(monkeypatch):
    import torch.distributed as dist

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        rng = np.random.default_rng(0)
        tabs = [_table(rng, 50, 3000), _table(rng, 0, 700), _table(rng, 120, 0), _table(rng, 10, 5000)]
        got = []

        class FakeTree:
            def delete_internal_nodes(self):
                pass

        def fake_fit(trees, tables):
            for _, parts in zip(trees, tables):
                for part, sub in parts:
                    assert (part.n_samples().astype(np.int64) == sub.counts).all()
                    assert part.n_head == 0 or part.n_tail == 0  # a chunk lies on one side of the head / tail boundary
                    got.append((part.numpy().copy(), sub.counts.copy(), sub.flat.copy()))

        monkeypatch.setattr(mr, "fit_buffers_concurrently", fake_fit)
        mine = [(str(i), "uint8", t, idx) for i, (t, idx, _) in enumerate(tabs)]
        ex = mr._Exchange(dist, torch.device("cpu"), torch.device("cpu"))
        owned, n_batches = ex.run_streaming(mine, None, lambda b: 0, 40_000, False, FakeTree)
        assert n_batches == 1 and len(owned) == 1 and len(got) > len(tabs)
        assert (np.concatenate([g[0] for g in got]) == np.concatenate([w for _, _, w in tabs])).all()
        assert (np.concatenate([g[1] for g in got]) == np.concatenate([i.counts for _, i, _ in tabs])).all()
        assert (np.concatenate([g[2] for g in got]) == np.concatenate([i.flat for _, i, _ in tabs])).all()
    finally:
        dist.destroy_process_group()
