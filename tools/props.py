"""Size-independent properties of a clustering (tests/test_configs45.py, tools/config45.py, bench.py's config-3 record):
the clusters partition 0..n-1, sizes / labels agree and are sorted, and the final cluster features' column sums equal the
column sums of all input fingerprints (linearity across fit, refinement, exchanges and merge rounds) - gathered on the
device, a slab of leaves at a time, so that it runs at 100 M rows."""
import numpy as np
import torch


def column_sums(rows_dev: torch.Tensor, acc: torch.Tensor | None = None) -> torch.Tensor:
    r"""int64 column sums (2048 features) of packed uint8 rows resident on the device, accumulated into `acc`."""
    dev = rows_dev.device
    nfeat = rows_dev.shape[1] * 8
    if acc is None:
        acc = torch.zeros(nfeat, dtype=torch.int64, device=dev)
    shifts = torch.arange(7, -1, -1, device=dev, dtype=torch.uint8)
    for lo in range(0, rows_dev.shape[0], 250_000):
        bits = (rows_dev[lo:lo + 250_000, :, None] >> shifts) & 1
        acc += bits.view(-1, nfeat).sum(dim=0, dtype=torch.int64)
    return acc


def check_clustering(tree, n: int, want_colsums: torch.Tensor, ids: np.ndarray | None = None) -> dict:
    r"""Raises AssertionError on the first violated property; returns cluster statistics."""
    if ids is None:
        ids = tree.get_assignments()
    assert ids.shape == (n,) and ids.min() == 1
    lv = tree._leaves()
    k = lv["ids"].size
    assert int(ids.max()) == k
    sizes = np.bincount(ids.astype(np.int64), minlength=k + 1)[1:]
    order = tree._leaf_order(True)
    assert (sizes == lv["n"][order].astype(np.int64)).all() and (np.diff(sizes) <= 0).all()
    srt = np.sort(lv["members"])
    assert srt.size == n and srt[0] == 0 and srt[-1] == n - 1 and (np.diff(srt) == 1).all()
    del srt
    dev = want_colsums.device
    shifts = torch.arange(7, -1, -1, device=dev, dtype=torch.uint8)
    total = torch.zeros_like(want_colsums)
    total_n = 0
    for name, pos in tree._group_positions(order).items():
        width = np.dtype(name).itemsize
        step = max(1, (1 << 30) // (2049 * width))
        for lo in range(0, pos.size, step):
            p = pos[lo:lo + step]
            ones = lv["n"][p] == 1
            n_tail = int(ones.size if ones.all() else np.argmax(~ones[::-1])) if width == 1 else 0
            tab = tree._engine.gather_buffers(p, width, device_out=True, n_tail=n_tail)
            if tab.n_head:
                v = tab.raw.view(tab.n_head, 2049, width)
                vals = v[:, :, 0].to(torch.int64)
                for b in range(1, width):
                    vals += v[:, :, b].to(torch.int64) << (8 * b)
                total += vals[:, :-1].sum(dim=0)
                total_n += int(vals[:, -1].sum())
                del v, vals
            if tab.tail is not None:
                for q in range(0, tab.n_tail, 250_000):
                    bits = (tab.tail[q:q + 250_000, :, None] >> shifts) & 1
                    total += bits.view(-1, 2048).sum(dim=0, dtype=torch.int64)
                total_n += tab.n_tail
            del tab
    assert total_n == n, (total_n, n)
    assert torch.equal(total, want_colsums)
    return {"clusters": int(k), "largest": int(sizes[0]), "singletons": int((sizes == 1).sum())}
