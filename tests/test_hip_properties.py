r"""Size-independent properties of the HIP engine at BASELINE.json's sizes (no oracle needed):
partition, cluster-feature checksums, centroid rule, determinism and stream continuity.  The
element-by-element comparison with the oracle at 1 M rows lives in test_hip_tree.py."""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))

from bblean_amd import BitBirch  # noqa: E402

pytestmark = pytest.mark.gpu


def _check_tree(tree: BitBirch, fps_host: np.ndarray, sample: int, seed: int) -> None:
    n = fps_host.shape[0]
    ids = tree.get_assignments()
    assert ids.shape == (n,) and ids.min() == 1
    members = tree.get_cluster_mol_ids()
    k = len(members)
    assert int(ids.max()) == k
    # partition: every row in exactly one cluster; labels follow the size-sorted order
    sizes = np.array([len(m) for m in members])
    assert sizes.sum() == n and (np.diff(sizes) <= 0).all()
    assert (np.bincount(ids.astype(np.int64), minlength=k + 1)[1:] == sizes).all()
    flat = np.concatenate([np.asarray(m, dtype=np.int64) for m in members])
    assert np.array_equal(np.sort(flat), np.arange(n))
    # checksum of checksums: the exported cluster features are the column sums of their members
    # (exact integers), n_samples their count, the packed centroid the majority vote 2*ls >= n
    bufs, mols = tree._bf_to_np()
    rng = np.random.default_rng(seed)
    total_ls = np.zeros(fps_host.shape[1] * 8, dtype=np.uint64)
    total_n = 0
    for name, table in bufs.items():
        table = np.asarray(table)
        total_ls += table[:, :-1].sum(axis=0, dtype=np.uint64)
        total_n += int(table[:, -1].sum(dtype=np.uint64))
        lists = mols[name]
        for i in rng.choice(len(lists), size=min(sample, len(lists)), replace=False):
            rows = fps_host[np.asarray(lists[i], dtype=np.int64)]
            ls = np.unpackbits(rows, axis=1).sum(axis=0, dtype=np.uint64)
            assert np.array_equal(table[i, :-1].astype(np.uint64), ls)
            assert int(table[i, -1]) == len(lists[i])
    assert total_n == n
    chunk = 200_000
    want = np.zeros_like(total_ls)
    for lo in range(0, n, chunk):
        want += np.unpackbits(fps_host[lo:lo + chunk], axis=1).sum(axis=0, dtype=np.uint64)
    assert np.array_equal(total_ls, want)
    cents = np.asarray(tree.get_centroids())
    lv = tree._leaves()
    order = tree._leaf_order(True)
    ns = lv["n"][order]
    for i in rng.choice(k, size=min(sample, k), replace=False):
        rows = fps_host[np.asarray(members[i], dtype=np.int64)]
        ls = np.unpackbits(rows, axis=1).sum(axis=0, dtype=np.uint64)
        bit = (2 * ls >= int(ns[i])) if ns[i] > 1 else ls.astype(bool)
        assert np.array_equal(cents[i], np.packbits(bit.astype(np.uint8)))


def test_properties_config2_1M_fake():
    import torch

    from bench import synth_fake_fps

    fps = synth_fake_fps(1_000_000, seed=4321, device=torch.device("cuda"))
    host = fps.cpu().numpy()
    tree = BitBirch(branching_factor=50, threshold=0.3).fit(fps)
    _check_tree(tree, host, sample=150, seed=1)
    # determinism and stream continuity: two calls continue the same tree
    again = BitBirch(branching_factor=50, threshold=0.3).fit(fps[:400_000]).fit(fps[400_000:])
    assert np.array_equal(tree.get_assignments(), again.get_assignments())


@pytest.mark.parametrize("bf", [254, 50])  # 254: the default of `bb run`, i.e. BASELINE config 3 as the CLI runs it
def test_properties_config3_2M_sparse_with_refine(bf):
    import torch

    from config3 import synth_ecfp

    fps = synth_ecfp(2_000_000, 11, torch.device("cuda"))
    host = fps.cpu().numpy()
    tree = BitBirch(branching_factor=bf, threshold=0.3).fit(fps)
    _check_tree(tree, host, sample=100, seed=2)
    tree.set_merge("tolerance-diameter", tolerance=0.05, threshold=0.3)
    tree.refine_inplace(host, n_largest=1)
    _check_tree(tree, host, sample=100, seed=3)
