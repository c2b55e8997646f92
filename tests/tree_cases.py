r"""Run one golden tree case (tests/golden/cases.py) through bblean_amd.BitBirch with a
given engine and compare every observable with the reference's recorded output."""
from __future__ import annotations

import hashlib
import json
from pathlib import Path

import numpy as np

from cases import TREE_CASES, make_input

from bblean_amd import BitBirch, make_fake_fingerprints

GOLD = Path(__file__).resolve().parent / "golden"
_TREES = None
_MANIFEST = None


def trees():
    global _TREES
    if _TREES is None:
        _TREES = dict(np.load(GOLD / "trees.npz"))
    return _TREES


def manifest():
    global _MANIFEST
    if _MANIFEST is None:
        _MANIFEST = json.loads((GOLD / "manifest.json").read_text())
    return _MANIFEST


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def case_by_name(name: str) -> dict:
    return next(c for c in TREE_CASES if c["name"] == name)


def _check_lists(tree: BitBirch, g: dict, prefix: str) -> None:
    lists = tree.get_cluster_mol_ids()
    sizes = np.array([len(x) for x in lists], dtype=np.int32)
    flat = np.array([i for x in lists for i in x], dtype=np.int32)
    assert sizes.tolist() == g[prefix + "_sizes"].tolist()
    assert (flat == g[prefix + "_members"]).all()


def run_case(case: dict, engine_factory, fps=None) -> BitBirch:
    g = trees()
    name = case["name"]
    if fps is None:
        fps = make_input(case, make_fake_fingerprints)
    assert sha(fps) == manifest()["inputs"][name], "input differs from the one the reference saw"
    kw = dict(branching_factor=case["bf"], threshold=case["thr"], merge_criterion=case["crit"])
    if case.get("tol") is not None:
        kw["tolerance"] = case["tol"]
    tree = BitBirch(_engine_factory=engine_factory, **kw)
    nf = case["n_features"]
    splits = case.get("fit_splits")
    if splits:
        lo = 0
        for hi in splits + [len(fps)]:
            tree.fit(fps[lo:hi], n_features=nf)
            lo = hi
    elif case.get("reinsert_offset") is not None:
        off = case["reinsert_offset"]
        tree.fit(fps, reinsert_indices=range(off, off + len(fps)), n_features=nf)
    else:
        tree.fit(fps, n_features=nf)
    if case.get("reinsert_offset") is None:
        assert (tree.get_assignments() == g[name + "_assign"]).all()
    _check_lists(tree, g, name)
    cents = np.array(tree.get_centroids())
    assert sha(cents) == bytes(g[name + "_cent_sha"]).hex()
    flat_u = np.array([i for x in tree.get_cluster_mol_ids(sort=False) for i in x], dtype=np.int32)
    assert (flat_u == g[name + "_members_unsorted"]).all()
    if case.get("bf_to_np"):
        bufs, mols = tree._bf_to_np()
        assert list(bufs.keys()) == manifest()["bf_groups"][name]
        for dt in bufs:
            arr = np.array(bufs[dt])
            assert arr.dtype.name == dt
            assert sha(arr) == bytes(g[f"{name}_bufs_{dt}_sha"]).hex()
            s2 = [len(x) for x in mols[dt]]
            f2 = [i for x in mols[dt] for i in x]
            assert s2 == g[f"{name}_bufmols_{dt}_sizes"].tolist()
            assert f2 == g[f"{name}_bufmols_{dt}_flat"].tolist()
    ref = case.get("refine")
    if ref is not None:
        if ref.get("set_merge"):
            tree.set_merge(**ref["set_merge"])
        tree.refine_inplace(fps, n_largest=ref.get("n_largest", 1))
        assert (tree.get_assignments() == g[name + "_refine_assign"]).all()
        _check_lists(tree, g, name + "_refine")
    rec = case.get("recluster")
    if rec:
        tree.recluster_inplace(**rec)
        assert (tree.get_assignments() == g[name + "_recluster_assign"]).all()
        _check_lists(tree, g, name + "_recluster")
    return tree
