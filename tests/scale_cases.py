r"""Shared helpers of the at-scale parity tests: digests of REFERENCE runs (tests/golden/scale.json, written by
tests/golden/make_golden_scale.py) compared with the same runs through bblean_amd on a given engine."""
from __future__ import annotations

import functools
import hashlib
import json
import pickle
from pathlib import Path

import numpy as np

from cases import MULTIROUND_CASES, MULTIROUND_SCALE_CASES, SCALE_CASES, make_input, multiround_shard

from bblean_amd import BitBirch, make_fake_fingerprints
from bblean_amd.multiround import run_multiround_bitbirch

GOLD = Path(__file__).resolve().parent / "golden"
SCALE = json.loads((GOLD / "scale.json").read_text())
ALL_MULTIROUND = list(MULTIROUND_CASES) + list(MULTIROUND_SCALE_CASES)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@functools.lru_cache(maxsize=3)
def scale_input(name: str) -> np.ndarray:
    case = next(c for c in SCALE_CASES if c["name"] == name)
    fps = make_input(case, make_fake_fingerprints)
    assert sha(fps) == SCALE["trees"][name]["input_sha"], "input differs from the one the reference saw"
    return fps


def tree_digest(tree: BitBirch) -> dict:
    assign = tree.get_assignments().astype("<u8")
    cents = np.array(tree.get_centroids(), dtype=np.uint8)
    sizes = np.bincount(assign.astype(np.int64))[1:]
    return {"clusters": int(cents.shape[0]), "assign_sha": sha(assign), "cent_sha": sha(cents),
            "sizes_sha": sha(sizes.astype("<i8")), "largest": int(sizes.max()), "singletons": int((sizes == 1).sum())}


def run_scale_tree(case: dict, engine_factory) -> None:
    name = case["name"]
    gold = SCALE["trees"][name]
    fps = scale_input(name)
    tree = BitBirch(branching_factor=case["bf"], threshold=case["thr"], merge_criterion=case["crit"],
                    _engine_factory=engine_factory)
    tree.fit(fps, n_features=case["n_features"])
    assert tree_digest(tree) == gold["fit"]
    ref = case.get("refine")
    if ref is not None:
        tree.set_merge(**ref["set_merge"])
        tree.refine_inplace(fps, n_largest=ref.get("n_largest", 1))
        assert tree_digest(tree) == gold["refine"]


def file_digest(path: Path) -> dict:
    raw = hashlib.sha256(path.read_bytes()).hexdigest()
    if path.suffix == ".npy":
        a = np.load(path)
        return {"raw": raw, "content": sha(a), "dtype": a.dtype.name, "shape": list(a.shape)}
    lists = pickle.load(open(path, "rb"))
    sizes = np.array([len(x) for x in lists], dtype="<i8")
    flat = np.array([i for x in lists for i in x], dtype="<i8")
    return {"raw": raw, "content": sha(sizes) + sha(flat), "n": len(lists)}


def write_shards(d: Path, case: dict) -> list[Path]:
    for s in case["seeds"]:
        np.save(d / f"fps.{str(s).zfill(4)}.npy", multiround_shard(case, s, make_fake_fingerprints))
    return sorted(d.glob("fps.*.npy"))


def check_final(case: dict, clusters, cents=None) -> None:
    gold = SCALE["multiround"][case["name"]]
    assert len(clusters) == gold["clusters"]
    assert sha(np.array([len(c) for c in clusters], dtype="<i8")) == gold["sizes_sha"]
    assert sha(np.array([i for c in clusters for i in c], dtype="<i8")) == gold["members_sha"]
    if cents is not None:
        assert sha(np.array(cents, dtype=np.uint8)) == gold["cent_sha"]


def run_multiround_files(case: dict, engine_factory, d: Path) -> None:
    r"""File-based multiround with every intermediate round-* table kept: each file must be byte-identical
    to the one the reference wrote (multiround.py:132-143), and so must the final clusters / centroids."""
    gold = SCALE["multiround"][case["name"]]
    files = write_shards(d, case)
    (d / "out").mkdir()
    run_multiround_bitbirch(files, d / "out", num_initial_processes=1, cleanup=False, _engine_factory=engine_factory,
                            **case["kwargs"])
    got = {p.name: file_digest(p) for p in sorted((d / "out").glob("round-*"))}
    assert sorted(got) == sorted(gold["files"])
    for name, dg in got.items():
        assert dg == gold["files"][name], name
    clusters = pickle.load(open(d / "out" / "clusters.pkl", "rb"))
    cents = pickle.load(open(d / "out" / "cluster-centroids-packed.pkl", "rb"))
    check_final(case, clusters, cents)
