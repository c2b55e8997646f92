#!/usr/bin/env python3
r"""Digests of REFERENCE runs at sizes where the tree is 4-5 levels deep and of
the reference's multiround with every intermediate round-* table kept (G6).

Runs only in the build container (needs /root/reference and oracle/_ref).  Only digests are written
(tests/golden/scale.json): cluster counts, sha256 of `get_assignments()` (little-endian uint64), of the
size-sorted packed centroids, of the sorted cluster sizes; per multiround file the sha256 of the raw file
bytes and of its content.  Inputs are regenerated from seeds by the parity tests
(tests/golden/cases.py generators; `make_fake_fingerprints` is pinned bit-identical to the reference's).

Usage:  python tests/golden/make_golden_scale.py [case-name ...]     (no names: everything, ~10 min)
"""
from __future__ import annotations

import hashlib
import json
import pickle
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parents[1]))

from _refimport import import_reference  # noqa: E402

import_reference(use_cpp=True)

from bblean import BitBirch  # noqa: E402
from bblean.fingerprints import make_fake_fingerprints  # noqa: E402
from bblean.multiround import run_multiround_bitbirch  # noqa: E402

from cases import (BBRUN_CASES, MULTIROUND_CASES, MULTIROUND_SCALE_CASES, SCALE_CASES, bb_run_sequence,  # noqa: E402
                   leaf_bfs_digest, make_input, multiround_shard)

OUT = HERE / "scale.json"


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def tree_digest(tree: BitBirch) -> dict:
    assign = tree.get_assignments().astype("<u8")
    cents = np.array(tree.get_centroids(), dtype=np.uint8)
    sizes = np.bincount(assign.astype(np.int64))[1:]
    return {"clusters": int(cents.shape[0]), "assign_sha": sha(assign), "cent_sha": sha(cents),
            "sizes_sha": sha(sizes.astype("<i8")), "largest": int(sizes.max()), "singletons": int((sizes == 1).sum())}


def file_digest(path: Path) -> dict:
    raw = hashlib.sha256(path.read_bytes()).hexdigest()
    if path.suffix == ".npy":
        a = np.load(path)
        return {"raw": raw, "content": sha(a), "dtype": a.dtype.name, "shape": list(a.shape)}
    lists = pickle.load(open(path, "rb"))
    sizes = np.array([len(x) for x in lists], dtype="<i8")
    flat = np.array([i for x in lists for i in x], dtype="<i8")
    return {"raw": raw, "content": sha(sizes) + sha(flat), "n": len(lists)}


def main() -> None:
    want = set(sys.argv[1:])
    rec = json.loads(OUT.read_text()) if OUT.is_file() else {"trees": {}, "multiround": {}}
    for case in SCALE_CASES:
        if want and case["name"] not in want:
            continue
        t0 = time.perf_counter()
        fps = make_input(case, make_fake_fingerprints)
        t1 = time.perf_counter()
        tree = BitBirch(branching_factor=case["bf"], threshold=case["thr"], merge_criterion=case["crit"])
        tree.fit(fps, n_features=case["n_features"])
        t2 = time.perf_counter()
        d = {"input_sha": sha(fps), "fit": tree_digest(tree), "ref_fit_seconds": round(t2 - t1, 2)}
        ref = case.get("refine")
        if ref is not None:
            tree.set_merge(**ref["set_merge"])
            tree.refine_inplace(fps, n_largest=ref.get("n_largest", 1))
            d["refine"] = tree_digest(tree)
            d["ref_refine_seconds"] = round(time.perf_counter() - t2, 2)
        rec["trees"][case["name"]] = d
        print(case["name"], f"gen {t1 - t0:.1f}s fit {t2 - t1:.1f}s", d["fit"]["clusters"], "clusters", flush=True)
        OUT.write_text(json.dumps(rec, indent=1, sort_keys=True))
    for case in list(MULTIROUND_CASES) + list(MULTIROUND_SCALE_CASES):
        if want and case["name"] not in want:
            continue
        with tempfile.TemporaryDirectory() as d:
            d = Path(d)
            for s in case["seeds"]:
                np.save(d / f"fps.{str(s).zfill(4)}.npy", multiround_shard(case, s, make_fake_fingerprints))
            (d / "out").mkdir()
            t0 = time.perf_counter()
            run_multiround_bitbirch(sorted(d.glob("*.npy")), d / "out", num_initial_processes=1, cleanup=False,
                                    **case["kwargs"])
            dt = time.perf_counter() - t0
            files = {p.name: file_digest(p) for p in sorted((d / "out").glob("round-*"))}
            clusters = pickle.load(open(d / "out" / "clusters.pkl", "rb"))
            cents = pickle.load(open(d / "out" / "cluster-centroids-packed.pkl", "rb"))
        sizes = np.array([len(c) for c in clusters], dtype="<i8")
        flat = np.array([i for c in clusters for i in c], dtype="<i8")
        rec["multiround"][case["name"]] = {
            "files": files, "clusters": len(clusters), "sizes_sha": sha(sizes), "members_sha": sha(flat),
            "cent_sha": sha(np.array(cents, dtype=np.uint8)), "ref_seconds": round(dt, 2)}
        print(case["name"], f"{dt:.1f}s", len(clusters), "clusters", len(files), "round files", flush=True)
        OUT.write_text(json.dumps(rec, indent=1, sort_keys=True))
    bbrun_goldens(rec, want)


def bbrun_goldens(rec: dict, want: set) -> None:
    from bblean.similarity import jt_stratified_sampling

    for name, case in BBRUN_CASES.items():
        if want and name not in want:
            continue
        with tempfile.TemporaryDirectory() as d:
            files = []
            arrays = []
            for i, (n, seed) in enumerate(case["files"]):
                f = Path(d) / f"fingerprints.{i}.npy"
                arr = make_fake_fingerprints(n, n_features=2048, seed=seed, pack=True)
                np.save(f, arr)
                files.append(f)
                arrays.append(arr)
            kw = {k: v for k, v in case.items() if k != "files"}
            out, leaf = bb_run_sequence(BitBirch, files, before_release=leaf_bfs_digest, **kw)
            fps = np.concatenate(arrays)
            # analysis helpers on the same tree / input (reference bitbirch.py:909-967, similarity.py:276-304)
            tree2 = BitBirch(branching_factor=case["branching_factor"], threshold=case["threshold"]).fit(fps)
            med = tree2.get_medoids(fps)
            samp = jt_stratified_sampling(fps[:500], 25)
        sizes = np.array([len(c) for c in out["mol_ids"]], dtype="<i8")
        flat = np.array([i for c in out["mol_ids"] for i in c], dtype="<i8")
        rec.setdefault("bbrun", {})[name] = {
            "clusters": len(out["mol_ids"]), "sizes_sha": sha(sizes), "members_sha": sha(flat),
            "cent_sha": sha(np.array(out["centroids"], dtype=np.uint8)), "first13": [list(map(int, c)) for c in out["mol_ids"][:13]],
            "leaf_bfs": leaf, "medoids_sha": sha(np.asarray(med, dtype=np.uint8)),
            "sampling": [int(i) for i in np.asarray(samp).tolist()],
        }
        print("bbrun", name, len(out["mol_ids"]), "clusters", flush=True)
        OUT.write_text(json.dumps(rec, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
