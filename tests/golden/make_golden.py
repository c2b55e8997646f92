#!/usr/bin/env python3
r"""Generate the committed golden fixtures by running the REFERENCE itself.

Runs only in the build container (needs /root/reference and oracle/_ref built by
oracle/build_ref.sh).  Outputs are data only (inputs are regenerated from seeds with
bblean_amd.fingerprints.make_fake_fingerprints, whose bit-identity with the reference
generator is itself pinned by `fake_digests`):

    tests/golden/similarity.npz   kernel-level known answers (C++ and NumPy backends)
    tests/golden/merges.npz       merge-criterion truth table
    tests/golden/trees.npz        end-to-end cluster assignments / member orders
    tests/golden/manifest.json    case descriptions + digests

Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import hashlib
import json
import sys
import warnings
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parents[1]))

from _refimport import import_reference  # noqa: E402

import_reference(use_cpp=True)

import bblean._cpp_similarity as csim  # noqa: E402
import bblean._py_similarity as pysim  # noqa: E402
from bblean import BitBirch  # noqa: E402
from bblean._merges import get_merge_accept_fn  # noqa: E402
from bblean.fingerprints import make_fake_fingerprints  # noqa: E402
from bblean.similarity import jt_isim_radius_compl_from_sum  # noqa: E402

from cases import TREE_CASES, make_input  # noqa: E402

SEED_A = 17408390758220920002
SEED_B = 12620509540149709235


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def similarity_goldens() -> dict[str, np.ndarray]:
    out: dict[str, np.ndarray] = {}
    rng = np.random.default_rng(7)
    shapes = [(10, 256), (1, 256), (7, 128), (51, 256), (255, 256), (33, 253), (64, 4), (5, 64)]
    for k, (n, nb) in enumerate(shapes):
        arr = rng.integers(0, 256, (n, nb), dtype=np.uint8)
        if n > 3:
            arr[1] = 0  # all-zero row
            arr[3] = arr[2]  # duplicate rows -> argmin/argmax ties
        # sparsify some rows
        arr[::3] &= rng.integers(0, 256, (len(arr[::3]), nb), dtype=np.uint8)
        vec = arr[min(2, n - 1)].copy()
        pre = f"s{k}_"
        out[pre + "arr"] = arr
        out[pre + "vec"] = vec
        out[pre + "popcount_cpp"] = csim._popcount_2d(arr)
        out[pre + "popcount_py"] = pysim._popcount(arr)
        out[pre + "sims_cpp"] = csim._jt_sim_arr_vec_packed(arr, vec)
        out[pre + "sims_py"] = pysim._jt_sim_arr_vec_packed(arr, vec)
        zero = np.zeros(nb, dtype=np.uint8)
        out[pre + "sims_zero_cpp"] = csim._jt_sim_arr_vec_packed(arr, zero)
        i1, i2, s1, s2 = csim.jt_most_dissimilar_packed(arr)
        j1, j2, t1, t2 = pysim.jt_most_dissimilar_packed(arr)
        assert (i1, i2) == (int(j1), int(j2)) and (s1 == t1).all() and (s2 == t2).all()
        out[pre + "md_idx"] = np.array([i1, i2], dtype=np.int64)
        out[pre + "md_s1"] = s1
        out[pre + "md_s2"] = s2
        if (nb * 8) % 8 == 0:
            un = csim.unpack_fingerprints(arr)
            assert (un == np.unpackbits(arr, axis=-1)).all()
            ls = un.sum(0, dtype=np.uint64)
            out[pre + "add_rows"] = csim.add_rows(un)
            assert (out[pre + "add_rows"] == ls).all()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out[pre + "isim_cpp"] = np.array([csim.jt_isim_from_sum(ls, n)])
                out[pre + "isim_py"] = np.array([pysim.jt_isim_from_sum(ls, n)], dtype=np.float64)
                out[pre + "isim_packed_cpp"] = np.array([csim.jt_isim_packed_u8(arr)])
            out[pre + "centroid_py"] = pysim.centroid_from_sum(ls, n, pack=True)
            out[pre + "centroid_cpp"] = csim.centroid_from_sum(ls, n, pack=True)
            out[pre + "centroid_unpacked_py"] = pysim.centroid_from_sum(ls, n, pack=False)
            if n >= 2:
                out[pre + "radius_compl"] = np.array([jt_isim_radius_compl_from_sum(ls, n)])
    # the reference's own known answers (tests/test_similarity.py)
    fps = make_fake_fingerprints(10, seed=SEED_A)
    out["ka_fps10"] = fps
    out["ka_popcount"] = csim._popcount_2d(fps)
    out["ka_sims_first"] = csim._jt_sim_arr_vec_packed(fps, fps[0])
    i1, i2, s1, s2 = csim.jt_most_dissimilar_packed(fps)
    out["ka_md_idx"] = np.array([i1, i2], dtype=np.int64)
    out["ka_md_s1"] = s1
    out["ka_md_s2"] = s2
    un100 = make_fake_fingerprints(100, seed=SEED_A, pack=False)
    out["ka_isim100"] = np.array([csim.jt_isim_from_sum(un100.sum(0), 100)])
    un10 = make_fake_fingerprints(10, seed=SEED_A, pack=False)
    out["ka_compl_isim10"] = pysim.jt_compl_isim(un10)
    # unpack with n_features = 2024 (tests/test_similarity.py:114-134)
    f2024 = make_fake_fingerprints(10, seed=SEED_A, pack=True, n_features=2024)
    out["ka_fps2024"] = f2024
    out["ka_unpack2024"] = csim.unpack_fingerprints(f2024)
    return out


def merge_goldens() -> dict[str, np.ndarray]:
    rng = np.random.default_rng(11)
    F = 256
    crits = ["diameter", "radius", "tolerance-diameter", "tolerance-radius", "tolerance-legacy", "never-merge"]
    fns = {c: get_merge_accept_fn(c, 0.05) for c in crits}
    old_ls, nom_ls, old_n, nom_n, thr = [], [], [], [], []
    for _ in range(600):
        on = int(rng.choice([1, 2, 3, 5, 17, 50, 300, 999, 1000, 1001, 1500]))
        nn = int(rng.choice([1, 1, 1, 2, 7]))
        dens = rng.uniform(0.05, 0.6, F)
        flip = rng.uniform(0, 0.5)
        ols = rng.binomial(on, dens).astype(np.uint64)
        d2 = np.where(rng.random(F) < flip, rng.uniform(0.05, 0.6, F), dens)
        nls = rng.binomial(nn, d2).astype(np.uint64)
        old_ls.append(ols)
        nom_ls.append(nls)
        old_n.append(on)
        nom_n.append(nn)
        thr.append(float(rng.choice([0.05, 0.1, 0.2, 0.3, 0.5, 0.65])))
    out = {
        "old_ls": np.array(old_ls),
        "nom_ls": np.array(nom_ls),
        "old_n": np.array(old_n, dtype=np.int64),
        "nom_n": np.array(nom_n, dtype=np.int64),
        "thr": np.array(thr),
        "tol_table": np.array(
            [max(0.05 * (np.exp(-1e-3 * n) - np.exp(-1e-3 * 1000)), 0.0) for n in range(1001)]
        ),
    }
    for c in crits:
        dec = []
        for i in range(len(old_n)):
            new_ls = old_ls[i] + nom_ls[i]
            new_n = old_n[i] + nom_n[i]
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                dec.append(bool(fns[c](thr[i], new_ls, new_n, old_ls[i], nom_ls[i], old_n[i], nom_n[i])))
        out["accept_" + c] = np.array(dec, dtype=np.uint8)
    return out


def flatten(lists: list[list[int]]) -> tuple[np.ndarray, np.ndarray]:
    sizes = np.array([len(x) for x in lists], dtype=np.int32)
    flat = np.array([i for x in lists for i in x], dtype=np.int32)
    return sizes, flat


def tree_goldens(manifest: dict) -> dict[str, np.ndarray]:
    out: dict[str, np.ndarray] = {}
    for case in TREE_CASES:
        name = case["name"]
        fps = make_input(case, make_fake_fingerprints)
        manifest["inputs"][name] = sha(fps)
        kw = dict(branching_factor=case["bf"], threshold=case["thr"], merge_criterion=case["crit"])
        if case.get("tol") is not None:
            kw["tolerance"] = case["tol"]
        tree = BitBirch(**kw)
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
            out[name + "_assign"] = tree.get_assignments().astype(np.uint32)
        sizes, flat = flatten(tree.get_cluster_mol_ids())
        out[name + "_sizes"] = sizes
        out[name + "_members"] = flat
        cents = np.array(tree.get_centroids())
        out[name + "_cent_sha"] = np.frombuffer(bytes.fromhex(sha(cents)), dtype=np.uint8)
        _, flat_u = flatten(tree.get_cluster_mol_ids(sort=False))
        out[name + "_members_unsorted"] = flat_u
        if case.get("bf_to_np"):
            bufs, mols = tree._bf_to_np()
            for dt in bufs:
                out[f"{name}_bufs_{dt}_sha"] = np.frombuffer(
                    bytes.fromhex(sha(np.array(bufs[dt]))), dtype=np.uint8)
                s2, f2 = flatten(mols[dt])
                out[f"{name}_bufmols_{dt}_sizes"] = s2
                out[f"{name}_bufmols_{dt}_flat"] = f2
            manifest["bf_groups"][name] = list(bufs.keys())
        ref = case.get("refine")
        if ref is not None:
            if ref.get("set_merge"):
                tree.set_merge(**ref["set_merge"])
            tree.refine_inplace(fps, n_largest=ref.get("n_largest", 1))
            out[name + "_refine_assign"] = tree.get_assignments().astype(np.uint32)
            sizes, flat = flatten(tree.get_cluster_mol_ids())
            out[name + "_refine_sizes"] = sizes
            out[name + "_refine_members"] = flat
        rec = case.get("recluster")
        if rec:
            tree.recluster_inplace(**rec)
            out[name + "_recluster_assign"] = tree.get_assignments().astype(np.uint32)
            sizes, flat = flatten(tree.get_cluster_mol_ids())
            out[name + "_recluster_sizes"] = sizes
            out[name + "_recluster_members"] = flat
        print(f"  {name}: {len(out[name + '_sizes'])} clusters")
    return out


def main() -> None:
    manifest: dict = {"inputs": {}, "bf_groups": {}, "fake_digests": {}}
    for n, seed, nf in [(10, SEED_A, 2048), (3000, SEED_B, 2048), (100, 1, 2048), (50, 5, 512)]:
        manifest["fake_digests"][f"{n}_{seed}_{nf}"] = sha(make_fake_fingerprints(n, n_features=nf, seed=seed))
    print("similarity goldens")
    np.savez_compressed(HERE / "similarity.npz", **similarity_goldens())
    print("merge goldens")
    np.savez_compressed(HERE / "merges.npz", **merge_goldens())
    print("tree goldens")
    np.savez_compressed(HERE / "trees.npz", **tree_goldens(manifest))
    with open(HERE / "manifest.json", "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("done")


if __name__ == "__main__":
    main()
