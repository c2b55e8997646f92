r"""`bb run` as the reference's CLI drives the tree (cli.py:1058-1121): the exact call sequence - BitBirch(...),
fit(path) per input file, set_merge(refine ...), refine_inplace(LIST OF PATHS, n_largest), recluster_inplace,
delete_internal_nodes, get_centroids_mol_ids - replayed against bblean_amd.bitbirch and compared with what the
reference produced for the same sequence (tests/golden/scale.json "bbrun", made by make_golden_scale.py; the first
case is the reference's own CLI golden, tests/test_cli.py:266-308).  Also pins `_get_leaf_bfs` (what bblean.sklearn
reads, sklearn.py:90), `get_medoids` and `jt_stratified_sampling` against the reference."""
from __future__ import annotations

import hashlib
import importlib
import json
from pathlib import Path

import numpy as np
import pytest

from cases import BBRUN_CASES, bb_run_sequence, leaf_bfs_digest
from oracle_engine import OracleEngine

from bblean_amd import make_fake_fingerprints

GOLD = json.loads((Path(__file__).resolve().parent / "golden" / "scale.json").read_text())["bbrun"]


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _variant():
    r"""`_import_bitbirch_variant` (reference utils.py:51-68): the module is named, BitBirch and set_merge taken from it."""
    mod = importlib.import_module("bblean_amd.bitbirch")
    return mod.BitBirch, mod.set_merge


def _run(name: str, engine_factory, tmp_path: Path) -> None:
    case, gold = BBRUN_CASES[name], GOLD[name]
    BitBirch, _ = _variant()
    files, arrays = [], []
    for i, (n, seed) in enumerate(case["files"]):
        f = tmp_path / f"fingerprints.{i}.npy"
        arr = make_fake_fingerprints(n, n_features=2048, seed=seed, pack=True)
        np.save(f, arr)
        files.append(f)
        arrays.append(arr)
    kw = {k: v for k, v in case.items() if k != "files"}
    out, leaf = bb_run_sequence(BitBirch, files, before_release=leaf_bfs_digest, _engine_factory=engine_factory, **kw)
    assert len(out["mol_ids"]) == gold["clusters"]
    assert [list(map(int, c)) for c in out["mol_ids"][:13]] == gold["first13"]
    assert sha(np.array([len(c) for c in out["mol_ids"]], dtype="<i8")) == gold["sizes_sha"]
    assert sha(np.array([i for c in out["mol_ids"] for i in c], dtype="<i8")) == gold["members_sha"]
    assert sha(np.array(out["centroids"], dtype=np.uint8)) == gold["cent_sha"]
    assert leaf == gold["leaf_bfs"]  # _get_leaf_bfs: order, n_samples, centroids, member lists, buffers, dtype names


@pytest.mark.parametrize("name", list(BBRUN_CASES))
def test_bb_run_sequence_oracle(name, tmp_path):
    _run(name, OracleEngine, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(BBRUN_CASES))
def test_bb_run_sequence_hip(name, tmp_path):
    _run(name, None, tmp_path)


def test_cli_golden_is_the_references_own():
    r"""The first clusters the reference's tests/test_cli.py:268-282 asserts for `bb run -b 50 -t 0.65`."""
    assert GOLD["cli_golden"]["first13"][0] == [2195, 2196, 2378, 2440, 2443, 2454, 2463, 2464, 2465, 2467, 2527, 2544]
    assert GOLD["cli_golden"]["first13"][12] == [614, 637]


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(BBRUN_CASES))
def test_medoids_and_sampling_hip(name):
    r"""`BitBirch.get_medoids` (bitbirch.py:909-967) and `jt_stratified_sampling` (similarity.py:276-304) on the HIP
    kernels against the reference's outputs."""
    from bblean_amd import BitBirch
    from bblean_amd.similarity import jt_stratified_sampling

    case, gold = BBRUN_CASES[name], GOLD[name]
    fps = np.concatenate([make_fake_fingerprints(n, n_features=2048, seed=seed, pack=True) for n, seed in case["files"]])
    tree = BitBirch(branching_factor=case["branching_factor"], threshold=case["threshold"]).fit(fps)
    assert sha(np.asarray(tree.get_medoids(fps), dtype=np.uint8)) == gold["medoids_sha"]
    assert [int(i) for i in np.asarray(jt_stratified_sampling(fps[:500], 25)).tolist()] == gold["sampling"]
