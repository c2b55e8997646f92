#!/usr/bin/env python3
r"""Golden values of the clustering metrics (reference bblean/metrics.py) produced by running
the REFERENCE in the build container.  Data only: seeds, cluster member indices, float64 results.

    python tests/golden/make_golden_metrics.py   ->  tests/golden/metrics.npz
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parents[1]))

from _refimport import import_reference  # noqa: E402

import_reference(use_cpp=True)

from bblean import BitBirch  # noqa: E402
from bblean.fingerprints import make_fake_fingerprints, unpack_fingerprints  # noqa: E402
from bblean.metrics import jt_dbi, jt_isim_chi, jt_isim_dunn  # noqa: E402
from bblean.similarity import estimate_jt_std, jt_sim_matrix_packed  # noqa: E402

CASES = [dict(seed=4242, n=700, thr=0.25, take=14), dict(seed=99, n=400, thr=0.3, take=9)]


def main() -> None:
    out: dict[str, np.ndarray] = {}
    for c, case in enumerate(CASES):
        fps = make_fake_fingerprints(case["n"], seed=case["seed"], pack=True)
        tree = BitBirch(branching_factor=50, threshold=case["thr"]).fit(fps)
        ids = tree.get_cluster_mol_ids()[: case["take"]]
        clusters = [fps[np.array(i)] for i in ids]
        unpacked = [unpack_fingerprints(x) for x in clusters]
        out[f"c{c}_members"] = np.concatenate([np.array(i, dtype=np.int64) for i in ids])
        out[f"c{c}_sizes"] = np.array([len(i) for i in ids], dtype=np.int64)
        vals = [
            jt_isim_chi(clusters), jt_isim_chi(unpacked, input_is_packed=False),
            jt_dbi(clusters), jt_dbi(clusters, centrals="medoid"), jt_dbi(unpacked, input_is_packed=False),
            jt_isim_dunn(clusters), jt_isim_dunn(unpacked, input_is_packed=False),
            estimate_jt_std(fps, n_samples=40),
        ]
        out[f"c{c}_values"] = np.array(vals, dtype=np.float64)
        out[f"c{c}_simmat"] = jt_sim_matrix_packed(fps[:37])
        out[f"c{c}_case"] = np.array([case["seed"], case["n"], case["take"]], dtype=np.int64)
        out[f"c{c}_thr"] = np.array([case["thr"]])
        print(case, vals)
    np.savez_compressed(HERE / "metrics.npz", **out)


if __name__ == "__main__":
    main()
