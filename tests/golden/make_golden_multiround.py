#!/usr/bin/env python3
r"""Golden outputs of the REFERENCE's run_multiround_bitbirch (build container only)."""
from __future__ import annotations

import pickle
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parents[1]))
from _refimport import import_reference  # noqa: E402

import_reference(use_cpp=True)
from bblean.fingerprints import make_fake_fingerprints  # noqa: E402
from bblean.multiround import run_multiround_bitbirch  # noqa: E402

from cases import MULTIROUND_CASES  # noqa: E402

out = {}
for case in MULTIROUND_CASES:
    with tempfile.TemporaryDirectory() as d:
        d = Path(d)
        for s in case["seeds"]:
            np.save(d / f"fps.{str(s).zfill(4)}.npy", make_fake_fingerprints(case["n_per_file"], seed=s))
        (d / "out").mkdir()
        run_multiround_bitbirch(sorted(d.glob("*.npy")), d / "out", num_initial_processes=1, **case["kwargs"])
        clusters = pickle.load(open(d / "out" / "clusters.pkl", "rb"))
        cents = pickle.load(open(d / "out" / "cluster-centroids-packed.pkl", "rb"))
    out[case["name"] + "_sizes"] = np.array([len(c) for c in clusters], dtype=np.int32)
    out[case["name"] + "_members"] = np.array([i for c in clusters for i in c], dtype=np.int32)
    out[case["name"] + "_cents"] = np.array(cents, dtype=np.uint8)
    print(case["name"], len(clusters), "clusters")
np.savez_compressed(HERE / "multiround.npz", **out)
