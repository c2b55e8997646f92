r"""Randomised parity: small trees of random shape (branching factor, width, threshold, merge
criterion, input distribution, one or several fit calls, BitFeature re-insertion) on the HIP
engine against the CPU oracle through the same host code.  Seeds are fixed: failures reproduce."""
import numpy as np
import pytest

from bblean_amd import BitBirch
from oracle_engine import OracleEngine

pytestmark = pytest.mark.gpu

MAX_DEPTH = 256  # MAXD of bb_tree.hip
CRITERIA = ["diameter", "radius", "tolerance-diameter", "tolerance-radius", "tolerance-legacy", "never-merge"]


def _rows(rng: np.random.Generator, n: int, nbytes: int, kind: int) -> np.ndarray:
    if kind == 0:  # dense uniform bits
        return rng.integers(0, 256, (n, nbytes), dtype=np.uint8)
    if kind == 1:  # sparse
        bits = rng.random((n, nbytes * 8)) < rng.uniform(0.01, 0.15)
        return np.packbits(bits, axis=1)
    if kind == 2:  # few prototypes with noise: many merges and exact duplicates
        k = int(rng.integers(2, 12))
        protos = rng.integers(0, 256, (k, nbytes), dtype=np.uint8)
        rows = protos[rng.integers(0, k, n)].copy()
        flip = rng.random((n, nbytes)) < 0.05
        rows[flip] ^= rng.integers(1, 256, int(flip.sum()), dtype=np.uint8)
        return rows
    rows = rng.integers(0, 256, (n, nbytes), dtype=np.uint8)  # dense with all-zero / all-one rows mixed in
    rows[rng.random(n) < 0.1] = 0
    rows[rng.random(n) < 0.05] = 255
    return rows


def _same(a: BitBirch, b: BitBirch, stats: bool = True) -> None:
    assert a.get_cluster_mol_ids() == b.get_cluster_mol_ids()
    assert (np.array(a.get_centroids()) == np.array(b.get_centroids())).all()
    ba, ma = a._bf_to_np()
    bo, mo = b._bf_to_np()
    assert list(ba) == list(bo)
    for k in ba:
        assert (np.array(ba[k]) == np.array(bo[k])).all() and ma[k] == mo[k]
    if stats:  # same number of comparisons, merges, appends and splits (counters restart at reset())
        assert a._engine.stats()[:7].tolist() == b._engine.stats()[:7].tolist()


@pytest.mark.parametrize("seed", range(48))
def test_fuzz_hip_vs_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    nbytes = int(rng.choice([8, 16, 24, 64, 100, 128, 256, 256, 256, 512]))
    bf = int(rng.choice([2, 3, 5, 8, 17, 50, 50, 64, 65, 120]))
    n = int(rng.integers(1, 2500))
    crit = CRITERIA[int(rng.integers(0, len(CRITERIA)))]
    thr = float(rng.uniform(0.1, 0.9))
    tol = float(rng.uniform(0.0, 0.2))
    rows = _rows(rng, n, nbytes, int(rng.integers(0, 4)))
    cuts = sorted(set(int(c) for c in rng.integers(0, n + 1, int(rng.integers(0, 3)))) | {0, n})
    def run(fac, stage):
        t = BitBirch(branching_factor=bf, threshold=thr, merge_criterion=crit, tolerance=tol, _engine_factory=fac)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            if hi > lo:
                t.fit(rows[lo:hi], n_features=nbytes * 8)
        if stage == 1 and seed % 3 == 0 and n > 20:  # refine / recluster re-insert BitFeature buffers
            t.set_merge("tolerance-diameter", tolerance=0.05, threshold=min(thr + 0.05, 0.95))
            t.refine_inplace(rows, n_largest=2)
        if stage == 1 and seed % 3 == 1 and n > 20:
            t.recluster_inplace(iterations=2, extra_threshold=0.02)
        return t

    for stage in (0, 1):
        ora = run(OracleEngine, stage)
        if int(ora._engine.stats()[6]) + 2 >= MAX_DEPTH:
            # degenerate branching factors make vines of hundreds of levels (the reference would hit
            # Python's recursion limit near 990); the device tree stops at MAX_DEPTH and says so
            with pytest.raises(MemoryError, match="deeper than"):
                run(None, stage)
            return
        _same(run(None, stage), ora, stats=stage == 0)
