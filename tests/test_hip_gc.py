r"""Node storage of the HIP engine (bb_tree.hip "Node storage"): nodes are addressed in blocks of rows, the compaction of the
node pools seals nodes that stopped changing (their length rounded up to a block instead of bf + 1 rows - the reference's node
is a Python list that grows, bitbirch.py:264-287) and renumbers every node; an insertion that reaches a sealed node moves it
back to full capacity first (the complete engine; the steady-state and the pipelined kernel hand such elements over).  None of
this may change a result: every test here forces compactions at the worst moments - at every pool exhaustion, with pools that
run out every few elements, between `fit` calls - and compares per element with the CPU oracle."""
import os

import numpy as np
import pytest

from bblean_amd import BitBirch
from oracle_engine import OracleEngine
from test_hip_pipe_fuzz import _rows, _same_tables, _segment

pytestmark = pytest.mark.gpu


class _Env:
    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _fit_both(rows, cuts, kw, compact_between=False):
    hip, ora = BitBirch(**kw), BitBirch(_engine_factory=OracleEngine, **kw)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        hip.fit(rows[lo:hi])
        ora.fit(rows[lo:hi])
        bad = np.nonzero(hip._log_leaf[-1] != ora._log_leaf[-1])[0]
        assert bad.size == 0, f"first differing element {lo + int(bad[0])} of [{lo}, {hi}) {kw}"
        assert hip._engine.stats()[:7].tolist() == ora._engine.stats()[:7].tolist(), (lo, hi)
        if compact_between:
            hip._engine.compact(True)
    return hip, ora


@pytest.mark.parametrize("seed", range(12))
def test_gc_at_every_growth_tiny_pools_vs_oracle(seed):
    r"""Pools that run out every few elements, every exhaustion of the node pools compacts them: nodes are sealed as soon as
    they go one interval without changing and thawed again when the next element reaches them - thousands of times per tree."""
    rng = np.random.default_rng(9100 + seed)
    bf = (50, 254, 50, 8)[seed % 4]
    n = int(rng.integers(9_000, 26_000)) if bf == 50 else (int(rng.integers(30_000, 60_000)) if bf == 254 else int(rng.integers(3_000, 7_000)))
    crit = "diameter" if seed % 3 else "tolerance-diameter"
    thr = float(rng.uniform(0.2, 0.75))
    rows = _rows(rng, n)
    cuts = sorted(set(int(c) for c in rng.integers(1, n + 1, 2)) | {0, n})
    kw = dict(branching_factor=bf, threshold=thr, merge_criterion=crit, tolerance=0.05)
    with _Env(BBHIP_TINY_POOLS="1", BBHIP_GC_MIN_MB="0"):
        hip, ora = _fit_both(rows, cuts, kw)
    mem = hip._engine.memory()
    if int(hip._engine.stats()[5]) >= 12:  # (a tree of a dozen nodes and more: rows that all merge into a handful of clusters never grow a pool)
        assert int(mem[4]) >= 3, "compactions were expected"
    # (sealed nodes that an insertion reached and thawed: most seeds have hundreds, sparse rows that only ever walk the
    # left-most path have none - test_gc_between_fit_calls_vs_oracle asserts them)
    _same_tables(hip, ora)


@pytest.mark.parametrize("bf,kind", [(50, 0), (254, 0), (50, 2), (254, 1), (50, 4), (254, 4)])
def test_gc_between_fit_calls_vs_oracle(bf, kind):
    r"""A compaction after every `fit` call (everything that did not change during the call is sealed), normal pools: the
    pipelined kernel meets sealed leaf-parents and sealed leaves, the steady-state kernel sealed nodes in its slot fills."""
    rng = np.random.default_rng(9200 + bf + kind)
    m = 1 if bf == 50 else 4  # (nodes of 255 rows: four times the rows for as many nodes)
    rows = np.concatenate([_segment(rng, 14_000 * m, kind), _segment(rng, 9_000 * m, kind), _segment(rng, 9_000 * m, (kind + 3) % 6)])
    cuts = [0, 9_000 * m, 9_500 * m, 15_000 * m, 15_000 * m + 1, 22_000 * m, len(rows)]
    kw = dict(branching_factor=bf, threshold=0.35 if kind in (0, 4) else 0.6, merge_criterion="diameter")
    with _Env(BBHIP_GC_MIN_MB="1024"):  # (only the explicit compactions, whatever the process's environment says)
        hip, ora = _fit_both(rows, cuts, kw, compact_between=True)
    mem = hip._engine.memory()
    assert int(mem[4]) == len(cuts) - 1 and int(mem[5]) > 0 and int(mem[7]) > 0, mem.tolist()
    _same_tables(hip, ora)
    # every node back to full capacity: same tree, nothing sealed
    hip._engine.compact(False)
    assert int(hip._engine.memory()[5]) == 0
    _same_tables(hip, ora)
    # ... and the tree goes on (refinement re-inserts through _fit_buffers: BitFeature buffers meet sealed nodes too)
    hip._engine.compact(True)
    hip._engine.compact(True)
    hip.set_merge("tolerance-diameter", tolerance=0.05)
    ora.set_merge("tolerance-diameter", tolerance=0.05)
    hip.refine_inplace(rows, n_largest=2)
    ora.refine_inplace(rows, n_largest=2)
    _same_tables(hip, ora)


def test_gc_sealed_leaves_give_their_rows_back():
    r"""Sparse rows that hardly merge at bf 254 (S-ecfp's shape: every fingerprint goes down the left-most path, splits are
    lopsided, the leaves end up a tenth full): with compaction the node pools' used part is a fraction of bf + 1 rows per node."""
    rng = np.random.default_rng(9300)
    n = 120_000
    bits = np.zeros((n, 2048), dtype=bool)
    cols = rng.integers(0, 2048, (n, 48))
    bits[np.arange(n)[:, None], cols] = True
    rows = np.packbits(bits, axis=1)
    kw = dict(branching_factor=254, threshold=0.3, merge_criterion="diameter")
    with _Env(BBHIP_GC_MIN_MB="1024"):
        plain = BitBirch(**kw).fit(rows)
    used_plain = int(plain._engine.memory()[1])
    with _Env(BBHIP_GC_MIN_MB="0"):
        hip = BitBirch(**kw)
        for lo in range(0, n, 20_000):
            hip.fit(rows[lo:lo + 20_000])
        hip._engine.compact(True)
        hip._engine.compact(True)
    used = int(hip._engine.memory()[1])
    assert (hip.get_assignments() == plain.get_assignments()).all()
    assert used < 0.35 * used_plain, (used, used_plain)
    assert used < 700 * n, f"{used / n:.0f} bytes of node rows per fingerprint"


def test_gc_batch_mode_thaws_everything_first():
    r"""The exact batch mode (concurrent gates, BBHIP_BATCH=1) never moves a node: a tree with sealed nodes is brought back to
    full capacity before its first batch."""
    rng = np.random.default_rng(9400)
    rows = np.concatenate([_segment(rng, 12_000, 1), _segment(rng, 8_000, 2)])
    kw = dict(branching_factor=50, threshold=0.6, merge_criterion="diameter")
    hip, ora = BitBirch(**kw), BitBirch(_engine_factory=OracleEngine, **kw)
    hip.fit(rows[:10_000])
    ora.fit(rows[:10_000])
    hip._engine.compact(True)
    hip._engine.compact(True)
    assert int(hip._engine.memory()[5]) > 0
    with _Env(BBHIP_BATCH="512"):
        hip.fit(rows[10_000:])
    ora.fit(rows[10_000:])
    assert (hip._log_leaf[-1] == ora._log_leaf[-1]).all()
    _same_tables(hip, ora)
