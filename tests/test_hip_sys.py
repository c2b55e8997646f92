r"""The level-systolic insertion kernel (`k_tree_sys`, bblean_amd/csrc/bb_tree_sys.inc): ONE tree over many workgroups - every
node has an owner workgroup, elements flow down the levels through one-way rings, the reference's order per node
(bblean/bitbirch.py:305-357) by construction, splits on a quiesced chain (bitbirch.py:162-211, :289-303).

Parity against the CPU oracle, per `fit` call: the leaf BitFeature every element ended in - the very ids, which the host
renumbers into insertion order after every launch - and the engine counters; at the end clusters, centroids and the
BitFeature tables.  `BBHIP_SYS=1` sends every tree the shape allows through the kernel (also the shapes the default policy
leaves to the pipelined kernel: all-zero upper levels, where every element goes down one path and most hand-overs of the
leaf-parent are ALONE); the default policy is checked separately.  The randomised suite of the pipelined kernel
(tests/test_hip_pipe_fuzz.py) was also run under `BBHIP_SYS=1` (tools/sys_soak.sh, profiles/r06/sys_soak.txt): that is where
the one ordering bug of the first version surfaced (a full node's children change producers when it splits)."""
import os

import numpy as np
import pytest

from bblean_amd import BitBirch
from oracle_engine import OracleEngine

pytestmark = pytest.mark.gpu


def _same_tables(a: BitBirch, b: BitBirch) -> None:
    assert a.get_cluster_mol_ids() == b.get_cluster_mol_ids()
    assert (a.get_assignments() == b.get_assignments()).all()
    assert (np.array(a.get_centroids()) == np.array(b.get_centroids())).all()
    ba, ma = a._bf_to_np()
    bo, mo = b._bf_to_np()
    assert list(ba) == list(bo)
    for k in ba:
        assert (np.array(ba[k]) == np.array(bo[k])).all() and ma[k] == mo[k]


def _workload(name: str, n: int, seed: int) -> np.ndarray:
    import torch

    from bench import WORKLOADS

    return WORKLOADS[name][0](n, seed, torch.device("cuda")).cpu().numpy()


def _fit_both(rows, cuts, **kw):
    hip, ora = BitBirch(**kw), BitBirch(_engine_factory=OracleEngine, **kw)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        hip.fit(rows[lo:hi])
        ora.fit(rows[lo:hi])
        bad = np.nonzero(hip._log_leaf[-1] != ora._log_leaf[-1])[0]
        assert bad.size == 0, f"first differing element {lo + int(bad[0])} of [{lo}, {hi})"
        assert hip._engine.stats()[:7].tolist() == ora._engine.stats()[:7].tolist(), (lo, hi)
    return hip, ora


@pytest.fixture
def force_sys(monkeypatch):
    monkeypatch.setenv("BBHIP_SYS", "1")


@pytest.mark.parametrize("bf", [50, 254])
@pytest.mark.parametrize("name", ["zipf", "hier", "rdkit", "fake", "ecfp"])
def test_sys_forced_vs_oracle(name, bf, force_sys):
    r"""Every bench workload, informative upper levels or not, at both branching factors: three `fit` calls (the kernel starts
    on a tree it did not build, relaunches after root splits), per-element ids and counters, final tables."""
    from bench import WORKLOADS

    n = 60_000
    rows = _workload(name, n, 777 + bf)
    hip, ora = _fit_both(rows, [0, 15_000, 40_000, n], branching_factor=bf, threshold=WORKLOADS[name][1], merge_criterion="diameter")
    _same_tables(hip, ora)
    sc = hip._engine.sys_counts()
    assert int(sc[0]) >= n - 15_000 and int(sc[1]) >= 2, sc.tolist()  # (the first elements build the tree's first levels elsewhere)
    assert int(sc[3]) >= 60  # workgroups of the last launch


@pytest.mark.parametrize("crit,tol", [("tolerance-diameter", 0.05), ("tolerance-legacy", 0.05), ("never-merge", None)])
def test_sys_forced_other_criteria_vs_oracle(crit, tol, force_sys):
    kw = dict(branching_factor=50, threshold=0.3, merge_criterion=crit)
    if tol is not None:
        kw["tolerance"] = tol
    n = 30_000 if crit != "never-merge" else 20_000
    rows = _workload("zipf", n, 4001)
    hip, ora = _fit_both(rows, [0, 12_000, n], **kw)
    _same_tables(hip, ora)
    assert int(hip._engine.sys_counts()[0]) > 0


@pytest.mark.parametrize("bf", [50, 254])
def test_sys_forced_pools_that_run_out_vs_oracle(bf, force_sys, monkeypatch):
    r"""Pools pre-grown by next to nothing: the kernel stops admitting when the worst case of the elements in flight no longer
    fits, drains, the host grows the pools and relaunches - dozens of launches per call."""
    monkeypatch.setenv("BBHIP_TINY_POOLS", "1")
    n = 30_000
    rows = _workload("zipf", n, 99 + bf)  # (45 % merges: a few dozen leaves at bf 254 - the root is not a leaf for long)
    hip, ora = _fit_both(rows, [0, 10_000, n], branching_factor=bf, threshold=0.3, merge_criterion="diameter")
    _same_tables(hip, ora)
    sc = hip._engine.sys_counts()
    assert int(sc[1]) >= 5, sc.tolist()


def test_sys_forced_duplicates_and_tier_promotions_vs_oracle(force_sys):
    r"""Runs of exact duplicates between distinct rows: the same leaf and the same row again and again (pending counts climb, the
    guard waits, full leaves are handed over ALONE and merge), BitFeatures that cross 255 members (uint8 -> uint16 cluster
    features, allocated by the leaf's owner through the tree's atomic counters), all-zero and all-one rows."""
    rng = np.random.default_rng(5150)
    F = 2048
    parts = []
    for k in range(40):
        base = rng.random((1, F)) < rng.uniform(0.02, 0.4)
        parts.append(np.repeat(base, int(rng.integers(100, 500)), axis=0))
        parts.append(rng.random((int(rng.integers(200, 900)), F)) < rng.uniform(0.02, 0.5))
    parts.append(np.zeros((50, F), dtype=bool))
    parts.append(np.ones((30, F), dtype=bool))
    rows = np.packbits(np.concatenate(parts), axis=1)
    rows = np.ascontiguousarray(rows[rng.permutation(rows.shape[0])] if False else rows)
    n = rows.shape[0]
    for bf in (50, 254):
        hip, ora = _fit_both(rows, [0, 9_000, n], branching_factor=bf, threshold=0.5, merge_criterion="diameter")
        _same_tables(hip, ora)
        assert int(np.bincount(hip.get_assignments()).max()) >= 256
        assert int(hip._engine.sys_counts()[0]) > 0


def test_sys_then_refine_and_buffers_vs_oracle(force_sys):
    r"""A tree the systolic kernel built goes on through the other paths: refinement (leaf export, BitFeature buffers through the
    steady-state kernel, packed singleton tails through whichever kernel takes them) and a further `fit`."""
    n = 50_000
    rows = _workload("zipf", n, 31337)
    kw = dict(branching_factor=50, threshold=0.3, merge_criterion="diameter")
    hip, ora = _fit_both(rows, [0, 30_000], **kw)
    for t in (hip, ora):
        t.set_merge("tolerance-diameter", tolerance=0.05)
        t.refine_inplace(rows[:30_000], n_largest=2)
    _same_tables(hip, ora)
    for t in (hip, ora):
        t.fit(rows[30_000:])
    _same_tables(hip, ora)


def test_sys_default_policy(monkeypatch):
    r"""Unset, `BBHIP_SYS` sends a tree to the systolic kernel when the pipelined kernel has handed it over AND its root is
    informative (zipf, hier: every level compares), and leaves trees with all-zero upper levels where they are fastest (S-fake:
    the single-level pipeline)."""
    monkeypatch.delenv("BBHIP_SYS", raising=False)
    from bench import WORKLOADS

    n = 80_000
    # (bf 254: the pipelined kernel refuses these trees' shape only once an upper level has turned informative - 96 % of a
    # 1 M-row zipf tree, not necessarily within 80 k rows: tests/test_hip_tree.py's 120 k-row cases print what took them)
    for name, bf, want_sys in (("zipf", 50, True), ("hier", 50, True), ("rdkit", 50, True), ("fake", 50, False), ("ecfp", 254, False)):
        rows = _workload(name, n, 2024)
        hip, ora = _fit_both(rows, [0, 40_000, n], branching_factor=bf, threshold=WORKLOADS[name][1], merge_criterion="diameter")
        sc, kc = hip._engine.sys_counts(), hip._engine.kernel_counts()
        if want_sys:
            assert int(sc[0]) > n // 2, (name, bf, sc.tolist(), kc.tolist())
        else:
            assert int(sc[0]) == 0, (name, bf, sc.tolist(), kc.tolist())
    monkeypatch.setenv("BBHIP_SYS", "0")
    rows = _workload("zipf", 30_000, 2025)
    hip, _ = _fit_both(rows, [0, 30_000], branching_factor=50, threshold=0.3, merge_criterion="diameter")
    assert int(hip._engine.sys_counts()[0]) == 0
