r"""Randomised parity of the PIPELINED insertion kernel (`k_tree_pipe`, bb_tree_pipe.inc) against the CPU oracle.

tests/test_hip_fuzz.py draws trees of a few thousand elements and mostly odd branching factors: none of its seeds ever
reaches the pipelined kernel (the host tries it only on 2048-bit rows at bf 50 / 254, and the first 8 192 elements of a
tree go through the steady-state kernel).  Here every seed does: bf 50 or 254, 12 k - 60 k fingerprints made of
segments of different kinds (sparse / dense planted prototypes, two-level planted families whose upper tree levels stay
informative, runs of exact duplicates, make_fake-like rows, all-zero and all-one rows), thresholds 0.15 - 0.8, diameter
or tolerance-diameter, random cuts into several `fit` calls (launch and run boundaries move), and - every fourth seed -
pools that are pre-grown by next to nothing (`BBHIP_TINY_POOLS`), so that the kernels stop on exhausted node / cluster
feature pools in the middle of their runs and are relaunched.  Compared with the oracle, per `fit` call: the leaf
BitFeature every element ended in (`_log_leaf`: a fresh id = appended, an existing one = merged) and the engine counters
(compares, rows compared, merges, appends, leaf / node / root splits); at the end clusters, centroids and the BitFeature
tables.  The reference's counterpart is tests/test_bb_consistency.py (fixed inputs, final clusters only).
Seeds are fixed: failures reproduce."""
import os

import numpy as np
import pytest

from bblean_amd import BitBirch
from oracle_engine import OracleEngine

pytestmark = pytest.mark.gpu

F = 2048


def _protos(rng, k, dens_lo, dens_hi):
    dens = rng.uniform(dens_lo, dens_hi, (k, 1))
    return rng.random((k, F)) < dens


def _segment(rng: np.random.Generator, m: int, kind: int) -> np.ndarray:
    if kind == 0:  # sparse ECFP-like rows around planted prototypes
        k = max(m // int(rng.integers(5, 80)), 1)
        p = _protos(rng, k, 0.01, 0.06)
        bits = p[rng.integers(0, k, m)]
        noise = rng.uniform(0.02, 0.2)
        bits = (bits & (rng.random((m, F)) > noise)) | (rng.random((m, F)) < noise * 0.03)
    elif kind == 1:  # dense prototypes, bits toggled: informative internal levels, routing spreads over the subtrees
        k = max(m // int(rng.integers(5, 80)), 1)
        p = _protos(rng, k, 0.35, 0.6)
        bits = p[rng.integers(0, k, m)] ^ (rng.random((m, F)) < rng.uniform(0.02, 0.12))
    elif kind == 2:  # two-level planted families (tests/golden/cases.py clustered_hier)
        ns = int(rng.integers(2, 14))
        sup = rng.random((ns, F)) < 0.5
        k = max(m // int(rng.integers(10, 60)), 1)
        p = sup[rng.integers(0, ns, k)] ^ (rng.random((k, F)) < rng.uniform(0.05, 0.15))
        bits = p[rng.integers(0, k, m)] ^ (rng.random((m, F)) < rng.uniform(0.01, 0.06))
    elif kind == 3:  # runs of exact duplicates (merges that grow one BitFeature: tier promotions at 256 members)
        base = rng.random((max(m // 40, 1), F)) < rng.uniform(0.02, 0.5)
        reps = rng.integers(1, 600 if rng.random() < 0.3 else 40, base.shape[0])
        bits = np.repeat(base, reps, axis=0)[:m]
        if bits.shape[0] < m:
            bits = np.concatenate([bits, rng.random((m - bits.shape[0], F)) < 0.2])
    elif kind == 4:  # make_fake_fingerprints-like: popcount ~ N(750, 400), uniform positions
        dens = np.clip(rng.normal(750 / F, 400 / F, (m, 1)), 1 / F, 1 - 1 / F)
        bits = rng.random((m, F)) < dens
    else:  # dense random rows with all-zero and all-one rows mixed in
        bits = rng.random((m, F)) < 0.5
        bits[rng.random(m) < 0.05] = False
        bits[rng.random(m) < 0.03] = True
    return np.packbits(bits, axis=1)


def _rows(rng: np.random.Generator, n: int) -> np.ndarray:
    parts, left = [], n
    kinds = rng.permutation(6)[: int(rng.integers(1, 4))]  # one to three kinds per seed
    while left > 0:
        m = int(min(left, rng.integers(500, 9000)))
        parts.append(_segment(rng, m, int(rng.choice(kinds))))
        left -= m
    rows = np.concatenate(parts)
    if rng.random() < 0.5:
        rows = rows[rng.permutation(n)]
    return np.ascontiguousarray(rows)


def _same_tables(a: BitBirch, b: BitBirch) -> None:
    assert (a.get_assignments() == b.get_assignments()).all()
    assert (np.array(a.get_centroids()) == np.array(b.get_centroids())).all()
    la, lb = a._leaves(), b._leaves()
    assert (la["members"] == lb["members"]).all() and (la["n"] == lb["n"]).all()
    ba, ma = a._bf_tables(a._leaf_order(True))
    bo, mo = b._bf_tables(b._leaf_order(True))
    assert list(ba) == list(bo)
    for k in ba:
        assert (np.asarray(ba[k]) == np.asarray(bo[k])).all()
        assert (ma[k].counts == mo[k].counts).all() and (ma[k].flat == mo[k].flat).all()


def _seed_range() -> range:
    r"""36 seeds by default; `BB_FUZZ_SEEDS=lo:hi` runs another range (soak runs after changes to the pipelined kernel)."""
    spec = os.environ.get("BB_FUZZ_SEEDS", "")
    if ":" in spec:
        lo, hi = spec.split(":")
        return range(int(lo), int(hi))
    return range(36)


@pytest.mark.parametrize("seed", _seed_range())
def test_pipe_fuzz_vs_oracle(seed):
    rng = np.random.default_rng(7000 + seed)
    bf = 50 if seed % 3 else 254
    n = int(rng.integers(12_000, 60_000))
    crit = "diameter" if rng.random() < 0.6 else "tolerance-diameter"
    thr = float(rng.uniform(0.15, 0.8))
    tol = float(rng.uniform(0.0, 0.1))
    rows = _rows(rng, n)
    cuts = sorted(set(int(c) for c in rng.integers(8_200, n + 1, int(rng.integers(0, 5)))) | {0, n})
    tiny = seed % 4 == 0
    kw = dict(branching_factor=bf, threshold=thr, merge_criterion=crit, tolerance=tol)
    hip, ora = BitBirch(**kw), BitBirch(_engine_factory=OracleEngine, **kw)
    old = os.environ.get("BBHIP_TINY_POOLS")
    try:
        if tiny:
            os.environ["BBHIP_TINY_POOLS"] = "1"
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            hip.fit(rows[lo:hi])
            ora.fit(rows[lo:hi])
            bad = np.nonzero(hip._log_leaf[-1] != ora._log_leaf[-1])[0]
            assert bad.size == 0, f"first differing element {lo + int(bad[0])} of [{lo}, {hi}), bf {bf} {crit} thr {thr:.3f}"
            assert hip._engine.stats()[:7].tolist() == ora._engine.stats()[:7].tolist(), (lo, hi)
    finally:
        if old is None:
            os.environ.pop("BBHIP_TINY_POOLS", None)
        else:
            os.environ["BBHIP_TINY_POOLS"] = old
    _same_tables(hip, ora)
    kc = hip._engine.kernel_counts()
    assert int(kc[:3].sum()) == n


@pytest.mark.parametrize("bf", [50, 254])
def test_pipe_tiny_pools_stop_and_resume(bf):
    r"""Pools pre-grown by next to nothing: the pipelined kernel (and the kernels it hands over to) stop on exhausted node /
    cluster-feature pools in the middle of their runs hundreds of times, the host grows the pool and relaunches from the
    element that did not fit - element by element the oracle's result."""
    rng = np.random.default_rng(99)
    rows = np.concatenate([_segment(rng, 10_000, 0), _segment(rng, 6_000, 4), _segment(rng, 6_000, 3)])
    kw = dict(branching_factor=bf, threshold=0.4, merge_criterion="diameter")
    old = os.environ.get("BBHIP_TINY_POOLS")
    os.environ["BBHIP_TINY_POOLS"] = "1"
    try:
        hip = BitBirch(**kw).fit(rows)
    finally:
        if old is None:
            os.environ.pop("BBHIP_TINY_POOLS", None)
        else:
            os.environ["BBHIP_TINY_POOLS"] = old
    ora = BitBirch(_engine_factory=OracleEngine, **kw).fit(rows)
    assert (hip._log_leaf[-1] == ora._log_leaf[-1]).all()
    assert hip._engine.stats()[:7].tolist() == ora._engine.stats()[:7].tolist()
    _same_tables(hip, ora)
    kc = hip._engine.kernel_counts()
    assert int(kc[7]) > 5, kc.tolist()  # launches that ended on an exhausted pool


def test_pipe_fuzz_reaches_the_pipeline():
    r"""The point of this file: most of the elements of most seeds are inserted by the pipelined kernel."""
    done = np.zeros(3, dtype=np.int64)
    for seed in (1, 2, 3, 5, 6, 7):
        rng = np.random.default_rng(7000 + seed)
        bf = 50 if seed % 3 else 254
        n = int(rng.integers(12_000, 60_000))
        crit = "diameter" if rng.random() < 0.6 else "tolerance-diameter"
        thr = float(rng.uniform(0.15, 0.8))
        tol = float(rng.uniform(0.0, 0.1))
        t = BitBirch(branching_factor=bf, threshold=thr, merge_criterion=crit, tolerance=tol).fit(_rows(rng, n))
        done += t._engine.kernel_counts()[:3].astype(np.int64)
    assert done[0] > done[1] + done[2], done.tolist()


@pytest.mark.parametrize("bf,tiny", [(50, False), (254, False), (254, True)])
def test_pipe_concurrent_trees_vs_oracle(bf, tiny):
    r"""`fit_concurrently` (multiround's first round: one workgroup per shard tree in shared launches) with shards large enough
    for the pipelined kernel: five trees of different sizes and kinds of rows - one of them with informative internal
    levels - leave and re-enter the pipeline at different elements (stretches of the steady-state kernel, pools that run
    out; bf 254: the trees share launches of the pipelined kernel of 2^15 .. 2^18 elements per tree, until a quarter of them
    leave one midway).  Every tree: the oracle's single-tree fit, element by element."""
    from bblean_amd import fit_concurrently

    rng = np.random.default_rng(4242 + bf)
    sizes = [30_000, 12_000, 300_000 if not tiny else 40_000, 9_000, 45_000]
    kinds = [0, 4, 4, 2, 1]
    shards = [np.ascontiguousarray(_segment(rng, m, k)) for m, k in zip(sizes, kinds)]
    kw = dict(branching_factor=bf, threshold=0.45, merge_criterion="diameter")
    old = os.environ.get("BBHIP_TINY_POOLS")
    try:
        if tiny:
            os.environ["BBHIP_TINY_POOLS"] = "1"
        hips = [BitBirch(**kw) for _ in shards]
        fit_concurrently(hips, shards)
    finally:
        if old is None:
            os.environ.pop("BBHIP_TINY_POOLS", None)
        else:
            os.environ["BBHIP_TINY_POOLS"] = old
    piped = 0
    for hip, rows in zip(hips, shards):
        ora = BitBirch(_engine_factory=OracleEngine, **kw).fit(rows)
        assert (hip._log_leaf[-1] == ora._log_leaf[-1]).all()
        assert hip._engine.stats()[:7].tolist() == ora._engine.stats()[:7].tolist()
        _same_tables(hip, ora)
        kc = hip._engine.kernel_counts()
        assert int(kc[:3].sum()) == rows.shape[0]
        piped += int(kc[0])
    if bf == 254:  # (shared launches of the pipelined kernel: bf 254 only - at bf 50 eight steady-state workgroups are as fast)
        assert piped > sum(sizes) // 4, piped


def test_pipe_moves_between_single_and_multi_level_instances():
    r"""S-fake rows at bf 50: informative levels above the leaf-parents come and go (a tracking row of a freshly split node has
    few members, its majority centroid is not all-zero yet).  The single-level instance hands the tree to the multi-level one
    when it meets such a level (STOP_PIPE_NEEDS_ML), the multi-level one hands it back after a stint of >= 256 elements when
    the shape allows (STOP_PIPE_PREFERS_SL) - every element as in the oracle's sequential fit, whichever instance inserted it."""
    import torch
    from bench import WORKLOADS

    gen, thr, _ = WORKLOADS["fake"]
    rows = gen(400_000, 5001, torch.device("cuda"))
    kw = dict(branching_factor=50, threshold=thr, merge_criterion="diameter")
    hip = BitBirch(**kw).fit(rows)
    ora = BitBirch(_engine_factory=OracleEngine, **kw).fit(rows.cpu().numpy())
    assert (hip._log_leaf[-1] == ora._log_leaf[-1]).all()
    assert hip._engine.stats()[:7].tolist() == ora._engine.stats()[:7].tolist()
    _same_tables(hip, ora)
    kc = hip._engine.kernel_counts()
    assert int(kc[0]) >= 390_000 and int(kc[3]) >= 4, kc.tolist()  # the pipelined kernel, in several launches (the hand-overs)


def test_pipe_run_end_audit():
    r"""The audit build of the pipelined kernel (`BBHIP_PIPE_AUDIT=1`: the phase-timer instances - diameter criterion - compare,
    whenever a run has drained, what the router holds about its tracking nodes (rows, popcounts, n, child ids, the children's
    lengths, nothing pending) and what the leaf engine holds about the leaves in its slots (rows, popcounts, row records,
    length) with HBM: every change is written through, so they must agree) over all workloads at bf 50 / 254, with pools
    that run out, element by element against the oracle as well.  The switch is read once per process: a subprocess.  And the
    audit audited: with `BBHIP_PIPE_AUDIT=corrupt` one bit of the router's slot is flipped before the comparison - the fit
    must fail with the audit's message."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, BBHIP_PIPE_AUDIT="1", BBHIP_TINY_POOLS="1")
    env.pop("BBHIP_LAUNCH_LOG", None)
    r = subprocess.run([sys.executable, str(root / "tools" / "pipe_check.py"), "30000", "50", "254"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    env.pop("BBHIP_TINY_POOLS")
    r = subprocess.run([sys.executable, str(root / "tools" / "pipe_check.py"), "60000", "50", "254"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch\nfrom bench import WORKLOADS\nfrom bblean_amd import BitBirch\n"
            "fps = WORKLOADS['fake'][0](40000, 5, torch.device('cuda'))\n"
            "t = BitBirch(branching_factor=50, threshold=0.3, merge_criterion='diameter').fit(fps)\n"
            "t._engine.stats()\n" % (str(root), str(root / "tests")))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stderr.splitlines() if "[bbhip pipe audit]" in ln]
    assert line and int(line[-1].split()[3]) > 100, r.stderr[-2000:]  # hundreds of runs ended and were audited
    r = subprocess.run([sys.executable, "-c", code], env=dict(env, BBHIP_PIPE_AUDIT="corrupt"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "run-end audit" in r.stderr, r.stderr[-2000:]
    # tolerance-diameter trees (refinement, the merge rounds) run audited instances too (until round 5 only the diameter
    # instances had one: a test "under the audit" could pass without a single slot audited)
    code_tol = code.replace("merge_criterion='diameter'", "merge_criterion='tolerance-diameter', tolerance=0.05")
    assert code_tol != code
    r = subprocess.run([sys.executable, "-c", code_tol], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stderr.splitlines() if "[bbhip pipe audit]" in ln]
    assert line and int(line[-1].split()[3]) > 100, r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-c", code_tol], env=dict(env, BBHIP_PIPE_AUDIT="corrupt"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "run-end audit" in r.stderr, r.stderr[-2000:]
