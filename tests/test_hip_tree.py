r"""GPU parity of the HBM-resident tree engine: every reference-generated end-to-end
fixture (tests/golden/trees.npz) and larger seeded runs against the CPU oracle.  Cluster
ids, member order, centroids and BitFeature tables must be identical."""
from __future__ import annotations

import numpy as np
import pytest

from cases import TREE_CASES, sparse_ecfp_like
from oracle_engine import OracleEngine
from tree_cases import run_case

from bblean_amd import BitBirch, make_fake_fingerprints

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", TREE_CASES, ids=[c["name"] for c in TREE_CASES])
def test_hip_tree_vs_reference_fixture(case):
    run_case(case, None)  # None -> the product's HIP engine


def _same(a: BitBirch, b: BitBirch) -> None:
    assert a.get_cluster_mol_ids() == b.get_cluster_mol_ids()
    assert (a.get_assignments() == b.get_assignments()).all()
    assert (np.array(a.get_centroids()) == np.array(b.get_centroids())).all()
    ba, ma = a._bf_to_np()
    bo, mo = b._bf_to_np()
    assert list(ba) == list(bo)
    for k in ba:
        assert (np.array(ba[k]) == np.array(bo[k])).all() and ma[k] == mo[k]


@pytest.mark.parametrize("crit,thr,bf,n,kind", [
    ("diameter", 0.3, 50, 100_000, "fake"),
    ("diameter", 0.3, 254, 40_000, "fake"),
    ("tolerance-diameter", 0.3, 50, 30_000, "fake"),
    ("diameter", 0.65, 50, 30_000, "fake"),
    ("radius", 0.5, 50, 10_000, "fake"),
    ("tolerance-radius", 0.4, 50, 10_000, "fake"),
    ("tolerance-legacy", 0.5, 50, 10_000, "fake"),
    ("diameter", 0.3, 50, 50_000, "sparse"),
    ("diameter", 0.6, 50, 20_000, "sparse"),
    ("tolerance-radius", 0.65, 8, 152_088, "tiers"),
    ("diameter", 0.3, 50, 40_000, "width512"),    # mirrored, one 16-byte piece per wave
    ("diameter", 0.3, 50, 40_000, "width1024"),   # mirrored, two pieces per wave
    ("diameter", 0.3, 50, 60_000, "width4096"),   # not mirrored, several byte groups per thread
])
def test_hip_tree_vs_oracle_large(crit, thr, bf, n, kind):
    if kind == "fake":
        fps = np.concatenate([make_fake_fingerprints(min(10_000, n), seed=500 + i) for i in range((n + 9999) // 10_000)])[:n]
    elif kind.startswith("width"):
        nf = int(kind[5:])
        rng = np.random.default_rng(77)
        dens = np.clip(rng.normal(750 / 2048, 400 / 2048, (n, 1)), 1 / nf, 1 - 1 / nf)
        fps = np.packbits(rng.random((n, nf)) < dens, axis=1)
    elif kind == "tiers":
        # exact duplicates in three weight classes so that uint8, uint16 and uint32 cluster
        # features meet inside the same leaves and splits (bf 8: a split every few new rows)
        protos = make_fake_fingerprints(120, seed=77)
        reps = np.array([70_000] * 2 + [400] * 30 + [1] * 88)
        # (tolerance-radius: diameter/radius let a huge tight cluster swallow strangers, which
        # would leave two clusters and no splits)
        order = np.random.default_rng(5).permutation(np.repeat(np.arange(120), reps))
        fps = protos[order]
        assert fps.shape[0] == n
    else:
        fps = sparse_ecfp_like(n, 2048, 99)
    hip = BitBirch(branching_factor=bf, threshold=thr, merge_criterion=crit).fit(fps)
    ora = BitBirch(branching_factor=bf, threshold=thr, merge_criterion=crit, _engine_factory=OracleEngine).fit(fps)
    _same(hip, ora)
    s_h, s_o = hip._engine.stats(), ora._engine.stats()
    assert s_h[:7].tolist() == s_o[:7].tolist()  # same comparisons, merges, appends, splits


@pytest.mark.parametrize("crit,thr,tol", [("diameter", 0.3, None), ("radius", 0.35, None), ("never-merge", 0.3, 0.05),
                                          ("tolerance-diameter", 0.3, 0.05), ("tolerance-radius", 0.3, 0.05),
                                          ("tolerance-legacy", 0.3, 0.05)])
def test_hip_per_insert_trace_vs_oracle(crit, thr, tol):
    r"""the insertion trace, element by element.  `out_leaf[e]` is the leaf BitFeature element e
    ended in at the moment it was inserted (a fresh id = appended, an existing id = merged), and the engine counters
    (descent calls, rows compared, merges, appends, leaf / node / root splits) are compared after every 250
    insertions - a descent that took another child, a different merge decision or a split at another element shows
    up in the chunk where it happens, not only in the final clusters."""
    fps = make_fake_fingerprints(3000, seed=77, pack=True)
    kw = dict(branching_factor=8, threshold=thr, merge_criterion=crit)
    if tol is not None:
        kw["tolerance"] = tol
    hip, ora = BitBirch(**kw), BitBirch(_engine_factory=OracleEngine, **kw)
    for lo in range(0, 3000, 250):
        hip.fit(fps[lo:lo + 250])
        ora.fit(fps[lo:lo + 250])
        assert (hip._log_leaf[-1] == ora._log_leaf[-1]).all(), lo
        assert hip._engine.stats()[:7].tolist() == ora._engine.stats()[:7].tolist(), lo


def test_hip_refine_pipeline_vs_oracle():
    r"""config[2] shape at test scale: fit, then refine with tolerance-diameter."""
    fps = np.concatenate([make_fake_fingerprints(10_000, seed=900 + i) for i in range(3)])
    trees = []
    for fac in (None, OracleEngine):
        t = BitBirch(branching_factor=50, threshold=0.3, merge_criterion="diameter", _engine_factory=fac).fit(fps)
        t.set_merge("tolerance-diameter", tolerance=0.05)
        t.refine_inplace(fps, n_largest=1)
        trees.append(t)
    _same(*trees)


def test_hip_device_resident_input():
    import torch

    fps = make_fake_fingerprints(5000, seed=4242)
    dev = torch.from_numpy(fps).cuda()
    a = BitBirch(branching_factor=50, threshold=0.3).fit(dev)
    b = BitBirch(branching_factor=50, threshold=0.3).fit(fps)
    assert a.get_cluster_mol_ids() == b.get_cluster_mol_ids()


@pytest.mark.parametrize("row_bytes", [272, 260])
def test_hip_device_resident_strided_rows(row_bytes):
    r"""Device-resident rows inside a wider allocation: a 16-byte aligned stride stays on the steady-state kernel
    (rows are read with 16-byte loads straight into registers), any other stride takes the complete engine."""
    import torch

    fps = make_fake_fingerprints(6000, seed=4243)
    wide = torch.zeros((6000, row_bytes), dtype=torch.uint8, device="cuda")
    wide[:, :256] = torch.from_numpy(fps).cuda()
    view = wide[:, :256]
    assert view.stride(0) == row_bytes
    a = BitBirch(branching_factor=50, threshold=0.3).fit(view)
    b = BitBirch(branching_factor=50, threshold=0.3, _engine_factory=OracleEngine).fit(fps)
    assert a.get_cluster_mol_ids() == b.get_cluster_mol_ids()
    assert a._engine.stats()[:7].tolist() == b._engine.stats()[:7].tolist()


def test_hip_edge_cases():
    with pytest.raises(ValueError):
        BitBirch().fit(np.zeros((0, 256), dtype=np.uint8), n_features=2048)
    for rep in (1, 2, 10):
        z = np.zeros((rep, 256), dtype=np.uint8)
        assert BitBirch().fit(z, n_features=2048).get_cluster_mol_ids() == [list(range(rep))]
        o = np.full((rep, 256), 255, dtype=np.uint8)
        assert BitBirch().fit(o, n_features=2048).get_cluster_mol_ids() == [list(range(rep))]
    t = BitBirch(branching_factor=50, threshold=0.3).fit(make_fake_fingerprints(3000, seed=1))
    t.delete_internal_nodes()
    with pytest.raises(ValueError):
        t.fit(make_fake_fingerprints(10, seed=2))
    t.reset()
    t.fit(make_fake_fingerprints(10, seed=2))
    assert t.num_fitted_fps == 10


def test_hip_concurrent_trees_equal_sequential():
    r"""One launch, one workgroup per tree (multiround round-1 shards) == one tree at a time."""
    from bblean_amd import fit_concurrently

    shards = [make_fake_fingerprints(3000 + 500 * i, seed=7000 + i) for i in range(9)]
    together = [BitBirch(branching_factor=50, threshold=0.3) for _ in shards]
    offs = np.cumsum([0] + [len(s) for s in shards])
    fit_concurrently(together, shards, reinsert_indices=[range(offs[i], offs[i + 1]) for i in range(len(shards))])
    for i, s in enumerate(shards):
        alone = BitBirch(branching_factor=50, threshold=0.3).fit(s, reinsert_indices=range(offs[i], offs[i + 1]))
        assert together[i].get_cluster_mol_ids() == alone.get_cluster_mol_ids()
        assert (np.array(together[i].get_centroids()) == np.array(alone.get_centroids())).all()


@pytest.mark.parametrize("bf,n_trees", [(50, 300), (20, 280)])
def test_hip_many_trees_two_workgroups_per_cu(bf, n_trees):
    r"""More trees than compute units: the launch switches to the two-workgroups-per-CU kernel
    (shape-specialised for bf 50, generic otherwise).  Every tree must equal the oracle's."""
    from bblean_amd import fit_concurrently

    pool = make_fake_fingerprints(12_000, seed=31)
    rng = np.random.default_rng(8)
    shards, ids = [], []
    start = 0
    for i in range(n_trees):
        m = int(rng.integers(150, 900))
        shards.append(pool[rng.integers(0, len(pool), m)])
        ids.append(range(start, start + m))
        start += m
    together = [BitBirch(branching_factor=bf, threshold=0.3) for _ in shards]
    fit_concurrently(together, shards, reinsert_indices=ids)
    for i in (0, 1, 57, 128, 199, 255, 256, 257, n_trees - 1):
        ora = BitBirch(branching_factor=bf, threshold=0.3, _engine_factory=OracleEngine).fit(shards[i], reinsert_indices=ids[i])
        assert together[i].get_cluster_mol_ids() == ora.get_cluster_mol_ids()  # global ids: no get_assignments
        assert (np.array(together[i].get_centroids()) == np.array(ora.get_centroids())).all()
        assert together[i]._engine.stats()[:7].tolist() == ora._engine.stats()[:7].tolist()
    assert [sum(len(c) for c in t.get_cluster_mol_ids()) for t in together] == [len(s) for s in shards]


def test_hip_tree_full_size_1M_vs_oracle():
    r"""BASELINE.json configs[1] at full size: 1 M synthetic 2048-bit fingerprints, thr 0.3,
    bf 50 - cluster ids of the HIP engine must equal the CPU oracle's, element by element."""
    import torch

    from bench import synth_fake_fps

    fps = synth_fake_fps(1_000_000, seed=1000, device=torch.device("cuda"))
    hip = BitBirch(branching_factor=50, threshold=0.3, merge_criterion="diameter").fit(fps)
    ora = BitBirch(branching_factor=50, threshold=0.3, merge_criterion="diameter", _engine_factory=OracleEngine).fit(fps.cpu().numpy())
    a, b = hip.get_assignments(), ora.get_assignments()
    assert (a == b).all()
    assert hip._engine.stats()[:7].tolist() == ora._engine.stats()[:7].tolist()
    lv_h, lv_o = hip._leaves(), ora._leaves()
    assert (lv_h["cents"] == lv_o["cents"]).all() and (lv_h["n"] == lv_o["n"]).all()


@pytest.mark.parametrize("bf", [50, 254])
def test_hip_pipelined_kernel_tier_promotions_vs_oracle(bf):
    r"""The pipelined kernel's own corners (bb_tree_pipe.inc): BitFeatures that move up a tier while elements are in flight
    (runs of 400 duplicates cross 255 members: uint8 -> uint16 cluster features, decided and applied by the leaf engine), full
    leaves, in-place splits and the tolerance criterion of the merge rounds, on sparse rows - the upper tree levels keep
    all-zero centroids, the shape the pipelined kernel takes (a 70 000-fold duplicate would dominate every ancestor's
    majority vote and hand the tree to k_tree_fast; the uint32 tier is covered by the "tiers" case above).  Element by
    element the oracle's leaf ids and counters."""
    protos = sparse_ecfp_like(60, 2048, 91)
    singles = sparse_ecfp_like(90_000, 2048, 92)
    # (runs of 400 copies of one row between stretches of distinct rows: shuffled, BitBIRCH scatters the copies over many
    # leaves and no cluster reaches 256 members)
    fps = np.concatenate([np.concatenate([singles[k * 1500:(k + 1) * 1500], np.repeat(protos[k:k + 1], 400, axis=0)]) for k in range(60)])
    kw = dict(branching_factor=bf, threshold=0.65, merge_criterion="tolerance-diameter", tolerance=0.05)
    hip = BitBirch(**kw).fit(fps)
    ora = BitBirch(_engine_factory=OracleEngine, **kw).fit(fps)
    assert (hip._log_leaf[-1] == ora._log_leaf[-1]).all()
    assert hip._engine.stats()[:7].tolist() == ora._engine.stats()[:7].tolist()
    _same(hip, ora)
    assert int(np.bincount(hip.get_assignments()).max()) >= 256  # (the uint16 tier was reached)


@pytest.mark.parametrize("bf", [50, 254])
def test_hip_pipelined_kernel_drifting_centroids_vs_oracle(bf):
    r"""Merges that move centroids while elements are in flight, on the tree shape whose leaf-parent is all-zero (sparse
    rows): short runs of NOISY copies of a prototype between stretches of distinct rows.  Consecutive elements go to the same
    leaf and the same row, every merge takes a majority vote that can raise or lower the row's similarity to the next
    element - the leaf engine's short cuts (tagged pre-compares; at bf 254 the helper's first-argmax with the changed rows
    folded in, including the case "the best row changed and its key went down") have to reproduce np.argmax on the
    current rows every time.  Element by element the oracle's leaf ids and counters."""
    rng = np.random.default_rng(931)
    n_protos, copies, between = 2000, 12, 20
    protos = np.unpackbits(sparse_ecfp_like(n_protos, 2048, 93), axis=1).astype(bool)
    singles = sparse_ecfp_like(n_protos * between, 2048, 94)
    # (few copies per prototype: no feature reaches half of a leaf's rows, the leaf-parent's centroids stay all-zero)
    keep = rng.random((n_protos, copies, 2048)) > 0.22
    add = rng.random((n_protos, copies, 2048)) < 0.003
    noisy = np.packbits((protos[:, None, :] & keep) | add, axis=2)
    fps = np.concatenate([np.concatenate([singles[k * between:(k + 1) * between], noisy[k]]) for k in range(n_protos)])
    kw = dict(branching_factor=bf, threshold=0.45, merge_criterion="diameter")
    hip = BitBirch(**kw).fit(fps)
    ora = BitBirch(_engine_factory=OracleEngine, **kw).fit(fps)
    assert (hip._log_leaf[-1] == ora._log_leaf[-1]).all()
    assert hip._engine.stats()[:7].tolist() == ora._engine.stats()[:7].tolist()
    _same(hip, ora)
    st = hip._engine.stats()
    assert int(st[2]) > 5000  # (the noisy copies do merge)


@pytest.mark.parametrize("bf,crit", [(256, "diameter"), (300, "diameter"), (520, "radius"), (770, "tolerance-diameter")])
def test_hip_tree_block_compare_boundaries_vs_oracle(bf, crit):
    r"""Branching factors above 255 compare a node in double-buffered blocks of 128 rows, requested through buffer loads bounded
    by the node's live bytes (node_best's block path, bb_tree.hip): node slots of 257 rows (two blocks and one row), leaves whose
    length crosses the block boundaries (128, 256, ... 768) as they fill and split, roots from a few rows up to bf, the
    first-argmin of the splits through the same path.  Per-element leaf ids, counters and centroids equal the oracle's."""
    import torch

    from bench import synth_fake_fps

    n = 150_000
    fps = synth_fake_fps(n, seed=3000 + bf, device=torch.device("cuda"))
    kw = dict(branching_factor=bf, threshold=0.3, merge_criterion=crit)
    if crit.startswith("tolerance"):
        kw["tolerance"] = 0.05
    hip = BitBirch(**kw).fit(fps)
    ora = BitBirch(**kw, _engine_factory=OracleEngine).fit(fps.cpu().numpy())
    assert (hip.get_assignments() == ora.get_assignments()).all()
    assert hip._engine.stats()[:7].tolist() == ora._engine.stats()[:7].tolist()
    lv_h, lv_o = hip._leaves(), ora._leaves()
    assert (lv_h["cents"] == lv_o["cents"]).all() and (lv_h["n"] == lv_o["n"]).all()


@pytest.mark.parametrize("bf", [254, 1000])
def test_hip_tree_1M_large_branching_factors_vs_oracle(bf):
    r"""The CLI default (bf 254) and the branching factor the reference recommends for 100-200 M molecules (bf 1000,
    docs/src/user-guide/parameters.rst:95-98) at 1 M rows: cluster ids, centroids and counters equal the oracle's.
    bf 1000 has no LDS-resident nodes (1001 x 272 B > 160 KiB): every level is compared straight from L2 / HBM."""
    import torch

    from bench import synth_fake_fps

    fps = synth_fake_fps(1_000_000, seed=2000 + bf, device=torch.device("cuda"))
    hip = BitBirch(branching_factor=bf, threshold=0.3, merge_criterion="diameter").fit(fps)
    ora = BitBirch(branching_factor=bf, threshold=0.3, merge_criterion="diameter", _engine_factory=OracleEngine).fit(fps.cpu().numpy())
    assert (hip.get_assignments() == ora.get_assignments()).all()
    assert hip._engine.stats()[:7].tolist() == ora._engine.stats()[:7].tolist()
    lv_h, lv_o = hip._leaves(), ora._leaves()
    assert (lv_h["cents"] == lv_o["cents"]).all() and (lv_h["n"] == lv_o["n"]).all()


@pytest.mark.parametrize("n,k,thr,bf,batch", [(60_000, 3000, 0.6, 50, 256), (40_000, 400, 0.7, 20, 128),
                                             (50_000, 5000, 0.5, 50, 512)])
def test_hip_batch_mode_concurrent_gates_vs_oracle(n, k, thr, bf, batch, monkeypatch):
    r"""Batch mode (BBHIP_BATCH): stable-level routing in parallel, admission by flip distance and
    gate slack, all gates of a batch inserting concurrently.  On clustered data the batches spread
    over many gates; the result must still be the sequential one."""
    from cases import clustered_dense

    fps = clustered_dense(n, 2048, k, seed=n + k)
    ora = BitBirch(branching_factor=bf, threshold=thr, _engine_factory=OracleEngine).fit(fps)
    monkeypatch.setenv("BBHIP_BATCH", str(batch))
    hip = BitBirch(branching_factor=bf, threshold=thr).fit(fps)
    monkeypatch.delenv("BBHIP_BATCH")
    _same(hip, ora)
    assert hip._engine.stats()[:7].tolist() == ora._engine.stats()[:7].tolist()


def test_hip_batch_mode_fake_vs_oracle(monkeypatch):
    r"""Batch mode on S-fake rows (every element lands in the same gate: windows of 1-3 elements) against the ORACLE, per element."""
    fps = np.concatenate([make_fake_fingerprints(10_000, seed=300 + i) for i in range(6)])
    ora = BitBirch(branching_factor=50, threshold=0.3, _engine_factory=OracleEngine).fit(fps)
    monkeypatch.setenv("BBHIP_BATCH", "256")
    bat = BitBirch(branching_factor=50, threshold=0.3).fit(fps)
    monkeypatch.delenv("BBHIP_BATCH")
    assert (bat._log_leaf[-1] == ora._log_leaf[-1]).all()
    _same(bat, ora)
    assert bat._engine.stats()[:7].tolist() == ora._engine.stats()[:7].tolist()


@pytest.mark.gpu
def test_hip_streaming_ingest_from_file(tmp_path, monkeypatch):
    r"""`fit(path)`: the memory-mapped file is streamed through two HBM slabs (pinned bounce
    buffers, copy stream, helper thread).  Small slabs so that many hand-overs happen; the
    second call continues the same tree; BitFeature buffers take the same route."""
    monkeypatch.setenv("BBHIP_SLAB_KB", "300")  # 1200 rows per slab
    fps = np.concatenate([make_fake_fingerprints(10_000, seed=300 + i) for i in range(3)])
    f1, f2 = tmp_path / "a.npy", tmp_path / "b.npy"
    np.save(f1, fps[:17_001])
    np.save(f2, fps[17_001:])
    hip = BitBirch(branching_factor=50, threshold=0.3).fit(f1).fit(f2)
    ora = BitBirch(branching_factor=50, threshold=0.3, _engine_factory=OracleEngine).fit(fps)
    _same(hip, ora)
    hip.set_merge("tolerance-diameter", tolerance=0.05)
    ora.set_merge("tolerance-diameter", tolerance=0.05)
    hip.refine_inplace([f1, f2], n_largest=2)  # re-inserts ~12 k BitFeature buffers from host tables
    ora.refine_inplace([f1, f2], n_largest=2)
    _same(hip, ora)


@pytest.mark.parametrize("kind,bf,n,thr", [
    ("hier", 50, 120_000, 0.6), ("dense", 50, 80_000, 0.6), ("rdkit", 50, 120_000, 0.6),
    ("hier", 254, 120_000, 0.6), ("dense", 254, 80_000, 0.6),
    ("hier", 50, 60_000, 0.35),
    # bench.py's Zipf-profile workload (the only generator with a realistic bit-frequency profile AND real merging, 45-53 %;
    # it ran in no parity test at bf 50 / 254)
    ("zipf", 50, 120_000, 0.3), ("zipf", 254, 120_000, 0.3),
])
def test_hip_pipelined_kernel_informative_internal_levels_vs_oracle(kind, bf, n, thr):
    r"""Trees whose INTERNAL levels stay informative: planted dense prototypes / two-level families (tests/golden/cases.py
    clustered_dense, clustered_hier: the generators that exist because their upper centroids are not all-zero) and
    S-rdkit-like rows.  Every level above the leaves compares (bitbirch.py:317-320) and every tracking row on the path takes
    the element and a new majority centroid (:352-357) - at bf 50 through the multi-level router of the pipelined kernel
    (pipe_router_ml: centroid drift at the leaf-parent AND above it while elements are in flight: the `dense` prototypes sit
    at 45-60 % density, so a tracking row's features hover around the majority threshold and its centroid changes with almost
    every element, two levels up as well), at bf 254 through whatever the host picks for the shape.  Element by element the
    oracle's leaf ids, chunk by chunk its counters; at the end clusters, centroids and BitFeature tables."""
    from cases import clustered_dense, clustered_hier, dense_rdkit_like

    if kind == "hier":
        fps = clustered_hier(n, 2048, 12, n // 50, 7)
    elif kind == "dense":
        fps = clustered_dense(n, 2048, n // 50, 7)
    elif kind == "zipf":
        import torch

        from bench import synth_zipf

        fps = synth_zipf(n, 4242, torch.device("cuda")).cpu().numpy()
    else:
        fps = dense_rdkit_like(n, 2048, 2026)
    kw = dict(branching_factor=bf, threshold=thr, merge_criterion="diameter")
    hip, ora = BitBirch(**kw), BitBirch(_engine_factory=OracleEngine, **kw)
    for lo in range(0, n, 20_000):
        hip.fit(fps[lo:lo + 20_000])
        ora.fit(fps[lo:lo + 20_000])
        bad = np.nonzero(hip._log_leaf[-1] != ora._log_leaf[-1])[0]
        assert bad.size == 0, f"first differing element {lo + int(bad[0])}"
        assert hip._engine.stats()[:7].tolist() == ora._engine.stats()[:7].tolist(), lo
    _same(hip, ora)
    kc = hip._engine.kernel_counts()
    assert int(kc[:3].sum()) == n
    if kind == "zipf":
        # which kernel took it (the level-systolic kernel - bb_tree_sys.inc, one tree over many workgroups - is opt-in, BBHIP_SYS=1:
        # tests/test_hip_sys.py; by default these trees are built by the pipelined kernel's multi-level instance at bf 50 and
        # mostly by the steady-state kernel at bf 254)
        sc = hip._engine.sys_counts()
        print(f"zipf bf {bf}: elements by kernel pipe/fast/complete = {kc[:3].tolist()}, systolic {sc[:4].tolist()}")
        assert int(sc[0]) == 0, sc.tolist()
    if bf == 50 and thr >= 0.5:
        # the pipelined kernel took the tree once it had a root above the leaves, several exact levels or not
        # (at threshold 0.35 everything merges into a handful of clusters: the root stays a leaf, nothing to pipeline)
        assert int(kc[0]) > n // 2, kc.tolist()
