r"""Case table shared by tests/golden/make_golden.py (reference side) and the parity
tests (oracle / HIP side).  Pure data + deterministic input builders."""
from __future__ import annotations

import numpy as np

SEED_B = 12620509540149709235


def sparse_ecfp_like(n: int, n_features: int, seed: int) -> np.ndarray:
    r"""S-ecfp: sparse rows (~48 of 2048 bits) scattered around
    n/50 planted prototypes with ~15 % of the bits flipped.  Packed uint8."""
    rng = np.random.default_rng(seed)
    k = max(n // 50, 1)
    scale = n_features / 2048.0
    pops = np.clip(np.rint(rng.normal(48 * scale, 12 * scale, k)), 8 * scale, 160 * scale).astype(np.int64)
    protos = np.zeros((k, n_features), dtype=bool)
    for i in range(k):
        protos[i, rng.choice(n_features, int(pops[i]), replace=False)] = True
    which = rng.integers(0, k, n)
    keep = rng.random((n, n_features)) > 0.15
    add = rng.random((n, n_features)) < (0.15 * pops[which] / n_features)[:, None]
    bits = (protos[which] & keep) | add
    return np.packbits(bits.astype(np.uint8), axis=1)


def dense_rdkit_like(n: int, n_features: int, seed: int) -> np.ndarray:
    r"""S-rdkit-like (BASELINE config 5): dense path-fingerprint-like rows, popcount
    ~ N(900, 250) clipped to [64, 1900] (scaled with the width), scattered around n/50 planted prototypes
    (12 % of a prototype's bits dropped, as many random bits added) so that threshold 0.6 with the
    diameter criterion gives non-trivial clusters.  Packed uint8."""
    rng = np.random.default_rng(seed)
    k = max(n // 50, 1)
    scale = n_features / 2048.0
    pops = np.clip(np.rint(rng.normal(900 * scale, 250 * scale, k)), 64 * scale, 1900 * scale).astype(np.int64)
    protos = rng.random((k, n_features)).argsort(axis=1) < pops[:, None]
    out = np.empty((n, n_features // 8), dtype=np.uint8)
    for lo in range(0, n, 20_000):  # chunks: n x n_features booleans would not fit for big n
        m = min(20_000, n - lo)
        which = rng.integers(0, k, m)
        keep = rng.random((m, n_features)) > 0.12
        add = rng.random((m, n_features)) < (0.12 * pops[which] / n_features)[:, None]
        out[lo:lo + m] = np.packbits(((protos[which] & keep) | add).astype(np.uint8), axis=1)
    return out


def sparse_ecfp_words(n: int, n_features: int, seed: int) -> np.ndarray:
    r"""S-ecfp at the size of BASELINE config 3 (10 M rows) in half a minute: the same construction as
    `sparse_ecfp_like` (n/50 planted prototypes of ~48 bits, about an eighth of a row's bits dropped, ~8 random bits
    added) built from 64-bit random WORDS - a bit is dropped where three random words agree on 1 (1/8), added where
    eight (three words and five rotated copies of them) do (~1/256) - instead of one random double per bit (160 GB of them at this size).  Packed uint8."""
    rng = np.random.default_rng(seed)
    k = max(n // 50, 1)
    w = n_features // 64
    pops = np.clip(np.rint(rng.normal(48, 12, k)), 8, 160)
    protos = np.empty((k, w), dtype=np.uint64)
    for lo in range(0, k, 20_000):
        m = min(20_000, k - lo)
        bits = rng.random((m, n_features)) < (pops[lo:lo + m] / n_features)[:, None]
        protos[lo:lo + m] = np.packbits(bits, axis=1).view(np.uint64)
    out = np.empty((n, n_features // 8), dtype=np.uint8)
    for lo in range(0, n, 100_000):
        m = min(100_000, n - lo)
        which = rng.integers(0, k, m)
        r = rng.integers(0, np.iinfo(np.uint64).max, size=(6, m, w), dtype=np.uint64, endpoint=True)
        drop = r[0] & r[1] & r[2]
        rot = lambda a, c: (a << np.uint64(c)) | (a >> np.uint64(64 - c))  # noqa: E731  (rotated copies stand in for more words)
        add = r[3] & r[4] & r[5] & rot(r[3], 17) & rot(r[4], 29) & rot(r[5], 7) & rot(r[3], 41) & rot(r[4], 3)
        out[lo:lo + m] = ((protos[which] & ~drop) | add).view(np.uint8)
    return out


def fake_chunks(n: int, seed0: int, make_fake, n_features: int = 2048) -> np.ndarray:
    r"""S-fake(N, seed): chunk c of 100 000 rows = make_fake_fingerprints(100_000, seed=seed0 + c)
    (the 1 M-row array of BASELINE.md section 2 is fake_chunks(1_000_000, 1000))."""
    parts = [make_fake(min(100_000, n - lo), n_features=n_features, seed=seed0 + c, pack=True)
             for c, lo in enumerate(range(0, n, 100_000))]
    return np.concatenate(parts) if len(parts) > 1 else parts[0]


def make_input(case: dict, make_fake) -> np.ndarray:
    kind = case.get("kind", "fake")
    n, nf = case["n"], case["n_features"]
    if kind == "fake":
        return make_fake(n, n_features=nf, seed=case["seed"], pack=True)
    if kind == "fake_chunks":
        return fake_chunks(n, case["seed"], make_fake, nf)
    if kind == "sparse":
        return sparse_ecfp_like(n, nf, case["seed"])
    if kind == "sparse_words":
        return sparse_ecfp_words(n, nf, case["seed"])
    if kind == "rdkit":
        return dense_rdkit_like(n, nf, case["seed"])
    if kind == "zeros":
        return np.zeros((n, nf // 8), dtype=np.uint8)
    if kind == "ones":
        return np.full((n, nf // 8), 255, dtype=np.uint8)
    if kind == "dups":
        rng = np.random.default_rng(case["seed"])
        base = rng.integers(0, 256, (7, nf // 8), dtype=np.uint8)
        return base[rng.integers(0, 7, n)]
    if kind == "uniform":
        rng = np.random.default_rng(case["seed"])
        return rng.integers(0, 256, (n, nf // 8), dtype=np.uint8)
    raise ValueError(kind)


def _c(name, n, bf, thr, crit, seed=SEED_B, n_features=2048, **kw):
    d = dict(name=name, n=n, bf=bf, thr=thr, crit=crit, seed=seed, n_features=n_features)
    d.update(kw)
    return d


TREE_CASES = [
    # the reference's own end-to-end goldens (tests/test_bb_consistency.py, test_refine.py)
    _c("diam065_3000", 3000, 50, 0.65, "diameter"),
    _c("radius065_1000", 1000, 50, 0.65, "radius"),
    _c("tollegacy065_500", 500, 50, 0.65, "tolerance-legacy", tol=0.05),
    _c("refine_100", 100, 50, 0.3, "diameter", refine={}),
    # configs of BASELINE.json at test scale
    _c("diam03_3000", 3000, 50, 0.3, "diameter", bf_to_np=True,
       refine={"set_merge": {"criterion": "tolerance-diameter", "tolerance": 0.05}}),
    _c("diam03_20000", 20000, 50, 0.3, "diameter", seed=1000),
    _c("toldiam03_3000", 3000, 50, 0.3, "tolerance-diameter", tol=0.05),
    _c("tolradius05_1000", 1000, 50, 0.5, "tolerance-radius", tol=0.05),
    _c("never_300", 300, 50, 0.3, "never-merge"),
    _c("bf254_4000", 4000, 254, 0.3, "diameter", seed=77, bf_to_np=True,
       refine={"n_largest": 3, "set_merge": {"criterion": "tolerance-diameter", "tolerance": 0.05}}),
    _c("bf10_2000", 2000, 10, 0.4, "diameter", seed=5),
    _c("f1024_1500", 1500, 20, 0.5, "diameter", seed=9, n_features=1024),
    _c("f512_800", 800, 16, 0.45, "tolerance-diameter", seed=3, n_features=512, tol=0.1),
    _c("f64_500", 500, 8, 0.5, "diameter", seed=4, n_features=64),
    _c("diam06_dense_3000", 3000, 50, 0.6, "diameter", seed=21,
       recluster={"iterations": 2, "extra_threshold": 0.025}),
    _c("sparse03_5000", 5000, 50, 0.3, "diameter", kind="sparse", seed=13,
       refine={"set_merge": {"criterion": "tolerance-diameter", "tolerance": 0.05}}),
    _c("sparse065_3000", 3000, 50, 0.65, "diameter", kind="sparse", seed=14),
    _c("uniform035_3000", 3000, 50, 0.35, "diameter", kind="uniform", seed=15),
    _c("bigclusters_6000", 6000, 50, 0.2, "diameter", seed=31, bf_to_np=True,
       refine={"set_merge": {"criterion": "tolerance-diameter", "tolerance": 0.05}}),
    _c("zeros_10", 10, 50, 0.65, "diameter", kind="zeros"),
    _c("ones_10", 10, 50, 0.65, "diameter", kind="ones"),
    _c("dups_400", 400, 5, 0.65, "diameter", kind="dups", seed=8),
    _c("splitfit_2500", 2500, 50, 0.3, "diameter", seed=41, fit_splits=[700, 1900]),
    _c("reinsert_1000", 1000, 50, 0.3, "diameter", seed=42, reinsert_offset=5000),
]


# multiround fixtures: files of make_fake_fingerprints(n_per_file, seed=s) for s in seeds
MULTIROUND_CASES = [
    # the reference's own test (tests/test_multiround.py:9-48)
    dict(name="mr_ref_test", seeds=list(range(1, 21, 2)), n_per_file=100,
         kwargs=dict(bin_size=2, threshold=0.65, midsection_merge_criterion="tolerance-legacy")),
    # CLI defaults at test scale (tolerance-diameter merge rounds, full refinement)
    dict(name="mr_defaults", seeds=[101, 102, 103, 104, 105, 106], n_per_file=400,
         kwargs=dict(bin_size=4, threshold=0.3, branching_factor=50)),
    dict(name="mr_split_2mid", seeds=[201, 202, 203, 204, 205], n_per_file=300,
         kwargs=dict(bin_size=2, threshold=0.3, branching_factor=30, num_midsection_rounds=2,
                     refinement_before_midsection="split")),
    dict(name="mr_none_big", seeds=[301, 302, 303], n_per_file=1500,
         kwargs=dict(bin_size=10, threshold=0.2, branching_factor=50, refinement_before_midsection="none")),
]


# Runs of the REFERENCE at sizes whose trees are 4-5 levels deep: only digests
# are committed (tests/golden/scale.json, written by make_golden_scale.py).  `gpu_only`: too slow for the
# CPU suite's oracle leg, checked on the HIP engine (and by the oracle when BB_SCALE_ORACLE=1).
_REFINE_TD = {"set_merge": {"criterion": "tolerance-diameter", "tolerance": 0.05}}
SCALE_CASES = [
    _c("fake_200k", 200_000, 50, 0.3, "diameter", seed=1000, kind="fake_chunks", refine=_REFINE_TD),
    _c("fake_bf254_100k", 100_000, 254, 0.3, "diameter", seed=1000, kind="fake_chunks"),
    _c("fake_bf1000_100k", 100_000, 1000, 0.3, "diameter", seed=1000, kind="fake_chunks"),
    _c("ecfp_100k", 100_000, 50, 0.3, "diameter", seed=2024, kind="sparse", refine=_REFINE_TD),
    _c("ecfp_bf254_100k", 100_000, 254, 0.3, "diameter", seed=2025, kind="sparse"),
    _c("rdkit_100k", 100_000, 50, 0.6, "diameter", seed=2026, kind="rdkit"),
    _c("rdkit_bf254_100k", 100_000, 254, 0.6, "diameter", seed=2027, kind="rdkit"),
    _c("fake_1M", 1_000_000, 50, 0.3, "diameter", seed=1000, kind="fake_chunks", gpu_only=True),
    # BASELINE config 3 (10 M S-ecfp rows, the CLI's default branching factor) as far as the reference runs in this
    # container's 62 GB: its fit of 10 M rows ran out of memory (stopped at a 58 GB address-space limit, twice), and so did
    # `--refine-num 1` at that size (old tree + every BitFeature buffer + new tree).  Pinned instead: the fit of
    # 5 M rows of the same generator (reference: 1072 s, 31 GB) and fit + refinement of 3 M rows of the same generator (565 s).  The 10 M instance
    # itself runs in tools/config3.py and profiles/r03/config3_10M_bf254.log (HIP only: timings and tree statistics).
    _c("ecfp_5M_bf254", 5_000_000, 254, 0.3, "diameter", seed=3003, kind="sparse_words", gpu_only=True),
    _c("ecfp_3M_bf254_refine", 3_000_000, 254, 0.3, "diameter", seed=3003, kind="sparse_words", refine=_REFINE_TD, gpu_only=True),
]

# BASELINE configs 4 and 5 at test scale: 8 shard files through multiround with the CLI defaults
# (bf 254, full refinement, one merge round in bins of 10, tolerance-diameter merges), intermediate
# round-* tables kept (cleanup=False) and digested file by file.
MULTIROUND_SCALE_CASES = [
    dict(name="mr_cfg4_ecfp_8x25k", kind="sparse", seeds=list(range(4000, 4008)), n_per_file=25_000,
         kwargs=dict(threshold=0.3)),
    dict(name="mr_cfg5_rdkit_8x25k", kind="rdkit", seeds=list(range(5000, 5008)), n_per_file=25_000,
         kwargs=dict(threshold=0.6, initial_merge_criterion="diameter")),
    dict(name="mr_fake_bf50_8x25k", kind="fake", seeds=list(range(6000, 6008)), n_per_file=25_000,
         kwargs=dict(threshold=0.3, branching_factor=50)),
]


def multiround_shard(case: dict, seed: int, make_fake) -> np.ndarray:
    kind = case.get("kind", "fake")
    if kind == "fake":
        return make_fake(case["n_per_file"], seed=seed)
    return make_input(dict(kind=kind, n=case["n_per_file"], n_features=2048, seed=seed), make_fake)


def clustered_dense(n: int, n_features: int, k: int, seed: int, flip: float = 0.08) -> np.ndarray:
    r"""k dense prototypes (45-60 % bits set), every row = a random prototype with `flip` of its
    bits toggled, in random order.  Unlike S-fake / S-ecfp the majority centroids of the upper tree
    levels stay informative, so consecutive fingerprints are routed to DIFFERENT subtrees - the
    workload that exercises concurrent gates in batch mode."""
    rng = np.random.default_rng(seed)
    dens = rng.uniform(0.45, 0.60, k)
    protos = rng.random((k, n_features)) < dens[:, None]
    which = rng.integers(0, k, n)
    bits = protos[which] ^ (rng.random((n, n_features)) < flip)
    return np.packbits(bits.astype(np.uint8), axis=1)


def clustered_hier(n: int, n_features: int, n_super: int, k: int, seed: int,
                   flip_cluster: float = 0.12, flip_member: float = 0.04) -> np.ndarray:
    r"""Two-level planted structure: `n_super` super-prototypes (50 % density), k cluster prototypes
    (a super-prototype with `flip_cluster` of its bits toggled), rows = a cluster prototype with
    `flip_member` toggled.  Trackers that cover one super-family have bit frequencies near 85 % / 15 %:
    informative AND far from the 50 % majority threshold, i.e. stable upper levels."""
    rng = np.random.default_rng(seed)
    supers = rng.random((n_super, n_features)) < 0.5
    sup_of = rng.integers(0, n_super, k)
    protos = supers[sup_of] ^ (rng.random((k, n_features)) < flip_cluster)
    which = rng.integers(0, k, n)
    bits = protos[which] ^ (rng.random((n, n_features)) < flip_member)
    return np.packbits(bits.astype(np.uint8), axis=1)


# ---- `bb run` as a call sequence (reference cli.py:1058-1121), shared by make_golden_scale.py and the parity tests ----
def bb_run_sequence(BitBirch_, files, *, branching_factor, threshold, merge_criterion, tolerance, refine_merge_criterion,
                    refine_threshold_change, refine_rounds, refine_num, recluster_rounds, before_release=None, **tree_kw):
    r"""The calls `bb run` makes, in its order (reference cli.py:1058-1121, "lean" variant): the parity tests replay
    exactly this function with bblean_amd.bitbirch.BitBirch."""
    tree = BitBirch_(branching_factor=branching_factor, threshold=threshold, merge_criterion=merge_criterion,
                     tolerance=tolerance, **tree_kw)
    for file in files:
        tree.fit(file, n_features=None, input_is_packed=True, max_fps=None)
    if recluster_rounds != 0 or refine_rounds != 0:
        tree.set_merge(refine_merge_criterion, tolerance=tolerance, threshold=threshold + refine_threshold_change)
        for _ in range(refine_rounds):
            tree.refine_inplace(files, input_is_packed=True, n_largest=refine_num)
        for _ in range(recluster_rounds):
            tree.recluster_inplace(shuffle=False)
    extra = before_release(tree) if before_release is not None else None
    tree.delete_internal_nodes()
    return tree.get_centroids_mol_ids(), extra


BBRUN_CASES = {
    # `bb run dir -b 50 -t 0.65` of the reference's tests/test_cli.py:266-308 (CLI defaults otherwise)
    "cli_golden": dict(files=[(3000, 12620509540149709235)], branching_factor=50, threshold=0.65, merge_criterion="diameter",
                       tolerance=0.05, refine_merge_criterion="tolerance-diameter", refine_threshold_change=0.0,
                       refine_rounds=0, refine_num=1, recluster_rounds=0),
    # two input files, one refinement round splitting the 2 largest clusters (re-read from the FILE LIST: ascending
    # global index, SURVEY.md section 8a rule 12), one recluster round
    "two_files_refine": dict(files=[(1800, 2001), (1700, 2002)], branching_factor=50, threshold=0.3, merge_criterion="diameter",
                             tolerance=0.05, refine_merge_criterion="tolerance-diameter", refine_threshold_change=0.0,
                             refine_rounds=1, refine_num=2, recluster_rounds=1),
}


def _sha(a) -> str:
    import hashlib

    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def leaf_bfs_digest(tree) -> dict:
    bfs = tree._get_leaf_bfs()
    ns = np.array([bf.n_samples for bf in bfs], dtype="<i8")
    cents = np.array([np.asarray(bf.packed_centroid) for bf in bfs], dtype=np.uint8)
    mols = np.array([i for bf in bfs for i in bf.mol_indices], dtype="<i8")
    ls = np.array([np.asarray(bf.linear_sum, dtype="<u8") for bf in bfs[:16]])
    names = [bf.dtype_name for bf in bfs[:16]]
    return {"k": len(bfs), "n_sha": _sha(ns), "cent_sha": _sha(cents), "mol_sha": _sha(mols), "ls16_sha": _sha(ls), "dtype16": names}


