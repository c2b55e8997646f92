"""One seed of tests/test_hip_pipe_fuzz.py under BBHIP_SYS=1 with more to look at than an assert.
    python tools/sys_fuzz_debug.py <seed> [<seed> ...]"""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from test_hip_pipe_fuzz import _rows
from bblean_amd import BitBirch
from oracle_engine import OracleEngine


def canon(logs):
    cat = np.concatenate([np.asarray(l) for l in logs]).astype(np.int64)
    _, first, inv = np.unique(cat, return_index=True, return_inverse=True)
    rank = np.empty(first.size, dtype=np.int64)
    rank[np.argsort(first, kind="stable")] = np.arange(first.size)
    return rank[inv]


for seed in [int(a) for a in sys.argv[1:]] or [6]:
    rng = np.random.default_rng(7000 + seed)
    bf = 50 if seed % 3 else 254
    n = int(rng.integers(12_000, 60_000))
    crit = "diameter" if rng.random() < 0.6 else "tolerance-diameter"
    thr = float(rng.uniform(0.15, 0.8))
    tol = float(rng.uniform(0.0, 0.1))
    rows = _rows(rng, n)
    cuts = sorted(set(int(c) for c in rng.integers(8_200, n + 1, int(rng.integers(0, 5)))) | {0, n})
    tiny = seed % 4 == 0
    if tiny:
        os.environ["BBHIP_TINY_POOLS"] = "1"
    else:
        os.environ.pop("BBHIP_TINY_POOLS", None)
    kw = dict(branching_factor=bf, threshold=thr, merge_criterion=crit, tolerance=tol)
    print(f"seed {seed}: bf {bf} n {n} {crit} thr {thr:.3f} tol {tol:.3f} cuts {cuts} tiny {tiny}", flush=True)
    for mode in ("1", "0"):
        os.environ["BBHIP_SYS"] = mode
        hip, ora = BitBirch(**kw), BitBirch(_engine_factory=OracleEngine, **kw)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            try:
                hip.fit(rows[lo:hi])
            except Exception as exc:
                print(f"   BBHIP_SYS={mode} [{lo},{hi}): {exc!r}"[:300])
                break
            ora.fit(rows[lo:hi])
            raw = np.nonzero(np.asarray(hip._log_leaf[-1]) != np.asarray(ora._log_leaf[-1]))[0]
            ch, co = canon(hip._log_leaf), canon(ora._log_leaf)
            cd = np.nonzero(ch != co)[0]
            sh, so = hip._engine.stats()[:7].tolist(), ora._engine.stats()[:7].tolist()
            print(f"   BBHIP_SYS={mode} [{lo},{hi}): raw id diffs {raw.size} (first {lo + int(raw[0]) if raw.size else -1}), canonical diffs {cd.size} (first {int(cd[0]) if cd.size else -1}), "
                  f"stats equal {sh == so}\n      hip {sh}\n      ora {so}\n      kernel_counts {hip._engine.kernel_counts().tolist()} sys {hip._engine.sys_counts().tolist()}", flush=True)
            if raw.size:
                i = int(raw[0])
                print(f"      around the first raw difference: hip {np.asarray(hip._log_leaf[-1])[i-3:i+6].tolist()} ora {np.asarray(ora._log_leaf[-1])[i-3:i+6].tolist()}")
