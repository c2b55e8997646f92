import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from test_hip_pipe_fuzz import _rows
from bblean_amd import BitBirch
from oracle_engine import OracleEngine
seed = int(sys.argv[1]); reps = int(sys.argv[2])
os.environ["BBHIP_SYS"] = "1"
def canon(logs):
    cat = np.concatenate([np.asarray(l) for l in logs]).astype(np.int64)
    _, first, inv = np.unique(cat, return_index=True, return_inverse=True)
    rank = np.empty(first.size, dtype=np.int64); rank[np.argsort(first, kind="stable")] = np.arange(first.size)
    return rank[inv]
rng = np.random.default_rng(7000 + seed)
bf = 50 if seed % 3 else 254
n = int(rng.integers(12_000, 60_000))
crit = "diameter" if rng.random() < 0.6 else "tolerance-diameter"
thr = float(rng.uniform(0.15, 0.8)); tol = float(rng.uniform(0.0, 0.1))
rows = _rows(rng, n)
cuts = sorted(set(int(c) for c in rng.integers(8_200, n + 1, int(rng.integers(0, 5)))) | {0, n})
if seed % 4 == 0: os.environ["BBHIP_TINY_POOLS"] = "1"
kw = dict(branching_factor=bf, threshold=thr, merge_criterion=crit, tolerance=tol)
ora = BitBirch(_engine_factory=OracleEngine, **kw)
for lo, hi in zip(cuts[:-1], cuts[1:]): ora.fit(rows[lo:hi])
so = ora._engine.stats()[:7].tolist()
bad = 0
for rep in range(reps):
    hip = BitBirch(**kw)
    try:
        for lo, hi in zip(cuts[:-1], cuts[1:]): hip.fit(rows[lo:hi])
    except Exception as exc:
        print(f"rep {rep}: {exc!r}"[:300], flush=True); bad += 1; continue
    raw = [int((np.asarray(a) != np.asarray(b)).sum()) for a, b in zip(hip._log_leaf, ora._log_leaf)]
    cd = int((canon(hip._log_leaf) != canon(ora._log_leaf)).sum())
    sh = hip._engine.stats()[:7].tolist()
    if any(raw) or cd or sh != so:
        bad += 1
        firsts = [int(np.argmax(np.asarray(a) != np.asarray(b))) if r else -1 for a, b, r in zip(hip._log_leaf, ora._log_leaf, raw)]
        print(f"rep {rep}: raw diffs per call {raw} (first {firsts}) canonical diffs {cd} stats equal {sh == so}\n   hip {sh}\n   ora {so}\n   sys {hip._engine.sys_counts().tolist()}", flush=True)
print(f"seed {seed}: {bad} bad of {reps}")
