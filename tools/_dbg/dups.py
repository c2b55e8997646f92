import os, sys
os.environ["BBHIP_SYS"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from bblean_amd import BitBirch
from oracle_engine import OracleEngine
rng = np.random.default_rng(5150)
F = 2048
parts = []
for k in range(40):
    base = rng.random((1, F)) < rng.uniform(0.02, 0.4)
    parts.append(np.repeat(base, int(rng.integers(100, 500)), axis=0))
    parts.append(rng.random((int(rng.integers(200, 900)), F)) < rng.uniform(0.02, 0.5))
parts.append(np.zeros((50, F), dtype=bool)); parts.append(np.ones((30, F), dtype=bool))
rows = np.ascontiguousarray(np.packbits(np.concatenate(parts), axis=1))
n = rows.shape[0]
print("rows", n)
for bf in (50, 254):
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
        kw = dict(branching_factor=bf, threshold=0.5, merge_criterion="diameter")
        hip, ora = BitBirch(**kw), BitBirch(_engine_factory=OracleEngine, **kw)
        for lo, hi in ((0, 9000), (9000, n)):
            try:
                hip.fit(rows[lo:hi])
            except Exception as exc:
                print(f"bf {bf} rep {rep} [{lo},{hi}): {exc!r}"[:250], "kc", hip._engine.kernel_counts().tolist(), "sys", hip._engine.sys_counts().tolist(), "stats", hip._engine.stats().tolist(), flush=True)
                break
            ora.fit(rows[lo:hi])
            d = np.nonzero(np.asarray(hip._log_leaf[-1]) != np.asarray(ora._log_leaf[-1]))[0]
            sh, so = hip._engine.stats()[:7].tolist(), ora._engine.stats()[:7].tolist()
            print(f"bf {bf} rep {rep} [{lo},{hi}): id diffs {d.size} stats equal {sh == so} {sh} {so} sys {hip._engine.sys_counts()[:4].tolist()}", flush=True)
