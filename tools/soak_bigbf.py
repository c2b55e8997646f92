"""Randomised parity soak of the bf > 255 node compare (node_best's block path, bb_tree.hip): random branching factors in
256..1023, all six merge criteria, four workloads, 20-70 k rows, HIP against the oracle (assignments, counters, centroids).
    python tools/soak_bigbf.py [first seed] [cases]"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import numpy as np
import torch

from bench import WORKLOADS
from oracle_engine import OracleEngine
from bblean_amd import BitBirch

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
CRITS = ["diameter", "radius", "tolerance-diameter", "tolerance-radius", "tolerance-legacy", "never-merge"]
bad = 0
for seed in range(first, first + cases):
    rng = np.random.default_rng(90_000 + seed)
    bf = int(rng.integers(256, 1024))
    crit = CRITS[seed % len(CRITS)]
    wl = ["fake", "ecfp", "zipf", "hier"][int(rng.integers(0, 4))]
    n = int(rng.integers(20_000, 70_001))
    gen, thr0, _ = WORKLOADS[wl]
    thr = float(np.clip(thr0 + rng.uniform(-0.1, 0.15), 0.15, 0.85))
    fps = gen(n, 5000 + seed, torch.device("cuda"))
    kw = dict(branching_factor=bf, threshold=thr, merge_criterion=crit)
    if crit.startswith("tolerance"):
        kw["tolerance"] = float(rng.choice([0.0, 0.05, 0.1]))
    if crit == "never-merge":
        n = min(n, 30_000)  # (every row its own cluster: the widest nodes per row inserted)
        fps = fps[:n]
    t0 = time.perf_counter()
    hip = BitBirch(**kw).fit(fps)
    t1 = time.perf_counter()
    ora = BitBirch(**kw, _engine_factory=OracleEngine).fit(fps.cpu().numpy())
    t2 = time.perf_counter()
    lv_h, lv_o = hip._leaves(), ora._leaves()
    ok = (bool((hip.get_assignments() == ora.get_assignments()).all())
          and hip._engine.stats()[:7].tolist() == ora._engine.stats()[:7].tolist()
          and bool((lv_h["cents"] == lv_o["cents"]).all()) and bool((lv_h["n"] == lv_o["n"]).all()))
    bad += not ok
    st = hip._engine.stats()
    print(f"seed {seed:3d} bf {bf:4d} {crit:18s} {wl:5s} n {n:6d} thr {thr:.3f}: {'ok' if ok else 'MISMATCH'}  "
          f"clusters {int(st[3])} splits {int(st[4])} depth {int(st[6])}  hip {t1 - t0:.2f} s oracle {t2 - t1:.2f} s", flush=True)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
