import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from bench import WORKLOADS
from bblean_amd import BitBirch
gen, thr, _ = WORKLOADS["fake"]
for seed in (1000, 5000, 5001, 5002, 7, 123456):
    fps = gen(1_000_000, seed, torch.device("cuda"))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t = BitBirch(branching_factor=50, threshold=thr, merge_criterion="diameter").fit(fps)
    dt = time.perf_counter() - t0
    kc = t._engine.kernel_counts()
    print(f"== seed {seed}: {1e6/dt:.0f} fps/s pipe/fast/complete {kc[:3].tolist()} launches {kc[3:6].tolist()} unsup {int(kc[6])} ml {t._engine.stats()[6]}", flush=True)
