"""Single-tree fit (and bb-run style refine) throughput per workload and branching factor, one launch log line each.
    python tools/fit_workloads.py [rows] [bf ...]"""
import os, sys, time
os.environ["BBHIP_LAUNCH_LOG"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from bench import WORKLOADS
from bblean_amd import BitBirch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
bfs = [int(a) for a in sys.argv[2:]] or [50, 254]
for name, (gen, thr, _) in WORKLOADS.items():
    fps = gen(n, 1000, torch.device("cuda"))
    torch.cuda.synchronize()
    for bf in bfs:
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            t = BitBirch(branching_factor=bf, threshold=thr, merge_criterion="diameter").fit(fps)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print(f"== {name} thr {thr} bf {bf}: {n / best:.0f} fps/s ({1e6 * best / n:.2f} us/insert) stats {t._engine.stats()[:7].tolist()}", flush=True)
