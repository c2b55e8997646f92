"""`fit_concurrently` of several large shards (multiround round 1 on one GPU): seconds, elements by kernel.
    python tools/multi_fit.py [workload] [shards] [rows per shard] [bf]        (BBHIP_NO_PIPE_MULTI=1: steady-state kernel only)"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from bench import WORKLOADS
from bblean_amd import BitBirch, fit_concurrently
w = sys.argv[1] if len(sys.argv) > 1 else "rdkit"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
bf = int(sys.argv[4]) if len(sys.argv) > 4 else 254
gen, thr, _ = WORKLOADS[w]
shards = [gen(n, 5000 + i, torch.device("cuda")) for i in range(k)]
torch.cuda.synchronize()
trees = [BitBirch(branching_factor=bf, threshold=thr, merge_criterion="diameter") for _ in shards]
t0 = time.perf_counter()
fit_concurrently(trees, shards)
dt = time.perf_counter() - t0
kc = np.sum([t._engine.kernel_counts() for t in trees], axis=0)
print(f"== {w} bf {bf}: {k} x {n} rows in {dt:.2f} s = {k * n / dt:.0f} fingerprints/s; elements pipe/fast/complete {kc[:3].tolist()} "
      f"launches {kc[3:6].tolist()} unsupported-shape stops {int(kc[6])} pool stops {int(kc[7])}")
