"""One tree, N rows of a workload at a branching factor: every leaf BitFeature must have n_samples >= 1, the n_samples must add
up to N and agree with the host's member lists (a tree whose pools were grown or whose chain was exported wrongly shows up here).
    python tools/tree_integrity.py [rows] [workload] [bf] [chunk rows, 0 = one fit call]"""
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from bench import WORKLOADS
from bblean_amd import BitBirch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
workload = sys.argv[2] if len(sys.argv) > 2 else "ecfp"
bf = int(sys.argv[3]) if len(sys.argv) > 3 else 254
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 0
gen, thr, _ = WORKLOADS[workload]
fps = gen(n, 1000, torch.device("cuda"))
torch.cuda.synchronize()
t0 = time.perf_counter()
t = BitBirch(branching_factor=bf, threshold=thr, merge_criterion="diameter")
for lo in range(0, n, chunk or n):
    t.fit(fps[lo:lo + (chunk or n)])
dt = time.perf_counter() - t0
lv = t._leaves()
sizes = (lv["end"] - lv["beg"]).astype(np.int64)
nn = lv["n"].astype(np.int64)
free, total = torch.cuda.mem_get_info()
print(f"HBM in use after the fit {(total - free) / 2**30:.1f} GiB (input {fps.numel() / 2**30:.1f} GiB)")
print(f"{workload} bf {bf}: {n / dt:.0f} fps/s; leaves {nn.size}, min n {nn.min()}, sum n {nn.sum()} (rows {n}), host/device size mismatches {(sizes != nn).sum()}; "
      f"stats {t._engine.stats().tolist()} kernels {t._engine.kernel_counts().tolist()}", flush=True)
bad = np.nonzero(sizes != nn)[0]
if bad.size:
    print("first bad leaf positions", bad[:10].tolist(), "device n", nn[bad[:10]].tolist(), "host", sizes[bad[:10]].tolist(), "ids", lv["ids"][bad[:10]].tolist())
    sys.exit(1)
