import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import numpy as np, torch
from cases import clustered_hier
from bblean_amd import BitBirch
n, S, k, thr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
B = sys.argv[5] if len(sys.argv) > 5 else "512"
fps = torch.from_numpy(clustered_hier(n, 2048, S, k, seed=1)).cuda()
os.environ.pop("BBHIP_BATCH", None)
t0 = time.perf_counter(); ser = BitBirch(branching_factor=50, threshold=thr).fit(fps); t1 = time.perf_counter()
os.environ["BBHIP_BATCH"] = B; os.environ["BBHIP_BATCH_VERBOSE"] = "1"
t2 = time.perf_counter(); bat = BitBirch(branching_factor=50, threshold=thr).fit(fps); t3 = time.perf_counter()
print(f"n={n} S={S} k={k} thr={thr}: serial {n/(t1-t0):.0f} fps/s  batch(B={B}) {n/(t3-t2):.0f} fps/s  clusters={len(ser._leaves()['ids'])} depth={ser._engine.stats()[6]}")
print("equal:", bool((ser.get_assignments() == bat.get_assignments()).all()), ser._engine.stats()[:7].tolist() == bat._engine.stats()[:7].tolist())
