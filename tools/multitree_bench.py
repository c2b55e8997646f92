import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from bench import synth_fake_fps
from bblean_amd import BitBirch, fit_concurrently
n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
fps = synth_fake_fps(n_total, 1000, torch.device("cuda"))
torch.cuda.synchronize()
for shards in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,8,64,256,512,1024".split(","))]:
    per = n_total // shards
    trees = [BitBirch(branching_factor=50, threshold=0.3) for _ in range(shards)]
    parts = [fps[i * per:(i + 1) * per] for i in range(shards)]
    t0 = time.perf_counter()
    fit_concurrently(trees, parts, reinsert_indices=[range(i * per, (i + 1) * per) for i in range(shards)])
    dt = time.perf_counter() - t0
    print(f"shards={shards:5d} rows/shard={per:8d} round-1 {dt:7.3f}s -> {shards*per/dt/1e6:8.3f} M fps/s", flush=True)
    del trees
