"""Round-1 multiround throughput: many independent shard trees in one launch.
Prints wall time, the tree kernel's own time (HIP events inside libbbhip) and launch count."""
import ctypes as C
import sys
import time

sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from bench import synth_fake_fps
from bblean_amd import BitBirch, fit_concurrently, _lib

lib = _lib.load()
n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
fps = synth_fake_fps(n_total, 1000, torch.device("cuda"))
torch.cuda.synchronize()
lib.bbh_profile_enable(1)
for shards in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,8,64,256,512,1024".split(","))]:
    per = n_total // shards
    for rep in range(2):
        t_c = time.perf_counter()
        trees = [BitBirch(branching_factor=50, threshold=0.3) for _ in range(shards)]
        parts = [fps[i * per:(i + 1) * per] for i in range(shards)]
        torch.cuda.synchronize()
        lib.bbh_profile_reset()
        t0 = time.perf_counter()
        fit_concurrently(trees, parts, reinsert_indices=[range(i * per, (i + 1) * per) for i in range(shards)])
        dt = time.perf_counter() - t0
        launches, ms = C.c_int64(0), C.c_double(0.0)
        lib.bbh_profile_get(b"tree_insert", C.byref(launches), C.byref(ms))
        del trees
    print(f"shards={shards:5d} rows/shard={per:8d} create {t0-t_c:6.3f}s fit wall {dt:7.3f}s kernel {ms.value/1e3:7.3f}s "
          f"in {launches.value} launches -> {shards*per/dt/1e6:8.3f} M fps/s wall, {shards*per/(ms.value/1e3)/1e6:8.3f} M fps/s kernel", flush=True)
