"""A/B of library builds on the two numbers that regressed between rounds 3 and 4 (one process per library):
    BBHIP_LIBRARY=var_so/lib_<commit>.so python tools/bisect_regress.py [rows]
Prints (a) 512 independent trees in one call (bench.py `concurrent_shards`), (b) one tree at bf 254 on the same S-fake rows.
Tolerates libraries that predate some of today's entry points (the prototypes of missing symbols are skipped)."""
import ctypes as C
import os
import sys
import time

ROOT = os.environ.get("BB_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import torch  # (first: its HIP runtime must be the one the library binds to)
from bblean_amd import _lib

_orig = dict(_lib._PROTOTYPES)
_probe = C.CDLL(str(_lib.library_path()))
for name in list(_lib._PROTOTYPES):
    if not hasattr(_probe, name):
        del _lib._PROTOTYPES[name]
        print(f"(library has no {name})")
from bench import synth_fake_fps
from bblean_amd import BitBirch, fit_concurrently

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
fps = synth_fake_fps(n, 1000, torch.device("cuda"))
torch.cuda.synchronize()
tag = os.path.basename(str(_lib.library_path()))
only = os.environ.get("BISECT_ONLY", "")
for shards in (() if only == "bf254" else (512, 64)):
    per = n // shards
    parts = [fps[i * per:(i + 1) * per] for i in range(shards)]
    best = None
    for _ in range(3):
        trees = [BitBirch(branching_factor=50, threshold=0.3) for _ in range(shards)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fit_concurrently(trees, parts, reinsert_indices=[range(i * per, (i + 1) * per) for i in range(shards)])
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        del trees
    print(f"== {tag} concurrent_shards {shards}: {best:.4f} s = {shards * per / best / 1e6:.2f} M fps/s", flush=True)
for bf in ((254,) if only == "bf254" else (254, 50)):
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        t = BitBirch(branching_factor=bf, threshold=0.3, merge_criterion="diameter").fit(fps)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"== {tag} single tree bf {bf}: {n / best:.0f} fps/s", flush=True)
