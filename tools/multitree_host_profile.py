"""Where the host time of fit_concurrently goes (prepare / C ABI call / commit)."""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from bench import synth_fake_fps
from bblean_amd import BitBirch
from bblean_amd._engine import HipEngine

n_total, shards = 4_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 512
fps = synth_fake_fps(n_total, 1000, torch.device("cuda"))
per = n_total // shards
for rep in range(2):
    trees = [BitBirch(branching_factor=50, threshold=0.3) for _ in range(shards)]
    parts = [fps[i * per:(i + 1) * per] for i in range(shards)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prepared = [t._prepare_fit(X, range(i * per, (i + 1) * per), True, None, None) for i, (t, X) in enumerate(zip(trees, parts))]
    t1 = time.perf_counter()
    leaves = HipEngine.fit_packed_many([t._engine for t in trees], [r for r, _ in prepared])
    t2 = time.perf_counter()
    for t, leaf, (_, ids) in zip(trees, leaves, prepared):
        t._commit_fit(leaf, ids)
    t3 = time.perf_counter()
    print(f"rep {rep}: prepare {t1-t0:.3f}s  fit_packed_many {t2-t1:.3f}s  commit {t3-t2:.3f}s")
    del trees
