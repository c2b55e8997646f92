"""Why bench.py's `concurrent_shards` (512 trees in one call) is 7x slower inside a full bench run than alone:
the same measurement before and after the kind of work bench.py does in between (a 5 M-row fit + refinement at bf 254,
trees created and dropped), and after the allocator caches are trimmed.  Per measurement: tree creation and fit wall time."""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from bench import synth_fake_fps, synth_ecfp
from bblean_amd import BitBirch, fit_concurrently, _lib

lib = _lib.load()
dev = torch.device("cuda")
n = 1_000_000
fps = synth_fake_fps(n, 1000, dev)
torch.cuda.synchronize()


def measure(tag):
    shards = 512
    per = n // shards
    parts = [fps[i * per:(i + 1) * per] for i in range(shards)]
    for rep in range(3):
        t0 = time.perf_counter()
        trees = [BitBirch(branching_factor=50, threshold=0.3) for _ in range(shards)]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fit_concurrently(trees, parts, reinsert_indices=[range(i * per, (i + 1) * per) for i in range(shards)])
        t2 = time.perf_counter()
        del trees
        t3 = time.perf_counter()
        free, total = torch.cuda.mem_get_info()
        print(f"== {tag} rep {rep}: create {t1 - t0:.4f} s, fit {t2 - t1:.4f} s, destroy {t3 - t2:.4f} s; device memory in use {(total - free) / 2**30:.1f} GiB", flush=True)


measure("fresh process")
c3 = synth_ecfp(5_000_000, 7, dev)
t = BitBirch(branching_factor=254, threshold=0.3, merge_criterion="diameter").fit(c3)
t.set_merge("tolerance-diameter", tolerance=0.05, threshold=0.3)
t.refine_inplace(c3, n_largest=1)
lab = t.get_assignments()
measure("big tree alive")
del t, lab, c3
measure("big tree dropped")
torch.cuda.empty_cache()
measure("torch cache emptied")
lib.bbh_trim_cache()
measure("library cache trimmed")
