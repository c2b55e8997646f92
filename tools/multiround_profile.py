"""Where a file-based multiround run spends its time: per-round wall, tree-kernel launches / elements / seconds
(bbh_profile_*), and the Python functions on top of the host profile.
    python tools/multiround_profile.py [n] [files] [branching_factor]"""
import cProfile, ctypes as C, pstats, sys, tempfile, time
from pathlib import Path

sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import torch

from bench import synth_fake_fps
from bblean_amd import _lib
from bblean_amd import multiround as mr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
files = int(sys.argv[2]) if len(sys.argv) > 2 else 64
bf = int(sys.argv[3]) if len(sys.argv) > 3 else 50
lib = _lib.load()
fps = synth_fake_fps(n, 1000, torch.device("cuda")).cpu().numpy()


def prof(tag):
    out = []
    for name in (b"tree_insert", b"gather_leaves"):
        l, ms, u = C.c_int64(0), C.c_double(0.0), C.c_int64(0)
        lib.bbh_profile_get(name, C.byref(l), C.byref(ms))
        lib.bbh_profile_units(name, C.byref(u))
        out.append(f"{name.decode()}: {l.value} launches {ms.value / 1e3:.3f}s {u.value} elems"
                   + (f" ({1e3 * ms.value / max(u.value, 1):.2f} us/elem)" if u.value else ""))
    lib.bbh_profile_reset()
    print(f"[{tag}] " + "; ".join(out), flush=True)


orig_merge, orig_init = mr._merge_rounds, mr._initial_rounds


def timed(fn, tag):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        print(f"[{tag}] {time.perf_counter() - t0:.3f}s", flush=True)
        prof(tag)
        return r
    return w


mr._merge_rounds = timed(orig_merge, "merge_rounds")
mr._initial_rounds = timed(orig_init, "initial_rounds")
with tempfile.TemporaryDirectory() as d:
    d = Path(d)
    per = n // files
    names = []
    for i in range(files):
        f = d / f"fps.{i:05d}.npy"
        np.save(f, fps[i * per:(i + 1) * per])
        names.append(f)
    out = d / "out"
    out.mkdir()
    lib.bbh_profile_enable(1)
    lib.bbh_profile_reset()
    pr = cProfile.Profile()
    pr.enable()
    t0 = time.perf_counter()
    timer = mr.run_multiround_bitbirch(names, out, branching_factor=bf, threshold=0.3, num_initial_processes=1)
    dt = time.perf_counter() - t0
    pr.disable()
print(f"total {dt:.2f}s", {k: round(v, 3) for k, v in timer.timings.items()})
pstats.Stats(pr).sort_stats("cumulative").print_stats(60)
