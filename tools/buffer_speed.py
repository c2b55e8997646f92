"""Throughput of BitFeature-buffer insertion (`_fit_buffers`, reference bitbirch.py:790-866) into one tree: the leaf
BitFeatures of a fitted tree re-inserted with the tolerance-diameter criterion (what refinement and the merge rounds
do), split into the buffer path (n_samples > 1) and the packed-singleton path.
    python tools/buffer_speed.py [n_rows]"""
import ctypes as C, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import torch
from bench import synth_fake_fps
from bblean_amd import BitBirch, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
lib = _lib.load()
fps = synth_fake_fps(n, 1000, torch.device("cuda"))
t = BitBirch(branching_factor=50, threshold=0.3).fit(fps)
bufs, mols = t._bf_tables(t._leaf_order(True))
u8 = np.asarray(bufs["uint8"])
multi = u8[u8[:, -1] > 1]
single = u8[u8[:, -1] == 1]
print(f"{len(u8)} uint8 BitFeatures: {len(multi)} with n > 1, {len(single)} singletons; other tables: "
      f"{[(k, len(v)) for k, v in bufs.items() if k != 'uint8']}")


def run(tag, table):
    lib.bbh_profile_enable(1)
    lib.bbh_profile_reset()
    tree = BitBirch(branching_factor=50, threshold=0.3, merge_criterion="tolerance-diameter", tolerance=0.05)
    tree._num_fitted_fps = 1 << 40  # "omit" index sequences
    t0 = time.perf_counter()
    tree._fit_buffers(table)
    dt = time.perf_counter() - t0
    l, ms, u = C.c_int64(0), C.c_double(0.0), C.c_int64(0)
    lib.bbh_profile_get(b"tree_insert", C.byref(l), C.byref(ms))
    lib.bbh_profile_units(b"tree_insert", C.byref(u))
    print(f"{tag}: {len(table)} elements, wall {dt:.3f}s, kernel {ms.value / 1e3:.3f}s in {l.value} launches = "
          f"{1e3 * ms.value / max(u.value, 1):.2f} us/element")


run("buffers n>1 (uint8)", multi)
run("singletons (packed path)", single)
for k, v in bufs.items():
    if k != "uint8":
        run(f"buffers {k}", np.asarray(v))
