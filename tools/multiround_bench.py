"""`bb multiround`-style clustering of N synthetic fingerprints split into FILES shard files on ONE
GPU (reference multiround.py:333-484 semantics: round 1 per shard with full refinement, one merge
round in bins of 10, final merge).  Prints the per-round wall times.

    python tools/multiround_bench.py [n] [files] [branching_factor]"""
import sys, tempfile, time
from pathlib import Path

sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import torch

from bench import synth_fake_fps
from bblean_amd.multiround import run_multiround_bitbirch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
files = int(sys.argv[2]) if len(sys.argv) > 2 else 64
bf = int(sys.argv[3]) if len(sys.argv) > 3 else 50
fps = synth_fake_fps(n, 1000, torch.device("cuda")).cpu().numpy()
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
    t0 = time.perf_counter()
    timer = run_multiround_bitbirch(names, out, branching_factor=bf, threshold=0.3, num_initial_processes=1)
    dt = time.perf_counter() - t0
    import pickle
    clusters = pickle.load(open(out / "clusters.pkl", "rb"))
print(f"multiround n={files * per} files={files} bf={bf}: total {dt:.2f}s -> {files * per / dt:.0f} fps/s; "
      f"{len(clusters)} clusters; rounds:", {k: round(v, 2) for k, v in getattr(timer, '_times', getattr(timer, 'times', {})).items()} if hasattr(timer, '_times') or hasattr(timer, 'times') else vars(timer))
