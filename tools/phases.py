import os, sys, time
os.environ["BBHIP_PHASES"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from bench import synth_fake_fps
from bblean_amd import BitBirch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
fps = synth_fake_fps(n, 1000, torch.device("cuda"))
torch.cuda.synchronize()
t0 = time.perf_counter()
t = BitBirch(branching_factor=50, threshold=0.3).fit(fps)
dt = time.perf_counter() - t0
print(f"{n/dt:.0f} fps/s  ({dt/n*1e6:.2f} us/insert)", t._engine.stats())
