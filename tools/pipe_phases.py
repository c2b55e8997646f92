"""Phase timers of the pipelined kernel (BBHIP_PIPE_PHASES=1: the PROF instance of k_tree_pipe).
    python tools/pipe_phases.py [rows] [workload: fake|ecfp|rdkit|hier] [branching factor: 50|254]"""
import os, sys, time
os.environ["BBHIP_PIPE_PHASES"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from bench import WORKLOADS
from bblean_amd import BitBirch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
workload = sys.argv[2] if len(sys.argv) > 2 else "fake"
bf = int(sys.argv[3]) if len(sys.argv) > 3 else 50
gen, thr, _ = WORKLOADS[workload]
fps = gen(n, 1000, torch.device("cuda"))
torch.cuda.synchronize()
t0 = time.perf_counter()
t = BitBirch(branching_factor=bf, threshold=thr, merge_criterion="diameter").fit(fps)
dt = time.perf_counter() - t0
print(f"{workload} bf {bf} thr {thr}: {n/dt:.0f} fps/s  ({dt/n*1e6:.2f} us/insert)", t._engine.stats())
