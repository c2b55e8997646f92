"""Phase profile (BBHIP_PHASES) of a final-merge-round-like insertion: the round-1 BitFeature tables of several S-fake
shards inserted into one tolerance-diameter tree, table after table (buffers with n > 1 first, then the packed
singleton runs, which are the launches the phase timers cover).
    python tools/merge_round_phases.py [rows per shard] [shards]"""
import os, sys, time
os.environ["BBHIP_PHASES"] = "1"
os.environ["BBHIP_LAUNCH_LOG"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from bench import synth_fake_fps
from bblean_amd import BitBirch
from bblean_amd.multiround import _files_range_tuples, _initial_rounds

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 4
inputs = [synth_fake_fps(n, 1000 + s, torch.device("cuda")) for s in range(shards)]
infos = _files_range_tuples(inputs)
tabs = _initial_rounds(infos, branching_factor=50, threshold=0.3, tolerance=0.05, merge_criterion="diameter", refinement="full",
                       refine_merge_criterion="tolerance-diameter", refine_threshold_change=0.0, n_features=None,
                       input_is_packed=True, max_fps=None, engine_factory=None, device=0)
print("round 1 done", flush=True)
tree = BitBirch(branching_factor=50, threshold=0.3, merge_criterion="tolerance-diameter", tolerance=0.05)
t0 = time.perf_counter()
k = 0
for bufs, mols in tabs:
    for name in sorted(bufs, reverse=True):
        tree._fit_buffers(bufs[name], mols[name])
        k += len(bufs[name])
dt = time.perf_counter() - t0
print(f"{k} BitFeatures in {dt:.2f}s = {1e6 * dt / k:.2f} us/element", flush=True)
print(tree._engine.stats())
