import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from bench import synth_fake_fps
from bblean_amd import BitBirch
from oracle_engine import OracleEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
B = sys.argv[2] if len(sys.argv) > 2 else "256"
fps = synth_fake_fps(n, 1000, torch.device("cuda"))
torch.cuda.synchronize()
os.environ.pop("BBHIP_BATCH", None)
t0 = time.perf_counter(); ser = BitBirch(branching_factor=50, threshold=0.3).fit(fps); t1 = time.perf_counter()
os.environ["BBHIP_BATCH"] = B
os.environ["BBHIP_BATCH_VERBOSE"] = "1"
t2 = time.perf_counter(); bat = BitBirch(branching_factor=50, threshold=0.3).fit(fps); t3 = time.perf_counter()
print(f"serial {n/(t1-t0):.0f} fps/s   batch(B={B}) {n/(t3-t2):.0f} fps/s")
a, b = ser.get_assignments(), bat.get_assignments()
print("assignments equal:", bool((a == b).all()), " stats equal:", ser._engine.stats()[:7].tolist() == bat._engine.stats()[:7].tolist())
print(ser._engine.stats()[:7].tolist()); print(bat._engine.stats()[:7].tolist())
print("cluster lists equal:", ser.get_cluster_mol_ids() == bat.get_cluster_mol_ids(), " centroids equal:", bool((np.array(ser.get_centroids()) == np.array(bat.get_centroids())).all()))
