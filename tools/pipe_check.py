"""HIP engine (pipelined kernel by default) against the CPU oracle, element by element, on the bench workloads.
    python tools/pipe_check.py [rows] [bf ...]"""
import os, sys, time
os.environ.pop("BBHIP_LAUNCH_LOG", None)  # (any value, "0" included, turns the log on)
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import torch
from bench import WORKLOADS
from bblean_amd import BitBirch
from oracle_engine import OracleEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000
bfs = [int(a) for a in sys.argv[2:]] or [50, 254]
bad = 0
for name, (gen, thr, _) in WORKLOADS.items():
    fps = gen(n, 4321, torch.device("cuda"))
    host = fps.cpu().numpy()
    for bf in bfs:
        for crit in ("diameter", "tolerance-diameter"):
            t0 = time.perf_counter()
            hip = BitBirch(branching_factor=bf, threshold=thr, merge_criterion=crit).fit(fps)
            dt = time.perf_counter() - t0
            ora = BitBirch(branching_factor=bf, threshold=thr, merge_criterion=crit, _engine_factory=OracleEngine).fit(host)
            lh, lo = hip._log_leaf[-1], ora._log_leaf[-1]
            same_leaf = bool((np.asarray(lh) == np.asarray(lo)).all())
            first = int(np.argmax(np.asarray(lh) != np.asarray(lo))) if not same_leaf else -1
            sh, so = hip._engine.stats()[:7].tolist(), ora._engine.stats()[:7].tolist()
            same_asg = bool((hip.get_assignments() == ora.get_assignments()).all())
            same_cent = bool((np.array(hip.get_centroids()) == np.array(ora.get_centroids())).all())
            ok = same_leaf and sh == so and same_asg and same_cent
            bad += not ok
            print(f"{'OK ' if ok else 'BAD'} {name} bf {bf} {crit}: {n / dt:.0f} fps/s leaf {same_leaf} (first diff {first}) asg {same_asg} cent {same_cent}\n    hip {sh}\n    ora {so}", flush=True)
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
