"""Where the level-systolic kernel's owners spend their cycles (the phase-timer instance: BBHIP_SYS_PHASES=1).
    python tools/sys_phases.py [rows] [workloads, comma separated] [bf ...]"""
import os, sys, time
os.environ["BBHIP_SYS"] = os.environ.get("BBHIP_SYS", "1")
os.environ["BBHIP_SYS_PHASES"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from bench import WORKLOADS
from bblean_amd import BitBirch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["zipf", "hier"]
bfs = [int(a) for a in sys.argv[3:]] or [50, 254]
for name in names:
    gen, thr, _ = WORKLOADS[name]
    fps = gen(n, 4321, torch.device("cuda"))
    for bf in bfs:
        print(f"== {name} bf {bf}, {n} rows", flush=True)
        sys.stderr.flush()
        t = BitBirch(branching_factor=bf, threshold=thr, merge_criterion="diameter")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t.fit(fps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sys.stderr.flush()
        print(f"   {n / dt:.0f} fingerprints/s (phase-timer instance); sys_counts {t._engine.sys_counts().tolist()} stats {t._engine.stats().tolist()}", flush=True)
