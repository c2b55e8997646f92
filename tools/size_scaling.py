"""How the cost of an insertion changes with the size of the tree (merge rounds at 20 M elements cost 1.35x
the fit of 1 M rows at the same bf): one tree, `chunks` fit calls of `rows` rows each; per call the rate, the engine counters of
the call (merges, appends, leaf splits, node splits), which kernel inserted how much, pool-exhaustion stops and compactions.
    python tools/size_scaling.py [workload=ecfp] [rows=1000000] [chunks=10] [bf=254]"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import torch
from bench import WORKLOADS
from bblean_amd import BitBirch

wl = sys.argv[1] if len(sys.argv) > 1 else "ecfp"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 10
bf = int(sys.argv[4]) if len(sys.argv) > 4 else 254
crit = sys.argv[5] if len(sys.argv) > 5 else "diameter"
gen, thr, _ = WORKLOADS[wl]
dev = torch.device("cuda")
t = BitBirch(branching_factor=bf, threshold=thr, merge_criterion=crit, tolerance=0.05)
prev_s = np.zeros(8, dtype=np.int64)
prev_k = np.zeros(8, dtype=np.int64)
for c in range(chunks):
    fps = gen(rows, 3000 + c, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t.fit(fps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s = t._engine.stats().astype(np.int64)
    k = t._engine.kernel_counts().astype(np.int64)
    m = t._engine.memory()
    ds, dk = s - prev_s, k - prev_k
    prev_s, prev_k = s, k
    print(f"== {wl} bf {bf} {crit} chunk {c}: {rows / dt:.0f} fps/s ({1e6 * dt / rows:.2f} us/insert); merges {ds[2]} appends {ds[3]} leaf+node splits {ds[4]} "
          f"nodes {s[5]} depth {s[6]}; by kernel pipe/fast/complete {dk[0]}/{dk[1]}/{dk[2]} launches {dk[3]}/{dk[4]}/{dk[5]} unsupported {dk[6]} pool stops {dk[7]}; "
          f"node pools {int(m[1]) / 1e9:.2f}/{int(m[0]) / 1e9:.2f} GB cf {int(m[2]) / 1e9:.2f} GB gc {int(m[4])} sealed {int(m[5])} thawed {int(m[7])}", flush=True)
    del fps
