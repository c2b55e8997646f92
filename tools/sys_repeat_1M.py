"""The systolic kernel (BBHIP_SYS=1) against the default path of the same build, repeated: 1 M rows, zipf / hier, bf 50 / 254.
    python tools/sys_repeat_1M.py [reps] [rows]
Labels, centroids and engine counters must be identical in every repeat (the default path is pinned against the oracle by the
test suite; this is about run-to-run stability of a kernel whose workgroups talk through memory)."""
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from bench import WORKLOADS
from bblean_amd import BitBirch
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
bad = 0
for name in ("zipf", "hier"):
    gen, thr, _ = WORKLOADS[name]
    fps = gen(n, 4321, torch.device("cuda"))
    for bf in (50, 254):
        os.environ["BBHIP_SYS"] = "0"
        ref = BitBirch(branching_factor=bf, threshold=thr, merge_criterion="diameter").fit(fps)
        ra, rs, rl = ref.get_assignments(), ref._engine.stats()[:7].tolist(), np.asarray(ref._log_leaf[-1])
        os.environ["BBHIP_SYS"] = "1"
        rates = []
        for rep in range(reps):
            t = BitBirch(branching_factor=bf, threshold=thr, merge_criterion="diameter")
            torch.cuda.synchronize(); t0 = time.perf_counter()
            try:
                t.fit(fps)
            except Exception as exc:
                print(f"BAD {name} bf {bf} rep {rep}: {exc!r}"[:300], flush=True); bad += 1; continue
            torch.cuda.synchronize(); rates.append(n / (time.perf_counter() - t0))
            ok = bool((t.get_assignments() == ra).all()) and t._engine.stats()[:7].tolist() == rs and bool((np.asarray(t._log_leaf[-1]) == rl).all())
            if not ok:
                bad += 1
                d = np.nonzero(np.asarray(t._log_leaf[-1]) != rl)[0]
                print(f"BAD {name} bf {bf} rep {rep}: differs from the default path (first element {int(d[0]) if d.size else -1}, {d.size} ids) stats {t._engine.stats()[:7].tolist()} vs {rs}", flush=True)
        print(f"{name} bf {bf}: {reps} repeats, {min(rates):.0f} - {max(rates):.0f} fingerprints/s, sys_counts {t._engine.sys_counts()[:4].tolist()}", flush=True)
print("FAILED" if bad else "ALL IDENTICAL", bad)
