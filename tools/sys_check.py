"""The level-systolic kernel (BBHIP_SYS=1) against the CPU oracle, element by element, and its speed next to the default path.
    python tools/sys_check.py [rows] [chunk] [workloads, comma separated] [bf ...]
BitFeature ids are handed out by whichever leaf owner appends first, so they are compared after renumbering both sides by first
appearance in the element stream (the oracle's ids ARE in that order: an id's first element is the one that created it)."""
import os, sys, time
os.environ.pop("BBHIP_LAUNCH_LOG", None)
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import torch
from bench import WORKLOADS
from bblean_amd import BitBirch
from oracle_engine import OracleEngine


def canon(logs):
    cat = np.concatenate([np.asarray(l) for l in logs]).astype(np.int64)
    _, first, inv = np.unique(cat, return_index=True, return_inverse=True)
    rank = np.empty(first.size, dtype=np.int64)
    rank[np.argsort(first, kind="stable")] = np.arange(first.size)
    return rank[inv]


n = int(sys.argv[1]) if len(sys.argv) > 1 else 120_000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
names = sys.argv[3].split(",") if len(sys.argv) > 3 else ["zipf", "hier"]
bfs = [int(a) for a in sys.argv[4:]] or [50, 254]
bad = 0
for name in names:
    gen, thr, _ = WORKLOADS[name]
    fps = gen(n, 4321, torch.device("cuda"))
    host = fps.cpu().numpy()
    for bf in bfs:
        kw = dict(branching_factor=bf, threshold=thr, merge_criterion="diameter")
        ora = BitBirch(_engine_factory=OracleEngine, **kw)
        t0 = time.perf_counter()
        for lo in range(0, n, chunk):
            ora.fit(host[lo:lo + chunk])
        t_ora = time.perf_counter() - t0
        res = {}
        for mode in ("1", "0"):
            os.environ["BBHIP_SYS"] = mode
            hip = BitBirch(**kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            err = None
            try:
                for lo in range(0, n, chunk):
                    hip.fit(fps[lo:lo + chunk])
            except Exception as exc:  # a wait that gave up, an internal error: report and go on
                err = repr(exc)[:300]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if err is not None:
                print(f"BAD {name} bf {bf} BBHIP_SYS={mode}: {err}", flush=True)
                bad += 1
                continue
            ch, co = canon(hip._log_leaf), canon(ora._log_leaf)
            same_leaf = bool((ch == co).all())
            raw_ids = all(bool((np.asarray(a) == np.asarray(b)).all()) for a, b in zip(hip._log_leaf, ora._log_leaf))  # (after the renumbering: the very ids)
            first = int(np.argmax(ch != co)) if not same_leaf else -1
            sh, so = hip._engine.stats()[:7].tolist(), ora._engine.stats()[:7].tolist()
            same_asg = bool((hip.get_assignments() == ora.get_assignments()).all())
            same_cent = bool((np.array(hip.get_centroids()) == np.array(ora.get_centroids())).all())
            same_ids = hip.get_cluster_mol_ids() == ora.get_cluster_mol_ids()
            ok = same_leaf and raw_ids and sh == so and same_asg and same_cent and same_ids
            bad += not ok
            kc = hip._engine.kernel_counts().tolist()
            sc = hip._engine.sys_counts().tolist()
            res[mode] = n / dt
            print(f"{'OK ' if ok else 'BAD'} {name} bf {bf} BBHIP_SYS={mode}: {n / dt:.0f} fps/s (oracle {n / t_ora:.0f}) leaf {same_leaf} (first diff {first}) ids {raw_ids} "
                  f"asg {same_asg} cent {same_cent} members {same_ids}\n    hip {sh}\n    ora {so}\n    kernel_counts {kc}\n    sys_counts {sc}", flush=True)
        if "1" in res and "0" in res:
            print(f"    speed: systolic {res['1']:.0f} vs default {res['0']:.0f} = {res['1'] / res['0']:.2f}x", flush=True)
os.environ.pop("BBHIP_SYS", None)
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
