import os, sys, re, tempfile
os.environ["BBHIP_LAUNCH_LOG"] = "1"
os.environ["BBHIP_SYS"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from test_hip_pipe_fuzz import _rows
from bblean_amd import BitBirch
from oracle_engine import OracleEngine
seed = int(sys.argv[1]); reps = int(sys.argv[2])
rng = np.random.default_rng(7000 + seed)
bf = 50 if seed % 3 else 254
n = int(rng.integers(12_000, 60_000))
crit = "diameter" if rng.random() < 0.6 else "tolerance-diameter"
thr = float(rng.uniform(0.15, 0.8)); tol = float(rng.uniform(0.0, 0.1))
rows = _rows(rng, n)
cuts = sorted(set(int(c) for c in rng.integers(8_200, n + 1, int(rng.integers(0, 5)))) | {0, n})
if seed % 4 == 0: os.environ["BBHIP_TINY_POOLS"] = "1"
kw = dict(branching_factor=bf, threshold=thr, merge_criterion=crit, tolerance=tol)
ora = BitBirch(_engine_factory=OracleEngine, **kw)
ora.fit(rows[:cuts[1]])
olog = np.asarray(ora._log_leaf[0])
bad = 0
for rep in range(reps):
    tf = tempfile.NamedTemporaryFile(delete=False); tf.close()
    sys.stderr.flush()
    saved = os.dup(2); fd = os.open(tf.name, os.O_WRONLY | os.O_TRUNC); os.dup2(fd, 2)
    hip = BitBirch(**kw)
    err = None
    try:
        hip.fit(rows[:cuts[1]])
    except Exception as exc:
        err = repr(exc)[:200]
    sys.stderr.flush(); os.dup2(saved, 2); os.close(fd); os.close(saved)
    log = open(tf.name).read().splitlines(); os.unlink(tf.name)
    if err:
        print(f"rep {rep}: {err}", flush=True); bad += 1; continue
    hl = np.asarray(hip._log_leaf[0])
    d = np.nonzero(hl != olog)[0]
    if d.size:
        bad += 1
        first = int(d[0])
        print(f"rep {rep}: first differing element {first} (hip {hl[first]} ora {olog[first]}), {d.size} diffs", flush=True)
        cum = 0
        for ln in log:
            m = re.search(r"launch\] (\w+) .* elems=(\d+) .* leaf_splits=(\d+) node_splits=(\d+) stop=(\d+)", ln)
            if not m: continue
            k, el, ls, ns, st = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5))
            if cum + el >= first - 400 and cum <= first + 200:
                print(f"     [{cum}, {cum + el}) {k} leaf_splits {ls} node_splits {ns} stop {st}")
            cum += el
print(f"seed {seed}: {bad} bad of {reps}")
