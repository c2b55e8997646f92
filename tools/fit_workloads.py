"""Single-tree fit throughput per workload and branching factor, one launch log line each, and which kernel did the work.
    python tools/fit_workloads.py [rows] [bf ...]
Workloads: the three generators of bench.py (S-fake, S-ecfp, S-rdkit-like) and two whose INTERNAL tree levels stay
informative (tests/golden/cases.py clustered_hier / clustered_dense: planted two-level families / dense prototypes)."""
import os, sys, time
os.environ["BBHIP_LAUNCH_LOG"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import numpy as np
import torch
from bench import WORKLOADS
from bblean_amd import BitBirch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
bfs = [int(a) for a in sys.argv[2:]] or [50, 254]


def hier(n, seed, device):
    r"""cases.clustered_hier on the GPU: 12 super-prototypes (50 %), n/50 cluster prototypes (12 % toggled), rows (4 % toggled)."""
    g = torch.Generator(device=device).manual_seed(seed)
    k = max(n // 50, 1)
    sup = torch.rand((12, 2048), device=device, generator=g) < 0.5
    protos = sup[torch.randint(0, 12, (k,), device=device, generator=g)] ^ (torch.rand((k, 2048), device=device, generator=g) < 0.12)
    return _rows(protos, n, 0.04, g, device)


def dense(n, seed, device):
    r"""cases.clustered_dense on the GPU: n/50 prototypes at 45-60 % density, rows with 8 % of the bits toggled."""
    g = torch.Generator(device=device).manual_seed(seed)
    k = max(n // 50, 1)
    dens = torch.rand((k, 1), device=device, generator=g) * 0.15 + 0.45
    protos = torch.rand((k, 2048), device=device, generator=g) < dens
    return _rows(protos, n, 0.08, g, device)


def _rows(protos, n, flip, g, device):
    w = torch.tensor([128, 64, 32, 16, 8, 4, 2, 1], dtype=torch.int32, device=device)
    out = torch.empty((n, 256), dtype=torch.uint8, device=device)
    for lo in range(0, n, 50_000):
        m = min(50_000, n - lo)
        which = torch.randint(0, protos.shape[0], (m,), device=device, generator=g)
        bits = (protos[which] ^ (torch.rand((m, 2048), device=device, generator=g) < flip)).to(torch.int32)
        out[lo:lo + m] = (bits.view(m, 256, 8) * w).sum(dim=2).to(torch.uint8)
    return out


work = dict(WORKLOADS)
work["hier"] = (hier, 0.6, "clustered_hier")
work["dense"] = (dense, 0.6, "clustered_dense")
only = os.environ.get("WORKLOADS")
for name, (gen, thr, _) in work.items():
    if only and name not in only.split(","):
        continue
    fps = gen(n, 1000, torch.device("cuda"))
    torch.cuda.synchronize()
    for bf in bfs:
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            t = BitBirch(branching_factor=bf, threshold=thr, merge_criterion="diameter").fit(fps)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        kc = t._engine.kernel_counts()
        print(f"== {name} thr {thr} bf {bf}: {n / best:.0f} fps/s ({1e6 * best / n:.2f} us/insert) stats {t._engine.stats()[:7].tolist()} "
              f"elements by kernel pipe/fast/complete {kc[:3].tolist()} launches {kc[3:6].tolist()} unsupported-shape stops {int(kc[6])}", flush=True)
