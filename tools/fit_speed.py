"""Single-tree fit throughput on S-fake rows with the production kernel (no phase timers).
    python tools/fit_speed.py [n] [reps] [branching_factor] [threshold] [n_features]"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from bench import synth_fake_fps
from bblean_amd import BitBirch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bf = int(sys.argv[3]) if len(sys.argv) > 3 else 50
thr = float(sys.argv[4]) if len(sys.argv) > 4 else 0.3
nf = int(sys.argv[5]) if len(sys.argv) > 5 else 2048
if nf == 2048:
    fps = synth_fake_fps(n, 1000, torch.device("cuda"))
else:  # same popcount distribution scaled to the width
    g = torch.Generator(device="cuda").manual_seed(1000)
    dens = (torch.randn(n, 1, device="cuda", generator=g) * (400 / 2048) + 750 / 2048).clamp(1 / nf, 1 - 1 / nf)
    bits = (torch.rand((n, nf), device="cuda", generator=g) < dens).to(torch.int32)
    w = torch.tensor([128, 64, 32, 16, 8, 4, 2, 1], dtype=torch.int32, device="cuda")
    fps = (bits.view(n, nf // 8, 8) * w).sum(dim=2).to(torch.uint8)
torch.cuda.synchronize()
best = None
for _ in range(reps):
    t0 = time.perf_counter()
    t = BitBirch(branching_factor=bf, threshold=thr).fit(fps)
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
print(f"bf={bf} thr={thr} bits={nf}: {n/best:.0f} fps/s  ({best/n*1e6:.2f} us/insert, best of {reps})", t._engine.stats()[:5])
