import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from bblean_amd import _lib
lib = _lib.load()
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
for nq, nc in [(1_000_000, 16), (1_000_000, 51), (1_000_000, 255), (4_000_000, 51), (65536, 51), (4096, 51)]:
    q = torch.randint(0, 256, (nq, 256), dtype=torch.uint8, device=dev, generator=g)
    c = torch.randint(0, 256, (nc, 256), dtype=torch.uint8, device=dev, generator=g)
    idx = torch.empty(nq, dtype=torch.int32, device=dev)
    inter = torch.empty(nq, dtype=torch.int32, device=dev)
    union = torch.empty(nq, dtype=torch.int32, device=dev)
    def run():
        _lib.check(lib.bbh_jt_best_match(q.data_ptr(), nq, c.data_ptr(), nc, 256, idx.data_ptr(), inter.data_ptr(), union.data_ptr(), None, None))
    for _ in range(2): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    pairs = nq * nc
    laneops = pairs * 128
    print(f"nq={nq:8d} nc={nc:4d}: {dt*1e3:8.3f} ms  {pairs/dt/1e9:8.2f} G pairs/s  {laneops/dt/1e12:6.2f} T lane-ops/s (peak ~39)  HBM {(nq*272+nc*256)/dt/1e9:7.1f} GB/s", flush=True)
