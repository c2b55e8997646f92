"""Host-side profile of the one-rank-per-GPU multiround (one RCCL rank, shards resident in HBM): per-round wall, tree
kernel seconds, and the Python functions that make up the rest.
    python tools/distributed_profile.py [rows per shard] [shards] [workload] [branching factor]"""
import cProfile, ctypes as C, os, pstats, sys, time

sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import torch.distributed as dist

from bench import WORKLOADS
from bblean_amd import _lib
from bblean_amd.multiround import run_multiround_distributed

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 4
workload = sys.argv[3] if len(sys.argv) > 3 else "fake"
bf = int(sys.argv[4]) if len(sys.argv) > 4 else 50  # the benchmark's; multiround's own default is 254
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
lib = _lib.load()
gen, thr = WORKLOADS[workload][0], WORKLOADS[workload][1]
inputs = [gen(n, 1000 + s, dev) for s in range(shards)]
torch.cuda.synchronize()


def step():
    tree, timer = run_multiround_distributed(inputs, None, threshold=thr, branching_factor=bf, device=0, return_tree=True)
    labels = tree.get_assignments()
    return timer, labels


step()  # warm-up (pools, RCCL)
lib.bbh_profile_enable(1)
lib.bbh_profile_reset()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
timer, labels = step()
pr.disable()
wall = time.perf_counter() - t0
l, ms, u = C.c_int64(0), C.c_double(0.0), C.c_int64(0)
lib.bbh_profile_get(b"tree_insert", C.byref(l), C.byref(ms))
lib.bbh_profile_units(b"tree_insert", C.byref(u))
print(f"{shards} x {n} rows ({workload}, bf {bf}): wall {wall:.3f}s = {shards * n / wall:.0f} fps/s; tree kernel {ms.value / 1e3:.3f}s in {l.value} launches, "
      f"{u.value} elements; rounds {({k: round(v, 3) for k, v in timer.timings.items()})}; clusters {int(labels.max())}")
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
dist.destroy_process_group()
