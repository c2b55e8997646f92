"""BASELINE.json configs[3] / configs[4] at their stated sizes on ONE MI355X: 8 shards through multiround
(`run_multiround_distributed`, one RCCL rank - the code path `bench.py --gpus N` runs with N ranks), CLI defaults
(bf 254, full refinement, one merge round in bins of 10, tolerance-diameter merges; reference multiround.py:333-484):

    python tools/config45.py 4 12500000        # config 4: S-ecfp, 8 x 12.5 M rows, threshold 0.3
    python tools/config45.py 5 6250000         # config 5: S-rdkit-like, 8 x 6.25 M rows, threshold 0.6, diameter
    python tools/config45.py 4 250000 --check  # small instance, also checked against the file-based run

The shards are generated on the GPU and handed over as HOST arrays (a rank's shard is streamed in: the input does not
occupy HBM next to the round tables); `--budget-mb` bounds what the merging rank receives at a time
(`recv_budget_mb`).  Prints per-round seconds, peak HBM (torch allocator + the library's pools, from hipMemGetInfo),
cluster statistics, and checks the size-independent properties of tests/test_configs45.py at full size: the clusters
partition the input, sizes / labels agree, the final cluster features add up to the column sums of all fingerprints."""
import argparse
import os
import socket
import sys
import threading
import time

sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tools")
import numpy as np
import torch


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class PeakHBM(threading.Thread):
    r"""Polls hipMemGetInfo (everything allocated on the device, whoever did it) ten times a second."""

    def __init__(self) -> None:
        super().__init__(daemon=True)
        self.peak = 0
        self.stop = False

    def run(self) -> None:
        while not self.stop:
            free, total = torch.cuda.mem_get_info()
            self.peak = max(self.peak, total - free)
            time.sleep(0.1)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("config", type=int, choices=(4, 5))
    ap.add_argument("rows_per_shard", type=int)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--budget-mb", type=float, default=4096.0)
    ap.add_argument("--bf", type=int, default=254)
    ap.add_argument("--check", action="store_true", help="also run the file-based multiround and compare the clusters")
    a = ap.parse_args()
    import torch.distributed as dist

    from bench import WORKLOADS
    from bblean_amd import _lib
    from bblean_amd.multiround import run_multiround_distributed

    workload, kwargs = (("ecfp", dict(threshold=0.3)) if a.config == 4
                        else ("rdkit", dict(threshold=0.6, initial_merge_criterion="diameter")))
    kwargs["branching_factor"] = a.bf
    dev = torch.device("cuda")
    gen = WORKLOADS[workload][0]
    n = a.shards * a.rows_per_shard
    t0 = time.perf_counter()
    from props import column_sums

    want = torch.zeros(2048, dtype=torch.int64, device=dev)
    shards = []
    for i in range(a.shards):
        s = gen(a.rows_per_shard, 4000 * a.config + i, dev)
        column_sums(s, want)
        shards.append(s.cpu().numpy())
        del s
    torch.cuda.empty_cache()
    print(f"config {a.config}: {a.shards} shards x {a.rows_per_shard} rows of S-{workload} = {n} fingerprints, bf {a.bf}, "
          f"recv budget {a.budget_mb:.0f} MB; generated in {time.perf_counter() - t0:.1f} s", flush=True)
    lib = _lib.load()
    lib.bbh_profile_enable(1)
    lib.bbh_profile_reset()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1)
    mon = PeakHBM()
    mon.start()
    t0 = time.perf_counter()
    tree, timer = run_multiround_distributed(shards, None, return_tree=True, recv_budget_mb=a.budget_mb, **kwargs)
    wall = time.perf_counter() - t0
    rounds = {k: round(v, 2) for k, v in timer.timings.items()}
    print(f"rounds (s): {rounds}", flush=True)
    print(f"exchange bytes: {timer.exchange}", flush=True)
    import ctypes as C
    l, ms = C.c_int64(0), C.c_double(0.0)
    lib.bbh_profile_get(b"tree_insert", C.byref(l), C.byref(ms))
    print(f"tree kernel: {ms.value / 1e3:.1f} s in {l.value} launches", flush=True)
    t1 = time.perf_counter()
    ids = tree.get_assignments()
    t_assign = time.perf_counter() - t1
    mon.stop = True
    print(f"whole job {wall:.1f} s = {n / wall:.0f} fingerprints/s (+ labels {t_assign:.1f} s); peak HBM in use {mon.peak / 2**30:.1f} GiB "
          f"of {torch.cuda.mem_get_info()[1] / 2**30:.0f}", flush=True)
    # ---- size-independent properties (tests/test_configs45.py:74-118) at this size
    from props import check_clustering

    cs = check_clustering(tree, n, want, ids)
    print(f"clusters {cs['clusters']}: largest {cs['largest']}, singletons {cs['singletons']}; partition of 0..{n - 1}: ok", flush=True)
    print("cluster-feature column sums == column sums of all fingerprints: ok", flush=True)
    mem = tree._engine.memory()
    print(f"final tree memory: node pools {int(mem[0]) / 1e9:.2f} GB ({int(mem[1]) / 1e9:.2f} GB used = {int(mem[1]) / 1e9 / (n / 1e6):.3f} GB per million "
          f"fingerprints), cluster-feature pools {int(mem[2]) / 1e9:.2f} GB, peak of this tree's allocations {int(mem[3]) / 1e9:.2f} GB; "
          f"{int(mem[4])} compactions, last one: {int(mem[5])} nodes sealed / {int(mem[6])} at full capacity; {int(mem[7])} sealed nodes thawed", flush=True)
    kc = tree._engine.kernel_counts()
    print(f"final tree: elements by kernel pipe/fast/complete {kc[:3].tolist()} launches {kc[3:6].tolist()} unsupported-shape stops {int(kc[6])} pool stops {int(kc[7])}", flush=True)
    st = tree._engine.stats()
    print(f"final tree: stats {st.tolist()}", flush=True)
    dist.destroy_process_group()
    if a.check:
        import pickle
        import tempfile
        from pathlib import Path

        from bblean_amd.multiround import run_multiround_bitbirch

        with tempfile.TemporaryDirectory() as td:
            files = []
            for i, s in enumerate(shards):
                f = Path(td) / f"fps.{i:04d}.npy"
                np.save(f, s)
                files.append(f)
            out = Path(td) / "out"
            out.mkdir()
            run_multiround_bitbirch(files, out, num_initial_processes=1, **kwargs)
            clusters = pickle.load(open(out / "clusters.pkl", "rb"))
        ids2 = np.zeros(n, dtype=np.uint64)
        for c, members in enumerate(clusters):
            ids2[np.asarray(members, dtype=np.int64)] = c + 1
        assert np.array_equal(ids, ids2)
        print("file-based multiround gives the same labels: ok", flush=True)


if __name__ == "__main__":
    main()
