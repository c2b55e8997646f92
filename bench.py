#!/usr/bin/env python3
r"""Benchmark of the BitBIRCH insertion hot path on MI355X.

One "step" = one pass of the hot path over one batch: `BitBirch.fit` of the whole
synthetic workload (BASELINE.json configs[1]: 1 M synthetic 2048-bit packed fingerprints,
threshold 0.3, branching factor 50, diameter merge) into a fresh HBM-resident tree, with
the fingerprints already resident in HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n-fps M]

For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank clusters its own
shard of the same size - the data-parallel first round of the reference's multiround scheme
(multiround.py:401-422) - with no collective inside the timed region (weak scaling).

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (the tree insertion
kernel; algorithmic bytes = 264 B per clustered fingerprint) timed
with HIP events on its launch stream inside libbbhip; `k1_roofline` is the arr-vec Tanimoto
kernel (264 B per row) that the north star's HBM target refers to.  `cpu_baseline` is the
CPU oracle (a C restatement of the reference path, kind "port") on a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_FP = 264  # 256 B row read once + 8 B label


def synth_fake_fps(n: int, seed: int, device):
    r"""S-fake(N, seed): same distribution as the reference's make_fake_fingerprints
    (fingerprints.py:70-108: popcount ~ rint(truncnorm(750, 400)) in [1, 2047], uniformly
    random bit positions), generated on the GPU in chunks.  Returns packed uint8 [n, 256]."""
    import torch

    g = torch.Generator(device=device).manual_seed(seed)
    out = torch.empty((n, 256), dtype=torch.uint8, device=device)
    weights = torch.tensor([128, 64, 32, 16, 8, 4, 2, 1], dtype=torch.int32, device=device)
    chunk = 50_000
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        pops = torch.empty(m, device=device)
        todo = torch.ones(m, dtype=torch.bool, device=device)
        while bool(todo.any()):
            draw = torch.randn(m, device=device, generator=g) * 400.0 + 750.0
            ok = (draw >= 1.0) & (draw <= 2047.0)
            take = todo & ok
            pops[take] = draw[take]
            todo &= ~ok
        pops = torch.round(pops).to(torch.int64)
        scores = torch.rand((m, 2048), device=device, generator=g)
        ranks = scores.argsort(dim=1).argsort(dim=1)
        bits = (ranks < pops[:, None]).to(torch.int32)
        out[lo : lo + m] = (bits.view(m, 256, 8) * weights).sum(dim=2).to(torch.uint8)
    return out


def cpu_baseline(fps_host, bf: int, thr: float, sample: int) -> dict:
    from oracle_engine import OracleEngine

    import numpy as np

    eng = OracleEngine(bf, thr, 0, 0.0, np.zeros(0), 2048)
    t0 = time.perf_counter()
    eng.fit_packed(fps_host[:sample])
    dt = time.perf_counter() - t0
    eng.close()
    return {
        "value": sample / dt,
        "unit": "fingerprints/s",
        "cores": 1,
        "kind": "port",
        "sample": f"oracle (C restatement of the reference path) fit of the first {sample} "
                  f"fingerprints of the same workload, {dt:.1f} s on one host core",
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-fps", type=int, default=1_000_000)
    ap.add_argument("--bf", type=int, default=50)
    ap.add_argument("--threshold", type=float, default=0.3)
    ap.add_argument("--cpu-sample", type=int, default=300_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--k1-rows", type=int, default=8_000_000)
    ap.add_argument("--shards", type=int, default=512)
    ap.add_argument("--multiround-files", type=int, default=64, help="0 skips the file-based multiround run")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # type: ignore[no-redef]

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from bblean_amd import BitBirch, _lib

    lib = _lib.load()
    n = args.n_fps
    fps = synth_fake_fps(n, seed=1000 + rank, device=dev)  # resident in HBM
    torch.cuda.synchronize()

    def one_step() -> BitBirch:
        tree = BitBirch(branching_factor=args.bf, threshold=args.threshold, merge_criterion="diameter",
                        device=local_rank)
        tree.fit(fps)
        return tree

    def barrier() -> None:
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    lib.bbh_profile_enable(1)
    lib.bbh_profile_reset()
    barrier()
    t0 = time.perf_counter()
    tree = None
    for _ in range(args.steps):
        tree = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    launches, total_ms = C.c_int64(0), C.c_double(0.0)
    lib.bbh_profile_get(b"tree_insert", C.byref(launches), C.byref(total_ms))
    lib.bbh_profile_enable(0)
    k_launches = max(int(launches.value), 1)
    avg_ms = total_ms.value / k_launches
    fps_per_launch = args.steps * n / k_launches
    achieved = BYTES_PER_FP * fps_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0

    # K1 (arr-vec Tanimoto): the HBM-bound kernel, on an array larger than the 256 MiB
    # Infinity Cache so that the rate is an HBM rate (the 1 M-row workload itself is 256 MB)
    from bblean_amd.similarity import _jt_sim_arr_vec_packed

    k1_rows = max(n, args.k1_rows)
    reps_fill = (k1_rows + n - 1) // n
    big = fps.repeat(reps_fill, 1)[:k1_rows].contiguous() if reps_fill > 1 else fps
    vec = fps[0].clone()
    for _ in range(3):
        _jt_sim_arr_vec_packed(big, vec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        _jt_sim_arr_vec_packed(big, vec)
    e1.record()
    torch.cuda.synchronize()
    k1_ms = e0.elapsed_time(e1) / reps
    k1_gbs = BYTES_PER_FP * k1_rows / (k1_ms * 1e-3) / 1e9
    # K2 (batched descent step: every query row against all centroids of a node, exact first-argmax):
    # VALU-bound - 2 ops (AND, BCNT) per query dword and centroid row on 256 CUs x 64 lanes
    nq2, nc2 = min(k1_rows, 4_000_000), args.bf + 1  # enough queries to hide the launch tail
    cents = fps[:nc2].contiguous()
    i_o = torch.empty(nq2, dtype=torch.int32, device=dev)
    n_o = torch.empty(nq2, dtype=torch.int32, device=dev)
    u_o = torch.empty(nq2, dtype=torch.int32, device=dev)

    def k2() -> None:
        _lib.check(lib.bbh_jt_best_match(big.data_ptr(), nq2, cents.data_ptr(), nc2, 256, i_o.data_ptr(), n_o.data_ptr(),
                                         u_o.data_ptr(), None, None))

    for _ in range(2):
        k2()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        k2()
    e1.record()
    torch.cuda.synchronize()
    k2_ms = e0.elapsed_time(e1) / 5
    k2_laneops = nq2 * nc2 * 128 / (k2_ms * 1e-3)  # 64 dwords x (AND + BCNT) per pair
    k2_peak = 256 * 64 * 2.4e9  # CUs x lanes x clock: one VALU op per lane and clock

    del big

    # independent trees in one launch (multiround round 1 with many shards on this GPU)
    shard_stats = None
    if args.shards > 1:
        from bblean_amd import fit_concurrently

        per = n // args.shards
        parts = [fps[i * per:(i + 1) * per] for i in range(args.shards)]
        best = None
        for _ in range(2):
            trees = [BitBirch(branching_factor=args.bf, threshold=args.threshold, device=local_rank)
                     for _ in range(args.shards)]
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            fit_concurrently(trees, parts, reinsert_indices=[range(i * per, (i + 1) * per) for i in range(args.shards)])
            dt = time.perf_counter() - t1
            best = dt if best is None else min(best, dt)
            del trees
        shard_stats = {"shards": args.shards, "rows_per_shard": per, "seconds": best,
                       "fingerprints_per_s": args.shards * per / best,
                       "note": "multiround round 1 (reference multiround.py:401-422) with this many input files: "
                               "one workgroup per shard tree, one kernel launch; results equal per-shard fit"}

    # the reference's answer for large sets, `bb multiround` (multiround.py:333-484), on this one GPU:
    # the same rows as shard files; all shards of round 1 and all batches of the merge round share
    # kernel launches, the final merge is one sequential tree
    mr_stats = None
    if args.multiround_files > 1 and rank == 0:
        import tempfile

        from bblean_amd.multiround import run_multiround_bitbirch

        host = fps.cpu().numpy()
        per = n // args.multiround_files
        with tempfile.TemporaryDirectory() as d:
            names = []
            for i in range(args.multiround_files):
                f = Path(d) / f"fps.{i:05d}.npy"
                np.save(f, host[i * per:(i + 1) * per])
                names.append(f)
            out_dir = Path(d) / "out"
            out_dir.mkdir()
            t1 = time.perf_counter()
            timer = run_multiround_bitbirch(names, out_dir, branching_factor=args.bf, threshold=args.threshold,
                                            num_initial_processes=1, device=local_rank)
            dt = time.perf_counter() - t1
        mr_stats = {"files": args.multiround_files, "rows": per * args.multiround_files, "seconds": dt,
                    "fingerprints_per_s": per * args.multiround_files / dt,
                    "rounds_s": {k: round(v, 3) for k, v in timer.timings.items()},
                    "note": "file-compatible multiround with the reference's defaults (full refinement, one merge "
                            "round in bins of 10, tolerance-diameter merges); files on tmpfs/disk inside the timing"}
        del host

    traffic = None
    pmc = REPO / "profiles" / "pmc_latest.json"
    if pmc.is_file():
        try:
            rec = json.loads(pmc.read_text())
            if int(rec.get("n_fps", -1)) == n:
                # FETCH_SIZE/WRITE_SIZE are in KB; FETCH_SIZE reads 1/2 on gfx950 (MI355X_MICROARCH.md)
                traffic = (2.0 * rec["tree_fetch_kb_total"] + rec["tree_write_kb_total"]) * 1024.0 / max(rec["tree_launches"], 1)
        except Exception:
            traffic = None
    if rank == 0:
        n_clusters = len(tree._leaves()["ids"]) if tree is not None else 0
        out = {
            "metric": "fingerprints/sec clustered (2048-bit, thr=0.3)",
            "value": world * args.steps * n / elapsed,
            "unit": "fingerprints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"{n} synthetic 2048-bit packed fingerprints per GPU (make_fake_fingerprints "
                            f"popcount distribution), threshold={args.threshold}, branching_factor={args.bf}, "
                            "merge=diameter, BitBirch.fit into a fresh HBM-resident tree, inputs resident in HBM",
                "clusters": n_clusters,
                "multi_gpu": "independent shard per GPU (multiround round 1), no collective in the timed region",
            },
            "roofline": {
                "kernel": "k_tree_insert",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "launches": k_launches,
                "avg_launch_ms": avg_ms,
                "note": "latency/dependency-bound sequential insertion; 264 algorithmic bytes per fingerprint",
            },
            "k1_roofline": {
                "kernel": "k_arr_vec<16,true> (arr-vec Tanimoto, 264 B/row)",
                "bound": "hbm",
                "achieved": k1_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": k1_gbs / HBM_PEAK_GBS,
                "rows": k1_rows,
                "avg_launch_ms": k1_ms,
            },
            "k2_valu": {
                "kernel": "k_best_match<64> (batched node compare, exact first-argmax)",
                "bound": "valu",
                "achieved": k2_laneops / 1e12,
                "peak": k2_peak / 1e12,
                "unit": "T lane-ops/s",
                "frac": k2_laneops / k2_peak,
                "queries": nq2, "centroids": nc2, "avg_launch_ms": k2_ms,
            },
            "concurrent_shards": shard_stats,
            "multiround_one_gpu": mr_stats,
        }
        if not args.no_cpu and world == 1:
            sample = min(args.cpu_sample, n)
            out["cpu_baseline"] = cpu_baseline(fps[:sample].cpu().numpy(), args.bf, args.threshold, sample)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
