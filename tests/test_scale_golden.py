r"""Parity at scale against the REFERENCE itself (SURVEY.md section 8c, G6 + G9): digests of reference runs at
100 k - 1 M rows (4-5-level trees, uint16 tiers, bf 254 and bf 1000 at depth) and of the reference's multiround
on 8 shard files with every intermediate round-* table kept - BASELINE configs 4 (S-ecfp, thr 0.3) and 5
(S-rdkit-like, thr 0.6, diameter) at test scale.  CPU tests pin the oracle; `-m gpu` tests run the HIP engine
through the same host code, file-based and one-rank-per-GPU."""
from __future__ import annotations

import os
import pickle
import socket
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np
import pytest

from cases import MULTIROUND_SCALE_CASES, SCALE_CASES
from oracle_engine import OracleEngine
from scale_cases import ALL_MULTIROUND, check_final, run_multiround_files, run_scale_tree, write_shards

REPO = Path(__file__).resolve().parents[1]
_CPU_TREES = [c for c in SCALE_CASES if not c.get("gpu_only") or os.environ.get("BB_SCALE_ORACLE")]


@pytest.mark.parametrize("case", _CPU_TREES, ids=[c["name"] for c in _CPU_TREES])
def test_oracle_vs_reference_at_scale(case):
    run_scale_tree(case, OracleEngine)


@pytest.mark.gpu
@pytest.mark.parametrize("case", SCALE_CASES, ids=[c["name"] for c in SCALE_CASES])
def test_hip_vs_reference_at_scale(case):
    run_scale_tree(case, None)


@pytest.mark.parametrize("case", ALL_MULTIROUND, ids=[c["name"] for c in ALL_MULTIROUND])
def test_multiround_round_files_oracle(case, tmp_path):
    run_multiround_files(case, OracleEngine, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ALL_MULTIROUND, ids=[c["name"] for c in ALL_MULTIROUND])
def test_multiround_round_files_hip(case, tmp_path):
    run_multiround_files(case, None, tmp_path)


# ---- one rank per GPU --------------------------------------------------------------------------------
_WORKER = r"""
import os, sys, pickle
from pathlib import Path
sys.path.insert(0, {repo!r}); sys.path.insert(0, {repo!r} + "/tests"); sys.path.insert(0, {repo!r} + "/tests/golden")
import torch, torch.distributed as dist
from bblean_amd.multiround import run_multiround_distributed
if {backend!r} == "nccl":
    torch.cuda.set_device(int(sys.argv[1]) % max(torch.cuda.device_count(), 1))  # one rank per GPU
engine = None
if {use_oracle}:
    from oracle_engine import OracleEngine as engine
dist.init_process_group({backend!r}, init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size={world})
files = sorted(Path({d!r}).glob("fps.*.npy"))
clusters, timer = run_multiround_distributed(files, Path({d!r}) / "out", _engine_factory=engine, **{kwargs!r})
if dist.get_rank() == 0:
    pickle.dump(clusters, open(Path({d!r}) / "clusters_rank0.pkl", "wb"))
pickle.dump(timer.exchange, open(Path({d!r}) / ("exchange_rank%d.pkl" % dist.get_rank()), "wb"))
dist.destroy_process_group()
"""


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_distributed(case: dict, d: Path, *, use_oracle: bool, backend: str, world: int, recv_budget_mb: float | None = None):
    write_shards(d, case)
    (d / "out").mkdir()
    kwargs = dict(case["kwargs"])
    if recv_budget_mb is not None:
        kwargs["recv_budget_mb"] = recv_budget_mb
    src = _WORKER.format(repo=str(REPO), use_oracle=use_oracle, backend=backend, port=_free_port(), world=world, d=str(d),
                         kwargs=kwargs)
    (d / "worker.py").write_text(src)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(d / "worker.py"), str(r)], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=1500) == 0
    clusters = pickle.load(open(d / "clusters_rank0.pkl", "rb"))
    cents = pickle.load(open(d / "out" / "cluster-centroids-packed.pkl", "rb"))
    assert clusters == pickle.load(open(d / "out" / "clusters.pkl", "rb"))
    check_final(case, clusters, cents)
    return [pickle.load(open(d / f"exchange_rank{r}.pkl", "rb")) for r in range(world)]


@pytest.mark.parametrize("case", MULTIROUND_SCALE_CASES[:2], ids=[c["name"] for c in MULTIROUND_SCALE_CASES[:2]])
def test_configs_4_5_distributed_gloo_world2_oracle(case, tmp_path):
    r"""BASELINE configs 4 / 5 at test scale through the one-rank-per-GPU path (2 gloo ranks, oracle engine): the
    point-to-point exchange must reproduce the reference's clusters."""
    ex = _run_distributed(case, tmp_path, use_oracle=True, backend="gloo", world=2)
    # every table travels at most once per round, and only towards the rank that merges it
    for r, per_round in enumerate(ex):
        for rnd, b in per_round.items():
            assert b["sent"] >= 0 and b["received"] >= 0
    assert sum(b["sent"] for per in ex for b in per.values()) == sum(b["received"] for per in ex for b in per.values())
    assert ex[1]["round-3"]["received"] == 0  # the final merge happens on rank 0 only


@pytest.mark.parametrize("world,case_i,budget", [(4, 0, None), (8, 1, None), (4, 1, 3), (8, 0, 3)],
                         ids=["cfg4_world4", "cfg5_world8", "cfg5_world4_bounded", "cfg4_world8_bounded"])
def test_configs_4_5_distributed_gloo_world4_world8_oracle(world, case_i, budget, tmp_path):
    r"""The plan changes shape with the world size - batch b is merged by rank b mod W, with 8 shard files at W = 8 every rank
    owns exactly one shard and the merge round's batches land on ranks 0 and 1, the slab schedule of the bounded receive has
    W - 1 senders per step: 4 and 8 gloo ranks (oracle engine) must reproduce the reference's digests of configs 4 / 5 with
    and without a receive budget, and account for every byte on the wire."""
    case = MULTIROUND_SCALE_CASES[case_i]
    ex = _run_distributed(case, tmp_path, use_oracle=True, backend="gloo", world=world, recv_budget_mb=budget)
    sent = sum(b["sent"] for per in ex for b in per.values())
    assert sent == sum(b["received"] for per in ex for b in per.values()) > 0
    for r in range(1, world):
        assert ex[r]["round-3"]["received"] == 0  # the final merge happens on rank 0 only
    # (round 2: the merge round's batches go to ranks b mod W - with 8 tables per dtype and bins of 10 that is ranks 0 and 1)
    assert all(ex[r]["round-2"]["received"] == 0 for r in range(2, world))


from conftest import rccl_world

_RCCL_W = rccl_world()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [_RCCL_W], ids=[f"world{_RCCL_W}"])
@pytest.mark.parametrize("case", MULTIROUND_SCALE_CASES[:2], ids=[c["name"] for c in MULTIROUND_SCALE_CASES[:2]])
def test_configs_4_5_distributed_rccl_hip(case, world, tmp_path):
    r"""The same on the real stack: HIP engine, torch.distributed "nccl" (= RCCL), one rank per visible GPU, BitFeature
    tables resident in HBM from the gather kernel to the next round's insertion kernel."""
    ex = _run_distributed(case, tmp_path, use_oracle=False, backend="nccl", world=world)
    if world > 1:
        assert sum(b["sent"] for per in ex for b in per.values()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", MULTIROUND_SCALE_CASES[:2], ids=[c["name"] for c in MULTIROUND_SCALE_CASES[:2]])
def test_configs_4_5_distributed_world2_hip_device_tables(case, tmp_path):
    r"""Two ranks with the HIP engine on the one GPU of the box (RCCL refuses two ranks per device, so the wire is
    gloo): tables are gathered into HBM, handed to the exchange as tensors and inserted from HBM by the receiving
    rank - no NumPy table on the way."""
    ex = _run_distributed(case, tmp_path, use_oracle=False, backend="gloo", world=2)
    assert sum(b["sent"] for per in ex for b in per.values()) > 0


# ---- bounded receive: the merging rank takes its tables in slabs (reference multiround.py:284-312 reads one pair at a time) ----
@pytest.mark.parametrize("case", MULTIROUND_SCALE_CASES[:2], ids=[c["name"] for c in MULTIROUND_SCALE_CASES[:2]])
def test_configs_4_5_distributed_gloo_world2_oracle_bounded_receive(case, tmp_path):
    r"""BASELINE configs 4 / 5 at test scale with a receive budget far below one table (a 25 k-row uint8 table is 51 MB):
    every table crosses the wire and enters the tree in slabs of 3 MB, slab s + 1 travelling while slab s is inserted -
    the reference's clusters, byte for byte the same outputs as the unbounded exchange."""
    ex = _run_distributed(case, tmp_path, use_oracle=True, backend="gloo", world=2, recv_budget_mb=3)
    assert sum(b["sent"] for per in ex for b in per.values()) == sum(b["received"] for per in ex for b in per.values()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("world", [_RCCL_W], ids=[f"world{_RCCL_W}"])
@pytest.mark.parametrize("case", MULTIROUND_SCALE_CASES[:2], ids=[c["name"] for c in MULTIROUND_SCALE_CASES[:2]])
def test_configs_4_5_distributed_rccl_hip_bounded_receive(case, world, tmp_path):
    r"""The same on the real stack (HIP engine, RCCL, tables resident in HBM, one rank per visible GPU), budget 3 MB."""
    _run_distributed(case, tmp_path, use_oracle=False, backend="nccl", world=world, recv_budget_mb=3)


@pytest.mark.gpu
@pytest.mark.parametrize("case", MULTIROUND_SCALE_CASES[:1], ids=[c["name"] for c in MULTIROUND_SCALE_CASES[:1]])
def test_config_4_distributed_world2_hip_bounded_receive(case, tmp_path):
    r"""Two ranks with the HIP engine on the one GPU of the box (gloo wire), budget 3 MB: chunks of device tables."""
    ex = _run_distributed(case, tmp_path, use_oracle=False, backend="gloo", world=2, recv_budget_mb=3)
    assert sum(b["sent"] for per in ex for b in per.values()) > 0


# ---- the round-* files as a wire format between engines -------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("first,rest", [("hip", "oracle"), ("oracle", "hip")])
def test_round_files_interoperate_between_engines(first, rest, tmp_path):
    r"""Round 1 on one engine, its round-1 files merged by the other: HIP shards feed CPU merge rounds and CPU shards
    feed HIP merge rounds, byte-identical files and the reference's clusters either way."""
    from cases import MULTIROUND_CASES
    from scale_cases import SCALE, file_digest

    from bblean_amd.multiround import run_multiround_bitbirch

    case = next(c for c in MULTIROUND_CASES if c["name"] == "mr_defaults")
    fac = {"hip": None, "oracle": OracleEngine}
    files = write_shards(tmp_path, case)
    (tmp_path / "out").mkdir()
    run_multiround_bitbirch(files, tmp_path / "out", num_initial_processes=1, cleanup=False, _engine_factory=fac[rest],
                            _round1_engine_factory=fac[first], **case["kwargs"])
    gold = SCALE["multiround"][case["name"]]
    for p in sorted((tmp_path / "out").glob("round-*")):
        assert file_digest(p) == gold["files"][p.name], p.name
    check_final(case, pickle.load(open(tmp_path / "out" / "clusters.pkl", "rb")),
                pickle.load(open(tmp_path / "out" / "cluster-centroids-packed.pkl", "rb")))
