r"""CPU: bench.py's `regressions` record (every throughput of a line against the previous round's committed line) on the two
lines that motivated it - round 4's against round 3's lists exactly the two regressions round 4 shipped without noticing -
and the distributed exchange's one definition of a packed tail row's size."""
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))


def test_regressions_record_finds_round_4s_two():
    import bench

    r04 = json.loads((REPO / "profiles" / "r04" / "bench_line_final.json").read_text())
    rec = bench.regressions(r04, [REPO / "profiles" / "r03" / "bench_line_final.json"])
    assert rec["against"] == ["profiles/r03/bench_line_final.json"] and rec["compared"] >= 8
    assert set(rec["worse"]) == {"bf254.fingerprints_per_s", "concurrent_shards.fingerprints_per_s"}
    assert rec["worse"]["concurrent_shards.fingerprints_per_s"]["ratio"] < 0.2
    # a line against itself: nothing worse, nothing better
    same = bench.regressions(r04, [REPO / "profiles" / "r04" / "bench_line_final.json"])
    assert same["worse"] == {} and same["better"] == {}
    # a missing file is reported, not raised
    assert "error" in bench.regressions(r04, [REPO / "profiles" / "r00" / "nothing.json"])


def test_round5_line_has_no_gpu_side_regression():
    import bench

    r05 = json.loads((REPO / "profiles" / "r05" / "bench_line_final.json").read_text())
    rec = bench.regressions(r05, [REPO / "profiles" / "r04" / "bench_line_final.json"])
    # (the file-based multiround follows its host: DESIGN section 8, round 5)
    assert set(rec["worse"]) <= {"multiround_one_gpu.fingerprints_per_s"}
    assert rec["better"]["bf254.fingerprints_per_s"] > 1.1 and rec["better"]["concurrent_shards.fingerprints_per_s"] > 5


def test_tail_row_bytes_single_definition():
    from bblean_amd.multiround import _tail_row_bytes

    assert _tail_row_bytes(2049) == 256 and _tail_row_bytes(65) == 8 and _tail_row_bytes(2041) == 255


def test_regressions_prefers_the_drivers_line_and_needs_two_of_three_repeats():
    r"""the driver's BENCH_rNN.json overrides the builder's line where it has the entry; an entry that
    carries its repeats is listed only when two of three are worse; the file-based multiround is compared on kernel time."""
    import bench

    r05 = json.loads((REPO / "profiles" / "r05" / "bench_line_final.json").read_text())
    drv = json.loads((REPO / "BENCH_r05.json").read_text())["parsed"]
    files = [REPO / "profiles" / "r05" / "bench_line_final.json", REPO / "BENCH_r05.json"]
    now = json.loads(json.dumps(r05))
    now["value"] = 0.9 * drv["value"]  # 10 % below the DRIVER's headline (the builder's line has another number)
    rec = bench.regressions(now, files)
    assert rec["against"] == ["profiles/r05/bench_line_final.json", "BENCH_r05.json"]
    assert rec["worse"]["value"]["before"] == drv["value"]
    before = r05["k1_roofline"]["achieved"]
    now = json.loads(json.dumps(r05))
    now["value"] = drv["value"]
    now["roofline"] = drv["roofline"]
    now["k1_roofline"]["achieved"] = 0.93 * before
    now["k1_roofline"]["achieved_repeats"] = [0.93 * before, 0.99 * before, 1.0 * before]  # one slow repeat: box noise
    assert "k1_roofline.achieved" not in bench.regressions(now, files)["worse"]
    now["k1_roofline"]["achieved_repeats"] = [0.93 * before, 0.92 * before, 1.0 * before]
    assert "k1_roofline.achieved" in bench.regressions(now, files)["worse"]
    # wall time of the file-based multiround is not an entry any more
    now = json.loads(json.dumps(r05))
    now["value"] = drv["value"]
    now["roofline"] = drv["roofline"]
    now["multiround_one_gpu"] = {"wall_fingerprints_per_s": 1.0, "kernel_fingerprints_per_s": 5e5, "seconds": 9.9}
    assert not any(k.startswith("multiround_one_gpu") for k in bench.regressions(now, files)["worse"])


def test_multi_gpu_plan_fits_the_drivers_window():
    r"""`bench.py --gpus N --steps 20 --warmup 5` must finish inside the driver's 1 800 s; the shard is
    sized so that the planned wall time of the steps is <= 1 200 s for W = 2, 4, 8, and is the SAME for every N (equal work
    per GPU along the scaling curve, and for the `scale_anchor` of the --gpus 1 line)."""
    import subprocess

    import bench

    rows = set()
    for w in (1, 2, 4, 8):
        plan = bench.plan_multi(w, 20, 5)
        assert plan["projected_wall_s"] <= bench.PLAN_BUDGET_S <= 1200.0, plan
        assert plan["rows_per_shard"] >= 50_000 and plan["rows_per_shard"] % 50_000 == 0
        rows.add(plan["rows_per_shard"])
    assert len(rows) == 1
    assert bench.plan_multi(8, 2, 1)["rows_per_shard"] == bench.PLAN_MAX_ROWS  # the default K / W keep round 5's 2 M rows
    assert bench.plan_multi(8, 200, 5)["rows_per_shard"] >= 50_000
    assert bench.plan_multi(8, 20, 5, n_fps=12_500_000)["overridden_by_n_fps"]
    out = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-run"],
                         capture_output=True, text=True, check=True).stdout
    plan = json.loads(out)
    assert plan["world"] == 8 and plan["projected_wall_s"] <= 1200.0 and plan["rows_per_shard_stated"] == 12_500_000


def test_workgroups_busy_record():
    r"""`roofline.workgroups_busy`: the headline kernel is one workgroup; opt-in systolic sub-records are carried over only when
    they ran (an `error` record has no workgroup figures) and keep their in-process check's verdict."""
    import bench

    others = {
        "zipf_bf50": {"fingerprints_per_s": 1.0, "systolic_opt_in": {"workgroups": 97, "parallel_workgroups": 5.5, "identical_to_default_path": True}},
        "zipf_bf254": {"fingerprints_per_s": 1.0, "systolic_opt_in": {"error": "level-systolic kernel: internal error"}},
        "ecfp_bf50": {"fingerprints_per_s": 1.0},
        "headline_other_seeds": {"5000": {"fingerprints_per_s": 1.0}},
    }
    out = bench.workgroups_busy(others)
    assert out["headline_kernel"] == 1 and out["compute_units"] == 256
    assert out["systolic_opt_in_zipf_bf50"] == {"workgroups_resident": 97, "parallel_workgroups": 5.5, "identical_to_default_path": True}
    assert "systolic_opt_in_zipf_bf254" not in out and "systolic_opt_in_ecfp_bf50" not in out
    assert bench.workgroups_busy(None) == {"headline_kernel": 1, "compute_units": 256}
