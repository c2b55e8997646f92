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
    rec = bench.regressions(r04, REPO / "profiles" / "r03" / "bench_line_final.json")
    assert rec["against"] == "profiles/r03/bench_line_final.json" and rec["compared"] >= 8
    assert set(rec["worse"]) == {"bf254.fingerprints_per_s", "concurrent_shards.fingerprints_per_s"}
    assert rec["worse"]["concurrent_shards.fingerprints_per_s"]["ratio"] < 0.2
    # a line against itself: nothing worse, nothing better
    same = bench.regressions(r04, REPO / "profiles" / "r04" / "bench_line_final.json")
    assert same["worse"] == {} and same["better"] == {}
    # a missing file is reported, not raised
    assert "error" in bench.regressions(r04, REPO / "profiles" / "r00" / "nothing.json")


def test_round5_line_has_no_gpu_side_regression():
    import bench

    r05 = json.loads((REPO / "profiles" / "r05" / "bench_line_final.json").read_text())
    rec = bench.regressions(r05, REPO / "profiles" / "r04" / "bench_line_final.json")
    # (the file-based multiround follows its host: DESIGN section 8, round 5)
    assert set(rec["worse"]) <= {"multiround_one_gpu.fingerprints_per_s"}
    assert rec["better"]["bf254.fingerprints_per_s"] > 1.1 and rec["better"]["concurrent_shards.fingerprints_per_s"] > 5


def test_tail_row_bytes_single_definition():
    from bblean_amd.multiround import _tail_row_bytes

    assert _tail_row_bytes(2049) == 256 and _tail_row_bytes(65) == 8 and _tail_row_bytes(2041) == 255
