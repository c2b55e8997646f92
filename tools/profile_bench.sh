#!/usr/bin/env bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py, then separate
# PMC passes (FETCH_SIZE / WRITE_SIZE) as MI355X_MICROARCH.md prescribes.  Summaries land
# under gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/.
set -uo pipefail
TAG="${1:-r02}"
NFPS="${2:-200000}"
K1ROWS="${3:-8000000}"   # the arr-vec kernel is profiled on an array larger than the 256 MiB Infinity Cache
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
FINAL="$ROOT/gpurun_out/prof_$TAG"
OUT="/tmp/prof_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT" "$FINAL"
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --steps 1 --warmup 0 --n-fps $NFPS --no-cpu --shards 0 --multiround-files 0 --no-extras --k1-rows $K1ROWS"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $CMD > "$OUT/bench_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- $CMD > "$OUT/bench_pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- $CMD > "$OUT/bench_pmc_write.log" 2>&1
for f in "$OUT"/*.log; do grep -m1 metric "$f" | cut -c1-200; done
find "$OUT" -name "*.csv" | head -50
BB_ROOT="$ROOT" python - "$OUT" "$NFPS" <<'PY'
import csv, sys, glob, collections, os
out = sys.argv[1]
rec = {}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats", f)
    print(open(f).read()[:3000])
for name in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(out + f"/{name}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
                agg[k][0] += 1
                agg[k][1] += float(row.get("Counter_Value", 0) or 0)
        with open(os.path.join(out, name + "_summary.txt"), "w") as w:
            for (k, c), (n, v) in sorted(agg.items()):
                line = f"{k:60s} {c:12s} dispatches={n:6d} total={v:.1f} per_dispatch={v/max(n,1):.1f}"
                print(line); w.write(line + "\n")
                if "k_tree" in k:
                    key = "tree_" + ("fetch" if c == "FETCH_SIZE" else "write")
                    rec[key + "_kb_total"] = rec.get(key + "_kb_total", 0.0) + v   # all tree kernels (k_tree_pipe, k_tree_fast ...)
                    rec[key + "_dispatches"] = rec.get(key + "_dispatches", 0) + n
                    rec["tree_launches"] = rec[key + "_dispatches"]
                if "k_arr_vec<16, true>" in k:
                    rec["k1_" + ("fetch" if c == "FETCH_SIZE" else "write") + "_kb_per_dispatch"] = v / max(n, 1)
import json
rec["n_fps"] = int(sys.argv[2])
rec["fits"] = 2  # the profiled command: --steps 1 --warmup 0 + bench.py's end-to-end step (fit + labels)
sys.path.insert(0, os.environ.get("BB_ROOT", "."))
try:
    import bench
    rec["kernel_src_sha"] = bench.kernel_src_sha()
except Exception as exc:
    rec["kernel_src_sha"] = None
rec["note"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), raw KB as reported; FETCH_SIZE must be doubled on gfx950 (MI355X_MICROARCH.md)"
open(os.path.join(out, "pmc_latest.json"), "w").write(json.dumps(rec, indent=1))
print(rec)
PY

cp "$OUT"/*.log "$OUT"/*_summary.txt "$OUT"/pmc_latest.json "$FINAL"/ 2>/dev/null
for f in $(find "$OUT/trace" -name "*kernel_stats.csv"); do cp "$f" "$FINAL/kernel_stats.csv"; done
ls -la "$FINAL"
