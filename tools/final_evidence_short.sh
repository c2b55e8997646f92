#!/usr/bin/env bash
# The reduced form of tools/final_evidence.sh for a change confined to the bf > 255 compare: its parity cases, the profile
# passes (kernel stats + FETCH_SIZE / WRITE_SIZE) and the bench line on the same sources; the whole GPU suite is NOT re-run.
set -u
TAG="${1:-r05}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_hip_tree.py -m gpu -q -x -k "block_compare" > gpurun_out/block_tests_final.txt 2>&1
timeout 120 bash tools/profile_bench.sh "$TAG" 1000000 > "gpurun_out/profile_$TAG.log" 2>&1
if [ -f "gpurun_out/prof_$TAG/pmc_latest.json" ]; then cp "gpurun_out/prof_$TAG/pmc_latest.json" profiles/pmc_latest.json; fi
cd "$ROOT"
( time timeout 270 python bench.py > gpurun_out/bench_line_final.json 2> gpurun_out/bench_final.err ) 2> gpurun_out/bench_final.time
tail -n 2 gpurun_out/block_tests_final.txt gpurun_out/bench_final.time
head -c 300 gpurun_out/bench_line_final.json
