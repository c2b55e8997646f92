#!/usr/bin/env bash
# Round-end evidence on the GPU box, one gpurun call (outputs under gpurun_out/; copy what should be judged into profiles/):
#   1. rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of the bench command (tools/profile_bench.sh)
#   2. the bench line itself, quoting those counters (profiles/pmc_latest.json must describe THIS kernel source)
#   3. smoke(), 4. the GPU test suite
# usage: bash tools/final_evidence.sh <tag, e.g. r05>
set -u
TAG="${1:-r05}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
mkdir -p gpurun_out
bash tools/profile_bench.sh "$TAG" 1000000 > "gpurun_out/profile_$TAG.log" 2>&1
if [ -f "gpurun_out/prof_$TAG/pmc_latest.json" ]; then cp "gpurun_out/prof_$TAG/pmc_latest.json" profiles/pmc_latest.json; fi
cd "$ROOT"
( time python bench.py > gpurun_out/bench_line_final.json 2> gpurun_out/bench_final.err ) 2> gpurun_out/bench_final.time
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
python -m pytest tests -m gpu -q -x > gpurun_out/gputest_final.log 2>&1
tail -n 2 gpurun_out/gputest_final.log gpurun_out/smoke.log gpurun_out/bench_final.time
head -c 400 gpurun_out/bench_line_final.json
