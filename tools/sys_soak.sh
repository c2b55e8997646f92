#!/usr/bin/env bash
# The randomised suites under BBHIP_SYS=1 (every tree the shape allows goes through the level-systolic kernel), on the GPU box:
#   bash tools/sys_soak.sh [lo:hi seeds of tests/test_hip_pipe_fuzz.py, default 0:200]
# test_pipe_moves_between_single_and_multi_level_instances / test_pipe_fuzz_reaches_the_pipeline assert launch counts of the
# PIPELINED kernel and are left out.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
BBHIP_SYS=1 BB_FUZZ_SEEDS="${1:-0:200}" python -m pytest tests/test_hip_pipe_fuzz.py -m gpu -q -x \
  -k "test_pipe_fuzz_vs_oracle" > gpurun_out/sys_soak.txt 2>&1
tail -n 5 gpurun_out/sys_soak.txt
