#!/usr/bin/env bash
# A/B of kernel builds on the GPU box: tools/ab.sh <reps> <rows> "<bf ...>" lib1.so lib2.so ...   (alternating, fresh process each)
REPS=$1; ROWS=$2; BFS=$3; shift 3
for r in $(seq 1 $REPS); do
  for lib in "$@"; do
    echo "## rep $r $(basename $lib)"
    BBHIP_LIBRARY=$lib timeout 300 python tools/fit_workloads.py $ROWS $BFS 2>&1 | grep "=="
  done
done
