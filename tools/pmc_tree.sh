#!/usr/bin/env bash
# SQ counter passes for the tree kernel (instruction mix, wait cycles); run on the GPU box.
set -uo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
N="${1:-200000}"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_tree; mkdir -p /tmp/pmc_tree
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" \
           ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_tree/p$i -o t -- python $ROOT/tools/fit_speed.py $N 1 > /tmp/pmc_tree/log$i.txt 2>&1
done
python - "$N" <<'PY'
import csv, glob, sys, collections
n = int(sys.argv[1])
agg = collections.defaultdict(float)
for f in glob.glob("/tmp/pmc_tree/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_tree" in row.get("Kernel_Name", ""):
            agg[row["Counter_Name"]] += float(row["Counter_Value"] or 0)
for k in sorted(agg):
    print(f"{k:24s} total={agg[k]:.4g}  per_insert={agg[k]/n:.1f}")
PY
