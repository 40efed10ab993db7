#!/usr/bin/env bash
# Two PMC passes (8 SQ counters each, never combined with trace domains other than --kernel-trace) over a list of
# scripts/kbench.py cases; prints one dict of summed counters per kernel and case. Used for profiles/pmc_summary.md.
#   scripts/pmc_pass.sh fc1_fwd_v10 fc1_bwd_fused_v5 ...
set -uo pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"
B="SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES"
for case in "$@"; do
  for pass in A B; do
    ctrs=$A; [ $pass = B ] && ctrs=$B
    echo "## $case (pass $pass)"
    rm -rf gpurun_out/pmc
    timeout -s KILL 120 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmc -o k -- \
      python scripts/kbench.py --only "$case" --eager --reps 3 > gpurun_out/pmc.log 2>&1 || { echo "pass failed"; exit 1; }
    python - <<'PY'
import collections, csv, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc/**/k_counter_collection.csv", recursive=True) + glob.glob("gpurun_out/pmc/k_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
    break
for k, v in acc.items():
    if any(t in k for t in ("k_multi", "k_gemm_glds", "k_lstm", "k_apex_td_dh", "k_conv1_direct", "k_head_fwd_dual")) and "void at::" not in k:
        print(k, dict(v))
PY
  done
done
rm -rf gpurun_out/pmc
