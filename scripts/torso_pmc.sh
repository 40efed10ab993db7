#!/usr/bin/env bash
# PMC passes over the fused torso kernel (scripts/torso_bench.py --eager): one 8-counter SQ pass per run, summed per kernel.
#   scripts/torso_pmc.sh "r2d2 chunk" [dbg]
set -uo pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
CASE="${1:-r2d2 chunk}"; DBG="${2:-0}"
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"
B="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
C="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"
for pass in A B C; do
  ctrs=$A; [ $pass = B ] && ctrs=$B; [ $pass = C ] && ctrs=$C
  echo "## $CASE dbg=$DBG (pass $pass)"
  rm -rf gpurun_out/pmc
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmc -o k -- \
    python scripts/torso_bench.py --only "$CASE" --eager --dbg "$DBG" > gpurun_out/pmc.log 2>&1 || { echo "pass failed"; tail -5 gpurun_out/pmc.log; exit 1; }
  python - <<'PY'
import collections, csv, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc/**/k_counter_collection.csv", recursive=True) + glob.glob("gpurun_out/pmc/k_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    break
for k, v in acc.items():
    if "k_torso" in k:
        print(k, dict(v))
PY
done
rm -rf gpurun_out/pmc
