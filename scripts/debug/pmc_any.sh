#!/usr/bin/env bash
# PMC passes over an arbitrary python target: scripts/debug/pmc_any.sh <kernel-substring> -- python <script> <args...>
# Three passes (<= 8 SQ + 2 GRBM counters each, never with trace domains other than --kernel-trace); prints the counters of
# the LAST launch of every kernel whose name contains the substring.
set -uo pipefail
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
sub=$1; shift; shift
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
B="SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
C="SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
for pass in A B C; do
  ctrs=$A; [ $pass = B ] && ctrs=$B; [ $pass = C ] && ctrs=$C
  rm -rf gpurun_out/pmc
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmc -o k -- "$@" > gpurun_out/pmc.log 2>&1 || { echo "pass $pass failed"; tail -5 gpurun_out/pmc.log; exit 1; }
  SUB="$sub" PASS=$pass python - <<'PY'
import collections, csv, glob, os
last = {}
for f in glob.glob("gpurun_out/pmc/**/k_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ["SUB"] in r["Kernel_Name"]:
            last.setdefault((r["Kernel_Name"][:60], int(r["Dispatch_Id"])), {})[r["Counter_Name"]] = float(r["Counter_Value"])
if last:
    k = max(last, key=lambda t: t[1])
    print("pass", os.environ["PASS"], k[0], {a: int(b) for a, b in last[k].items()})
PY
done
rm -rf gpurun_out/pmc
