#!/usr/bin/env bash
# Roofline inputs of the two learner steps, per kernel, measured in situ (run on the GPU box from the repo root):
#   1. un-profiled kernel timing      rocprofv3 --kernel-trace --stats          (eager steps: one dispatch per kernel)
#   2. PMC pass A (matrix pipe)       MFMA busy cycles / MFMA ops / wave cycles / GRBM_GUI_ACTIVE
#   3. PMC pass B / C (memory)        FETCH_SIZE / WRITE_SIZE (bytes the L2 exchanged with HBM / Infinity Cache)
# Counter passes are runs of their own, never combined with trace domains other than --kernel-trace, and the program
# itself follows "--". Output: gpurun_out/roofline/<algo>_{stats,pmcA,pmcB,pmcC}.csv -> scripts/roofline.py
set -uo pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/roofline
mkdir -p $out
A="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
B="FETCH_SIZE GRBM_GUI_ACTIVE"
C="WRITE_SIZE GRBM_GUI_ACTIVE"
for algo in apex r2d2; do
  if [ $algo = apex ]; then prog="python bench.py --steps 40 --warmup 5 --settle 0 --no-extra --no-graph --no-system"; steps=47
  else prog="python scripts/r2d2_bench.py --steps 6 --warmup 2 --no-graph"; steps=8; fi
  rm -rf $out/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tmp -o k -- $prog > $out/$algo.log 2>&1 || { echo "$algo: timing run failed"; tail -3 $out/$algo.log; exit 1; }
  cp $out/tmp/k_kernel_stats.csv $out/${algo}_stats.csv
  for pass in A B C; do
    ctrs=$A; [ $pass = B ] && ctrs=$B; [ $pass = C ] && ctrs=$C
    rm -rf $out/tmp
    timeout -s KILL 240 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/tmp -o k -- $prog > $out/$algo.pmc$pass.log 2>&1 || { echo "$algo pass $pass failed"; tail -3 $out/$algo.pmc$pass.log; exit 1; }
    f=$(find $out/tmp -name "k_counter_collection.csv" | head -1)
    python - "$f" "$out/${algo}_pmc$pass.csv" <<'PY'
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen:
        seen.add((k, r["Dispatch_Id"])); n[k] += 1
names = sorted({c for v in acc.values() for c in v})
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["kernel", "dispatches"] + names)
for k, v in acc.items():
    w.writerow([k, n[k]] + [v.get(c, 0.0) for c in names])
PY
  done
  echo "$algo: done ($steps steps incl. warm-up and capture warm-up)"
done
rm -rf $out/tmp
