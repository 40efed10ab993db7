#!/usr/bin/env bash
# rocprofv3 recipes used for the summaries under profiles/ (run on the GPU box, from the repo root).
#   scripts/profile.sh apex   -> per-kernel time of the Ape-X learner step      -> gpurun_out/prof_apex/
#   scripts/profile.sh r2d2   -> per-kernel time of the R2D2 learner step       -> gpurun_out/prof_r2d2/
#   scripts/profile.sh pmc <kernel-substring> <counters...>  -> one PMC pass over scripts/kbench.py --only <substring>
# Kernel-trace runs and PMC runs are separate on purpose (counters + trace domains in one run are refused on this
# pool); the program itself follows "--" (no env / bash -c hop under the profiler).
set -euo pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mode=${1:-apex}
case "$mode" in
  apex)
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_apex -o apex -- \
      python bench.py --steps 200 --warmup 20 --no-extra --no-graph > gpurun_out/prof_apex.log 2>&1
    rm -f gpurun_out/prof_apex/apex_kernel_trace.csv
    python scripts/prof_summary.py gpurun_out/prof_apex/apex_kernel_stats.csv "Ape-X learner step" 222 ;;
  r2d2)
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2d2 -o r2d2 -- \
      python scripts/r2d2_bench.py --steps 10 --warmup 3 --no-graph > gpurun_out/prof_r2d2.log 2>&1
    rm -f gpurun_out/prof_r2d2/r2d2_kernel_trace.csv
    python scripts/prof_summary.py gpurun_out/prof_r2d2/r2d2_kernel_stats.csv "R2D2 learner step" 13 ;;
  pmc)
    only=$2; shift 2
    timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc -o k -- \
      python scripts/kbench.py --only "$only" --eager --reps 5 > gpurun_out/pmc.log 2>&1
    python - <<'PY'
import collections, csv
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("gpurun_out/pmc/k_counter_collection.csv")):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    if k.startswith("void k_") or k.startswith("k_"):
        print(k, dict(v))
PY
    ;;
  *) echo "usage: $0 apex|r2d2|pmc <substr> <counters...>"; exit 2 ;;
esac
