#!/usr/bin/env bash
# Starts of the backward time chunks of the R2D2 step (R2D2Engine argument bwd_cuts, window steps of 80): ms per step.
out=${1:-gpurun_out/r2d2_bwd_cut_sweep.txt}
: > "$out"
run() {
  echo "BWD=$1" >> "$out"
  a=""
  if [ -n "$1" ]; then a="--engine-arg bwd_cuts=$1"; fi
  timeout -k 10 120 python scripts/r2d2_bench.py --steps ${STEPS:-60} --warmup 5 $a 2>&1 | grep "ms/step" >> "$out" || exit 1
}
for c in ${CUTS:-"" 40 52 56 60 64 "40,64" "36,60" "44,66" "48,68" "32,56,70" ""}; do run "$c" || exit 1; done
