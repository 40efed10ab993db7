#!/usr/bin/env bash
# Cut of the pipelined forward of the R2D2 step (R2D2Engine argument fwd_chunks = "cut,", time steps of 120): ms per step.
out=${1:-gpurun_out/r2d2_fwd_cut_sweep.txt}
: > "$out"
for c in ${CUTS:-"" 56 60 64 68 72 76 ""}; do
  echo "FWD=$c" >> "$out"
  a=""
  if [ -n "$c" ]; then a="--engine-arg fwd_chunks=$c,"; fi
  timeout -k 10 120 python scripts/r2d2_bench.py --steps ${STEPS:-150} --warmup 5 $a 2>&1 | grep "ms/step" >> "$out" || exit 1
done
