#!/usr/bin/env bash
# Forward chunk cut points x backward chunk starts of the R2D2 step (R2D2Engine arguments fwd_chunks / bwd_cuts): ms per step.
out=${1:-gpurun_out/r2d2_chunk_sweep.txt}
: > "$out"
run() {
  echo "FWD=$1 BWD=$2" >> "$out"
  a=""
  if [ -n "$1" ]; then a="$a --engine-arg fwd_chunks=$1"; fi
  if [ -n "$2" ]; then a="$a --engine-arg bwd_cuts=$2"; fi
  timeout -k 10 120 python scripts/r2d2_bench.py --steps 40 --warmup 5 $a 2>&1 | grep "ms/step" >> "$out" || exit 1
}
run "" "" && run "16,64," "" && run "24,72," "" && run "16,56," "" && run "8,64," "" && run "32,64," "" && run "16,48,84," "" && run "" ""
