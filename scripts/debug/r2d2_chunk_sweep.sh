#!/bin/bash
# A/B sweep of the R2D2 step's time chunking (forward cut list / backward cut list): one r2d2_bench.py run per setting.
# usage: scripts/debug/r2d2_chunk_sweep.sh <out-file>
out=${1:-gpurun_out/r2d2_chunk_sweep.txt}
: > "$out"
run() {
  echo "FWD=$1 BWD=$2" >> "$out"
  if [ -n "$1" ]; then export DRL_FWD_CHUNKS="$1"; else unset DRL_FWD_CHUNKS; fi
  if [ -n "$2" ]; then export DRL_BWD_CUTS="$2"; else unset DRL_BWD_CUTS; fi
  timeout -k 10 120 python scripts/r2d2_bench.py --steps 40 --warmup 5 2>&1 | grep "ms/step" >> "$out" || exit 1
}
run "" "" && run "16,64," "" && run "24,72," "" && run "16,56," "" && run "8,64," "" && run "32,64," "" && run "16,48,84," "" && run "" ""
