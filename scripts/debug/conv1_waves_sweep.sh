#!/bin/bash
# Sweep of the wave slots per CU the conv1 forward tiles are dealt to (DRL_CONV1_WAVES): Ape-X and R2D2 step times.
out=${1:-gpurun_out/conv1_waves_sweep.txt}
: > "$out"
for w in 8 4 6 12 16 8; do
  echo "DRL_CONV1_WAVES=$w" >> "$out"
  DRL_CONV1_WAVES=$w timeout -k 10 120 python bench.py --steps 3000 --warmup 200 --no-extra --no-system 2>/dev/null | python -c "import json,sys; print('apex ms/step', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" >> "$out" || exit 1
  DRL_CONV1_WAVES=$w timeout -k 10 120 python scripts/r2d2_bench.py --steps 40 --warmup 5 2>&1 | grep "ms/step" >> "$out" || exit 1
done
