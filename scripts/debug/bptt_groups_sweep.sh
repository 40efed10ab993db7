#!/bin/bash
# Row groups of the persistent BPTT x cuts of the backward window: whole R2D2 step (r2d2_bench, graph replay).
set -e
out=gpurun_out/bptt_groups_cuts.txt
: > $out
for g in 1 2 4; do
  for cuts in 48 56 64 40,64 32,56 48,64 56,68; do
    echo "== DRL_BPTT_GROUPS=$g DRL_BWD_CUTS=$cuts" >> $out
    DRL_BPTT_GROUPS=$g DRL_BWD_CUTS=$cuts timeout -k 10 200 python scripts/r2d2_bench.py --steps 200 --warmup 20 2>&1 | grep r2d2 >> $out
  done
done
