#!/usr/bin/env bash
# Row groups of the persistent BPTT x backward chunk starts (R2D2Engine arguments bptt_groups / bwd_cuts): ms per step.
set -e
out=gpurun_out/bptt_groups_cuts.txt
: > $out
for g in 1 2 4; do
  for cuts in 48 56 64 40,64 32,56 48,64 56,68; do
    echo "== bptt_groups=$g bwd_cuts=$cuts" >> $out
    timeout -k 10 200 python scripts/r2d2_bench.py --steps 200 --warmup 20 --engine-arg bptt_groups=$g --engine-arg bwd_cuts=$cuts 2>&1 | grep r2d2 >> $out
  done
done
