#!/bin/bash
# conv1 forward prefetch depth A/B: kernel alone (Ape-X- and R2D2-sized launches) and both learner steps.
set -e
out=gpurun_out/conv1_prefetch.txt
: > $out
timeout -k 10 200 python -u -m pytest tests/test_hip_kernels.py -k "conv_fwd or conv1_fwd" -x -q --timeout 120 --timeout-method thread >> $out 2>&1
timeout -k 10 120 python scripts/kbench.py --frames 1536 --only conv1_fwd_d >> $out 2>&1
timeout -k 10 120 python scripts/kbench.py --frames 3840 --only conv1_fwd_d >> $out 2>&1
timeout -k 10 200 python scripts/r2d2_bench.py --steps 200 --warmup 20 2>&1 | grep r2d2 >> $out
timeout -k 10 200 python bench.py --no-extra --steps 2000 --warmup 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('apex ms/step', d['ms_per_step'])" >> $out
