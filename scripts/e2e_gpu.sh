#!/usr/bin/env bash
# End-to-end on ONE GPU box: ./run.sh 2 with the gfx950 engines in all three processes (learner: HBM replay +
# graph-captured step; actors: batched ActorEngine / R2D2ActorEngine), shm transport, bounded learner steps.
#   scripts/e2e_gpu.sh [ape_x|r2d2]
set -euo pipefail
cd "$(dirname "$0")/.."
algo=${1:-ape_x}
conf=config/all.conf; [ "$algo" = r2d2 ] && conf=config/all_r2d2.conf; [ -n "${DRL_E2E_CONF:-}" ] && conf=$DRL_E2E_CONF
save=$(mktemp -d /tmp/drl_e2e_XXXX)
extra="--batch-size 32 --start-memory-size 64"
abatch=16
if [ "$algo" = r2d2 ]; then extra="--batch-size 16 --start-memory-size 16"; abatch=4; fi
DRL_NUM_ENVS=8 timeout -k 10 280 ./run.sh 2 "$conf" "" --run-id "e2e$$" --timeout 240 \
  --learner-args "--max-steps 30 $extra --save-timing 10 --save-dir $save --quiet --engine hip" \
  --actor-args "--quiet --engine hip --actor-batch $abatch"
ls -l "$save"
test -f "$save/model_10.pth" && echo "E2E $algo OK"
