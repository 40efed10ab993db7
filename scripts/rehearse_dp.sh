#!/usr/bin/env bash
# Rehearse the N-rank data-parallel bench on ONE GPU (run on the GPU box through gpurun): ranks share the device, the
# process group is gloo (RCCL refuses two ranks on one GPU), the gradient all-reduce is the direct xGMI kernel over IPC.
#   scripts/rehearse_dp.sh [ranks=2] [steps=200]
set -uo pipefail
cd "$(dirname "$0")/.."
n=${1:-2}; steps=${2:-200}
export DRL_DIST_BACKEND=gloo DRL_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus "$n" --steps "$steps" --warmup 20 --replay 100000 2> gpurun_out/rehearse_dp.err | tail -1
