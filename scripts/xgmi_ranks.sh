#!/bin/bash
# Start W ranks of the direct all-reduce self-check (python -m distributed_rl_amd.parallel.xgmi_check) on this node:
#   scripts/xgmi_ranks.sh W [args of xgmi_check...]
# Ranks share GPUs round-robin when there are fewer GPUs than ranks.
W=$1; shift
PORT=${MASTER_PORT:-29577}
pids=()
for ((r = 0; r < W; r++)); do
  RANK=$r WORLD_SIZE=$W MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT HSA_ENABLE_IPC_MODE_LEGACY=0 \
    timeout -k 10 150 python -m distributed_rl_amd.parallel.xgmi_check "$@" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
exit $rc
