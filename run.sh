#!/bin/bash
# ./run.sh <n_actors> [conf] [server_host] [-- extra launcher flags]
# Same contract as the reference's run.sh:4-13 (conf keys actor/learner/visdom/algorithm/redis_server/
# visdom_server/actor_device). The conf file is *parsed*, never sourced, and `learner=false` is honoured
# (the reference tests `$leaner`, a typo, and always starts a learner — run.sh:26). Supervision, fail-fast
# teardown (run.sh:3,44) and heartbeat watchdog live in distributed_rl_amd/runtime/launcher.py.
cd "$(dirname "$0")"
n_actors=${1:-4}
config=${2:-"config/all.conf"}
server=${3:-""}
shift; shift; shift
exec python -m distributed_rl_amd.runtime.launcher --num-actors "$n_actors" --conf "$config" --server "$server" "$@"
