"""Direct xGMI all-reduce (csrc/hip/xgmi_ar.hip): W ranks as W processes. On a 1-GPU box they share the device - the IPC
mapping, the epoch-flag protocol, slices / channels / padding and graph replay are what is covered; link bandwidth is not."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(world, args, **extra_env):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
        procs.append(subprocess.Popen([sys.executable, "-m", "distributed_rl_amd.parallel.xgmi_check"] + args,
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=100)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res = []
    for rc, o, e in outs:
        assert rc == 0, (o[-2000:], e[-4000:])
        res.append(json.loads(o.strip().splitlines()[-1]))
    return res


@pytest.mark.parametrize("world", [2, 4])
def test_xgmi_all_reduce_processes(world):
    for res in _run_ranks(world, ["--iters", "2"]):
        assert res["ok"] and res["graph_ok"] and res["error_flag"] == 0 and res["checks"] == 10


@pytest.mark.parametrize("flag", ["--engine", "--engine-r2d2"])
def test_engine_data_parallel_over_direct_all_reduce(flag):
    """Two learner processes (Ape-X / R2D2), sharded replay, the all-reduce kernels captured in the one-graph step:
    identical parameters on both ranks after 12 updates. DRL_XGMI_TIMEOUT_S: the bound of the kernels' flag polls once the
    start-up verification (which runs with 1 s) has passed."""
    res = _run_ranks(2, [flag], DRL_XGMI_TIMEOUT_S="7.5")
    assert all(r["ok"] and r["graph"] == "one" and r["xgmi_timeout_s"] == 7.5 for r in res)
    assert res[0]["sums"] == res[1]["sums"]


def test_xgmi_all_reduce_soak_500_steps_on_changing_data():
    """500 collectives back to back on inputs that kernels rewrite before every one (no host barrier in between),
    alternating 16 / 32 / 64 workgroups and the channels; every result is checked against the host's sum over the ranks'
    inputs. Two ranks share the box's GPU: protocol and epochs are covered, the links are not."""
    for res in _run_ranks(2, ["--iters", "1", "--soak", "500", "--numel", "1048576"]):
        assert res["ok"] and res["error_flag"] == 0 and res["soak_steps"] == 500 and res["soak_worst_rel_err"] < 1e-4, res


def test_xgmi_all_reduce_on_fine_grained_memory():
    """The fallback memory kind parallel/dist.py switches to when the cached buffer fails its verification on a node."""
    for res in _run_ranks(2, ["--iters", "1", "--mem", "fine", "--soak", "60", "--numel", "262144"]):
        assert res["ok"] and res["graph_ok"] and res["error_flag"] == 0, res
