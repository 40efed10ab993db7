"""Ape-X step with the FC backward launch's GEMM bodies in either order behind the riders (ops.hip.FC_BWD_VARIANT: 5 = dW tiles
first, 2 = dX tiles first, 4 / 1 = the same with a deeper ring for dX)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_rl_amd.config import ApexConfig  # noqa: E402
from distributed_rl_amd.engine.apex import ApexEngine  # noqa: E402
from distributed_rl_amd.ops import hip as H  # noqa: E402
from distributed_rl_amd.replay.device_per import DevicePER  # noqa: E402

rep = DevicePER(2_000_000, "cuda", batch_size=512, seed=1)
rep.fill_synthetic(2_000_000, n_action=4, seed=0)
# arguments: fc,conv1,conv2,conv3[,wgrad_splits] variant tuples, e.g. 2,0,4,2 or 2,0,4,2,48
cfgs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(5, 0, 4, 2), (2, 0, 4, 2)]
for v, c1, c2, c3, *rest in cfgs:
    sp = (rest[0] if len(rest) == 1 else tuple(rest)) if rest else 32
    H.FC_BWD_VARIANT = v
    H.CONV_BWD_VARIANT = {1: c1, 2: c2, 3: c3}
    torch.manual_seed(0)
    e = ApexEngine(4, ApexConfig(batch_size=512), rep, "cuda", wgrad_splits=sp, ride_blocks=int(os.environ.get("RB", "512")),
                   steps_per_graph=int(os.environ.get("KG", "4")))
    e.step()
    e.run_steps(400)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.run_steps(2000)
    torch.cuda.synchronize()
    print("variants fc %d conv1 %d conv2 %d conv3 %d splits %s: %.4f ms/step" % (v, c1, c2, c3, sp, 1e3 * (time.perf_counter() - t0) / 2000), flush=True)
    del e
