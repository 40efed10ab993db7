"""Resident memory of a learner process over many graph replays (host-side growth of the HIP runtime per hipGraphLaunch?):
   python scripts/debug/graph_rss.py r2d2|apex [steps] [recapture_every]"""
import os
import sys
import time

import psutil
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
algo = sys.argv[1] if len(sys.argv) > 1 else "r2d2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
recap = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if algo == "r2d2":
    from distributed_rl_amd.config import R2D2Config
    from distributed_rl_amd.engine.r2d2 import R2D2Engine
    from distributed_rl_amd.replay.device_seq import DeviceSequenceReplay
    cfg = R2D2Config()
    rep = DeviceSequenceReplay(4096, "cuda", cfg, seed=0)
    rep.fill_synthetic(4096, n_action=4, seed=0)
    eng = R2D2Engine(4, cfg, rep, "cuda")
    chunk, many = 4096, None
else:
    from distributed_rl_amd.config import ApexConfig
    from distributed_rl_amd.engine.apex import ApexEngine
    from distributed_rl_amd.replay.device_per import DevicePER
    cfg = ApexConfig(batch_size=512)
    rep = DevicePER(200000, "cuda", batch_size=512, seed=0)
    rep.fill_synthetic(200000, n_action=4, seed=0)
    eng = ApexEngine(4, cfg, rep, "cuda")
    chunk, many = 16384, eng.run_steps
eng.step()
proc = psutil.Process(os.getpid())
done, t0 = 0, time.time()
while done < steps:
    if many is not None:
        many(chunk)
    elif os.environ.get("LOOP") == "events":          # the learner loop's run-ahead bound: an event per step, wait for the 3rd last
        from collections import deque
        infl = deque()
        for i in range(chunk):
            eng.step()
            ev = torch.cuda.Event()
            ev.record()
            infl.append(ev)
            if len(infl) > 3:
                infl.popleft().synchronize()
    else:
        for i in range(chunk):
            eng.step()
            if i % 64 == 63:
                torch.cuda.synchronize()
    torch.cuda.synchronize()
    done += chunk
    if recap and done % recap == 0:
        eng._graphs = None
        if hasattr(eng, "_graph_k"):
            eng._graph_k = None
        eng.step()
    print("%s: %7d steps, %5.0f s, rss %d MB" % (algo, done, time.time() - t0, proc.memory_info().rss >> 20), flush=True)
