"""Host cost of the learner-side ingest (engine/ingest.py service()) per experience block, by stage.

Plane-pool blocks shaped like the vectorised actors' (one run of 50 transitions = 56 planes) go through a real shm ring
into a FULL replay (every reservation evicts), with the GPU otherwise idle: what is measured is the host time the
learner thread spends per block - the number that shows up as ``learner/ingest`` in the whole-system bench.

    python scripts/ingest_bench.py [--blocks 3000] [--algo ape_x|r2d2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=3000)
    ap.add_argument("--run", type=int, default=50)
    ap.add_argument("--per-service", type=int, default=2, help="blocks pushed between two service() calls")
    a = ap.parse_args()
    from distributed_rl_amd.engine.ingest import AsyncIngest
    from distributed_rl_amd.replay.device_per import DevicePER
    from distributed_rl_amd.transport import ShmExperience, codec, shm_names
    dev = torch.device("cuda")
    rep = DevicePER(5000, dev, batch_size=512, seed=0)
    rep.fill_synthetic(5000, chunk=50, run=50)
    ring = ShmExperience(shm_names("ingbench%d" % os.getpid())[0], create=True, n_slots=64, slot_bytes=1 << 20)
    pipe = AsyncIngest(ring, rep, dev, depth=48, slot_bytes=1 << 20)
    rng = np.random.default_rng(0)
    per = a.run + 6
    K = a.run
    ws = np.arange(K, dtype=np.int32)
    blk = dict(planes=rng.integers(0, 256, (per, 84, 84), dtype=np.uint8), win_s=ws, win_s2=ws + 3,
               action=rng.integers(0, 4, K).astype(np.int32), reward=rng.standard_normal(K).astype(np.float32),
               done=np.zeros(K, np.float32), prio=(rng.random(K) + 0.5).astype(np.float32))
    acc = {}

    def timed(obj, name):
        fn = getattr(obj, name)

        def w(*x, **k):
            t0 = time.perf_counter()
            r = fn(*x, **k)
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
            return r
        setattr(obj, name, w)
    for n in ("reserve_block", "retire_ranges", "write_staged", "commit_blocks"):
        timed(rep, n)
    from distributed_rl_amd.ops import hip as H
    from distributed_rl_amd import native
    timed(H, "ingest_block")
    timed(H, "sumtree_ingest")

    class _Lib:                                                 # time spent inside the native calls themselves
        def __init__(self, lib):
            self._lib = lib

        def __getattr__(self, name):
            fn = getattr(self._lib, name)
            if name not in ("drl_ingest_block", "drl_sumtree_ingest"):
                return fn

            def w(*x):
                t0 = time.perf_counter()
                r = fn(*x)
                acc["native:" + name] = acc.get("native:" + name, 0.0) + time.perf_counter() - t0
                return r
            return w
    lib = _Lib(native.hip())
    native.hip = lambda: lib
    try:
        sent = 0
        t_service = 0.0
        calls = 0
        warm = 200
        t_end = time.time() + 120
        while pipe.blocks < a.blocks + warm and time.time() < t_end:
            for _ in range(a.per_service):
                if ring.push(blk, codec.KIND_APEX_PLANES, 1, sent):
                    sent += 1
            time.sleep(0.0002)                                  # the learner's step launch + the thread's staging
            if pipe.blocks >= warm and "t0" not in acc:
                acc.clear(); acc["t0"] = 1.0; t_service = 0.0; calls = 0; warm = pipe.blocks
            t0 = time.perf_counter()
            pipe.service()
            t_service += time.perf_counter() - t0
            calls += 1
        torch.cuda.synchronize()
        nb = max(pipe.blocks - warm, 1)
        out = dict(blocks=pipe.blocks, service_calls=calls, us_per_block=round(t_service / nb * 1e6, 1),
                   us_per_call=round(t_service / max(calls, 1) * 1e6, 1),
                   stages_us_per_block={k: round(v / nb * 1e6, 1) for k, v in acc.items() if k != "t0"})
        print(json.dumps(out))
    finally:
        pipe.close()
        ring.close()


if __name__ == "__main__":
    main()
