"""Resident memory of a HipLearner process (loop of engine/learner.py: ingest service, run-ahead events, publisher) without
actors: python scripts/debug/learner_rss.py r2d2|ape_x [steps] [transport]"""
import os
import sys
import tempfile
import time

import psutil

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_rl_amd.config import make_config  # noqa: E402
from distributed_rl_amd.engine.learner import HipLearner  # noqa: E402
from distributed_rl_amd.transport import connect  # noqa: E402

algo = sys.argv[1] if len(sys.argv) > 1 else "r2d2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
transport = sys.argv[3] if len(sys.argv) > 3 else "inproc"
cfg = make_config(algo, start_memory_size=64, fit_timing=1 << 30, save_timing=1 << 30)
exp, par = connect(transport, "rss%d" % os.getpid(), create=True)
pub = float(os.environ.get("PUB", "0.05"))
lr = HipLearner(algo, 4, cfg, exp, par, None, 4096 if algo == "r2d2" else 200000, "cuda", publish_interval=pub)
lr.replay.fill_synthetic(4096 if algo == "r2d2" else 200000, n_action=4, seed=0)
lr.publish(wait=True)
proc, t0, tmp = psutil.Process(os.getpid()), time.time(), tempfile.mkdtemp()
done = 0
if os.environ.get("ONE_CALL"):                        # one long optimize_loop call (what a real learner process does)
    import threading

    def watch():
        while True:
            time.sleep(10)
            mi = proc.memory_info()
            print("%s/%s one call: step %7d, %5.0f s, rss %d MB (shared %d)" % (algo, transport, lr.step, time.time() - t0,
                                                                               mi.rss >> 20, mi.shared >> 20), flush=True)
    threading.Thread(target=watch, daemon=True).start()
    lr.optimize_loop(max_steps=steps, save_model_dir=tmp)
    done = steps
while done < steps:
    lr.optimize_loop(max_steps=4096, save_model_dir=tmp)
    done += 4096
    mi = proc.memory_info()
    print("%s/%s pub %.2f: %7d steps, %5.0f s, rss %d MB (shared %d)" % (algo, transport, pub, done, time.time() - t0, mi.rss >> 20,
                                                                         mi.shared >> 20), flush=True)
lr.close()
