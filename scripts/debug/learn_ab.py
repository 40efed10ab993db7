"""A/B of scripts/learn_catch.py (HIP learner, Ape-X) with pieces of the learner switched: which one changed the curve?"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scripts"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import learn_catch  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "default"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
from distributed_rl_amd.engine import apex, ingest  # noqa: E402

if variant == "noride":
    apex.ApexEngine.ride_norm = property(lambda self: False)
if variant == "slowingest":
    ingest.AsyncIngest.fast_path = False
    _init = ingest.AsyncIngest.__init__

    def init(self, *a, **k):
        _init(self, *a, **k)
        self.fast_path = False
    ingest.AsyncIngest.__init__ = init
r = learn_catch.train("ape_x", "hip", rounds=30, seed=seed)
print(json.dumps({"variant": variant, "seed": seed, "final": r["final"], "curve": r["curve"]}), flush=True)
