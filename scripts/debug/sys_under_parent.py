"""Does freeing tens of GB of HBM in the parent right before the system bench slow its children down?"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def run(tag):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "system_bench.py"), "--algorithm", "ape_x", "--seconds", "5"],
                       capture_output=True, text=True, timeout=240, cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT))
    last = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    d = json.loads(last[-1]) if last else {"error": r.stderr[-300:]}
    print(tag, d.get("error"), d.get("learner_updates_per_s"), d.get("env_steps_per_s"), d.get("actor_phase_ms"), flush=True)

import torch
x = torch.zeros(100 << 30, dtype=torch.uint8, device="cuda")
x.fill_(3)
torch.cuda.synchronize()
del x
torch.cuda.empty_cache()
run("right after freeing 100 GB ")
run("second run                 ")
