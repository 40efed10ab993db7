"""Per-seed spread of the production-shape R2D2 gradient comparison (engine vs fp32 autograd): is a tensor's error the
data's or the kernels'? Prints cosine / relative error / reference norm of every gradient tensor for several seeds."""
import sys
import torch
sys.path.insert(0, "tests")
from test_engine_gpu import _r2d2_setup  # noqa: E402
from test_production_gpu import _cmp  # noqa: E402
from distributed_rl_amd.models import DuelingLSTMDQN  # noqa: E402
from distributed_rl_amd.ops import oracle  # noqa: E402

for seed in (1, 2, 3, 4):
    model, cfg, rep, eng = _r2d2_setup(B=64, burn=40, seq=80, n=5, A=4, cap=128, seed=seed)
    target = DuelingLSTMDQN(eng.A, 64, cfg.burn_in, cfg.nstep)
    target.load_state_dict(model.state_dict())
    model.to("cuda"); target.to("cuda")
    eng._forward_backward()
    torch.cuda.synchronize()
    idx = rep.idx.long()
    frames = rep.gather_stacks(idx).permute(1, 0, 2, 3, 4).contiguous()
    a, r, d = rep.act[idx].t().long(), rep.rew[idx].t(), rep.done[idx].t()
    st = eng.stats()
    w = oracle.is_weights(rep.prio.cpu(), st["total_prio"], st["n_items"], st["beta"]).cuda()
    model.zero_grad()
    td = model.sequence_td(target, frames, a, r, d, rep.length[idx].long(), rep.h0[idx], rep.c0[idx], w, cfg.eta, cfg.gamma_n,
                           cfg.rescale_eps)
    td.loss.backward()
    rows = _cmp(eng.G.state_dict(), model)
    norms = {n: float(p.grad.norm()) for n, p in model.named_parameters()}
    print("seed", seed, "unique idx", int(idx.unique().numel()),
          {k: ("%.5f" % c, "%.4f" % e, "%.3g" % norms[k]) for k, (c, e) in rows.items() if k.endswith("weight")}, flush=True)
