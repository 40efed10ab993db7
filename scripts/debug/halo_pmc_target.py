"""PMC target: a few launches of one conv forward kernel at a fixed shape (run under rocprofv3 --pmc).
   python scripts/debug/halo_pmc_target.py <layer> <frames1> <frames2> <halo variant | -1 = tiled>"""
import sys

import torch

sys.path.insert(0, ".")
from distributed_rl_amd.ops import hip as H  # noqa: E402

layer, f1, f2, v = (int(a) for a in sys.argv[1:5])
ishape, oshape, K = {2: ((20, 20, 32), (9, 9, 64), 512), 3: ((9, 9, 64), (7, 7, 64), 576)}[layer]
g = torch.Generator().manual_seed(0)
nets = []
for f in (f1, f2):
    x = torch.relu(torch.randn(f, *ishape, generator=g)).to(torch.bfloat16).cuda()
    w = (torch.randn(64, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    b = (torch.randn(64, generator=g) * 0.1).cuda()
    nets.append((x, f, w, b, torch.zeros(f, *oshape, dtype=torch.bfloat16, device="cuda")))
(x1, _, w1, b1, o1), (x2, _, w2, b2, o2) = nets
for _ in range(5):
    if v >= 0:
        H.conv_halo_fwd(layer, x1, f1, w1, b1, o1, x2, f2, w2, b2, o2, variant=v)
    else:
        H.native.hip().drl_conv_fwd_dual(layer, H._p(x1), f1, H._p(w1), H._p(b1), H._p(o1), H._p(x2), f2, H._p(w2), H._p(b2),
                                         H._p(o2), H._s())
torch.cuda.synchronize()
