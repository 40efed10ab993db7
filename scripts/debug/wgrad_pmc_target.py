"""PMC target: a few launches of one LDS-halo weight-gradient kernel at a fixed shape (run under rocprofv3 --pmc).
   python scripts/debug/wgrad_pmc_target.py <layer 1|2|3> <frames>"""
import sys

import torch

sys.path.insert(0, ".")
from distributed_rl_amd.ops import hip as H  # noqa: E402

layer, f = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator().manual_seed(0)
ws = H.conv_wgrad_ws(layer, "cuda")
if layer == 1:
    store = torch.randint(0, 256, (f + 3, 7056), dtype=torch.uint8, generator=g).cuda()
    off = (torch.randperm(f, generator=g) * 7056).cuda()
    dy = (torch.randn(f, 20, 20, 32, generator=g) * 0.1).to(torch.bfloat16).cuda()
    dw, db = torch.zeros(32, 256, device="cuda"), torch.zeros(32, device="cuda")
    fn = lambda: H.conv1_wgrad_halo(store, off, dy, dw, db, f, ws)                     # noqa: E731
else:
    ishape, oshape, K = {2: ((20, 20, 32), (9, 9, 64), 512), 3: ((9, 9, 64), (7, 7, 64), 576)}[layer]
    x = torch.relu(torch.randn(f, *ishape, generator=g)).to(torch.bfloat16).cuda()
    dy = (torch.randn(f, *oshape, generator=g) * 0.1).to(torch.bfloat16).cuda()
    dw, db = torch.zeros(64, K, device="cuda"), torch.zeros(64, device="cuda")
    fn = lambda: H.conv_wgrad_halo(layer, x, dy, dw, db, f, ws)                        # noqa: E731
for _ in range(4):
    fn()
torch.cuda.synchronize()
