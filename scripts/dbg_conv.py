import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_rl_amd.ops import hip as H
torch.manual_seed(0)
frames = 1
def run(x, w, b):
    o1 = torch.zeros(frames, 20, 20, 32, dtype=torch.bfloat16, device="cuda"); o2 = torch.zeros_like(o1)
    H.conv_fwd(1, x, w, b, o1, frames, direct=False); H.conv_fwd(1, x, w, b, o2, frames, direct=True)
    return o1.float(), o2.float()
b = torch.zeros(32, device="cuda")
# 1) ones input, weight = 1 at a single k -> output 1 everywhere if that k is read
x = torch.ones(frames, 84, 84, 4, dtype=torch.uint8, device="cuda")
w = torch.ones(32, 8, 8, 4, device="cuda").to(torch.bfloat16)
o1, o2 = run(x, w, b)
print("ones: tiled", o1[0, 0, 0, :4].tolist(), "direct", o2[0, 0, 0, :4].tolist(), o2[0, 5, 7, 28:].tolist())
# 2) one-hot weight: which input element does each k pick up? x = unique values along (y, x, c) mod 251
x = (torch.arange(84 * 84 * 4, device="cuda") % 251).to(torch.uint8).view(1, 84, 84, 4)
for k in (0, 1, 4, 8, 31, 32, 33, 255):
    w = torch.zeros(32, 256, device="cuda"); w[:, k] = 1
    o1, o2 = run(x, w.view(32, 8, 8, 4).to(torch.bfloat16), b)
    print("k", k, "tiled", o1[0, 0, 0, 0].item(), o1[0, 3, 2, 5].item(), "direct", o2[0, 0, 0, 0].item(), o2[0, 3, 2, 5].item())
