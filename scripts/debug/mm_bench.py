import torch, sys
sys.path.insert(0, "/root/repo/scripts")
DEV="cuda"
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
feat = torch.randn(7680, 3136, device=DEV).bfloat16()
wih = torch.randn(2048, 3136, device=DEV).bfloat16()
dG = torch.randn(5120, 2048, device=DEV).bfloat16()
out1 = torch.empty(7680, 2048, device=DEV, dtype=torch.bfloat16)
out2 = torch.empty(2048, 3136, device=DEV, dtype=torch.bfloat16)
out3 = torch.empty(5120, 3136, device=DEV, dtype=torch.bfloat16)
f5 = feat[:5120]
for name, fn, fl in [("xg NT 7680x2048x3136", lambda: torch.mm(feat, wih.t(), out=out1), 2*7680*2048*3136),
                     ("dWih TN 2048x3136x5120", lambda: torch.mm(dG.t(), f5, out=out2), 2*5120*2048*3136),
                     ("dX NN 5120x3136x2048", lambda: torch.mm(dG, wih, out=out3), 2*5120*2048*3136)]:
    us = timeit(fn)
    print("%-28s %8.1f us %8.1f TFLOP/s" % (name, us, fl / us / 1e6))
try:
    o = torch.mm(feat, wih.t(), out_dtype=torch.float32)
    print("out_dtype ok", o.dtype, o.shape)
    us = timeit(lambda: torch.mm(feat, wih.t(), out_dtype=torch.float32))
    print("xg NT f32 out (alloc)  %8.1f us" % us)
    o2 = torch.empty(7680, 2048, device=DEV)
    try:
        us = timeit(lambda: torch.mm(feat, wih.t(), out_dtype=torch.float32, out=o2))
        print("xg NT f32 out= %8.1f us" % us)
    except Exception as e:
        print("out= with out_dtype failed:", str(e)[:200])
    us = timeit(lambda: torch.mm(dG.t(), f5, out_dtype=torch.float32))
    print("dWih TN f32 out  %8.1f us" % us)
    ref = feat.float() @ wih.float().t()
    print("max err f32-out vs fp32 ref", float((o - ref).abs().max()), "ref absmax", float(ref.abs().max()))
except Exception as e:
    print("out_dtype failed:", str(e)[:300])
hs = torch.randn(5120, 512, device=DEV).bfloat16()
dh2 = torch.randn(5120, 1024, device=DEV).bfloat16()
wfc = torch.randn(1024, 512, device=DEV).bfloat16()
g_hh = torch.empty(2048, 512, device=DEV)
g_fc = torch.empty(1024, 512, device=DEV)
dhs = torch.empty(5120, 512, device=DEV, dtype=torch.bfloat16)
bias = torch.randn(2048, device=DEV)
for name, fn, fl in [("dWhh TN 2048x512x5120 f32", lambda: torch.mm(dG.t(), hs, out_dtype=torch.float32, out=g_hh), 2*5120*2048*512),
                     ("fc1 dW TN 1024x512x5120 f32", lambda: torch.mm(dh2.t(), hs, out_dtype=torch.float32, out=g_fc), 2*5120*1024*512),
                     ("fc1 dX NN 5120x512x1024", lambda: torch.mm(dh2, wfc, out=dhs), 2*5120*1024*512)]:
    us = timeit(fn)
    print("%-28s %8.1f us %8.1f TFLOP/s" % (name, us, fl / us / 1e6))
try:
    us = timeit(lambda: torch.addmm(bias, feat, wih.t(), out_dtype=torch.float32, out=o2))
    print("xg addmm f32 out= %8.1f us" % us)
    print("addmm err", float((o2 - (ref + bias)).abs().max()))
except Exception as e:
    print("addmm out_dtype failed:", str(e)[:200])
def mm_bias():
    torch.mm(feat, wih.t(), out_dtype=torch.float32, out=o2)
    o2.add_(bias)
print("xg mm + add_ %8.1f us" % timeit(mm_bias))
