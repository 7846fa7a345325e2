# 5x5 heads (64 -> 120 at 4 x 180 x 320): direct MFMA kernel vs Winograd F(2x2,5x5)
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from eavsr_amd import ops
dev = torch.device("cuda:0")
n, h, w = 4, 180, 320
x = torch.randn(n, 64, h, w, device=dev)
wt = torch.randn(120, 64, 5, 5, device=dev) * 0.02
b = torch.randn(120, device=dev)
ref = None
for mode in ("direct", "winograd4"):
    ops.set_conv_mode(mode)
    for _ in range(3): y = ops.conv2d(x, wt, b)
    torch.cuda.synchronize()
    if ref is None: ref = y
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv2d(x, wt, b)
    e1.record(); torch.cuda.synchronize()
    print(f"{mode:9s} {e0.elapsed_time(e1)/10*1000:8.1f} us   rel diff vs direct {(y-ref).abs().max().item()/ref.abs().max().item():.2e}")
