# F(4x4,3x3) vs F(2x2,3x3) vs direct at the backbone's conv shape (4 x 64 x 180 x 320) and a cin sweep
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from eavsr_amd import ops
dev = torch.device("cuda:0")
n, h, w = 4, 180, 320
for cin in (8, 32, 64, 128, 320):
    x = torch.randn(n, cin, h, w, device=dev)
    wt = torch.randn(64, cin, 3, 3, device=dev) * 0.05
    b = torch.randn(64, device=dev)
    ref = None
    for mode in ("direct", "winograd", "winograd4"):
        ops.set_conv_mode(mode)
        for _ in range(3): y = ops.conv2d(x, wt, b, act="relu")
        torch.cuda.synchronize()
        if ref is None: ref = y
        err = (y - ref).abs().max().item() / ref.abs().max().item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.conv2d(x, wt, b, act="relu")
        e1.record(); torch.cuda.synchronize()
        print(f"cin {cin:4d} {mode:9s} {e0.elapsed_time(e1)/20*1000:8.1f} us   rel diff vs direct {err:.2e}")
