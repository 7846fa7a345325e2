import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from eavsr_amd import ops
dev = torch.device("cuda:0")
ops.set_conv_mode("direct")
for (n, h, w) in ((2, 96, 96), (4, 45, 80), (4, 90, 160), (28, 45, 80)):
    for cin in (8, 64, 128):
        x = torch.randn(n, cin, h, w, device=dev)
        wt = torch.randn(64, cin, 3, 3, device=dev) * 0.05
        b = torch.randn(64, device=dev)
        for _ in range(3): ops.conv2d(x, wt, b, act="relu")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ops.conv2d(x, wt, b, act="relu")
        e1.record(); torch.cuda.synchronize()
        rows = ops.lib().eavsr_conv2d_tile_rows(n, h, w, 3)
        print(f"{n}x{h}x{w} cin {cin:4d} tile rows {rows:2d}: {e0.elapsed_time(e1)/50*1000:8.1f} us")
