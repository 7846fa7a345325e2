import sys, os
sys.path.insert(0, "/root/repo")
import torch
from eavsr_amd import ops
dev = torch.device("cuda:0")
wt = torch.randn(64, 64, 3, 3, device=dev) * 0.04
b = torch.randn(64, device=dev)
def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (n, h, w) in [(1, 8, 32), (1, 8, 64), (1, 64, 128), (1, 128, 256), (1, 128, 512), (2, 180, 320), (2, 128, 512), (4, 128, 512), (8, 128, 512), (16, 128, 512)]:
    x = torch.randn(n, 64, h, w, device=dev)
    xh = ops.to_nhwc_h16(x, "bf16")
    tiles = ((h + 7) // 8) * ((w + 31) // 32) * n
    t = timed(lambda: ops.conv3x3_c64_h16(xh, wt, b, relu=True))
    print(f"n={n} {h}x{w}: tiles {tiles} ({tiles / 256:.2f} per CU)  {t:.1f} us", flush=True)
