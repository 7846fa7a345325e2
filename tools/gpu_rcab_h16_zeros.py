import os, sys
sys.path.insert(0, "/root/repo")
import torch
from eavsr_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def med(fn, reps=9, inner=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return sorted(ts)[len(ts) // 2]
n, h, w = 4, 256, 256
for name, scale in (("random", 1.0), ("zeros", 0.0)):
    x = ops.to_nhwc_h16(torch.randn(n, 64, h, w, device=dev) * scale, "bf16")
    w1, w2 = torch.randn(64, 64, 3, 3, device=dev) / 24 * scale, torch.randn(64, 64, 3, 3, device=dev) / 24 * scale
    b1, b2 = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    one = lambda: ops.rcab_convs_h16(x, w1, b1, w2, b2, chan_partial=True)
    def two():
        t = ops.conv3x3_c64_h16(x, w1, b1, relu=True)
        return ops.conv3x3_c64_h16(t, w2, b2, chan_partial=True)
    for _ in range(2):
        print(name, "one", round(med(one), 1), "two", round(med(two), 1), flush=True)
