"""3x3 64 -> 64 at the bench's launch shape: conv_x6_kernel<3> (exact bf16x6, direct) against the fp32 Winograd F(4x4,3x3) kernel.
  python tools/gpu_conv3_x6_time.py [n h w]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n, h, w = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else (2, 180, 320)
REPS = int(os.environ.get("REPS", "100"))
torch.manual_seed(0)
x = torch.randn(n, 64, h, w, device=dev)
wt = torch.randn(64, 64, 3, 3, device=dev) / 24
b = torch.randn(64, device=dev) * 0.1
p = lambda t: C.c_void_p(t.data_ptr())
L = ops.lib()
wp = torch.empty(L.eavsr_conv_weight_x6_bytes(3, 64, 64), device=dev, dtype=torch.uint8)
assert L.eavsr_pack_conv_weight_x6(p(wt), p(wp), 3, 64, 64, None) == 0
out = torch.empty(n, 64, h, w, device=dev)
st = torch.cuda.current_stream().cuda_stream


def x6():
    assert L.eavsr_conv_f32x6(p(x), p(wp), p(b), p(out), n, 64, 64, h, w, 3, 1, C.c_float(0.0), -1, C.c_void_p(st)) == 0
    return out


def wino():
    return ops.conv2d(x, wt, b, act="relu")


ref = torch.nn.functional.conv2d(x.double(), wt.double(), b.double(), padding=1).relu()
for name, f in (("x6", x6), ("wino4", wino)):
    y = f()
    torch.cuda.synchronize()
    print(name, "max rel err vs fp64", ((y.double() - ref).abs().max() / ref.abs().max()).item())
for rot in range(3):
    line = f"rot {rot}:"
    for name, f in (("x6", x6), ("wino4", wino)):
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            f()
        e1.record()
        torch.cuda.synchronize()
        line += f"  {name} {e0.elapsed_time(e1) * 1e3 / REPS:7.2f} us"
    print(line, flush=True)
