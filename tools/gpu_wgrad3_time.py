"""3x3 64 x 64 weight gradient over 7 segments of 2 x 64 x 96 x 96 (one use per frame of a training clip): the bf16x6 kernel against
the fp32-MFMA kernel (EAVSR_WGRAD3=fp32 in a child process: the switch is read once per process).
  python tools/gpu_wgrad3_time.py"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from eavsr_amd import ops  # noqa: E402

if "--child" not in sys.argv and os.environ.get("EAVSR_WGRAD3") != "fp32":
    env = dict(os.environ, EAVSR_WGRAD3="fp32")
    subprocess.run([sys.executable, __file__, "--child"], env=env)
dev = torch.device("cuda:0")
nseg, n, h, w = 7, 2, 96, 96
REPS = int(os.environ.get("REPS", "50"))
torch.manual_seed(0)
dys = [torch.randn(n, 64, h, w, device=dev) for _ in range(nseg)]
xs = [[torch.randn(n, 64, h, w, device=dev)] for _ in range(nseg)]
out = torch.empty(64, 64, 3, 3, device=dev)
want = torch.zeros(64, 64, 3, 3, dtype=torch.float64)
for d, x in zip(dys, xs):
    want += torch.nn.grad.conv2d_weight(x[0].double().cpu(), (64, 64, 3, 3), d.double().cpu(), padding=1)
ops.conv_wgrad_multi(dys, xs, 3, out=out)
torch.cuda.synchronize()
err = ((out.cpu().double() - want).abs().max() / want.abs().max()).item()
mode = ops.lib().eavsr_wgrad3_mode()
for rot in range(3):
    for _ in range(3):
        ops.conv_wgrad_multi(dys, xs, 3, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        ops.conv_wgrad_multi(dys, xs, 3, out=out)
    e1.record()
    torch.cuda.synchronize()
    print(f"wgrad3 mode {mode} ({'bf16x6' if mode else 'fp32 MFMA'}): {e0.elapsed_time(e1) * 1e3 / REPS:7.1f} us per launch + reduce, max rel err vs fp64 {err:.2e}", flush=True)
