#!/usr/bin/env python3
"""eavsr_conv_wgrad_multi_f32 (3x3) variants (tools/build_wgrad_diag.sh): time per launch (+ reduce) over 7 segments of
2 x 64 x 96 x 96 and equality with the shipped kernel for libwg_v_*."""
import ctypes as C
import glob
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

dev = torch.device("cuda:0")
nseg, n, h, w = 7, 2, 96, 96
REPS = int(os.environ.get("REPS", "50"))
p = lambda t: C.c_void_p(t.data_ptr())
paths = sorted(glob.glob(os.path.join(ROOT, "eavsr_amd", "lib", "libwg_*.so")))
paths.sort(key=lambda q: (not q.endswith("libwg_full.so"), q))
torch.manual_seed(0)
dys = [torch.randn(n, 64, h, w, device=dev) for _ in range(nseg)]
xs = [torch.randn(n, 64, h, w, device=dev) for _ in range(nseg)]
PA = C.c_void_p * nseg
dl, xl = PA(*[t.data_ptr() for t in dys]), PA(*[t.data_ptr() for t in xs])
ents = []
for path in paths:
    lib = C.CDLL(path)
    blocks = lib.eavsr_conv_wgrad_blocks(n * nseg, h, w, 3)
    ws = torch.empty(blocks * 64 * 64 * 9, device=dev)
    out = torch.zeros(64, 64, 3, 3, device=dev)
    ents.append((os.path.basename(path)[6:-3], lib, ws, out))


def call(e):
    return e[1].eavsr_conv_wgrad_multi_f32(dl, xl, nseg, p(e[3]), p(e[2]), n, h, w, 64, 0, 64, 0, 64, 0, 3, 0, None)


ref = None
for e in ents:
    assert call(e) == 0
    torch.cuda.synchronize()
    if e[0] == "full":
        ref = e[3].clone()
    else:
        print(f"{e[0]}: max |diff| vs full {(e[3] - ref).abs().max().item():.2e}")
def timed(e, ns):
    f = lambda: e[1].eavsr_conv_wgrad_multi_f32(dl, xl, ns, p(e[3]), p(e[2]), n, h, w, 64, 0, 64, 0, 64, 0, 3, 0, None)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


for rot in range(3):
    line = f"rot {rot}:"
    for e in ents:
        line += f"  {e[0]} {timed(e, nseg):6.1f}"
    print(line, flush=True)
# fixed cost vs cost per tile: 1, 2, 4, 7 segments (144 four-row tiles each on 128 x 4 workgroups)
for e in ents:
    print(f"{e[0]:12s} us by segments: " + "  ".join(f"{ns}: {timed(e, ns):6.1f}" for ns in (1, 2, 4, 7)), flush=True)
