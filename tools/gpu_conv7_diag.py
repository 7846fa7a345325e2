#!/usr/bin/env python3
"""eavsr_conv7x7_f32x6 variants (tools/build_c7_diag.sh): time per launch at the bench's two largest pyramid levels, equality with
the shipped kernel for libc7_v_*, and the phase stamps of libc7_stamps (shader cycles of wave 0, summed over workgroups)."""
import ctypes as C
import glob
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from eavsr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("N", 24))
p = lambda t: C.c_void_p(t.data_ptr())
paths = sorted(glob.glob(os.path.join(ROOT, "eavsr_amd", "lib", "libc7_*.so")))
paths.sort(key=lambda q: (not q.endswith("libc7_full.so"), q))
ref = {}
for path in paths:
    lib = C.CDLL(path)
    lib.eavsr_conv_weight_x6_bytes.restype = C.c_size_t
    name = os.path.basename(path)
    line = f"{name:24s}"
    for (h, w) in ((192, 320), (96, 160), (24, 40)):
        for cin, cout in ((32, 64), (64, 32)):
            torch.manual_seed(1)
            x = torch.randn(N, cin, h, w, device=dev)
            wt = torch.randn(cout, cin, 7, 7, device=dev) * 0.02
            b = torch.randn(cout, device=dev) * 0.1
            wp = torch.empty(lib.eavsr_conv_weight_x6_bytes(7, cout, cin), device=dev, dtype=torch.uint8)
            assert lib.eavsr_pack_conv_weight_x6(p(wt), p(wp), 7, cout, cin, None) == 0
            out = torch.zeros(N, cout, h, w, device=dev)
            call = lambda: lib.eavsr_conv_f32x6(p(x), p(wp), p(b), p(out), N, cin, cout, h, w, 7, 1, C.c_float(0.0), None)
            for _ in range(2):
                assert call() == 0
            torch.cuda.synchronize()
            key = (h, w, cin, cout)
            if name == "libc7_full.so":
                ref[key] = out.clone()
            elif name.startswith("libc7_v_"):
                d = (out - ref[key]).abs().max().item()
                if d != 0.0:
                    line += f" [DIFF {d:.2e}]"
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 * 1e3
            line += f"  {h}x{w} {cin}->{cout} {us:7.1f}"
            if "stamps" in name and h == 192:
                buf = (C.c_ulonglong * 8)()
                lib.eavsr_debug_c7_stamps(buf, 1)
                call()
                torch.cuda.synchronize()
                lib.eavsr_debug_c7_stamps(buf, 1)
                nwg = N * ((h + 15) // 16) * ((w + 31) // 32)
                names = ["prologue", "k-steps", "own vmcnt", "barrier", "DMA issue", "epilogue issue", "stores ack", "-"]
                tot = sum(buf[i] for i in range(8))
                print(f"  {cin}->{cout}: " + "  ".join(f"{names[i]} {buf[i] / nwg:.0f}" for i in range(7)) + f"  total {tot / nwg:.0f} cycles per workgroup", flush=True)
    print(line, flush=True)
