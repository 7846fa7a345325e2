#!/usr/bin/env python3
"""Known-answer fixtures that do NOT come from the reference.

DCNv2's arithmetic lives in mmcv 1.x, which is absent from /root/reference and from this image: there is no reference
binary to generate from (parity UNPINNED against mmcv, see oracle/eavsr_oracle.py).  What can be made regression-proof is
the restatement itself:

  G4  g4_dcnv2.npz       forward outputs of the plain-C restatement oracle/dcnv2_ref.c (double accumulation, rounded to
                         fp32 once) on tests/golden/cases.py::g4_inputs -- borders, |offset| > 1, dg = 8, sigma in
                         {0.5, 2, 8}, exact validity-boundary positions.  The two PyTorch restatements and every HIP
                         DCNv2 kernel are tested against it.
  G9  g9_gradients.npz   gradients (d input, d offset, d mask, d weight, d bias) of the G4 cases and (d x, d flow) of the
                         G1 flow_warp cases, by fp64 autograd through oracle.dcnv2 / oracle.flow_warp with the seeded
                         cotangent cases.g9_cotangent, stored as fp32.  The HIP backward kernels are tested against it.

Run from the repo root:  python tests/golden/gen_known_answers.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import c_ref  # noqa: E402
from oracle import eavsr_oracle as O  # noqa: E402
from tests.golden import cases  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    c_ref.build()
    g4, g9 = {}, {}
    for name in cases.G4_CASES:
        x, off, mask, wt, b, dg = cases.g4_inputs(name)
        out_c = c_ref.dcnv2(x, off, mask, wt, b, 1, 1, 1, dg)
        g4[name] = out_c.numpy()
        # cross-check at generation time: the fp64 PyTorch restatement agrees with the C one
        out64 = O.dcnv2(x.double(), off.double(), mask.double(), wt.double(), b.double(), 1, 1, 1, 1, dg)
        err = (out64 - out_c.double()).abs().max().item()
        assert err <= 5e-6 * max(1.0, out64.abs().max().item()), (name, err)
        leaves = [t.double().requires_grad_(True) for t in (x, off, mask, wt, b)]
        out = O.dcnv2(*leaves, 1, 1, 1, 1, dg)
        G = cases.g9_cotangent(name, out.shape).double()
        grads = torch.autograd.grad((out * G).sum(), leaves)
        for key, gr in zip(("dx", "doffset", "dmask", "dweight", "dbias"), grads):
            g9[f"dcn_{name}__{key}"] = gr.float().numpy()
        print(f"G4/G9 {name}: |out| max {out_c.abs().max():.3f}, C vs fp64 torch {err:.2e}")
    for name, (x, flow, pad) in cases.g1_flow_warp_cases().items():
        leaves = [x.double().requires_grad_(True), flow.double().requires_grad_(True)]
        out = O.flow_warp(leaves[0], leaves[1], pad)
        G = cases.g9_cotangent(name, out.shape).double()
        dx, dflow = torch.autograd.grad((out * G).sum(), leaves)
        g9[f"warp_{name}__dx"] = dx.float().numpy()
        g9[f"warp_{name}__dflow"] = dflow.float().numpy()
    np.savez_compressed(os.path.join(HERE, "g4_dcnv2.npz"), **g4)
    np.savez_compressed(os.path.join(HERE, "g9_gradients.npz"), **g9)
    for f in ("g4_dcnv2.npz", "g9_gradients.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
