"""ctypes loader of oracle/_build/liboracle_ref.so (the plain-C restatement, oracle/dcnv2_ref.c).
TEST INFRASTRUCTURE ONLY -- see oracle/eavsr_oracle.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "liboracle_ref.so")


def build():
    subprocess.run(["make", "-C", HERE, "-s"], check=True)
    return SO


def _lib():
    if not os.path.exists(SO):
        build()
    return C.CDLL(SO)


def _f(t):
    a = np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))
    return a, a.ctypes.data_as(C.c_void_p)


def dcnv2(x, offset, mask, weight, bias, stride=1, padding=1, dilation=1, deform_groups=1):
    n, c, h, w = x.shape
    co, _, k, _ = weight.shape
    ho, wo = offset.shape[-2:]
    out = np.zeros((n, co, ho, wo), np.float32)
    keep = [_f(t) for t in (x, offset, mask, weight)]
    b = _f(bias) if bias is not None else (None, None)
    rc = _lib().eavsr_ref_dcnv2(keep[0][1], keep[1][1], keep[2][1], keep[3][1], b[1], out.ctypes.data_as(C.c_void_p),
                                n, c, h, w, co, k, stride, padding, dilation, deform_groups)
    assert rc == 0
    return torch.from_numpy(out)


def flow_warp(x, flow, padding_mode="zeros"):
    n, c, h, w = x.shape
    out = np.zeros((n, c, h, w), np.float32)
    a, b = _f(x), _f(flow)
    rc = _lib().eavsr_ref_flow_warp(a[1], b[1], out.ctypes.data_as(C.c_void_p), n, c, h, w,
                                    1 if padding_mode == "border" else 0)
    assert rc == 0
    return torch.from_numpy(out)
