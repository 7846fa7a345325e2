#!/usr/bin/env python3
"""How long does the host need to ENQUEUE one forward (Python + ctypes launches), against the device time?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from argparse import Namespace
from eavsr_amd.eavsrp_model import EAVSRP
from eavsr_amd.utils.synthetic import fill_state_dict, shapes_of, synthetic_clip
dev = torch.device("cuda:0")
net = EAVSRP(Namespace(predict=False, n_frame=7, n_flow=5, scale=4), None)
sd0 = net.state_dict()
net.load_state_dict(fill_state_dict(shapes_of(sd0), "trained_like", fixed=sd0))
net = net.to(dev).eval()
clips = synthetic_clip(4, 7, 180, 320, 0).to(dev)
with torch.no_grad():
    for _ in range(2):
        net(clips)
    torch.cuda.synchronize()
    for _ in range(3):
        t0 = time.perf_counter()
        net(clips)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"enqueue {1e3 * (t1 - t0):.1f} ms, until done {1e3 * (t2 - t0):.1f} ms")
