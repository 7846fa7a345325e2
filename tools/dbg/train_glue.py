#!/usr/bin/env python3
"""Which Python lines launch the ATen kernels (fill / add / copy / cat) of one eager training step (configs[3])?"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from argparse import Namespace
from eavsr_amd.eavsrp_model import EAVSRPModel
from eavsr_amd.utils.synthetic import fill_state_dict, shapes_of, synthetic_clip
dev = torch.device("cuda:0")
opt = Namespace(predict=False, n_frame=7, n_flow=5, scale=4, isTrain=True, gpu_ids=[0], lr=1e-4, beta1=0.9, beta2=0.999, weight_decay=0.0, npost=350)
model = EAVSRPModel(opt)
sd0 = model.netEAVSRP.state_dict()
model.netEAVSRP.load_state_dict(fill_state_dict(shapes_of(sd0), "trained_like", fixed=sd0), strict=True)
model.set_input({"lr_seq": synthetic_clip(2, 7, 96, 96, seed=0), "hr_seq": synthetic_clip(2, 7, 384, 384, seed=100), "fname": "s"}, epoch=0)
for _ in range(2):
    model.optimize_parameters()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    model.optimize_parameters()
torch.cuda.synchronize()
want = ("aten::zeros_like", "aten::zeros", "aten::zero_", "aten::fill_", "aten::add", "aten::add_", "aten::copy_", "aten::cat", "aten::clone", "aten::contiguous", "aten::mul", "aten::sum", "aten::empty_like")
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in want:
        st = [s for s in (ev.stack or []) if "eavsr_amd" in s or "autograd" in s.lower()]
        top = st[0] if st else (ev.stack[0] if ev.stack else "<no python frame: autograd engine>")
        shp = str(ev.input_shapes)[:60] if ev.input_shapes else ""
        cnt[(ev.name, top[-90:], shp)] += 1
import os
only = os.environ.get('ONLY')
for (name, where, shp), c in cnt.most_common(int(os.environ.get('TOP', '60'))):
    if only and name not in only.split(','):
        continue
    print(f"{c:5d}  {name:18s} {where}  {shp}")
