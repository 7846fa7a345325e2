"""ATen operators of one eager training step (BASELINE configs[3] shape) that are NOT library launches: name, output shape, count,
and the innermost eavsr_amd frame that issued it -- where autograd's glue (gradient sums, copies, fills) comes from.
  python tools/gpu_train_aten_census.py"""
import collections
import os
import sys
import traceback
from argparse import Namespace

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from eavsr_amd.eavsrp_model import EAVSRPModel  # noqa: E402
from eavsr_amd.utils.synthetic import fill_state_dict, shapes_of, synthetic_clip  # noqa: E402

dev = torch.device("cuda:0")
opt = Namespace(predict=False, n_frame=7, n_flow=5, scale=4, isTrain=True, gpu_ids=[0], lr=1e-4, beta1=0.9, beta2=0.999,
                weight_decay=0.0, npost=350)
model = EAVSRPModel(opt)
sd0 = model.netEAVSRP.state_dict()
model.netEAVSRP.load_state_dict(fill_state_dict(shapes_of(sd0), "trained_like", fixed=sd0), strict=True)
model.set_input({"lr_seq": synthetic_clip(2, 7, 96, 96, seed=0), "hr_seq": synthetic_clip(2, 7, 384, 384, seed=100), "fname": "s"}, epoch=0)
model.optimize_parameters()
torch.cuda.synchronize()

counts = collections.Counter()
SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.alias", "aten.t.", "aten.slice", "aten.select", "aten.expand", "aten.as_strided",
        "aten.reshape", "aten.permute", "aten.transpose", "aten.unsqueeze", "aten.squeeze", "aten.unbind", "aten.split", "aten.empty", "aten.sym_",
        "aten._local_scalar", "aten.is_", "aten.stride", "aten.size", "aten.new_empty", "aten.lift_fresh", "aten.chunk", "aten.narrow", "prim.")


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not name.startswith(SKIP):
            shp = tuple(out.shape) if isinstance(out, torch.Tensor) else None
            where = "?"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if "eavsr_amd" in fr.filename and "tools" not in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            counts[(name, shp, where)] += 1
        return out


with Census():
    model.optimize_parameters()
torch.cuda.synchronize()
tot = collections.Counter()
for (name, shp, where), c in counts.items():
    tot[name] += c
print("by operator:", tot.most_common(25))
print()
for (name, shp, where), c in counts.most_common(70):
    print(f"{c:6d}  {name:32s} {str(shp):28s} {where}")
