# end-to-end difference of the conv modes on the bench clip (configs[1] shape, 1 clip) against the direct mode
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from argparse import Namespace
from eavsr_amd import ops
from eavsr_amd.eavsrp_model import EAVSRP
from eavsr_amd.utils.synthetic import fill_state_dict, shapes_of, synthetic_clip
dev = torch.device("cuda:0")
net = EAVSRP(Namespace(predict=False, n_frame=7, n_flow=5, scale=4), None)
sd0 = net.state_dict()
net.load_state_dict(fill_state_dict(shapes_of(sd0), "trained_like", fixed=sd0), strict=True)
net = net.to(dev).eval()
clip = synthetic_clip(2, 7, 180, 320, seed=0).to(dev)
outs = {}
with torch.no_grad():
    for mode in ("direct", "winograd", "winograd4"):
        ops.set_conv_mode(mode)
        outs[mode] = net(clip).double()
for mode in ("winograd", "winograd4"):
    d = (outs[mode] - outs["direct"]).abs()
    print(f"{mode:9s} vs direct: max abs {d.max().item():.3e}  mean abs {d.mean().item():.3e}  (output range {outs['direct'].min().item():.2f}..{outs['direct'].max().item():.2f})")
