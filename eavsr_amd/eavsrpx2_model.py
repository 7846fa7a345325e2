"""Drop-in counterpart of models/eavsrpx2_model.py: the x2 (RealVSR) twin.  The network differs from
the x4 one only in the upsampling tail (one pixel-shuffle stage, x2 bilinear skip:
eavsrpx2_model.py:154-159,356-360), so it is the same class with scale=2."""
from .eavsrp_model import (EAVSRPx2 as EAVSRP, EAVSRPModel as _Model, ResidualBlocksWithInputConv,  # noqa: F401
                           SPyNet, SPyNetBasicModule, flow_warp)


class EAVSRPx2Model(_Model):
    def __init__(self, opt):
        if getattr(opt, "scale", 2) != 2:
            raise ValueError("EAVSRPx2Model is the x2 model")
        super().__init__(opt)
