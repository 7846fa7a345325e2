"""eavsr_amd -- MI355X-native (gfx950) implementation of EAVSR's inter-frame alignment and
feature-propagation hot path behind the reference's own module API.

    from eavsr_amd.eavsrp_model import EAVSRP          # drop-in for models/eavsrp_model.py::EAVSRP
    from eavsr_amd.networks import MultiAdSTN, flow_warp, modulated_deform_conv2d, RCAGroup

Kernels live in eavsr_amd/csrc/*.hip, are compiled into eavsr_amd/lib/libeavsr_hip.so
(`python -m eavsr_amd.build`) and are reached through the C ABI of include/eavsr_hip.h.
"""
__version__ = "0.1.0"
