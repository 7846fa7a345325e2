"""The oracle (oracle/eavsr_oracle.py) against the golden vectors produced by the real
reference modules (tests/golden/gen_golden.py).  CPU only."""
import pytest
import torch

from oracle import eavsr_oracle as O
from tests import helpers as H
from tests.golden import cases

TOL = 2e-6  # same torch ops in a different composition: float re-association only


def test_g1_flow_warp_both_layouts():
    g = H.golden("g1_flow_warp")
    for name, (x, flow, pad) in cases.g1_flow_warp_cases().items():
        a = O.flow_warp(x, flow, pad)
        b = O.flow_warp_nhwc(x, flow.permute(0, 2, 3, 1).contiguous(), pad)
        assert H.maxabs(a, g[name + "__nchw"]) <= TOL, name
        assert H.maxabs(b, g[name + "__nhwc"]) <= TOL, name
        # the direct (no grid_sample) restatement agrees up to the normalise round trip
        d = O.flow_warp_direct(x, flow, pad)
        assert H.maxabs(d, g[name + "__nchw"]) <= 5e-5 * max(1.0, x.abs().max().item()), name


@pytest.mark.parametrize("preset", ["default", "trained_like"])
def test_g2_adapt3x3_and_trans(preset):
    g = H.golden(f"g2_adapt3x3_{preset}")
    sd = H.filled({**H.adapt3x3_shapes("g2.flow."), **H.trans_shapes("g2.trans.")}, preset)
    x, h = cases.g2_inputs()
    off = O.adapt_block2_3x3(sd, "g2.flow.", x, h)
    assert off.shape == g["offset18"].shape
    assert H.maxabs(off, g["offset18"]) <= 1e-5
    assert H.maxabs(O.trans_offset(sd, "g2.trans.", off), g["flow2"]) <= 1e-5


@pytest.mark.parametrize("preset", ["default", "trained_like"])
def test_g3_adapt_block_offset_layout(preset):
    g = H.golden(f"g3_adaptoffset_{preset}")
    sd = H.filled(H.adaptoffset_shapes("g3.adastn."), preset)
    x, h = cases.g3_inputs()
    off, mask = O.adapt_block_offset(sd, "g3.adastn.", x, h, 8)
    assert off.shape == g["offset"].shape == (2, 144, 10, 12)
    assert mask.shape == g["mask"].shape == (2, 72, 10, 12)
    assert H.maxabs(off, g["offset"]) <= 1e-5
    assert H.maxabs(mask, g["mask"]) <= 1e-6


@pytest.mark.parametrize("preset", ["default", "trained_like"])
def test_g5_multiadstn(preset):
    g = H.golden(f"g5_multiadstn_{preset}")
    sd = H.filled(H.multiadstn_shapes("g5.align."), preset)
    nbr, ref, fp, flow = cases.g5_inputs()
    out = O.multi_adstn(sd, "g5.align.", nbr, ref, fp, flow, 8)
    assert H.maxabs(out, g["out"]) <= 2e-5


@pytest.mark.parametrize("preset", ["default", "trained_like"])
def test_g6_backbone_blocks(preset):
    g = H.golden(f"g6_backbone_{preset}")
    x64, x128 = cases.g6_inputs()
    sd = H.filled({**H.rcab_shapes("g6.rcab."), **H.rcagroup_shapes("g6.group.", 2),
                   **H.rbic_shapes("g6.rbic.", 128, 2)}, preset)
    assert H.maxabs(O.rcab(sd, "g6.rcab.", x64), g["rcab"]) <= 1e-5
    assert H.maxabs(O.rca_group(sd, "g6.group.", x64, 2), g["group"]) <= 1e-5
    assert H.maxabs(O.resblocks_with_input_conv(sd, "g6.rbic.", x128), g["rbic"]) <= 1e-5


@pytest.mark.parametrize("preset", ["default", "trained_like"])
def test_g7_propagate(preset):
    g = H.golden(f"g7_propagate_{preset}")
    sd = H.filled(H.model_shapes("x4"), preset)
    feats, flows, prev = cases.g7_inputs()
    f1 = {k: list(v) for k, v in feats.items()}
    f1["backward_1"] = []
    r1 = torch.stack(O.propagate(sd, f1, flows, "backward_1")["backward_1"], 1)
    assert H.maxabs(r1, g["backward_1"]) <= 5e-5
    f2 = {k: list(v) for k, v in feats.items()}
    for k in ("backward_1", "forward_1", "backward_2"):
        f2[k] = list(prev[k])
    f2["forward_2"] = []
    r2 = torch.stack(O.propagate(sd, f2, flows, "forward_2")["forward_2"], 1)
    assert H.maxabs(r2, g["forward_2"]) <= 5e-5


@pytest.mark.slow
@pytest.mark.parametrize("tag,scale", [("x4", 4), ("x2", 2)])
@pytest.mark.parametrize("preset", ["default", "trained_like"])
def test_g8_end_to_end(tag, scale, preset):
    g = H.golden(f"g8_e2e_{tag}_{preset}")
    sd = H.filled(H.model_shapes(tag), preset)
    clip = cases.g8_clip()
    ff, fb = O.compute_flow(sd, clip)
    assert H.maxabs(ff, g["flows_forward"]) <= 1e-5
    assert H.maxabs(fb, g["flows_backward"]) <= 1e-5
    y = O.eavsrp_forward(sd, clip, scale)
    assert tuple(y.shape) == tuple(g["shape"].tolist())
    assert H.maxabs(cases.subsample(y), g["sub"]) <= 2e-5
    assert H.maxabs(y.mean(dim=(-1, -2)), g["mean"]) <= 1e-5


def test_state_dict_contract_counts():
    k = H.golden_keys("x4")
    assert k["n_tensors"] == 1296 and k["n_params"] == 13718099 + 12 + 16 * 18  # params + mean/std + regular_matrix buffers
