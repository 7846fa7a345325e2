"""Host-side logic that needs no GPU: the C-ABI library loads and exports every symbol the header
declares, the drop-in modules construct with the reference's state_dict keys, and the product path
fails loudly instead of falling back when it is given CPU tensors."""
import os
import re
from argparse import Namespace

import pytest
import torch

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    from eavsr_amd import _native, build
    build.build_native()
    lib = _native.load()
    header = open(os.path.join(ROOT, "include", "eavsr_hip.h")).read()
    stable_part, lab_part = header.split(" * EXPERIMENTAL -- exported by the LAB build only")
    find = lambda text: set(re.findall(r"^(?:int|int32_t|int64_t|size_t|const char\*)\s+(eavsr_[a-z0-9_]+)\s*\(", text, flags=re.M))
    declared, declared_lab = find(stable_part), find(lab_part)
    # the header's two sections are exactly the binding's two tables ...
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    assert declared_lab == set(_native.LAB_SIGNATURES), declared_lab ^ set(_native.LAB_SIGNATURES)
    # ... and the DEFAULT build (what build() makes and the GPU box runs) exports exactly the stable section: product, not notebook
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (eavsr_[a-z0-9_]+)$", nm, flags=re.M))
    assert lib.eavsr_lab_build() == 0 and not _native.lab_build()
    assert exported == declared, exported ^ declared
    assert not (exported & declared_lab)
    for name in declared:
        assert hasattr(lib, name), name
    from eavsr_amd import ops
    with pytest.raises(ops.LabBuildRequired):
        ops.set_conv_mode("bf16x9")
    with pytest.raises(ops.LabBuildRequired):
        ops.set_dcn_il_impl("ws")
    assert lib.eavsr_abi_version() == _native.ABI_VERSION
    assert b"gfx950" in lib.eavsr_version()
    # pure host helpers (no device work)
    assert lib.eavsr_conv2d_ck(3) == 8 and lib.eavsr_conv2d_ck(7) == 2 and lib.eavsr_conv2d_ck(5) == 4 and lib.eavsr_conv2d_ck(1) == 16
    assert lib.eavsr_conv2d_tile_rows(4, 180, 320, 3) == 32 and lib.eavsr_conv2d_tiles(4, 180, 320, 3) == 6 * 10
    # a training crop launches too few 32-row tiles for 256 CUs: 8-row tiles
    assert lib.eavsr_conv2d_tile_rows(2, 96, 96, 3) == 8 and lib.eavsr_conv2d_tiles(2, 96, 96, 3) == 12 * 3
    assert lib.eavsr_conv2d_tile_rows(2, 96, 96, 7) == 32
    assert lib.eavsr_packed_weight_elems(64, 64, 3) == 64 * 64 * 9
    assert lib.eavsr_packed_weight_elems(120, 64, 5) == 2 * 64 * 25 * 64
    assert lib.eavsr_packed_weight_elems(2, 18, 3) == 24 * 9 * 32


def test_host_side_argument_errors_are_reported_without_a_gpu():
    from eavsr_amd import _native
    lib = _native.load()
    assert lib.eavsr_flow_warp_f32(None, None, None, None, 1, 1, 1, 1, 0, 0, None) == -1
    assert b"NULL" in lib.eavsr_last_error()
    assert lib.eavsr_dcnv2_f32(1, 1, 1, 1, None, 1, 1, 12, 8, 8, 8, 3, None) == -2  # 4 channels / group
    assert b"multiple of 8" in lib.eavsr_last_error()
    assert lib.eavsr_pyramid_f32(16, 16, 16, 1, 10, 12, None) == -2


@pytest.mark.parametrize("tag", ["x4", "x2"])
def test_drop_in_state_dict_contract(tag):
    from eavsr_amd.eavsrp_model import EAVSRP
    from eavsr_amd.utils.synthetic import keys_digest, shapes_of
    net = EAVSRP(Namespace(predict=False, n_frame=7, n_flow=5, scale=4 if tag == "x4" else 2), None)
    shapes = shapes_of(net.state_dict())
    assert shapes == H.model_shapes(tag)
    assert keys_digest(shapes) == H.golden_keys(tag)["digest"]
    assert not any(p.requires_grad for p in net.spynet.parameters())  # eavsrp_model.py:132-133
    # round trip through the reference's checkpoint format {'state_dict': ...}
    sd = H.filled(H.model_shapes(tag), "default")
    net.load_state_dict(sd, strict=True)
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_no_cpu_fallback():
    from eavsr_amd import networks as N, ops
    from eavsr_amd.eavsrp_model import EAVSRP
    x = torch.zeros(1, 2, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.flow_warp(x, torch.zeros(1, 2, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU path"):
        N.RCABlock(64, 64)(torch.zeros(1, 64, 8, 8))
    net = EAVSRP(Namespace(predict=False, n_frame=7, n_flow=5, scale=4), None).eval()
    with pytest.raises(RuntimeError, match="GPU only"):
        net(torch.zeros(1, 3, 3, 64, 64))


def test_product_never_imports_the_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg = os.path.join(ROOT, "eavsr_amd")
    pat = re.compile(r"^\s*(from|import)\s+[.\w]*oracle|[\"']oracle[\"'/]|liboracle|_oracle", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not pat.search(src), os.path.join(dirpath, f)


def test_synthetic_fill_is_order_independent_and_deterministic():
    from eavsr_amd.utils.synthetic import fill_state_dict, synthetic_clip
    shapes = {"b.weight": (4, 3, 3, 3), "a.weight": (2, 2, 1, 1), "a.bias": (2,)}
    x = fill_state_dict(shapes, "default")
    y = fill_state_dict(dict(reversed(list(shapes.items()))), "default")
    assert all(torch.equal(x[k], y[k]) for k in shapes)
    assert torch.equal(synthetic_clip(1, 2, 64, 64, 3), synthetic_clip(1, 2, 64, 64, 3))
    c = synthetic_clip(2, 3, 64, 96, 1)
    assert c.shape == (2, 3, 3, 64, 96) and 0 <= c.min() and c.max() < 1


# ---- harness (f4): window index maps and PSNR -----------------------------------------
def test_harness_window_maps_and_psnr():
    import numpy as np
    from eavsr_amd import harness as H
    # test items: scenes of 14 frames cut into two windows of 7 (mvsr4x_dataset.py:130-136)
    assert H.test_window_starts(42, 14, 7) == [0, 7, 14, 21, 28, 35]
    with pytest.raises(ValueError):
        H.test_window_starts(30, 10, 7)
    # interior key frame: contiguous window
    assert H.train_window(idx=103, frame=3, n_frame=7, n_seq=50) == [100, 101, 102, 103, 104, 105, 106]
    # scene start: mirrored about the key frame, never below the first frame of the scene (image 100)
    assert H.train_window(idx=100, frame=0, n_frame=7, n_seq=50) == [103, 102, 101, 100, 101, 102, 103]
    assert H.train_window(idx=101, frame=1, n_frame=7, n_seq=50) == [104, 103, 100, 101, 102, 103, 104]
    # scene end (last frame is image 149): mirrored, never above it
    w = H.train_window(idx=149, frame=49, n_frame=7, n_seq=50)
    assert w[:4] == [146, 147, 148, 149] and max(w) == 149 and w[4:] == [148, 147, 146]
    # PSNR formula (util/util.py:302-320): uniform error of 1 grey level -> 20 log10(255)
    a = torch.zeros(1, 2, 3, 8, 8)
    assert abs(H.calc_psnr(a + 1.0, a) - 20 * np.log10(255.0)) < 1e-4
    assert H.crop_center(torch.arange(36.0).view(1, 6, 6), 2).flatten().tolist() == [14.0, 15.0, 20.0, 21.0]


def test_harness_ssim_follows_skimage_and_frames_are_written_as_png(tmp_path):
    """psnr_total.py:39-44's SSIM (skimage `structural_similarity(win_size=11, data_range=255,
    multichannel=True, gaussian_weights=True)`) and the frame writing of test_basic.py:85-92.  skimage is not installed here: its
    published algorithm is restated independently with scipy.ndimage (gaussian_filter sigma 1.5 / truncate 3.5 / reflect, sample
    covariance, crop by 5) and compared; known answers: identical images -> 1, a constant shift -> the luminance term alone."""
    import numpy as np
    from scipy.ndimage import gaussian_filter
    from eavsr_amd import harness as Hn
    g = torch.Generator().manual_seed(5)
    a = (torch.rand(3, 40, 52, generator=g) * 255).round()
    b = (a + torch.randn(3, 40, 52, generator=g) * 12).clamp(0, 255).round()

    def skimage_ssim(x, y):      # skimage/metrics/_structural_similarity.py, the gaussian_weights branch, per channel
        vals = []
        for c in range(x.shape[0]):
            im1, im2 = x[c].numpy().astype(np.float64), y[c].numpy().astype(np.float64)
            f = lambda t: gaussian_filter(t, sigma=1.5, truncate=3.5, mode="reflect")
            ux, uy = f(im1), f(im2)
            cn = 121.0 / 120.0
            vx, vy, vxy = cn * (f(im1 * im1) - ux * ux), cn * (f(im2 * im2) - uy * uy), cn * (f(im1 * im2) - ux * uy)
            c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
            S = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
            vals.append(S[5:-5, 5:-5].mean())
        return float(np.mean(vals))
    got, want = Hn.calc_ssim(b, a), skimage_ssim(b, a)
    assert abs(got - want) < 1e-9 and 0.2 < got < 0.99, (got, want)
    assert abs(Hn.calc_ssim(a, a) - 1.0) < 1e-12
    flat = torch.full((1, 32, 32), 100.0)
    lum = (2 * 100 * 110 + 6.5025) / (100 ** 2 + 110 ** 2 + 6.5025)      # contrast / structure terms are 1 for constants
    assert abs(Hn.calc_ssim(flat + 10, flat) - lum) < 1e-12
    assert abs(Hn.calc_ssim(torch.stack([b, a]), torch.stack([a, a])) - 0.5 * (got + 1.0)) < 1e-9      # leading dims are averaged
    with pytest.raises(ValueError):
        Hn.calc_ssim(torch.zeros(3, 8, 8), torch.zeros(3, 8, 8))
    # PNG: round trip through our own decoder, and through PIL where it is installed
    path = Hn.write_png(a, str(tmp_path / "x" / "frame.png"))
    assert torch.equal(Hn.read_png(path), a.to(torch.uint8))
    grey = (torch.rand(17, 9, generator=g) * 255)
    assert torch.equal(Hn.read_png(Hn.write_png(grey, str(tmp_path / "g.png")))[0], grey.to(torch.uint8))      # astype(uint8): truncation
    try:
        from PIL import Image
        assert np.array_equal(np.asarray(Image.open(path)), a.to(torch.uint8).permute(1, 2, 0).numpy())
        Image.fromarray(b.to(torch.uint8).permute(1, 2, 0).numpy()).save(str(tmp_path / "pil.png"))      # filtered scanlines
        assert torch.equal(Hn.read_png(str(tmp_path / "pil.png")), b.to(torch.uint8))
    except ImportError:
        pass
    # the folder layout of test_basic.py:85-92
    res = {"data_sr_seq": torch.stack([a, b]).unsqueeze(0)}
    fn = [["000/00000.png"], ["000/00001.png"]]
    out = Hn.save_visuals(res, fn, str(tmp_path / "ckpt" / "run"), load_iter="200", full_res=True)
    assert [os.path.relpath(o, str(tmp_path)) for o in out] == ["ckpt/run/sr_full_200/000/00000.png", "ckpt/run/sr_full_200/000/00001.png"]
    assert torch.equal(Hn.read_png(out[1]), b.to(torch.uint8))


# ---- the bench contract, checked on the committed line of the last GPU visit -------------------------------------
def test_committed_bench_line_follows_the_contract():
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_bench_line.json")
    line = json.loads(open(path).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["unit"] == "frames/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["data"] == "synthetic" and line["dtype"].startswith("f32")
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - line["n_gpus"] * 4 * 7 / (line["ms_per_step"] * 1e-3)) < 1e-3 * line["value"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.0 < r["frac"] <= 1.0                      # performed FLOP/s over the peak; the algorithmic-equivalent rate is its own key
    assert r["algorithmic_equivalent"]["achieved"] >= r["achieved"]
    for k in line["kernels"]:
        assert 0.0 < k["frac"] <= 1.0, k
    assert abs(line["value_median"] - line["n_gpus"] * 4 * 7 / (line["ms_per_step_median"] * 1e-3)) < 1e-3 * line["value"]
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1
    assert "no extrapolation" in c["sample"] and "180 x 320" in c["sample"]      # one full-size clip, not a scaled crop


def test_round4_bench_line_names_its_evidence():
    """round 4's committed line: how the per-kernel figures were taken, that the PMC traffic is replayed, every rank's time and
    device, and what the DCNv2 sampler saw"""
    import json
    path = os.path.join(ROOT, "profiles", "r04_bench_line.json")
    line = json.loads(open(path).read().strip().splitlines()[-1])
    assert "device-side delay" in line["kernel_timing"]
    assert "REPLAYED" in line["traffic_source"] and "not measured in this run" in line["traffic_source"]
    assert len(line["per_rank_ms"]) == line["n_gpus"] == len(line["per_rank_device"]) and line["host_threads_per_rank"] >= 1
    assert abs(max(line["per_rank_ms"]) - line["ms_per_step"]) < 0.02 * line["ms_per_step"]
    assert "lds_fill" not in line["roofline"]                     # the ingest-bound reading was withdrawn
    dcn = line["kernels"][0]
    assert dcn["kernel"] == "dcnv2_il_heads" and dcn["schedule"] == "il2" and 0.2 < dcn["frac"] < 1.0
    st = dcn["offset_stats"]
    assert st["calls"] > 0 and 0.0 <= st["frac_outside_lds_window"] <= 1.0 and st["max_abs"] >= st["mean_abs_dy"] >= 0.0
    syn = dcn["synthetic_offsets"]
    assert syn["sigma_4.0"]["frac_outside_lds_window"] > syn["sigma_0.5"]["frac_outside_lds_window"] >= 0.0
    assert syn["sigma_4.0"]["avg_ms"] > syn["sigma_0.5"]["avg_ms"] > 0.0
    assert dcn["on_torch_rand_clips"]["calls"] == dcn["calls"] and dcn["on_torch_rand_clips"]["avg_ms"] > 0.0
    assert line["config"]["dcnv2_schedule"].startswith("eavsr_dcnv2_il2_f32") and "conv3x3_wino4_schedule" in line["config"]
    # the default run also measures the other BASELINE configurations (child processes): their lines are part of the evidence
    oc = {e["config"]: e for e in line["other_configs"]}
    assert set(oc) == {2, 3, 4} and not any("error" in e or "skipped" in e for e in oc.values())
    assert "configs[2]" in oc[2]["metric"] and "bf16" in oc[2]["dtype"] and oc[2]["psnr_vs_fp32"]["psnr_db"] > 50.0
    assert "configs[4]" in oc[4]["metric"] and "fp16" in oc[4]["dtype"] and oc[4]["psnr_vs_fp32"]["psnr_db"] > 60.0
    assert "training" in oc[3]["metric"] and oc[3]["value"] > 0 and "HIP graph" in oc[3]["launch"]
    for k in (2, 4):
        assert oc[k]["timed_output_check"]["bit_identical"] and 0.5 < oc[k]["share_of_step_in_16bit"] < 1.0
        assert "h16" in oc[k]["roofline"]["kernel"] and oc[k]["roofline"]["peak"] == 2500.0 and 0.0 < oc[k]["roofline"]["hbm"]["frac"] < 1.0
    for tag, frames in (("config2_bf16", 7), ("config4_fp16", 15)):
        l2 = json.loads(open(os.path.join(ROOT, "profiles", f"r04_bench_line_{tag}.json")).read().strip().splitlines()[-1])
        c = l2["cpu_baseline"]
        assert c["value"] and c["kind"] == "port" and c["cores"] >= 1 and f"{frames} frames" in c["sample"].replace(" x ", " ").replace(f"x {frames} ", f"{frames} frames ")


def test_dcn_offset_stats_expand_the_heads_as_the_reference_does():
    """`ops.dcn_offset_stats` (measurement helper behind `kernels[0].offset_stats`) against the oracle's affine expansion
    (networks.py:302-311) and a hand-count of samples outside the kernel's LDS window"""
    from eavsr_amd import ops
    from oracle import eavsr_oracle as O
    D, n, h, w = 2, 1, 16, 64
    g = torch.Generator().manual_seed(5)
    heads = torch.cat([torch.randn(n, 4 * D, h, w, generator=g) * 0.3 + torch.tensor([1.0, 0, 0, 1.0]).repeat(D).view(1, 4 * D, 1, 1),
                       torch.randn(n, 2 * D, h, w, generator=g) * 3.0, torch.randn(n, 9 * D, h, w, generator=g)], 1)
    st = ops.dcn_offset_stats(heads, D)
    off = O.affine_offsets(heads[:, :4 * D], heads[:, 4 * D:6 * D], D).view(n, D, 9, 2, h, w)      # (dy, dx) per tap
    dy, dx = off[:, :, :, 0], off[:, :, :, 1]
    assert abs(st["mean_abs_dy"] - dy.abs().mean().item()) < 1e-5 and abs(st["mean_abs_dx"] - dx.abs().mean().item()) < 1e-5
    assert abs(st["max_abs"] - max(dy.abs().max().item(), dx.abs().max().item())) < 1e-5
    ky = torch.tensor([-1., -1., -1., 0., 0., 0., 1., 1., 1.]).view(1, 1, 9, 1, 1)
    kx = torch.tensor([-1., 0., 1., -1., 0., 1., -1., 0., 1.]).view(1, 1, 9, 1, 1)
    ry = 6 + (torch.arange(h).view(1, 1, 1, h, 1) % 8) + ky + torch.floor(dy)
    rx = 8 + (torch.arange(w).view(1, 1, 1, 1, w) % 32) + kx + torch.floor(dx)
    outside = ((ry < 0) | (ry > 18) | (rx < 0) | (rx > 46)).float().mean().item()
    assert outside > 0.0 and abs(st["frac_outside_lds_window"] - outside) < 1e-6


def test_heads_mask_activation_is_chosen_only_where_the_kernel_takes_it():
    from eavsr_amd import ops
    prev = (ops.DCN_IL_IMPL, ops.HEADS_MASK_ACTIVATED)
    try:
        ops.set_dcn_il_impl("il2")
        assert ops.heads_mask_activated(64) == (ops.CONV5_MODE == "bf16x6")
        assert ops.heads_mask_activated(24) is False              # cin % 16 != 0: the round-2 kernel runs, which wants logits
        assert ops.heads_mask_activated(64, 4) == ops.heads_mask_activated(64, 8)
        assert ops.heads_mask_activated(64, 2) is False           # 6 D = 12 is not a whole octet of heads channels
        assert ops.heads_mask_activated(64, 1) is False
        ops.set_dcn_il_impl("il")
        assert ops.heads_mask_activated(64) is False
        ops.set_dcn_il_impl("il2")
        ops.HEADS_MASK_ACTIVATED = False
        assert ops.heads_mask_activated(64) is False
    finally:
        ops.set_dcn_il_impl(prev[0])
        ops.HEADS_MASK_ACTIVATED = prev[1]


# ---- optimizer files are interchangeable with the reference's ------------------------------------------
def test_optimizer_groups_follow_the_reference_grouping_and_state_dicts_round_trip():
    """models/eavsrp_model.py:45-59 builds Adam's groups from ALL parameters() (frozen SPyNet tensors included): group 0 =
    every non-alignment parameter in registration order, group 1 = the alignment modules at lr 1e-5.  The parameter
    indices of an optimizer state_dict depend on exactly that, so the same construction is required here."""
    from eavsr_amd.eavsrp_model import EAVSRP, make_optimizer
    opt = Namespace(predict=False, n_frame=7, n_flow=5, scale=4, lr=1e-4, beta1=0.9, beta2=0.999, weight_decay=0.0)
    net = EAVSRP(opt, None)
    ours = make_optimizer(net, opt)
    # the reference's construction, restated from the cited lines (filter over module.parameters())
    align_id = []
    for name in ["backward_1", "forward_1", "backward_2", "forward_2"]:
        align_id += list(map(id, net.deform_align[name].parameters()))
    basic = list(filter(lambda p: id(p) not in align_id, net.parameters()))
    align = list(filter(lambda p: id(p) in align_id, net.parameters()))
    ref = torch.optim.Adam([{"params": basic}, {"params": align, "lr": 1e-5}], lr=opt.lr, betas=(opt.beta1, opt.beta2),
                           weight_decay=opt.weight_decay)
    g_ours, g_ref = ours.state_dict()["param_groups"], ref.state_dict()["param_groups"]
    assert [g["params"] for g in g_ours] == [g["params"] for g in g_ref]
    assert [g["lr"] for g in g_ours] == [1e-4, 1e-5]
    n_frozen = sum(1 for p in net.spynet.parameters())
    assert n_frozen == 60 and len(g_ours[0]["params"]) == len(basic) and len(basic) + len(align) == len(list(net.parameters()))
    assert all(p1 is p2 for p1, p2 in zip(ours.param_groups[0]["params"], basic))
    # one step of the reference-grouped optimizer (frozen tensors get no gradient, hence no state), then its state_dict
    # loads here and ours loads there
    for p in net.parameters():
        if p.requires_grad:
            p.grad = torch.full_like(p, 1e-3)
    ref.step()
    ours.load_state_dict(ref.state_dict())
    assert len(ours.state_dict()["state"]) == len(ref.state_dict()["state"]) == len(list(net.parameters())) - n_frozen
    ours.step()
    ref.load_state_dict(ours.state_dict())


def test_learning_rate_schedulers_follow_the_reference_policies():
    """models/networks.py:16-37 restated (get_scheduler): step halves every lr_decay_iters epochs, linear decays to zero over
    niter_decay after niter, cosine reaches zero at niter; unknown policies raise."""
    from eavsr_amd.eavsrp_model import get_scheduler
    mk = lambda: torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
    opt = Namespace(lr_policy="step", lr_decay_iters=2, niter=4, niter_decay=4)
    o = mk(); sch = get_scheduler(o, opt)
    lrs = []
    for _ in range(5):
        o.step(); sch.step(); lrs.append(o.param_groups[0]["lr"])
    assert lrs == pytest.approx([1e-4, 5e-5, 5e-5, 2.5e-5, 2.5e-5])
    opt.lr_policy = "linear"
    o = mk(); sch = get_scheduler(o, opt)
    lrs = []
    for _ in range(8):
        o.step(); sch.step(); lrs.append(o.param_groups[0]["lr"])
    assert lrs[:4] == pytest.approx([1e-4] * 4) and lrs[-1] == pytest.approx(0.0) and lrs[5] == pytest.approx(5e-5)
    opt.lr_policy = "cosine"
    o = mk(); sch = get_scheduler(o, opt)
    for _ in range(4):
        o.step(); sch.step()
    assert o.param_groups[0]["lr"] == pytest.approx(0.0, abs=1e-12)
    opt.lr_policy = "nope"
    with pytest.raises(NotImplementedError):
        get_scheduler(mk(), opt)


def test_every_derived_weight_cache_is_registered_for_clearing():
    """graph.clear_weight_caches() must drop EVERY cache keyed by (parameter, version); caches register themselves in
    ops.WEIGHT_CACHES -- no module-level `*_cache` dict of ops / autograd may stay outside it."""
    from eavsr_amd import autograd, graph, ops
    regs = {id(d) for d in ops.WEIGHT_CACHES}
    found = 0
    for mod in (ops, autograd):
        for name, val in vars(mod).items():
            if name.endswith("_cache") and isinstance(val, dict):
                assert id(val) in regs, f"{mod.__name__}.{name} is not registered (ops.register_weight_cache)"
                found += 1
    assert found >= 17 and id(ops.pack_cache) in regs
    ops._dgrad_w_cache["x"] = 1
    autograd._dcn_wt_cache["y"] = 2
    graph.clear_weight_caches()
    assert not ops._dgrad_w_cache and not autograd._dcn_wt_cache


def test_scoped_mode_switches_restore_the_previous_modes():
    """`ops.modes(...)` / `networks.backbone_dtype(...)`: scoped kernel selection that never leaks (also on an exception)."""
    from eavsr_amd import networks as Nw, ops
    before = (ops.CONV_MODE, ops.DCN_MODE, ops.DCN_IL_IMPL, Nw.BACKBONE_DTYPE)
    with ops.modes(conv="direct", dcn="native", dcn_il_impl="il"):
        assert (ops.CONV_MODE, ops.DCN_MODE, ops.DCN_IL_IMPL) == ("direct", "native", "il")
        with ops.modes(dcn="il9"):
            assert (ops.CONV_MODE, ops.DCN_MODE) == ("direct", "il9")
        assert ops.DCN_MODE == "native"
    assert (ops.CONV_MODE, ops.DCN_MODE, ops.DCN_IL_IMPL) == before[:3]
    with pytest.raises(ValueError):
        with ops.modes(conv="direct"):
            raise ValueError("boom")
    assert ops.CONV_MODE == before[0]
    with pytest.raises(ops.LabBuildRequired):      # a lab-only mode on the default build: refused, nothing left behind
        with ops.modes(dcn="native", conv="winograd"):
            pass
    assert (ops.CONV_MODE, ops.DCN_MODE) == before[:2]
    with pytest.raises(ValueError):
        with ops.modes(dcn="no-such-mode"):
            pass
    assert (ops.CONV_MODE, ops.DCN_MODE) == before[:2]
    with Nw.backbone_dtype("bf16"):
        assert Nw.BACKBONE_DTYPE == "bf16"
    assert Nw.BACKBONE_DTYPE == before[3]
