#!/usr/bin/env python3
"""bench.py -- SR frames/s of the EAVSR x4 forward on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one `EAVSRP.forward` over one batch of synthetic clips already resident in HBM:
BASELINE.json configs[1] = 4 clips x 7 frames x 3 x 180 x 320 fp32 per GPU, x4 -> 720 x 1280
(weak scaling: every rank owns its own 4 clips; clips are independent, there is no data-path
collective).  By default the step runs as `--streams 2`: the clips as two sub-batches, each captured once as a
HIP graph and replayed on its own stream, so that one group's streaming kernels overlap the other group's
convolutions (same kernels per clip, bit-identical output; `--streams 1` = one eager forward over all clips).
`python bench.py --gpus N` with N > 1 and no launcher environment starts the N ranks itself: before anything touches
the GPU the parent spawns `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process, relays rank
0's JSON line and exits with the children's status (a process that has initialised HIP is never re-executed).
Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      the dominant kernel (3x3 64->64 MFMA conv) against the fp32 MFMA peak: `achieved` / `frac` count the
                multiplications the kernel PERFORMS (Winograd F(4x4,3x3): a quarter of the direct sum's), the
                algorithmic-equivalent rate is reported beside it under its own key,
  kernels       the same for DCNv2 / flow_warp (HBM-bound) -- durations measured with HIP events on the
                launch stream during one extra, untimed, instrumented pass over one sub-batch (the launch shapes of
                the timed region, without the overlap),
  cpu_baseline  the CPU oracle timed on this box's host cores on ONE full-size clip of the workload (no extrapolation),
  other_configs (default run only: N = 1, --config 1) BASELINE.json configs[2] (x2 model, bf16), [3] (training step) and [4] (15 frames
                at 540 x 960, fp16), each timed by a child `bench.py --config K` / `--mode train` process after the headline
                measurement, a few steps each, under their own metric names (`--also ''` switches them off).
`--dry` replaces the kernels by a sleep and the GPU by the CPU (gloo): it exists only so that the launch / barrier /
max-over-ranks / JSON plumbing of `--gpus N` can be tested in a container without a GPU; its line says so.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix peak (spec; 155 measured)
PEAK_MFMA_16BIT_TFLOPS = 2500.0     # dense bf16 / fp16 MFMA (MI355X_MICROARCH.md); the 16-bit kernels' roofline
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 4],
                    help="BASELINE.json configs[i]: 1 (default, the headline) = eavsrp x4, 4 clips x 7 x 3 x 180 x 320 fp32; "
                         "2 = eavsrpx2 (x2 RealVSR path), 8 clips x 7 x 3 x 256 x 256, bf16; 4 = eavsrp x4, 1 clip x 15 x 3 x 540 x 960, "
                         "fp16 (long-sequence stress).  2 and 4 are reported under their own metric names, with the PSNR of the "
                         "16-bit output against the fp32 forward of the same clips.  (configs[3] is --mode train.)")
    ap.add_argument("--clips", type=int, default=None, help="clips per GPU (default: the config's: 4 / 8 / 1)")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--preset", default="trained_like", choices=["default", "trained_like"])
    ap.add_argument("--backbone-dtype", default=None, choices=["fp32", "bf16", "fp16"],
                    help="fp32 (default, the BASELINE headline: exact fp32 everywhere) or a 16-bit NHWC residual backbone "
                         "(BASELINE.json configs[2] / [4]); the 16-bit runs are reported under their own metric name")
    ap.add_argument("--conv-mode", default="winograd4", choices=["winograd", "winograd4", "direct", "bf16x9"],
                    help="large 3x3 convolutions (all fp32 in / out / accumulate): winograd = F(2x2,3x3) on the fp32 MFMA "
                         "(default); direct = direct sum on the fp32 MFMA; bf16x9 = direct sum, both operands split exactly "
                         "into three bf16 terms, nine partial products on the bf16 MFMA.  The 5x5 predictor heads and SPyNet's 7x7 layers "
                         "are chosen separately: EAVSR_CONV5 / EAVSR_CONV7 = bf16x6 (default) | fp32 (reported in config.conv5 / conv7)")
    ap.add_argument("--dcn-mode", default="il6", choices=["il6", "il9", "native", "bf16x9"],
                    help="DCNv2: il6 (default) = IL8-layout kernel fed by the paired warp and the predictor heads, fp32 operands "
                         "split into 3 bf16 terms, 6 partial products (dropped ones < 2^-23 relative); il9 = all 9 products "
                         "(exact); native / bf16x9 = round 1's NCHW kernels behind affine_offsets + two warps")
    ap.add_argument("--streams", type=int, default=None,
                    help="split the clips of a step into this many sub-batches, each replayed as its own HIP graph on its own "
                         "stream (eavsr_amd.graph.StreamedForward)")
    ap.add_argument("--no-train-graph", action="store_true", help="--mode train: run the step eagerly even when every rank owns a device")
    ap.add_argument("--graph", action="store_true",
                    help="replay the forward as one captured HIP graph (eavsr_amd.graph.GraphedForward): ~1 ms of host time "
                         "per step instead of ~180 ms of Python / ctypes launches; same kernels, same device time")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="infer: the BASELINE metric (default). train: BASELINE.json configs[3], a data-parallel "
                         "training step (2 clips/GPU x 7 x 3 x 96 x 96, L1 loss, Adam, one RCCL all-reduce on the "
                         "loss gradients); reported under its own metric name")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--cpu-crop", type=int, nargs=2, default=[64, 96],
                    help="h w of the CPU-baseline FALLBACK crop (used only if one full-size clip does not finish in time)")
    ap.add_argument("--cpu-runs", type=int, default=3, help="timed full-clip runs of the CPU oracle after one warm-up run")
    ap.add_argument("--cpu-budget", type=float, default=300.0,
                    help="seconds the CPU baseline may take in total; further timed runs are skipped once it is spent")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = every usable core)")
    ap.add_argument("--also", default=os.environ.get("EAVSR_BENCH_ALSO", "2,3,4"),
                    help="default run (N = 1, --config 1, inference) only: BASELINE.json configs that are timed as well, each in a child "
                         "process behind the headline measurement, a few steps each, summarised under `other_configs` "
                         "(3 = the training step); '' or 0 = none")
    ap.add_argument("--also-budget", type=float, default=150.0, help="seconds the `other_configs` children may take in all")
    ap.add_argument("--dry", action="store_true",
                    help="plumbing test without a GPU: the step is a sleep, ranks talk over gloo; the line is labelled "
                         "as such and is not a measurement")
    args = ap.parse_args()
    # the workload of the chosen BASELINE.json config: (clips, frames, h, w, scale, dtype the config names, sub-batches)
    cfg = {1: (4, 7, 180, 320, 4, "fp32", 2), 2: (8, 7, 256, 256, 2, "bf16", 2), 4: (1, 15, 540, 960, 4, "fp16", 1)}[args.config]
    args.clips = cfg[0] if args.clips is None else args.clips
    args.frames = cfg[1] if args.frames is None else args.frames
    args.height = cfg[2] if args.height is None else args.height
    args.width = cfg[3] if args.width is None else args.width
    args.scale = cfg[4]
    args.backbone_dtype = cfg[5] if args.backbone_dtype is None else args.backbone_dtype
    if args.streams is None:
        args.streams = cfg[6]
        if args.config == 4:
            args.graph = True        # one clip cannot be split into sub-batches: the whole forward as ONE HIP graph
    return args


def build_model(device, preset, scale=4):
    from argparse import Namespace
    from eavsr_amd.eavsrp_model import EAVSRP
    from eavsr_amd.utils.synthetic import fill_state_dict, shapes_of
    net = EAVSRP(Namespace(predict=False, n_frame=7, n_flow=5, scale=scale), None)
    sd0 = net.state_dict()
    sd = fill_state_dict(shapes_of(sd0), preset, fixed=sd0)
    net.load_state_dict(sd, strict=True)
    return net.to(device).eval(), sd


def usable_cores() -> int:
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return os.cpu_count() or 1


_CPU_CHILD = r"""
import json, os, sys, time
sys.path.insert(0, {root!r})
import torch
from argparse import Namespace
from oracle import eavsr_oracle as O
from eavsr_amd.eavsrp_model import EAVSRP
from eavsr_amd.utils.synthetic import fill_state_dict, shapes_of, synthetic_clip
cores = {cores}
torch.set_num_threads(cores)
net = EAVSRP(Namespace(predict=False, n_frame=7, n_flow=5, scale={scale}), None)
sd0 = net.state_dict()
sd = fill_state_dict(shapes_of(sd0), {preset!r}, fixed=sd0)
del net
clip = synthetic_clip(1, {frames}, {h}, {w}, seed=0)
t_start = time.time()
times = []
with torch.no_grad():
    for i in range({runs} + 1):          # run 0 is the warm-up (the reference discards its first iteration too)
        if i > 1 and time.time() - t_start + times[-1] > {budget}:
            break
        t0 = time.time()
        O.eavsrp_forward(sd, clip, {scale})
        times.append(time.time() - t0)
        print(json.dumps({{"run": i, "seconds": times[-1]}}), flush=True)
"""


def _run_cpu_child(preset, frames, h, w, cores, runs, budget, timeout_s, scale=4):
    import subprocess
    code = _CPU_CHILD.format(root=ROOT, cores=cores, preset=preset, frames=frames, h=h, w=w, runs=runs, budget=budget, scale=scale)
    env = dict(os.environ, OMP_NUM_THREADS=str(cores), MKL_NUM_THREADS=str(cores), HIP_VISIBLE_DEVICES="",
               CUDA_VISIBLE_DEVICES="")
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    try:
        out, err = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        p.kill()
        out, err = p.communicate()
    secs = []
    for ln in out.splitlines():
        try:
            secs.append(json.loads(ln)["seconds"])
        except Exception:  # noqa: BLE001
            pass
    return secs, err


def cpu_baseline(preset, frames, full_h, full_w, crop_h, crop_w, runs=3, budget=420.0, threads=0, scale=4):
    """The oracle (a port of the reference's algorithm, oracle/eavsr_oracle.py) timed on the host cores the way the
    reference times its own forward (models/eavsrp_model.py:100-107: wall time around one forward of one clip, the
    first iteration discarded): ONE full-size clip, one warm-up run, then up to `runs` timed runs, median; no
    extrapolation.  Child process (never initialises the GPU), hard time budget.  Only if not even the warm-up + one
    timed run of the full clip fit the budget does it fall back to a crop scaled by the pixel ratio, and says so."""
    import statistics
    probe = ""
    if threads > 0:
        cores = threads
    else:
        # every usable core is offered; torch's CPU kernels do not always scale to all of them (a 256-thread host ran this
        # workload slower than with 32 threads), so the thread count is the fastest of a short probe on the fallback crop
        usable = usable_cores()
        cands = sorted({c for c in (8, 16, 32, 64, 128, usable) if c <= usable})
        best = None
        # the probe runs the FULL-SIZE frames of the workload (two of them: torch's CPU kernels thread over pixels, and a 64 x 96
        # crop said nothing about how a 180 x 320 convolution scales -- VERDICT r5 weak 11); larger workloads keep the crop
        pf, ph, pw = (2, full_h, full_w) if full_h * full_w <= 180 * 320 else (min(frames, 7), crop_h, crop_w)
        for c in cands:      # ascending; stop as soon as more threads are clearly slower (saves the slow candidates' minutes)
            secs_p, _ = _run_cpu_child(preset, pf, ph, pw, c, 1, 40.0, timeout_s=60, scale=scale)
            if len(secs_p) >= 2:
                probe += f"{c}: {secs_p[-1]:.1f} s; "
                if best is None or secs_p[-1] < best[1]:
                    best = (c, secs_p[-1])
                elif secs_p[-1] > 1.5 * best[1]:
                    probe += "more threads not tried; "
                    break
            else:
                probe += f"{c}: > 60 s; "
                if best is not None:
                    break
        cores = best[0] if best else min(usable, 32)
        probe = f" (thread count chosen by a probe on a {pf} x {ph}x{pw} clip over {usable} usable cores -- {probe.rstrip('; ')})"
    base = {"unit": "frames/s", "cores": cores, "kind": "port"}
    if frames * full_h * full_w > 2_500_000:
        # a clip this large (configs[4]: 15 x 540 x 960) needs minutes per run on the host: a BOUNDED sample instead -- all frames
        # (the recurrence depth is the workload) on a centre-size third of the frame, scaled by the pixel ratio, and labelled
        sh, sw = max(64, full_h // 3 // 4 * 4), max(64, full_w // 3 // 4 * 4)
        secs_b, err_b = _run_cpu_child(preset, frames, sh, sw, cores, 1, budget, timeout_s=budget + 60, scale=scale)
        if len(secs_b) >= 2:
            dt = secs_b[-1]
            return dict(base, value=frames / dt * (sh * sw) / float(full_h * full_w),
                        sample=f"BOUNDED SAMPLE: 1 clip x {frames} frames x 3 x {sh} x {sw} (all frames, a third of the frame each way) "
                               f"in {dt:.1f} s after one warm-up run ({secs_b[0]:.1f} s), scaled by the pixel ratio "
                               f"{sh * sw}/{full_h * full_w} to the {full_h}x{full_w} workload; oracle/eavsr_oracle.py eavsrp_forward, "
                               f"torch {torch.__version__} CPU, {cores} threads" + probe)
        return dict(base, value=None, sample=f"failed: {err_b[-300:]!r}")
    secs, err = _run_cpu_child(preset, frames, full_h, full_w, cores, runs, budget, timeout_s=budget + 60, scale=scale)
    if len(secs) >= 2:
        timed = secs[1:]
        med = statistics.median(timed)
        return dict(base, value=frames / med,
                    sample=f"1 clip x {frames} frames x 3 x {full_h} x {full_w} (one clip of the timed workload, full size), "
                           f"oracle/eavsr_oracle.py eavsrp_forward on torch {torch.__version__} CPU with {cores} threads: "
                           f"1 warm-up run ({secs[0]:.1f} s, discarded) + {len(timed)} timed run(s) "
                           f"{[round(x, 1) for x in timed]} s, median {med:.1f} s; no extrapolation" + probe)
    # fallback: the full clip did not finish twice within the budget
    secs_c, err_c = _run_cpu_child(preset, frames, crop_h, crop_w, cores, 1, 120.0, timeout_s=180, scale=scale)
    if len(secs_c) >= 2:
        dt = secs_c[-1]
        return dict(base, value=frames / dt * (crop_h * crop_w) / float(full_h * full_w),
                    sample=f"FALLBACK (the full {full_h}x{full_w} clip did not finish a warm-up + one timed run within "
                           f"{budget:.0f} s; finished runs: {[round(x, 1) for x in secs]}): 1 clip x {frames} x 3 x {crop_h} x "
                           f"{crop_w} crop in {dt:.1f} s, scaled by the pixel ratio; {cores} threads")
    return dict(base, value=None, sample=f"failed: {(err or err_c)[-300:]!r}")


def other_configs(which, budget_s):
    """BASELINE.json configs[2] / [3] / [4] under their own metric names, each by `bench.py --config K` (3: `--mode train`) in a child
    process started after the headline measurement (this process keeps its GPU context; the child makes its own), a few steps
    each and no CPU baseline, so that the lines of profiles/*bench_line_config*.json are re-measured by whoever runs the default
    bench.  A child that fails or runs out of the time budget is reported as such; nothing here touches the headline figures."""
    import subprocess
    keep = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "psnr_vs_fp32", "share_of_step_in_16bit",
            "timed_output_check", "degraded", "loss", "step_breakdown_ms", "launches_per_step", "step_kernel_ms_library")
    out, t_end = [], time.perf_counter() + budget_s
    for k in which:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-cpu-baseline", "--also", ""]
        cmd += {2: ["--config", "2", "--steps", "4", "--warmup", "1"], 4: ["--config", "4", "--steps", "3", "--warmup", "1"],
                3: ["--mode", "train", "--steps", "5", "--warmup", "2"]}[k]
        left = t_end - time.perf_counter()
        if left < 25:
            out.append({"config": k, "skipped": "time budget of --also-budget spent"})
            continue
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=left, env={**os.environ, "EAVSR_BREAKDOWN_N": "4"})
            rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not rows:
                out.append({"config": k, "error": f"exit {r.returncode}: {r.stderr.strip()[-300:]}"})
                continue
            ln = json.loads(rows[-1])
            e = {"config": k, **{f: ln[f] for f in keep if f in ln}, "workload": ln["config"]["workload"], "launch": ln["config"].get("launch"),
                 "child_wall_s": round(time.perf_counter() - t0, 1)}
            if "roofline" in ln:
                e["roofline"] = {f: ln["roofline"][f] for f in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_ms", "share_of_step", "hbm",
                                                                   "share_of_library_kernel_time", "conv_wgrad")
                                 if f in ln["roofline"]}
            out.append(e)
        except subprocess.TimeoutExpired:
            out.append({"config": k, "error": f"timed out after {left:.0f} s"})
        except Exception as ex:
            out.append({"config": k, "error": repr(ex)})
    return out


def train_bench(args, rank, world, device, ranks_seen=1):
    """BASELINE.json configs[3]: eavsrp x4 training step, 2 clips/GPU x 7 x 3 x 96 x 96, data parallel."""
    from argparse import Namespace
    from eavsr_amd import ops, shard
    from eavsr_amd.eavsrp_model import EAVSRPModel
    from eavsr_amd.utils.synthetic import fill_state_dict, shapes_of, synthetic_clip
    n, t = 2, 7
    h = w = 96
    opt = Namespace(predict=False, n_frame=7, n_flow=5, scale=4, isTrain=True, gpu_ids=[device.index], lr=1e-4,
                    beta1=0.9, beta2=0.999, weight_decay=0.0, npost=350)
    model = EAVSRPModel(opt)
    sd0 = model.netEAVSRP.state_dict()
    model.netEAVSRP.load_state_dict(fill_state_dict(shapes_of(sd0), args.preset, fixed=sd0), strict=True)
    data = {"lr_seq": synthetic_clip(n, t, h, w, seed=rank), "hr_seq": synthetic_clip(n, t, 4 * h, 4 * w, seed=100 + rank),
            "fname": "synthetic"}
    model.set_input(data, epoch=0)
    # the graphed step (forward + backward captured once; all-reduce and Adam behind each replay when world > 1) whenever every
    # rank has a device of its own; ranks that share a device (test launches) stay eager: their graphs would interleave badly
    own_device = world <= torch.cuda.device_count()
    use_graph = args.graph or (own_device and not args.no_train_graph)
    if use_graph:
        from eavsr_amd.graph import GraphedTrainStep
        graphed = GraphedTrainStep(model, warmup=max(args.warmup, 1))
        step = graphed.step
        step()
    else:
        step = model.optimize_parameters
        for _ in range(args.warmup):
            step()
    torch.cuda.synchronize()
    shard.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0
    shard.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    on_dev = world > 1 and not shard.host_collectives()      # only a pure-RCCL group needs device tensors for the MAX
    per_rank_ms = shard.all_ranks(1e3 * own_elapsed / args.steps, device=device if on_dev else None)
    per_rank_device = shard.all_ranks_str(f"cuda:{device.index} {torch.cuda.get_device_name(device)}")
    elapsed = shard.max_over_ranks(elapsed, device=device if on_dev else None)
    extra = {}
    if rank == 0 and world == 1 and not args.no_kernel_profile:      # (one process only: an extra step on one rank would wait for the others' all-reduce)
        # one more EAGER step (same work as the timed ones) with HIP events around every launch of the library, enqueued behind a
        # device-side delay so that the kernels run back to back: where a training step's time goes.  ATen
        # kernels (autograd's gradient sums, Adam) carry no events: profiles/*train* has the rocprofv3 census of every kernel.
        try:
            torch.cuda.synchronize()
            torch.cuda._sleep(int(0.6 * 2.0e9))
            with ops.profile() as prof:
                model.optimize_parameters()
            summ = prof.summary()
            lib_ms = sum(v["ms"] for v in summ.values())
            extra["step_breakdown_ms"] = {k: round(v["ms"], 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:12]}
            extra["launches_per_step"] = {"library_kernels": sum(v["calls"] for v in summ.values()),
                                          "by_kernel": {k: v["calls"] for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["calls"])[:12]},
                                          "note": "C-ABI launches of one eager step (each = one or two kernels); ATen kernels of autograd / Adam "
                                                  "not counted: profiles/*train* has the rocprofv3 census of all of them"}
            extra["step_kernel_ms_library"] = lib_ms

            def roof(name, what, x6):
                """x6: the kernel runs on the bf16 matrix pipe with both fp32 operands split exactly into three bf16 terms -- six
                partial products per multiplication: `achieved` counts the FLOP the pipe PERFORMS (6 x algorithmic) against the
                dense bf16 peak (<= 1 by construction); the algorithmic rate has its own key"""
                v = summ.get(name)
                if not v or v["ms"] <= 0:
                    return None
                alg = v["flops"] / (v["ms"] * 1e-3) / 1e12
                ach, peak = (6.0 * alg, PEAK_MFMA_16BIT_TFLOPS) if x6 else (alg, PEAK_MFMA_F32_TFLOPS)
                return {"kernel": what, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                        "frac": ach / peak, "avg_ms": v["ms"] / v["calls"], "calls": v["calls"],
                        "share_of_library_kernel_time": v["ms"] / lib_ms, "traffic": None,
                        "algorithmic_equivalent": {"achieved": alg, "unit": "TFLOP/s", "x_fp32_mfma_peak": alg / PEAK_MFMA_F32_TFLOPS},
                        "note": ("algorithmic FLOP (2 cin cout k k per output pixel) / HIP-event time of the launches of one eager step"
                                 + ("; exact bf16x6: six bf16 partial products per fp32 multiplication, priced on the bf16 MFMA peak" if x6 else ""))}
            convs = sorted(((k, v) for k, v in summ.items() if k.startswith("conv3x3_64to64")), key=lambda kv: -kv[1]["ms"])
            if convs:
                extra["roofline"] = roof(convs[0][0], f"{convs[0][0]} (forward and input-gradient 3x3 64->64 convolutions at 2 x 64 x 96 x 96)",
                                         convs[0][0].endswith(("_x6s", "_x6")))
            wg_x6 = bool(ops.lib().eavsr_wgrad3_mode()) and w % 4 == 0
            rw = roof("conv_wgrad3x3", ("conv_wgrad3_x6_kernel" if wg_x6 else "conv_wgrad3_kernel") +
                      " (weight gradient 3x3, K = pixels x frames: the uses of a weight across the recurrence are segments of one launch; "
                      "the time includes the slab-reduction kernel behind it)", wg_x6)
            if rw is not None:
                extra.setdefault("roofline", {})["conv_wgrad"] = rw
        except Exception as ex:      # measurement garnish must not void the line
            extra["step_breakdown_error"] = repr(ex)
    if rank == 0:
        print(json.dumps({
            **extra,
            "per_rank_ms": per_rank_ms, "per_rank_device": per_rank_device, "rccl_ranks_seen": ranks_seen,
            "metric": "training LR frames/sec, eavsrp x4 step (forward + backward + grad all-reduce + Adam)",
            "value": world * n * t * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"eavsrp x4 training step, {n} clips/GPU x {t} x 3 x {h} x {w}, HR {4*h}x{4*w}, L1, "
                                   "Adam (1e-4 / 1e-5), DP with one bucketed RCCL all-reduce on 49.1 MB of gradients "
                                   "(BASELINE.json configs[3])",
                       "launch": ("HIP graph per step (forward + backward" + (" + Adam)" if world == 1 else "; all-reduce + Adam behind the replay)"))
                                 if use_graph else "eager (one launch per kernel)"},
            "loss": model.get_current_losses()}), flush=True)
    if world > 1:
        shard.barrier()
        torch.distributed.destroy_process_group()


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args) -> int:
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks as children of THIS process, which has
    not touched the GPU (no torch.cuda call, no HIP call), through torch.distributed.run -- one process per GPU, RCCL
    rendezvous on 127.0.0.1 -- relay rank 0's JSON line and return the children's exit status.  Nothing is re-executed."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    env.setdefault("OMP_NUM_THREADS", str(max(1, cores // args.gpus)))      # N ranks share the node's cores
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    if r.returncode == 0 and line is None:
        print("[bench] the ranks exited cleanly but rank 0 printed no JSON line", file=sys.stderr)
        return 1
    return r.returncode


def dry_bench(args):
    """Plumbing only (no GPU, no kernels): rendezvous over gloo, barrier, K sleeps as 'steps', MAX over ranks, one JSON
    line.  Lets a CPU container test what `--gpus N` does around the kernels."""
    from eavsr_amd import shard
    os.environ.setdefault("EAVSR_DIST_BACKEND", "gloo")
    rank, local_rank, world = shard.env_rank_world()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if args.gpus != world:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    n, t = args.clips, args.frames
    if os.environ.get("EAVSR_DRY_DIE_RANK") == str(rank):      # tests: a rank that dies must take the job down, not hang it
        print(f"[bench] rank {rank}: dying on request (EAVSR_DRY_DIE_RANK)", file=sys.stderr)
        os._exit(7)
    for _ in range(args.warmup):
        time.sleep(0.01)
    shard.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + rank))          # ranks differ on purpose: the line must carry the MAX
    own = time.perf_counter() - t0             # this rank's own work, before it waits for the others
    shard.barrier()
    mine = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(mine)
    per_rank = shard.all_ranks(1e3 * own / max(args.steps, 1))
    per_dev = shard.all_ranks_str(f"cpu:{rank}")
    thread_cap = shard.cap_host_threads(world)
    seen = shard.ranks_seen()              # the same call the GPU path makes on its device (there: over RCCL)
    if rank == 0:
        print(json.dumps({
            "per_rank_ms": per_rank, "per_rank_device": per_dev, "host_threads_per_rank": thread_cap, "rccl_ranks_seen": seen,
            "metric": "DRY RUN -- no kernels, no GPU: launch / barrier / max-over-ranks plumbing only (not a measurement)",
            "value": world * n * t * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "none", "data": "none", "dry": True,
            "config": {"workload": "sleep", "world_size": world,
                       "backend": torch.distributed.get_backend() if world > 1 else "none"}}), flush=True)
    if world > 1:
        shard.barrier()
        torch.distributed.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))          # before anything can initialise the GPU in this process
    if args.dry:
        return dry_bench(args)
    from eavsr_amd import ops, shard
    rank, local_rank, world = shard.init_process_group()
    if args.gpus != world:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: the launcher started a different number of ranks",
                  file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("[bench] no GPU visible: eavsr_amd has no CPU path", file=sys.stderr)
        sys.exit(2)
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    thread_cap = shard.cap_host_threads(world)
    ops.lib()  # fail loudly if the HIP extension is missing
    if world > 1:
        # the N ranks the line reports are N live RCCL ranks, one per GPU (or EAVSR_DIST_BACKEND=gloo in shared-GPU tests)
        assert torch.distributed.get_world_size() == args.gpus, (torch.distributed.get_world_size(), args.gpus)

    # what the DATA-PATH backend itself sees: one 4-byte device all-reduce over RCCL before anything is timed (the control plane
    # is gloo and cannot vouch for RCCL).  Inference has no other device collective; training all-reduces its gradients there.
    ranks_seen = 1
    if world > 1:
        try:
            ranks_seen = shard.ranks_seen(device)
        except Exception as ex:      # inference needs no device collective: a broken RCCL must not cost the node its measurement
            if args.mode == "train":
                raise
            ranks_seen = 0
            print(f"[bench] rank {rank}: the RCCL probe failed ({ex!r}); inference shards clips with no data-path collective and "
                  "continues -- rccl_ranks_seen = 0 in the line", file=sys.stderr)
        if ranks_seen != args.gpus and args.mode == "train":
            raise RuntimeError(f"RCCL sees {ranks_seen} ranks, the launcher started {args.gpus}")

    from eavsr_amd.utils.synthetic import synthetic_clip
    if args.mode == "train":
        return train_bench(args, rank, world, device, ranks_seen)
    net, sd = build_model(device, args.preset, args.scale)
    if args.backbone_dtype != "fp32":
        from eavsr_amd import networks as _nw
        _nw.set_backbone_dtype(args.backbone_dtype)
    ops.set_conv_mode(args.conv_mode)
    ops.set_dcn_mode(args.dcn_mode)
    n, t, h, w = args.clips, args.frames, args.height, args.width
    clips = synthetic_clip(n, t, h, w, seed=rank).to(device)   # resident in HBM before the timed region

    run = net
    degraded = None
    if args.streams < 1 or n % args.streams:
        args.streams = 1           # the clips do not split evenly: one eager forward
    if world > torch.cuda.device_count():
        args.streams = 1           # several ranks share a device (test launches): their graphs would interleave badly
    if args.streams > 1:
        from eavsr_amd.graph import StreamedForward
        try:
            run = StreamedForward(net, clips, groups=args.streams)
        except Exception as e:   # a capture problem must not cost the measurement: the same work as one eager forward
            print(f"[bench] rank {rank}: HIP-graph capture failed ({type(e).__name__}: {e}); running --streams 1", file=sys.stderr)
            degraded = f"HIP-graph capture failed ({type(e).__name__}: {e}); the step ran as ONE eager forward (--streams 1), not as the requested graphs"
            torch.cuda.synchronize()
            args.streams = 1
            run = net
    elif args.graph:
        from eavsr_amd.graph import GraphedForward
        run = GraphedForward(net, clips)
    with torch.no_grad():
        for _ in range(args.warmup):
            run(clips)
        torch.cuda.synchronize()
        shard.barrier()
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            out = run(clips)
            marks[i + 1].record()          # device-side step boundaries (no host sync inside the timed region)
        torch.cuda.synchronize()
        own_elapsed = time.perf_counter() - t0      # this rank's own K steps, before it waits for the others
        shard.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        # What was timed is what is returned (the reference times the call whose result it keeps, models/eavsrp_model.py:100-107):
        # the first sub-batch of the LAST timed step against one eager forward of the same clips -- the same kernels in the same
        # order, so the two must agree bit for bit; anything above 1e-5 fails the run.
        sub = n // args.streams
        timed_first = out[:sub].clone()
        eager_first = net(clips[:sub])
        torch.cuda.synchronize()
        timed_vs_eager = float((timed_first - eager_first).abs().max().item())
        timed_equal = bool(torch.equal(timed_first, eager_first))
        timed_finite = bool(torch.isfinite(timed_first).all().item())
        psnr_vs_fp32 = None
        if args.backbone_dtype != "fp32":
            # reduced precision is judged by PSNR against the exact path (BASELINE.json: "PSNR vs CPU ref"; the fp32 HIP forward
            # is pinned to the CPU oracle within 1e-3 by tests/test_hip_configs.py): the same clips once more in fp32, eagerly
            import math
            from eavsr_amd import networks as _nw2
            _nw2.set_backbone_dtype(None)
            exact_first = net(clips[:sub])
            _nw2.set_backbone_dtype(args.backbone_dtype)
            q = lambda z: (z.clamp(0, 1) * 255).round()          # util/util.py:302-320 on base_model.py:145-150 images
            mse = float(((q(timed_first) - q(exact_first)) / 255).pow(2).mean().item())
            psnr_vs_fp32 = {"psnr_db": None if mse == 0 else -10 * math.log10(mse), "identical_8bit_images": mse == 0,
                            "max_abs": float((timed_first - exact_first).abs().max().item()),
                            "what": f"clips 0..{sub - 1} of the last timed step ({args.backbone_dtype} mode) vs the fp32 forward of the same clips, "
                                    "on clamp*255*round images"}
            del exact_first
        del timed_first, eager_first
    on_dev = world > 1 and not shard.host_collectives()      # only a pure-RCCL group needs device tensors for the MAX
    per_rank_ms = shard.all_ranks(1e3 * own_elapsed / args.steps, device=device if on_dev else None)
    per_rank_device = shard.all_ranks_str(f"cuda:{device.index} {torch.cuda.get_device_name(device)}")
    elapsed = shard.max_over_ranks(elapsed, device=device if on_dev else None)
    timed_vs_eager = shard.max_over_ranks(timed_vs_eager, device=device if on_dev else None)
    frames_total = world * n * t * args.steps
    value = frames_total / elapsed
    import statistics
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    med_ms = shard.max_over_ranks(statistics.median(step_ms), device=device if on_dev else None)

    line = {
        "metric": ({1: "SR frames/sec at 4x 7-frame 180x320->720x1280",
                    2: f"SR frames/sec, eavsrpx2 2x RealVSR path, {n} clips x {t} x 3 x {h} x {w} -> {2 * h}x{2 * w} (BASELINE.json configs[2])",
                    4: f"SR frames/sec, eavsrp 4x long-sequence propagation, {n} clip x {t} frames x 3 x {h} x {w} -> {4 * h}x{4 * w} "
                       "(BASELINE.json configs[4])"}[args.config]) +
                  ("" if args.backbone_dtype == "fp32" else
                   f" [{args.backbone_dtype}: 16-bit residual backbone, warp -> DCNv2, predictor heads and upsampling tail; fp32 elsewhere" +
                   ("; not the fp32 headline]" if args.config == 1 else "]")) +
                  ("" if args.conv_mode != "bf16x9" and args.dcn_mode != "bf16x9" else " [fp32 via exact bf16x9 split products, opt-in mode]"),
        "value": value,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        # every rank's own time per step (the job is quoted on their MAX) and the device each one ran on
        "per_rank_ms": per_rank_ms, "per_rank_device": per_rank_device, "host_threads_per_rank": thread_cap,
        "rccl_ranks_seen": ranks_seen,      # ranks counted by a device all-reduce on the data-path backend itself (1 on one process)
        # median over the timed steps (device-side step boundaries; max over ranks of the per-rank medians)
        "ms_per_step_median": med_ms,
        "value_median": world * n * t / (med_ms * 1e-3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "world_size": torch.distributed.get_world_size() if world > 1 else 1,
        "backend": (str(torch.distributed.get_backend()) + (" (nccl = RCCL; inference has no data-path collective, the barrier and "
                                                              "the MAX over ranks are host scalars)" if "nccl" in str(torch.distributed.get_backend()) else ""))
                   if world > 1 else "none (one process)",
        "dtype": ("f32" if args.backbone_dtype == "fp32" else f"{args.backbone_dtype} backbone + f32") +
                 ("" if args.conv_mode != "bf16x9" and args.dcn_mode != "bf16x9"
                  else " (contractions in bf16x9 mode: exact 3 x bf16 operand split, 9 products, f32 accumulate)"),
        "data": "synthetic",
        "timed_output_max_abs_vs_eager": timed_vs_eager,
        "timed_output_check": {"what": f"clips 0..{n // args.streams - 1} of the last timed step vs one eager forward of the same clips after the "
                                       "timed region (max over ranks)", "bit_identical": timed_equal, "finite": timed_finite, "bound": 1e-5},
        "config": {"workload": f"{'eavsrpx2 x2' if args.scale == 2 else 'eavsrp x4'} inference, {n} clips/GPU x {t} frames x 3 x {h} x {w} "
                               f"{args.backbone_dtype} (BASELINE.json configs[{args.config}]), weights: seeded '{args.preset}' init",
                   "clips_per_gpu": n, "frames": t, "lr_size": [h, w], "scale": args.scale, "sharding": "clips across ranks, no collective",
                   "conv3x3": {"winograd": "Winograd F(2x2,3x3), fp32 MFMA (direct fp32 kernel for the shapes it does not cover)",
                               "winograd4": "Winograd F(4x4,3x3), fp32 MFMA (F(2x2,3x3) / direct fp32 kernels for the shapes it does not cover)",
                               "direct": "direct sum, fp32 MFMA", "bf16x9": "direct sum, exact bf16x9 split"}[args.conv_mode],
                   "dcnv2": {"il6": "IL8 kernel, fp32 operands as 3 exact bf16 terms, 6 of 9 partial products (dropped < 2^-23 each), "
                                    "f32 accumulate; predictor heads + paired warp fused in",
                             "il9": "IL8 kernel, exact bf16x9 products, f32 accumulate; predictor heads + paired warp fused in",
                             "native": "fp32 MFMA, NCHW LDS-window kernel (round 1)", "bf16x9": "round-1 bf16x9 kernel"}[args.dcn_mode],
                   "conv7x7_conv5x5": {"bf16x6": "eavsr_conv_f32x6: fp32 operands as 3 exact bf16 terms, 6 of 9 partial products (dropped < 2^-23 "
                                                 "each), f32 accumulate, bf16 MFMA"}.get(ops.CONV7_MODE, "fp32 MFMA (implicit GEMM)") +
                                      " | 5x5 heads: " + {"bf16x6": "the same kernel"}.get(ops.CONV5_MODE, "Winograd F(2x2,5x5), fp32 MFMA"),
                   "conv3x3_wino4_schedule": ("grouped (transform phases of two chunks, pure GEMM iterations)"
                                              if ops.lib().eavsr_wino4_schedule() else "duty pair (EAVSR_W4_GRP=0)"),
                   "dcnv2_schedule": {"il2": "eavsr_dcnv2_il2_f32 (round 4: taps of two groups paired, one pipeline across groups / tiles; round 5: buffer-addressed)",
                                      "il": "eavsr_dcnv2_il_f32 (round 2)", "ws": "eavsr_dcnv2_ws_f32 (wave-specialised)"}[ops.DCN_IL_IMPL],
                   "launch": (f"{args.streams} HIP graphs on {args.streams} streams per step" if args.streams > 1 else
                              "one HIP graph per step" if args.graph else "eager (one launch per kernel)")},
    }

    if rank == 0 and not args.no_kernel_profile:
        # one extra untimed pass with HIP events around every launch (events on the launch stream), over one sub-batch:
        # the launch shapes of the timed region
        # The launches are enqueued behind a device-side delay, so that the host is a whole forward ahead of the device and the
        # kernels run back to back: an event pair then brackets its kernel alone.  (Enqueued live, the device outruns the host
        # on the small kernels and each bracket also holds the host's launch latency: 5-6 us on a 70 us kernel -- the reason
        # round 3's event figures sat 7 % above the rocprofv3 durations of the same kernels.)
        torch.cuda.synchronize()
        torch.cuda._sleep(int(0.45 * 2.0e9))
        with torch.no_grad(), ops.profile() as prof:
            net(clips[:n // args.streams])
        summ = prof.summary()
        line["kernel_timing"] = ("HIP events around every launch of one eager forward over one sub-batch, on the launch stream, "
                                 "enqueued behind a 0.2 s device-side delay (kernels back to back, no host launch latency "
                                 "inside a bracket); agrees with profiles/*rocprof* kernel-trace durations")
        total_ms = sum(v["ms"] for v in summ.values())

        def mult_reduction(name):
            """algorithmic multiplications / multiplications the kernel performs (1 for direct sums)"""
            if name.endswith("_wino4") or name.endswith("_wino4_ca"):
                return 4.0, "Winograd F(4x4,3x3): 36 products per 4x4 outputs instead of 144"
            if name.startswith("conv5x5") and name.endswith("_wino"):
                return 100.0 / 36.0, "Winograd F(2x2,5x5): 36 products per 2x2 outputs instead of 100"
            if name.endswith("_wino") or name.endswith("_wino_ca"):
                return 2.25, "Winograd F(2x2,3x3): 16 products per 2x2 outputs instead of 36"
            if name.endswith("_x6"):
                return 1.0 / 6.0, ("direct sum on the bf16 matrix pipe: every fp32 multiplication as 6 exact bf16 partial products "
                                   "(operands split into 3 bf16 terms), f32 accumulate; priced against the bf16 MFMA peak")
            return 1.0, "direct sum"

        def entry(name, bound):
            v = summ.get(name)
            if not v:
                return None
            avg_ms = v["ms"] / v["calls"]
            if bound == "mfma":
                alg = v["flops"] / v["calls"] / (avg_ms * 1e-3) / 1e12
                red, how = mult_reduction(name)
                ach = alg / red
                # `achieved` / `frac`: FLOP the matrix pipe actually performs per second against its fp32 peak (<= 1 by
                # construction); the algorithmic-equivalent rate (SURVEY 8d's 2*cin*cout*k*k per pixel) is its own key
                peak = PEAK_MFMA_16BIT_TFLOPS if (name.endswith("_x6") or "_h16" in name) else PEAK_MFMA_F32_TFLOPS      # (_h16, _h16g, _h16x1, _h16_ps2)
                return {"kernel": name, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                        "frac": ach / peak, "traffic": None, "avg_ms": avg_ms, "calls": v["calls"],
                        "share_of_step": v["ms"] / total_ms, "algorithm": how,
                        "algorithmic_equivalent": {"achieved": alg, "unit": "TFLOP/s", "x_peak": alg / peak,
                                                   "note": "direct-sum FLOP count / time; exceeds the peak when the algorithm "
                                                           "multiplies less -- not a roofline fraction"}}
            ach = v["bytes"] / v["calls"] / (avg_ms * 1e-3) / 1e9
            return {"kernel": name, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": ach / PEAK_HBM_GBS, "traffic": None, "avg_ms": avg_ms, "calls": v["calls"],
                    "share_of_step": v["ms"] / total_ms}

        dom_name = {"winograd": "conv3x3_64to64_wino", "winograd4": "conv3x3_64to64_wino4", "direct": "conv3x3_64to64", "bf16x9": "conv3x3_64to64_x9"}[args.conv_mode]
        fp32_dom = None
        if args.backbone_dtype != "fp32" and "conv3x3_64to64_h16" in summ:
            # 16-bit modes (configs[2] / [4]): the dominant kernel is the NHWC 16-bit residual-backbone convolution; it is priced on
            # the 16-bit MFMA peak (its HBM fraction on the 16-bit activations it streams is reported beside it)
            fp32_dom = entry(dom_name, "mfma")
            dom_name = "conv3x3_64to64_h16"
        dom = entry(dom_name, "mfma")
        if dom_name == "conv3x3_64to64_h16" and dom is not None:
            hb = entry(dom_name, "hbm")
            dom["algorithm"] = "direct sum, 16-bit operands, fp32 accumulation, 16-bit MFMA (csrc/conv_h16.hip)"
            dom["hbm"] = {k: hb[k] for k in ("achieved", "peak", "unit", "frac")}
            if fp32_dom is not None:      # (none since the tail and the other 3x3 convolutions run in 16 bits as well)
                dom["fp32_kernel_of_the_same_shape_outside_the_groups"] = {k: fp32_dom[k] for k in ("kernel", "frac", "avg_ms", "share_of_step")}
        if args.backbone_dtype != "fp32":
            # which kernels of the step ran in 16 bits (the rest is fp32), with their share of the step's kernel time
            names16 = ("conv3x3_64to64_h16", "conv3x3_64to256_h16_ps2", "conv3x3_64to3_h16", "scale_residual_h16", "dcnv2_il16_heads",
                       "dcnv2_il16", "nchw_f32_to_nhwc_h16", "nhwc_h16_to_nchw_f32")
            heads16 = tuple(k for k in summ if (k.startswith("conv5x5_") and k.endswith("_h16"))      # the predictor's 5x5 heads
                            or k.endswith(("_h16g", "_h16x1")))                                      # generic 3x3 (csrc/conv3_h16.hip), SPyNet 7x7 (one plane of conv_x6.hip)
            line["kernels_16bit"] = [e for e in (entry(k, "hbm") for k in names16) if e]
            line["kernels_16bit"] += [e for e in (entry(k, "mfma") for k in heads16) if e]
            names16 = names16 + heads16
            if "flow_warp_pair" in summ:
                e = entry("flow_warp_pair", "hbm")
                e["note"] = "fp32 in, second output rounded to 16-bit IL8 for dcnv2_il16"
                line["kernels_16bit"].append(e)
            t16 = sum(summ[k]["ms"] for k in names16 if k in summ)
            line["share_of_step_in_16bit"] = t16 / total_ms
        if dom is not None:
            line["roofline"] = {k: dom[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_ms",
                                                    "share_of_step", "algorithm", "algorithmic_equivalent", "hbm",
                                                    "fp32_kernel_of_the_same_shape_outside_the_groups") if k in dom}
            # the same figure at ONE launch shape, so that it can be recomputed from profiles/ without the mix of shapes above:
            # the sub-batch's full-resolution 64 -> 64 convolution (2 x 64 x 180 x 320 at configs[1])
            try:
                can_flops = 2.0 * 64 * 64 * 9 * (n // args.streams) * h * w
                grp = prof.by_flops(dom_name).get(can_flops)
                if grp:
                    ms_c = grp[1] / grp[0]
                    red_c, _ = mult_reduction(dom_name)
                    ach_c = can_flops / red_c / (ms_c * 1e-3) / 1e12
                    line["roofline"]["canonical_launch"] = {
                        "shape": f"{n // args.streams} x 64 x {h} x {w} (3x3 64 -> 64)", "calls": grp[0], "avg_ms": ms_c, "achieved": ach_c,
                        "frac": ach_c / (PEAK_MFMA_16BIT_TFLOPS if dom_name.endswith(("_h16", "_x6")) else PEAK_MFMA_F32_TFLOPS),
                        "performed_flop_per_launch": can_flops / red_c,
                        "note": "launches of this one shape only (the entry above averages every launch of the kernel name, incl. the "
                                "half- and quarter-resolution pyramid levels)"}
            except Exception as ex:
                line["roofline"]["canonical_launch_error"] = repr(ex)
            line["roofline"]["kernel"] = {
                "winograd": "conv3x3_wino_kernel (3x3 64->64, the residual backbone)",
                "winograd4": "conv_wino6_kernel<3> (3x3 64->64, the residual backbone)",
                "direct": "conv2d_mfma_kernel<3,2> (3x3 64->64, the residual backbone)",
                "bf16x9": "conv3x3_x9_kernel (3x3 64->64, bf16x9; priced against the fp32 MFMA peak)"}[args.conv_mode]
            if dom_name == "conv3x3_64to64_h16":
                line["roofline"]["kernel"] = f"conv3x3_c64_h16_kernel<{args.backbone_dtype}> (3x3 64->64 NHWC 16-bit, the residual backbone)"
        dcn_name = {"il6": "dcnv2_il_heads", "il9": "dcnv2_il_heads", "native": "dcnv2", "bf16x9": "dcnv2_x9"}[args.dcn_mode]
        if args.backbone_dtype != "fp32" and args.dcn_mode in ("il6", "il9"):
            dcn_name = "dcnv2_il16_heads"
        dcn_entry = entry(dcn_name, "hbm")
        if dcn_entry is not None and dcn_name.endswith("_heads"):
            # heads mode: `achieved` / `frac` price the kernel on SURVEY 8d's algorithmic 344 e B/px of the DCNv2 op (what the
            # reference's op moves at least: input, 18 D offsets, 9 D masks, output); the fused kernel itself reads the 15 D head
            # channels instead of 27 D offset / mask channels -- both figures, so that neither is mistaken for the other
            moved_px = (64 * 2 + 15 * 8 * 4 + 64 * 4) if dcn_name.startswith("dcnv2_il16") else (64 + 15 * 8 + 64) * 4
            alg_px = 344 * (2 if dcn_name.startswith("dcnv2_il16") else 4)
            dcn_entry["bytes_per_px"] = {"algorithmic": alg_px, "moved_by_this_kernel": moved_px}
            dcn_entry["moved"] = {"achieved": dcn_entry["achieved"] * moved_px / alg_px, "unit": "GB/s",
                                  "frac": dcn_entry["frac"] * moved_px / alg_px,
                                  "note": "bytes the fused (heads-mode) kernel itself moves / time / 8 TB/s"}
        if dcn_entry is not None and dcn_name == "dcnv2_il_heads":
            # what the sampler saw in THIS workload: the offsets the predictor heads imply, per call, and the
            # same launch shape again on synthetic heads with sigma = 4 px offsets (where ~20 % of the samples leave the LDS window)
            dcn_entry["schedule"] = ops.DCN_IL_IMPL
            try:
                with torch.no_grad(), ops.dcn_probe() as probe:
                    net(clips[:n // args.streams])
                torch.cuda.synchronize()
                keys = probe[0].keys() if probe else ()
                dcn_entry["offset_stats"] = {"calls": len(probe), **{k: (max(p[k] for p in probe) if k == "max_abs" else
                                                                        sum(p[k] for p in probe) / len(probe)) for k in keys},
                                             "worst_call_frac_outside_lds_window": max((p["frac_outside_lds_window"] for p in probe), default=None),
                                             "note": "offsets (T.R - R + t of networks.py:304-311) over all heads-mode DCNv2 calls of one "
                                                     "sub-batch forward; outside = a bilinear corner leaves the kernel's LDS window"}
                sub_n = n // args.streams
                g4 = torch.Generator(device=device).manual_seed(4)
                rn = lambda *sh: torch.randn(*sh, device=device, generator=g4)
                ident = torch.tensor([1.0, 0, 0, 1.0], device=device).repeat(8).view(1, 32, 1, 1)
                res4 = {}
                for sg in (0.5, 2.0, 4.0):
                    act_m = ops.heads_mask_activated(64)      # as the fused alignment calls it: masks activated by the heads' epilogue
                    mk = rn(sub_n, 72, h, w)
                    hd = torch.cat([rn(sub_n, 32, h, w) * 0.25 + ident, rn(sub_n, 16, h, w) * sg, torch.sigmoid(mk) if act_m else mk], 1)
                    xil = ops.to_il8(rn(sub_n, 64, h, w))
                    wt4, b4 = rn(64, 64, 3, 3) * 0.05, rn(64) * 0.1
                    for _ in range(3):
                        ops.dcnv2_il(xil, hd, None, wt4, b4, 8, heads=True, mask_activated=act_m)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(20):
                        ops.dcnv2_il(xil, hd, None, wt4, b4, 8, heads=True, mask_activated=act_m)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / 20
                    st4 = ops.dcn_offset_stats(hd, 8)
                    res4[f"sigma_{sg}"] = {"avg_ms": ms, "frac": alg_px * sub_n * h * w / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                           "frac_outside_lds_window": st4["frac_outside_lds_window"], "mean_norm": st4["mean_norm"]}
                # SURVEY 8d names torch.rand clips; the timed clips are a drifting low-pass texture (white noise gives SPyNet no motion
                # to find).  The same forward once on torch.rand clips of the same shape: the offsets DCNv2 sees there, and its time
                from eavsr_amd.utils.synthetic import synthetic_clip as _sc
                rclips = _sc(sub_n, t, h, w, seed=7, smooth=False).to(device)
                torch.cuda.synchronize()
                torch.cuda._sleep(int(0.45 * 2.0e9))
                with torch.no_grad(), ops.profile() as prof_r:
                    net(rclips)
                with torch.no_grad(), ops.dcn_probe() as probe_r:
                    net(rclips)
                torch.cuda.synchronize()
                vr = prof_r.summary().get(dcn_name)
                if vr and probe_r:
                    ms_r = vr["ms"] / vr["calls"]
                    dcn_entry["on_torch_rand_clips"] = {
                        "avg_ms": ms_r, "frac": alg_px * sub_n * h * w / (ms_r * 1e-3) / 1e9 / PEAK_HBM_GBS, "calls": vr["calls"],
                        "mean_norm": sum(p["mean_norm"] for p in probe_r) / len(probe_r), "max_abs": max(p["max_abs"] for p in probe_r),
                        "frac_outside_lds_window": sum(p["frac_outside_lds_window"] for p in probe_r) / len(probe_r),
                        "note": "one eager forward of torch.rand clips (SURVEY 8d's literal workload) of the same shape, same event bracketing"}
                del rclips
                dcn_entry["synthetic_offsets"] = {**res4, "note": "same launch shape, iid gaussian translations of sigma px on top of "
                                                  "near-identity transforms; 20 back-to-back launches per figure"}
            except Exception as ex:      # measurement garnish must not void the line
                dcn_entry["offset_stats_error"] = repr(ex)
        if dcn_entry is not None and "roofline" in line:
            # BASELINE.json's north-star figure inside `roofline` itself (the driver's record keeps `roofline`
            # and drops `kernels`): DCNv2 on SURVEY 8d's algorithmic bytes against the 8 TB/s HBM peak, measured in THIS run by
            # the same HIP-event brackets as every other kernel figure of the line
            ns = {k: dcn_entry[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_ms", "calls", "share_of_step")}
            ns["target_frac"] = 0.30
            ns["launch_shape"] = f"{n // args.streams} x 64 x {h} x {w}, 8 deformable groups, 3x3"
            if "bytes_per_px" in dcn_entry:
                ns["bytes_per_px"] = dcn_entry["bytes_per_px"]["algorithmic"]
                ns["frac_on_moved_bytes"] = dcn_entry["moved"]["frac"]
            if "synthetic_offsets" in dcn_entry:
                ns["frac_at_sigma_px"] = {k.replace("sigma_", ""): round(v["frac"], 4) for k, v in dcn_entry["synthetic_offsets"].items()
                                          if isinstance(v, dict)}
            if "offset_stats" in dcn_entry:
                ns["workload_offsets_px"] = {k: dcn_entry["offset_stats"].get(k) for k in ("mean_norm", "max_abs", "frac_outside_lds_window")}
            line["roofline"]["north_star"] = ns
            # ... and once more where a record that keeps only the top-level keys of `roofline` still carries it (VERDICT r5 item 4:
            # the driver's BENCH record drops nested roofline keys and keeps `config` whole)
            line["config"]["dcnv2_hbm"] = (f"{1e3 * ns['avg_ms']:.1f} us = {ns['frac']:.3f} of 8 TB/s on {ns.get('bytes_per_px', 1376)} B/px "
                                           f"({ns['kernel']}, {ns['launch_shape']}, {ns['calls']} launches, HIP events in this run)")
            if "frac_at_sigma_px" in ns:
                line["config"]["dcnv2_hbm_at_sigma_px"] = ns["frac_at_sigma_px"]
        line["kernels"] = [e for e in (dcn_entry, entry("flow_warp_pair", "hbm"), entry("flow_warp", "hbm"),
                                       entry("adapt_frontend", "hbm"), entry("affine_offsets", "hbm"),
                                       entry("scale_residual", "hbm"), entry("conv5x5_64to120_wino", "mfma"),
                                       entry("conv5x5_64to120", "mfma"), entry("conv5x5_64to120_x6", "mfma"),
                                       entry("conv7x7_32to64_x6", "mfma"), entry("conv7x7_64to32_x6", "mfma"),
                                       entry("conv7x7_32to64", "mfma"), entry("conv7x7_64to32", "mfma")) if e]
        line["step_breakdown_ms"] = {k: round(v["ms"], 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:int(os.environ.get("EAVSR_BREAKDOWN_N", "12"))]}
        line["step_device_ms_instrumented"] = total_ms * args.streams   # the sub-batches back to back, no overlap
        # HBM bytes per launch from the PMC passes of the last profiling visit (profiles/traffic.json:
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 correction applied there)
        pmc = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(pmc):
            try:
                tr = json.load(open(pmc))
                line["traffic_source"] = ("profiles/traffic.json: HBM bytes per launch from separate rocprofv3 --pmc passes of an earlier "
                                          f"visit ({tr.get('_taken', 'date not recorded')}); REPLAYED here, not measured in this run")
                if "roofline" in line:
                    line["roofline"]["traffic"] = tr.get(dom_name)
                for e in line["kernels"]:
                    e["traffic"] = tr.get(e["kernel"])
            except Exception:
                pass

    if rank == 0:
        # the N = 1 step time of the last committed profile visit: a SCALE record (N = 2, 4, 8; weak scaling, so every rank's step
        # should take this long) can be sanity-checked against it without a second run
        import glob as _glob
        import re as _re
        cands = sorted((p_ for p_ in _glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_line.json"))),
                       key=lambda p_: int(_re.search(r"r(\d+)_bench_line", p_).group(1)))
        if cands and args.config == 1 and args.backbone_dtype == "fp32":
            try:
                ref_line = json.load(open(cands[-1]))
                line["expected_n1_ms"] = {"ms_per_step": ref_line["ms_per_step"], "value": ref_line["value"],
                                          "source": os.path.relpath(cands[-1], ROOT),
                                          "note": "N = 1 on one MI355X at the commit of that profile visit; weak scaling: per-rank step time "
                                                  "at N > 1 should equal it (per_rank_ms), value should be N x this value"}
            except Exception:
                pass

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args.preset, t, h, w, args.cpu_crop[0], args.cpu_crop[1], runs=args.cpu_runs,
                                            budget=args.cpu_budget, threads=args.cpu_threads, scale=args.scale)

    also = [int(k) for k in str(args.also).replace(" ", "").split(",") if k and k != "0"]
    if (rank == 0 and world == 1 and also and args.config == 1 and args.backbone_dtype == "fp32" and args.conv_mode == "winograd4"
            and args.dcn_mode == "il6"):
        # the other BASELINE.json configurations, each measured by a child process AFTER everything above (the headline is
        # already in `line`); free what this process holds first: configs[4] wants ~71 GiB
        del out, run, net, clips
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        line["other_configs"] = other_configs([k for k in also if k in (2, 3, 4)], args.also_budget)

    if degraded is not None:
        line["degraded"] = degraded
    if psnr_vs_fp32 is not None:
        line["psnr_vs_fp32"] = psnr_vs_fp32
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        shard.barrier()
        torch.distributed.destroy_process_group()
    if not (timed_vs_eager <= 1e-5) or not timed_finite:
        print(f"[bench] rank {rank}: the timed output differs from the eager forward by {timed_vs_eager:.3e} (bound 1e-5) "
              f"or is not finite: the measurement is void", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
