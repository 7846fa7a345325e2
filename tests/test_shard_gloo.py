"""The N>1 path on CPU: two processes, gloo backend, the same shard / barrier / max-over-ranks code
bench.py uses on RCCL.  No GPU needed."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from eavsr_amd import shard
    r, lr, w = shard.init_process_group("gloo")
    assert (r, w) == (rank, world)
    clips = torch.arange(5 * 2, dtype=torch.float32).view(5, 2, 1, 1, 1)  # 5 "clips" of 2 "frames"
    mine = shard.shard_clips(clips, r, w)
    shard.barrier()
    job_time = shard.max_over_ranks(1.0 + r)          # slowest rank defines the job time
    frames = shard.sum_over_ranks(float(mine.shape[0] * mine.shape[1]))
    out = shard.gather_outputs(mine * 2.0, 5, r, w)   # DataParallel-style gather of per-clip results
    q.put((rank, [int(v) for v in mine[:, 0].flatten()], job_time, frames,
           None if out is None else out.flatten().tolist()))
    shard.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_clip_sharding_over_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, own0, t0, f0, out0), (r1, own1, t1, f1, out1) = res
    assert own0 == [0, 4, 8] and own1 == [2, 6]         # clips 0,2,4 / 1,3 (first frame ids)
    assert t0 == t1 == 2.0                               # MAX over ranks
    assert f0 == f1 == 10.0                              # every frame counted exactly once
    assert out0 == [2.0 * v for v in range(10)] and out1 is None


def test_clip_indices_partition_every_clip_once():
    from eavsr_amd.shard import clip_indices
    for n in (0, 1, 7, 8, 33):
        for world in (1, 2, 4, 8):
            seen = sorted(i for r in range(world) for i in clip_indices(n, r, world))
            assert seen == list(range(n))
            sizes = [len(clip_indices(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        clip_indices(4, 2, 2)


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from eavsr_amd import shard
    shard.init_process_group("gloo")
    torch.manual_seed(0)                       # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 4))
    net[3].weight.requires_grad_(False)        # a frozen tensor, like SPyNet
    sync = shard.GradientAllReducer(net.parameters(), bucket_bytes=1024)   # several small buckets
    g = torch.Generator().manual_seed(100 + rank)                          # different data per rank
    for step in range(2):
        x = torch.randn(5, 16, generator=g)
        net.zero_grad(set_to_none=True)
        net(x).pow(2).mean().backward()        # hooks launch the bucket all-reduces during backward
        sync.finish()
    # plain nested lists, not tensors: a tensor travels through the queue as a shared-memory handle that dies with
    # this process (a race with the parent's q.get)
    q.put((rank, [p.grad.tolist() if p.grad is not None else None for p in net.parameters()], len(sync.buckets)))
    shard.barrier()
    torch.distributed.destroy_process_group()


def test_gradient_allreduce_averages_over_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, g0, nb0), (_, g1, nb1) = res
    g0 = [None if a is None else torch.tensor(a) for a in g0]
    g1 = [None if a is None else torch.tensor(a) for a in g1]
    assert nb0 == nb1 and nb0 > 1
    # every rank ends with the same (averaged) gradients ...
    for a, b in zip(g0, g1):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
    # ... equal to the mean of the per-rank gradients computed independently
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 4))
    net[3].weight.requires_grad_(False)
    per_rank = []
    for r in range(world):
        g = torch.Generator().manual_seed(100 + r)
        for step in range(2):
            x = torch.randn(5, 16, generator=g)
            net.zero_grad(set_to_none=True)
            net(x).pow(2).mean().backward()
        per_rank.append([p.grad.clone() if p.grad is not None else None for p in net.parameters()])
    for a, p0, p1 in zip(g0, per_rank[0], per_rank[1]):
        if a is None:
            assert p0 is None
        else:
            assert torch.allclose(a, (p0 + p1) / 2, atol=1e-7)


# ---- bench.py --gpus N starts its own ranks (VERDICT r1 / ADVICE r1) ---------------------------------------------------
def _run_bench(*argv, env=None):
    import json
    import subprocess
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=e,
                       timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_gpus_2_launches_its_own_ranks_and_reports_the_max_over_ranks():
    """`python bench.py --gpus 2` with no launcher environment: the parent (which never touches the GPU) starts two
    ranks through torch.distributed.run, they rendezvous on 127.0.0.1, and rank 0's single JSON line comes back through
    the parent.  --dry: sleeps instead of kernels, gloo instead of RCCL -- the plumbing around the kernels is the same
    code path as the measured run."""
    r, lines = _run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["config"]["backend"] == "gloo"
    assert line["dry"] is True and "DRY RUN" in line["metric"]
    # rank r sleeps 10 ms * (1 + r) per step: the job time is the slower rank's
    assert line["ms_per_step"] >= 19.0, line
    assert abs(line["value"] - 2 * 4 * 7 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]   # whole-job aggregate


def test_bench_single_rank_line_is_unchanged_by_the_launcher_and_child_failures_propagate():
    r, lines = _run_bench("--steps", "2", "--dry")
    assert r.returncode == 0 and len(lines) == 1 and lines[0]["n_gpus"] == 1
    # a launcher that started a different number of ranks than --gpus is an error, not a silently different job
    r, lines = _run_bench("--gpus", "2", "--dry", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and not lines
    # no GPU here: the real (non-dry) 2-rank job must fail loudly, through the parent, with a non-zero status
    r, lines = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                          env={"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert r.returncode != 0 and not lines


def _worker8(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    torch.set_num_threads(1)
    from eavsr_amd import shard
    r, _, w = shard.init_process_group("gloo")
    mine = shard.clip_indices(32, r, w)
    cap = shard.cap_host_threads(w)
    shard.barrier()
    times = shard.all_ranks(10.0 + r)                 # per-rank step times, rank order, on every rank
    names = shard.all_ranks_str(f"cpu:{r}")
    total = shard.sum_over_ranks(float(len(mine)))
    q.put((rank, mine, times, names, total, cap, torch.get_num_threads()))
    shard.barrier()
    torch.distributed.destroy_process_group()


def test_eight_ranks_partition_32_clips_and_report_per_rank():
    """the node's shape (one process per GPU, 8 per node): 32 clips over 8 ranks, 4 each, every clip exactly once;
    per-rank times / device names arrive in rank order on every rank; the host-thread cap is cores // 8"""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = sorted(i for _, mine, *_ in res for i in mine)
    assert owned == list(range(32)) and all(len(mine) == 4 for _, mine, *_ in res)
    for rank, mine, times, names, total, cap, nthreads in res:
        assert mine == list(range(rank, 32, 8))
        assert times == [10.0 + r for r in range(8)] and names == [f"cpu:{r}" for r in range(8)]
        assert total == 32.0 and cap >= 1 and nthreads <= cap


def test_bench_a_rank_that_dies_takes_the_job_down_within_a_timeout():
    """`bench.py --gpus 2` where rank 1 exits before the first barrier: the launcher tears the other rank down and the parent
    returns a non-zero status with no JSON line -- it does not hang in the barrier"""
    import time as _t
    t0 = _t.time()
    r, lines = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry", env={"EAVSR_DRY_DIE_RANK": "1"})
    assert r.returncode != 0 and not lines, (r.returncode, r.stdout[-500:])
    assert _t.time() - t0 < 240


def test_bench_dry_line_carries_per_rank_fields():
    r, lines = _run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry")
    assert r.returncode == 0 and len(lines) == 1
    line = lines[0]
    assert len(line["per_rank_ms"]) == 2 and line["per_rank_ms"][1] > line["per_rank_ms"][0] >= 9.0      # rank r sleeps 10 (1 + r) ms
    assert line["per_rank_device"] == ["cpu:0", "cpu:1"] and line["host_threads_per_rank"] >= 1
    assert line["rccl_ranks_seen"] == 2      # counted by an all-reduce on the collective backend itself (gloo here, RCCL on GPUs)
    assert abs(max(line["per_rank_ms"]) - line["ms_per_step"]) < 5.0


def _bcast_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from eavsr_amd import shard
    shard.init_process_group("gloo")
    torch.manual_seed(1000 + rank)             # DIFFERENT replicas on purpose (a checkpoint loaded on rank 0 only looks like this)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Linear(8, 4))
    net[1].running_mean.fill_(float(rank))     # buffers travel too, also the integer one (num_batches_tracked)
    net[1].num_batches_tracked.fill_(7 + rank)
    before = [v.clone() for v in net.state_dict().values()]
    nbytes = shard.broadcast_module(net, src=0, bucket_bytes=256)      # several buckets
    after = list(net.state_dict().values())
    unchanged = all(torch.equal(a, b) for a, b in zip(before, after))
    seen = shard.ranks_seen()
    q.put((rank, [v.tolist() for v in after], nbytes, unchanged, seen))
    shard.barrier()
    torch.distributed.destroy_process_group()


def test_parameters_are_broadcast_from_rank_0_once():
    """ranks agreed only because every rank seeded identically.  Two ranks with different seeds end with rank
    0's parameters AND buffers after `broadcast_module` (the call `EAVSRPModel` makes at construction and after
    `load_networks`); rank 0 itself is unchanged; `ranks_seen` counts the ranks over the collective backend itself."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bcast_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, sd0, nb0, same0, seen0), (_, sd1, nb1, same1, seen1) = res
    assert sd0 == sd1                              # identical replicas afterwards
    assert same0 and not same1                     # ... and they are rank 0's
    assert nb0 == nb1 > 0 and seen0 == seen1 == 2
    torch.manual_seed(1000)
    want = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Linear(8, 4))
    assert sd0[0] == want[0].weight.tolist()
    assert sd1[-2] == want[2].weight.tolist() and sd1[4] == [0.0] * 8 and sd1[6] == 7      # running_mean / num_batches_tracked of rank 0


def _load_worker(rank, world, port, q, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from argparse import Namespace
    from eavsr_amd import shard
    from eavsr_amd.eavsrp_model import EAVSRPModel
    shard.init_process_group("gloo")
    assert shard.all_ranks_ok(True) is True and shard.all_ranks_ok(rank == 0) is False
    torch.manual_seed(7 + rank)
    m = object.__new__(EAVSRPModel)      # load_networks needs a network, options and the broadcast -- not a GPU
    m.netEAVSRP = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Linear(4, 2))
    m.opt = Namespace(load_path="")
    good = os.path.join(tmp, "good.pth")
    if rank == 0:
        torch.manual_seed(99)
        ref = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Linear(4, 2))
        torch.save({"state_dict": ref.state_dict()}, good)
        torch.save({"state_dict": {k: v for k, v in ref.state_dict().items() if k != "1.bias"}}, os.path.join(tmp, "bad_1.pth"))
    shard.barrier()
    # (1) every rank reads a good file: identical replicas = the file's
    m.load_networks(good)
    ok_sd = [v.tolist() for v in m.netEAVSRP.state_dict().values()]
    # (2) rank 1's file is broken (a key short), rank 0's is fine: BOTH ranks raise, nobody hangs in the broadcast
    raised = None
    try:
        m.load_networks(good if rank == 0 else os.path.join(tmp, "bad_1.pth"))
    except RuntimeError as e:
        raised = str(e)
    # (3) the group is still usable and in step afterwards
    seen = shard.ranks_seen()
    q.put((rank, ok_sd, raised, seen))
    shard.barrier()
    torch.distributed.destroy_process_group()


def test_load_networks_is_a_collective_and_a_failing_rank_aborts_every_rank(tmp_path):
    """`load_networks` ends in the start-up broadcast; a rank that failed to read its file used to leave the others
    blocked in it (or paired with their next collective).  Now the ranks agree on success first (`shard.all_ranks_ok`)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_load_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, sd0, r0, seen0), (_, sd1, r1, seen1) = res
    assert sd0 == sd1
    assert r0 is not None and "another rank" in r0          # rank 0 loaded fine and still aborts
    assert r1 is not None and "1.bias" in r1                # rank 1 reports its own error
    assert seen0 == seen1 == 2


class _StubGraph:
    """what a replay of the captured forward + backward leaves behind: gradients in the bound `.grad` tensors"""

    def __init__(self, fn):
        self.fn, self.replays = fn, 0

    def replay(self):
        self.replays += 1
        self.fn()


def _graph_step_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from argparse import Namespace
    from eavsr_amd import shard
    from eavsr_amd.graph import GraphedTrainStep
    shard.init_process_group("gloo")
    torch.manual_seed(50 + rank)               # different replicas; the broadcast below makes them one
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
    shard.broadcast_module(net)
    opt = torch.optim.Adam([{"params": net[0].parameters(), "lr": 1e-2}, {"params": net[2].parameters(), "lr": 1e-3}])
    model = Namespace(netEAVSRP=net, optimizer_EAVSRP=opt, grad_sync=shard.GradientAllReducer(net.parameters(), bucket_bytes=64),
                      data_lr_seq=None, data_hr_seq=None)
    g = torch.Generator().manual_seed(900 + rank)      # every rank its own data
    st = object.__new__(GraphedTrainStep)              # the capture needs a GPU; the control flow around a replay does not
    st.model, st.world = model, world
    st.static_lr, st.static_hr = torch.zeros(4, 6), torch.zeros(4, 2)
    grads = [(p, torch.zeros_like(p)) for p in net.parameters()]
    st._grads = grads

    def replayed():      # forward + backward of the "captured" step, writing into the captured gradient tensors
        out = net(st.static_lr)
        gs = torch.autograd.grad((out - st.static_hr).abs().mean(), list(net.parameters()))
        for (_, buf), gi in zip(grads, gs):
            buf.copy_(gi)
    st.graph = _StubGraph(replayed)
    for _ in range(3):
        # an eager step in between rebinds .grad (zero_grad(set_to_none=True)): step() must bind the captured tensors back
        for p in net.parameters():
            p.grad = None
        st.step({"lr_seq": torch.randn(4, 6, generator=g), "hr_seq": torch.randn(4, 2, generator=g)})
    q.put((rank, [p.detach().tolist() for p in net.parameters()], st.graph.replays,
           all(p.grad is buf for p, buf in grads)))
    shard.barrier()
    torch.distributed.destroy_process_group()


def test_graphed_train_step_world_2_keeps_the_replicas_identical():
    """`GraphedTrainStep.step()` with world size 2 (eavsr_amd/graph.py): the graph ends after backward, the bucketed all-reduce
    and Adam run eagerly behind each replay.  The capture itself needs a GPU, so the graph is a stub that does what a replay
    does (gradients written into the captured `.grad` tensors); everything around it is the product's code.  After 3 steps on
    DIFFERENT data both ranks hold the same parameters, equal to a single-process run on the averaged gradients."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_graph_step_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0, n0, bound0), (_, p1, n1, bound1) = res
    assert n0 == n1 == 3 and bound0 and bound1
    assert p0 == p1                                   # bit-identical replicas after three data-parallel steps
    # single-process restatement: mean of the two ranks' gradients, same Adam
    torch.manual_seed(50)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
    opt = torch.optim.Adam([{"params": net[0].parameters(), "lr": 1e-2}, {"params": net[2].parameters(), "lr": 1e-3}])
    gens = [torch.Generator().manual_seed(900 + r) for r in range(world)]
    for _ in range(3):
        acc = [torch.zeros_like(p) for p in net.parameters()]
        for g in gens:
            x, y = torch.randn(4, 6, generator=g), torch.randn(4, 2, generator=g)
            gs = torch.autograd.grad((net(x) - y).abs().mean(), list(net.parameters()))
            for a, gi in zip(acc, gs):
                a.add_(gi)
        for p, a in zip(net.parameters(), acc):
            p.grad = a / world
        opt.step()
    for got, want in zip(p0, net.parameters()):
        assert torch.allclose(torch.tensor(got), want.detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_bench_gpus_2_over_rccl_when_two_devices_are_visible():
    """The real N > 1 path: `python bench.py --gpus 2` starts two ranks itself, one per GPU, over RCCL (backend 'nccl'); also
    the data-parallel training step with its gradient all-reduce.  Skipped on a single-GPU box (the round's test box has one
    GPU: there the 2-rank code is covered by the gloo tests above and by tests/test_hip_backward.py's shared-GPU test)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    r, lines = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-profile")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["world_size"] == 2
    assert lines[0]["backend"].startswith("nccl"), lines[0]["backend"]
    assert lines[0]["value"] > 0 and lines[0]["scaling"] == "weak"
    r, lines = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--mode", "train")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["loss"]["EAVSRP_L1"] == lines[0]["loss"]["EAVSRP_L1"]
