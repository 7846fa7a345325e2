#!/usr/bin/env python3
"""When do the two sub-batch streams of a bench step finish?  Events at the end of each stream's graph: the time one stream runs
alone at the end of every step (and would not with steps pipelined across the join) is the gap between them."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from eavsr_amd.graph import StreamedForward  # noqa: E402
from eavsr_amd.utils.synthetic import synthetic_clip  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
net, _ = bench.build_model(dev, "trained_like")
clips = synthetic_clip(4, 7, 180, 320, seed=0).to(dev)
run = StreamedForward(net, clips, groups=2)
cur = torch.cuda.current_stream(dev)
with torch.no_grad():
    for _ in range(2):
        run(clips)
    torch.cuda.synchronize()
    for step in range(6):
        e0 = torch.cuda.Event(enable_timing=True)
        ends = [torch.cuda.Event(enable_timing=True) for _ in run.streams]
        e0.record(cur)
        for g, (part, st) in enumerate(zip(run.parts, run.streams)):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                part.static_in.copy_(clips[g * run.per:(g + 1) * run.per])
                part.graph.replay()
                ends[g].record(st)
        for st in run.streams:
            cur.wait_stream(st)
        torch.cuda.synchronize()
        t = [e0.elapsed_time(e) for e in ends]
        print(f"step {step}: stream 0 done at {t[0]:7.2f} ms, stream 1 at {t[1]:7.2f} ms, one stream alone for {abs(t[0] - t[1]):5.2f} ms", flush=True)
