#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py: over the last timed step, how long is a CU-owning kernel (the Winograd / bf16x6
convolutions, DCNv2: 120-157 KB of LDS, one workgroup per CU) running, how long only co-resident kernels, how long nothing; and where
the gaps are (start / end of the step, between the streams).   usage: trace_occupancy.py <dir with *kernel_trace.csv>
CAVEAT (measured, tools/visits/r3_ai.sh): under `rocprofv3 --kernel-trace` the dispatches of the two streams do not overlap -- the traced
step is 279 ms against 222 ms untraced, every kernel shows its stand-alone duration -- so this is the SERIALISED budget of a step
(CU-owning kernels 211 ms + co-resident kernels 68 ms per step at the end of round 3), not the real two-stream schedule."""
import csv
import glob
import os
import sys

paths = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for p in paths:
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
HOG = ("conv_wino6_kernel", "conv_x6_kernel", "dcnv2_il_kernel", "conv3x3_wino_kernel", "conv2d_mfma_kernel<7", "conv2d_mfma_kernel<3")
is_hog = lambda n: any(h in n for h in HOG)
# steps: the bench replays two graphs per step; split the trace at gaps > 200 us with no kernel at all
t0 = rows[0][0]
segs, cur_s, cur_e = [], rows[0][0], rows[0][1]
for s, e, n, q in rows[1:]:
    if s > cur_e + 200_000:
        segs.append((cur_s, cur_e))
        cur_s = s
    cur_e = max(cur_e, e)
segs.append((cur_s, cur_e))
print(f"{len(rows)} kernels, {len(segs)} busy segments (split at idle gaps > 200 us); the longest ones (ms):",
      [round((e - s) / 1e6, 2) for s, e in sorted(segs, key=lambda x: x[0] - x[1])[:6]])


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    out = []
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            out.append((cs, ce))
            cs, ce = s, e
    if cs is not None:
        out.append((cs, ce))
    return out


# every segment over 50 ms, in time order: the streamed steps are the ones with two queues of (almost) equal kernel counts
for s0, e0 in [x for x in segs if x[1] - x[0] > 50_000_000]:
    ks = [r for r in rows if r[0] >= s0 and r[1] <= e0]
    hog = union([(s, e) for s, e, n, q in ks if is_hog(n)])
    anyk = union([(s, e) for s, e, n, q in ks])
    th = sum(e - s for s, e in hog)
    ta = sum(e - s for s, e in anyk)
    span = e0 - s0
    print(f"segment of {span / 1e6:.2f} ms: a CU-owning kernel runs {th / 1e6:.2f} ms ({100 * th / span:.1f} %), only co-resident kernels "
          f"{(ta - th) / 1e6:.2f} ms, nothing {(span - ta) / 1e6:.2f} ms; {len(ks)} kernels")
    # gaps between hog intervals, by size
    gaps = [(hog[i + 1][0] - hog[i][1], hog[i][1] - s0) for i in range(len(hog) - 1)]
    big = sorted(gaps, reverse=True)[:8]
    print("   largest hog-free gaps (us @ ms into the segment):", [(round(g / 1e3, 1), round(at / 1e6, 1)) for g, at in big])
    print(f"   hog-free gaps: {len([g for g, _ in gaps if g > 20_000])} over 20 us, sum of all {sum(g for g, _ in gaps) / 1e6:.2f} ms; "
          f"head {(hog[0][0] - s0) / 1e3:.0f} us, tail {(e0 - hog[-1][1]) / 1e3:.0f} us")
    # per queue / stream: first and last kernel
    qs = {}
    for s, e, n, q in ks:
        a = qs.setdefault(q, [s, e, 0])
        a[0] = min(a[0], s); a[1] = max(a[1], e); a[2] += 1
    for q, (s, e, c) in sorted(qs.items()):
        print(f"   queue/stream {q}: {c} kernels, from +{(s - s0) / 1e3:.0f} us to +{(e - s0) / 1e3:.0f} us")

# a window of the longest segment as a timeline (DUMP_AT=<ms into the segment>, DUMP_US=<length>)
at = float(os.environ.get("DUMP_AT", "150")) * 1e6
ln = float(os.environ.get("DUMP_US", "700")) * 1e3
two_q = [x for x in segs if x[1] - x[0] > 50_000_000 and len({r[3] for r in rows if r[0] >= x[0] and r[1] <= x[1]}) >= 2]
s0, e0 = two_q[-1] if two_q else sorted(segs, key=lambda x: x[0] - x[1])[0]      # the last segment that two queues share
short = lambda n: n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:34]
print(f"--- timeline from +{at / 1e6:.1f} ms, {ln / 1e3:.0f} us (start us, duration us, queue, kernel; * = CU-owning)")
for s, e, n, q in rows:
    if s >= s0 + at and s < s0 + at + ln:
        print(f"{(s - s0 - at) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  q{q} {'*' if is_hog(n) else ' '} {short(n)}")
