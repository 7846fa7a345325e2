#!/usr/bin/env python3
"""Condense rocprofv3 output dirs (kernel stats + PMC counter CSVs) into a text summary + traffic.json."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
out = {}
stats = glob.glob(os.path.join(root, "bench", "*", "*kernel_stats.csv"))
if stats:
    print("== rocprofv3 --kernel-trace --stats of `bench.py --steps 2 --warmup 1` (top kernels)")
    rows = list(csv.DictReader(open(stats[0])))
    for r in rows[:16]:
        print(f"{float(r['Percentage']):6.2f}%  calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:10.1f} us  {r['Name'][:110]}")
print()
hot = ("conv_wino6_kernel<3, false, true, false>", "conv_wino6_kernel<3, false, true, true>", "ca_scale_pre_kernel<0>", "conv_wino6_kernel<5, false", "conv_x6_kernel<5", "conv_x6_kernel<7", "conv3x3_wino_kernel", "conv2d_mfma_kernel", "dcnv2_grp_kernel", "dcnv2_il_kernel<6, true>", "dcnv2_il_kernel<6, false>", "dcnv2_il2_kernel<6, 1>", "dcnv2_il2_kernel<6, 0>", "dcnv2_il2_kernel<6, 2>", "flow_warp_kernel", "flow_warp_pair_kernel")
if os.environ.get("HOT"):      # another set of kernels (tools/gpu_profile_h16.sh): substrings of the kernel names, ';'-separated
    hot = tuple(os.environ["HOT"].split(";"))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(root, "pmc_*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(d)):
        for hname in hot:
            if hname in r["Kernel_Name"]:
                agg[hname][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(root, "pmc_sq", "*", "*kernel_trace.csv"))):
    for r in csv.DictReader(open(d)):
        for hname in hot:
            if hname in r["Kernel_Name"]:
                dur[hname].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(os.environ.get("PMC_TITLE", "== PMC (tools/bench_kernels.py: 2 x 64 x 180 x 320 fp32 = one sub-batch of the default bench; one counter group per pass)"))
traffic = {}
for hname in hot:
    c = {k: sum(v) / len(v) for k, v in agg[hname].items()}
    if not c:
        continue
    t = sum(dur[hname]) / max(1, len(dur[hname]))
    print(f"{hname}: avg duration under the profiler {t:.1f} us")
    for k, v in sorted(c.items()):
        print(f"    {k:30s} {v:.4g}")
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        print(f"    -> shader clock ~ {cyc / t / 1e3:.2f} GHz; MFMA pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.1%} of SIMD-cycles")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # rocprofv3 reports KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) streams:
        # doubled here as MI355X_MICROARCH.md (HBM section) prescribes; WRITE_SIZE is exact.
        fetch = c["FETCH_SIZE"] * 1024 * 2
        write = c["WRITE_SIZE"] * 1024
        traffic[hname] = {"fetch_bytes_corrected": fetch, "write_bytes": write, "total_bytes": fetch + write,
                          "note": "per launch; FETCH_SIZE x2 (gfx950 wide-read correction), WRITE_SIZE as is"}
        print(f"    -> HBM traffic per launch: fetch {fetch / 1e6:.1f} MB (x2 corrected) + write {write / 1e6:.1f} MB")
json.dump(traffic, open(os.path.join(root, "traffic.json"), "w"), indent=1)
