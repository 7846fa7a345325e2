#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p gpurun_out/r2e
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "il or pair or known_answers" > gpurun_out/r2e/il_tests.log 2>&1; echo "il rc=$?" > gpurun_out/r2e/rc.txt
timeout 300 python tools/gpu_il_ablate.py > gpurun_out/r2e/ablate.log 2>&1
cd /tmp && export TMPDIR=/tmp
export WHICH=dcn,dcnil REPS=3 EAVSR_DCN_MODE=native
rm -rf $R/gpurun_out/r2e/pmc_*
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/r2e/pmc_a -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r2e/pmc_*/*/*counter_collection.csv")):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        if "dcnv2_il" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][40:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, {c: f"{sum(x)/len(x):.4g}" for c, x in v.items()})
for d in sorted(glob.glob("gpurun_out/r2e/pmc_a/*/*kernel_trace.csv")):
    rows=[r for r in csv.DictReader(open(d)) if "dcnv2_il" in r["Kernel_Name"]]
    for r in rows[::3]:
        print(r["Kernel_Name"][40:70], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["VGPR_Count"], r.get("SGPR_Count"), r.get("Scratch_Size"))
PY
find gpurun_out/r2e -name "*.csv" -size +5M -delete
cat gpurun_out/r2e/rc.txt gpurun_out/r2e/ablate.log; tail -n 3 gpurun_out/r2e/il_tests.log
