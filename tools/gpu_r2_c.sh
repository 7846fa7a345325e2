#!/bin/bash
# DCNv2 IL kernel: ablations + PMC
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p gpurun_out/r2c
true
true
cd /tmp && export TMPDIR=/tmp
export WHICH=dcn,dcnil REPS=3 EAVSR_DCN_MODE=native
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/r2c/pmc_a -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r2c/pmc_b -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $R/gpurun_out/r2c/pmc_c -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r2c/pmc_*/*/*counter_collection.csv")):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        if "dcnv2" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(d.split("/")[2], k, {c: f"{sum(x)/len(x):.4g}" for c, x in v.items()})
for d in sorted(glob.glob("gpurun_out/r2c/pmc_a/*/*kernel_trace.csv")):
    rows=[r for r in csv.DictReader(open(d)) if "dcnv2" in r["Kernel_Name"]]
    for r in rows[::3]:
        print(r["Kernel_Name"][:70], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["VGPR_Count"], r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"))
PY
find gpurun_out/r2c -name "*.csv" -size +5M -delete
