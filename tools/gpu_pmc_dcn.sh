R=$PWD
rm -rf $R/gpurun_out/prof2; mkdir -p $R/gpurun_out/prof2
cd /tmp && export TMPDIR=/tmp
export WHICH=dcn REPS=3
for v in ${MODES:-native bf16x9}; do
  export EAVSR_DCN_MODE=$v
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/prof2/${v}_a -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof2/${v}_b -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/prof2/*/*/*counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "dcnv2" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d.split("/")[2], {k: f"{sum(v)/len(v):.4g}" for k,v in agg.items()})
for d in sorted(glob.glob("gpurun_out/prof2/*_a/*/*kernel_trace.csv")):
    rows=[r for r in csv.DictReader(open(d)) if "dcnv2" in r["Kernel_Name"]]
    print(d.split("/")[2], [ (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows], rows[0]["VGPR_Count"] if rows else None, rows[0].get("LDS_Block_Size"))
PY
