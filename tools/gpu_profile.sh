# rocprofv3 passes: kernel-trace stats of bench.py, then PMC passes on the hot kernels
R=$PWD
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/bench_stdout.log 2>&1
echo "trace exit $?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/prof/pmc_sq -- python3 $R/tools/bench_kernels.py > $R/gpurun_out/prof/pmc_sq.log 2>&1
echo "pmc1 exit $?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof/pmc_sq2 -- python3 $R/tools/bench_kernels.py > $R/gpurun_out/prof/pmc_sq2.log 2>&1
echo "pmc2 exit $?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof/pmc_fetch -- python3 $R/tools/bench_kernels.py > $R/gpurun_out/prof/pmc_fetch.log 2>&1
echo "pmc3 exit $?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof/pmc_write -- python3 $R/tools/bench_kernels.py > $R/gpurun_out/prof/pmc_write.log 2>&1
echo "pmc4 exit $?"
cd $R
find gpurun_out/prof -name "*.csv" | head -30
du -sh gpurun_out/prof
# keep the merge small: drop the per-dispatch trace of the full bench (thousands of rows), keep stats
find gpurun_out/prof/bench -name "*kernel_trace.csv" -size +20M -delete
