# rocprofv3 PMC passes on the training step's two dominant kernels (tools/bench_kernels.py WHICH=train: 2 x 64 x 96 x 96 convolution,
# 7-segment 3x3 weight gradient)
R=$PWD
rm -rf $R/gpurun_out/prof_trk; mkdir -p $R/gpurun_out/prof_trk
cd /tmp && export TMPDIR=/tmp
export WHICH=train REPS=5
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/prof_trk/pmc_sq -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof_trk/pmc_sq2 -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/prof_trk/pmc_sq3 -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_trk/pmc_fetch -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_trk/pmc_write -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
cd $R
HOT="conv3x3_x6s_kernel;conv_wgrad3_x6_kernel;wgrad_reduce_kernel" \
PMC_TITLE="== PMC of the training step's dominant kernels (tools/bench_kernels.py WHICH=train: conv3x3_x6s_kernel at 2 x 64 x 96 x 96; conv_wgrad3_x6_kernel + wgrad_reduce_kernel over 7 segments of 2 x 64 x 96 x 96)" \
python3 tools/summarize_prof.py gpurun_out/prof_trk > gpurun_out/prof_trk/summary.txt 2>&1
cat gpurun_out/prof_trk/summary.txt
