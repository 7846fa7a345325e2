# rocprofv3 PMC passes on the 16-bit modes' kernels (tools/bench_kernels.py WHICH=h16: configs[2]'s sub-batch shape, 4 x 64 x 256 x 256 bf16)
R=$PWD
rm -rf $R/gpurun_out/prof_h16; mkdir -p $R/gpurun_out/prof_h16
cd /tmp && export TMPDIR=/tmp
export WHICH=h16 REPS=5
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/prof_h16/pmc_sq -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof_h16/pmc_sq2 -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_h16/pmc_fetch -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_h16/pmc_write -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
cd $R
HOT="conv3x3_c64_h16_kernel;conv3x3_h16g_kernel;conv5x5_c64_h16_kernel;conv3x3_c64to3_h16_kernel" \
PMC_TITLE="== PMC of the 16-bit modes' kernels (tools/bench_kernels.py WHICH=h16: 4 x 64 x 256 x 256 bf16 = one sub-batch of configs[2]; conv3x3_c64_h16_kernel averages the backbone launch and the four-slice pixel-shuffle launch; conv3x3_h16g_kernel averages 320 -> 64 and 128 -> 256)" \
python3 tools/summarize_prof.py gpurun_out/prof_h16 > gpurun_out/prof_h16/summary.txt 2>&1
cat gpurun_out/prof_h16/summary.txt
# the backbone launch ALONE (one launch shape per kernel name) and the one-launch RCAB convolutions
rm -rf $R/gpurun_out/prof_h16b; mkdir -p $R/gpurun_out/prof_h16b
cd /tmp
export WHICH=h16b REPS=5
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/prof_h16b/pmc_sq -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof_h16b/pmc_sq2 -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_h16b/pmc_fetch -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_h16b/pmc_write -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
cd $R
HOT="conv3x3_c64_h16_kernel;rcab_convs_h16_kernel" \
PMC_TITLE="== PMC of the 16-bit backbone launch ALONE (tools/bench_kernels.py WHICH=h16b: 4 x 64 x 256 x 256 bf16; conv3x3_c64_h16_kernel = the two launches of an RCAB, one shape; rcab_convs_h16_kernel = the same two convolutions as one launch)" \
python3 tools/summarize_prof.py gpurun_out/prof_h16b > gpurun_out/prof_h16b/summary.txt 2>&1
cat gpurun_out/prof_h16b/summary.txt
