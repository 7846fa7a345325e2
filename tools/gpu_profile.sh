# rocprofv3 passes: kernel-trace stats of bench.py, then PMC passes on the hot kernels
R=$PWD
TAG=${1:-r02}
rm -rf $R/gpurun_out/prof; mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --also '' > $R/gpurun_out/prof/bench_stdout.log 2>&1
echo "trace exit $?"
export WHICH=conv,rcab,dcn,dcnil,warp REPS=5 SIGMA=0.5
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/prof/pmc_sq -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof/pmc_sq2 -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/prof/pmc_sq3 -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof/pmc_fetch -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof/pmc_write -- python3 $R/tools/bench_kernels.py > /dev/null 2>&1
cd $R
find gpurun_out/prof/bench -name "*kernel_trace.csv" -size +30M -delete
cp $(ls -t gpurun_out/prof/bench/*/*kernel_stats.csv | head -1) gpurun_out/prof/bench_kernel_stats.csv   # THIS visit's (gpurun_out keeps older runs)
python3 tools/summarize_prof.py gpurun_out/prof > gpurun_out/prof/summary.txt 2>&1
cat gpurun_out/prof/summary.txt | head -60
# bench.py reads profiles/traffic.json (bench kernel names -> bytes per launch): regenerate it from this visit's PMC passes
python3 - <<PY
import json
r = json.load(open('gpurun_out/prof/traffic.json'))
g = lambda k: r.get(k, {}).get('total_bytes')
out = {'conv3x3_64to64_wino4': g('conv_wino6_kernel<3, false, true, false>'), 'conv5x5_64to120_wino': g('conv_wino6_kernel<5, false'), 'conv5x5_64to120_x6': g('conv_x6_kernel<5'), 'conv3x3_64to64_wino': g('conv3x3_wino_kernel'), 'conv3x3_64to64': g('conv2d_mfma_kernel'), 'dcnv2': g('dcnv2_grp_kernel'), 'dcnv2_il_heads': g('dcnv2_il2_kernel<6, 2>') or g('dcnv2_il2_kernel<6, 1>') or g('dcnv2_il2_kernel<6, true>') or g('dcnv2_il_kernel<6, true>'), 'dcnv2_il': g('dcnv2_il2_kernel<6, 0>') or g('dcnv2_il2_kernel<6, false>') or g('dcnv2_il_kernel<6, false>'), 'dcnv2_il_heads_round2_kernel': g('dcnv2_il_kernel<6, true>'), 'flow_warp': g('flow_warp_kernel'), 'flow_warp_pair': g('flow_warp_pair_kernel'),
       '_taken': __import__('datetime').date.today().isoformat() + ' (' + '${TAG}' + ', commit ' + '${COMMIT:-unknown}' + ')',
       'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2 on gfx950; bytes per launch at 2x64x180x320 (one sub-batch of the default bench.py --streams 2)'}
json.dump(out, open('profiles/traffic.json', 'w'), indent=1)
json.dump(out, open('gpurun_out/prof/traffic_bench.json', 'w'), indent=1)
PY
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err
tail -c 1200 gpurun_out/bench_${TAG}.json
