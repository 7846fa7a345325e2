#!/bin/bash
# round 3, visit o: rocprofv3 kernel trace + PMC passes of HEAD, the default bench line (with the CPU baseline), configs[2] / [4] lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3o
bash tools/gpu_profile.sh r03 > gpurun_out/r3o/profile.log 2>&1
cp gpurun_out/prof/summary.txt gpurun_out/r3o/rocprof_summary.txt
cp gpurun_out/prof/bench_kernel_stats.csv gpurun_out/r3o/bench_kernel_stats.csv
cp gpurun_out/prof/traffic_bench.json gpurun_out/r3o/traffic.json
cp gpurun_out/prof/traffic.json gpurun_out/r3o/traffic_raw.json
cp gpurun_out/prof/traffic_bench.json profiles/traffic.json
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r3o/bench_line.json 2> gpurun_out/r3o/bench_line.err
timeout 900 python bench.py --config 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3o/bench_line_config2_bf16.json 2> gpurun_out/r3o/c2.err
timeout 900 python bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3o/bench_line_config4_fp16.json 2> gpurun_out/r3o/c4.err
timeout 900 python bench.py --backbone-dtype bf16 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3o/bench_line_bf16.json 2> gpurun_out/r3o/bf16.err
tail -30 gpurun_out/r3o/profile.log
python - <<'PY'
import json
for f in ("bench_line", "bench_line_config2_bf16", "bench_line_config4_fp16", "bench_line_bf16"):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r3o/{f}.json') if l.startswith('{')][-1])
        print(f, d['ms_per_step'], d['value'], d['timed_output_max_abs_vs_eager'], d.get('roofline', {}).get('frac'), d.get('cpu_baseline', {}).get('value'), d.get('psnr_vs_fp32', {}).get('psnr_db') if d.get('psnr_vs_fp32') else None)
    except Exception as e:
        print(f, 'failed', e)
PY
