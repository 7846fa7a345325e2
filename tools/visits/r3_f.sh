#!/bin/bash
# round 3, visit f: lock-step kernel with the duty pair exempt from the weight DMA; bench --config 2 / 4 smoke
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3f
timeout 300 python tools/gpu_wino4_timeline.py > gpurun_out/r3f/timeline.log 2>&1
timeout 300 python tools/gpu_wino4_itstamp.py > gpurun_out/r3f/itstamp.log 2>&1
timeout 900 python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3f/bench_c2.json 2> gpurun_out/r3f/bench_c2.err
timeout 900 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3f/bench_c4.json 2> gpurun_out/r3f/bench_c4.err
grep -v amdgpu.ids gpurun_out/r3f/timeline.log | grep -v "^    \|^  \["
grep -v amdgpu.ids gpurun_out/r3f/itstamp.log | head -24
python - <<'PY'
import json
for c in ("c2", "c4"):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r3f/bench_{c}.json') if l.startswith('{')][-1])
        print(c, d['metric'], d['ms_per_step'], d['value'], d['timed_output_max_abs_vs_eager'], d.get('psnr_vs_fp32'), d.get('share_of_step_in_16bit'), d['config']['launch'])
    except Exception as e:
        print(c, 'failed', e)
PY
tail -3 gpurun_out/r3f/bench_c2.err gpurun_out/r3f/bench_c4.err
