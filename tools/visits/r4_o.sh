#!/bin/bash
# round 4, visit o: masks activated by the heads convolution (heads = 2): tests, model-level tests, bench DCN line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4o
timeout 1200 python3 -m pytest tests/test_hip_ops.py tests/test_hip_model.py -q -x -m gpu -k "dcnv2 or conv5x5 or conv7x7 or golden or adstn or MultiAd or heads" > gpurun_out/r4o/pytest.log 2>&1
tail -4 gpurun_out/r4o/pytest.log
for sg in 1 0; do
EAVSR_HEADS_SIGMOID=$sg timeout 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r4o/bench_$sg.log 2>&1
tail -1 gpurun_out/r4o/bench_$sg.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k={e['kernel']:e for e in d['kernels']}
print('heads sigmoid $sg:', round(d['ms_per_step'],2), 'ms; dcn', round(k['dcnv2_il_heads']['avg_ms']*1e3,1), 'us frac', round(k['dcnv2_il_heads']['frac'],3), '; heads conv', round(k['conv5x5_64to120_x6']['avg_ms']*1e3,1), 'us; sigma0.5', round(k['dcnv2_il_heads']['synthetic_offsets']['sigma_0.5']['avg_ms']*1e3,1), d['timed_output_check']['bit_identical'])
"
done
