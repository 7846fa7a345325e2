#!/bin/bash
# round 6: the sampler-side DCNv2 backward as two instantiations (data / weight) -- kernel trace, parity, the training line both ways
R=$PWD
mkdir -p $R/gpurun_out/r6l; rm -rf $R/gpurun_out/r6l/prof
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6l/prof -- python3 $R/tools/dbg/dcn_bwd_check.py > $R/gpurun_out/r6l/run.log 2>&1
cd $R
grep "sampler - columns\|backward via" gpurun_out/r6l/run.log | head -12
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r6l/prof/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:10]:
        print(r.get('Name', '')[:100], r.get('Calls'), r.get('AverageNs'), r.get('Percentage'))
PY
find gpurun_out/r6l/prof -name "*kernel_trace.csv" -delete
timeout 900 python -m pytest tests/test_hip_backward.py -x -q -m gpu > gpurun_out/r6l/tests.log 2>&1
tail -2 gpurun_out/r6l/tests.log
for mode in columns sampler columns sampler; do
  EAVSR_DCN_BWD=$mode timeout 600 python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r6l/train_$mode.json 2> gpurun_out/r6l/train_$mode.err
  python3 - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r6l/train_$mode.json').read().strip().splitlines()[-1])
    print('$mode:', round(d['ms_per_step'], 2), 'ms', {k: v for k, v in (d.get('step_breakdown_ms') or {}).items() if 'dcn' in k or 'il8' in k})
except Exception as e:
    print('$mode failed', e, open('gpurun_out/r6l/train_$mode.err').read()[-500:])
PY
done | tee gpurun_out/r6l/train.txt
