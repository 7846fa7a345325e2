#!/bin/bash
# round 6: kernel trace of the sampler-side DCNv2 backward (which of its three kernels takes the time)
R=$PWD
mkdir -p $R/gpurun_out/r6k; rm -rf $R/gpurun_out/r6k/prof
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6k/prof -- python3 $R/tools/dbg/dcn_bwd_check.py > $R/gpurun_out/r6k/run.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r6k/prof/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:16]:
        print(r.get('Name', '')[:100], r.get('Calls'), r.get('AverageNs'), r.get('Percentage'))
PY
find gpurun_out/r6k/prof -name "*kernel_trace.csv" -delete
