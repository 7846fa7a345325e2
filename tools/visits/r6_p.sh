#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6p
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r6p/prof -o dcnb -- python3 tools/dbg/dcn_bwd_time.py > gpurun_out/r6p/time.txt 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r6p/prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']}%")
PY
timeout 900 python bench.py --mode train 2>&1 | tail -1 | tee gpurun_out/r6p/train_line.json | cut -c1-400
