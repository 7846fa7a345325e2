#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6q
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_hip_h16.py -m gpu -x -q -k "rcab_attention" 2>&1 | tail -4
export TMPDIR=/tmp
WHICH=rcab REPS=20 rocprofv3 --kernel-trace --stats -d gpurun_out/r6q/prof -o k -- python3 tools/bench_kernels.py > gpurun_out/r6q/k.log 2>&1
python3 - <<'PY'
import sqlite3, glob
for f in glob.glob("gpurun_out/r6q/prof/**/*.db", recursive=True):
    c=sqlite3.connect(f)
    tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
    for r in c.execute(f"select s.kernel_name, count(*), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 3 desc"): print(r[0][:90], r[1], round(r[2]/1e3,1))
PY
for c in 1 4 2; do
for s in 1 0; do
EAVSR_CA_PRE_SPLIT=$s timeout 900 python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --also '' 2>/dev/null | tail -1 | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('config $c split $s', round(d['ms_per_step'],2), d.get('timed_output_check',{}).get('bit_identical'), d.get('step_breakdown_ms'))"
done; done
