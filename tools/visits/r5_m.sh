#!/bin/bash
# round 5: kernel census of the configs[4] / configs[2] bench children (what is NOT a library kernel there?)
cd "$(dirname "$0")/../.."
R=$PWD
mkdir -p gpurun_out
for cfg in 4 2; do
  rm -rf $R/gpurun_out/prof_c$cfg; mkdir -p $R/gpurun_out/prof_c$cfg
  (cd /tmp && TMPDIR=/tmp timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c$cfg -- python3 $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --also '' > $R/gpurun_out/prof_c$cfg/stdout.log 2>&1)
  find gpurun_out/prof_c$cfg -name "*kernel_trace.csv" -size +30M -delete
  python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_c$cfg/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows if "spin_kernel" not in r["Name"])
print(f"config $cfg: total non-spin kernel time {tot/1e6:.1f} ms")
for r in rows[:32]:
    if "spin_kernel" in r["Name"]: continue
    print(f"{100*float(r['TotalDurationNs'])/tot:6.2f}%  calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}")
PY
done
