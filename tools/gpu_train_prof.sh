# rocprofv3 kernel stats of the training step (bench.py --mode train)
R=$PWD
rm -rf $R/gpurun_out/prof_train; mkdir -p $R/gpurun_out/prof_train
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -- python3 $R/bench.py --mode train --steps 2 --warmup 1 > $R/gpurun_out/prof_train/stdout.log 2>&1
cd $R
find gpurun_out/prof_train -name "*kernel_trace.csv" -size +30M -delete
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_train/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over 3 steps")
for r in rows[:28]:
    print(f"{float(r['Percentage']):6.2f}%  calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:100]}")
PY
tail -2 gpurun_out/prof_train/stdout.log | cut -c1-400
