#!/bin/bash
# round 4, visit i: default bench line with the device-bound kernel-profile pass, offset statistics and sigma = 4 figures
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4i
timeout 900 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4i/bench.log 2>&1
echo "exit $?" >> gpurun_out/r4i/bench.log
tail -2 gpurun_out/r4i/bench.log | head -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step') if k in d})
print(json.dumps(d['kernels'][0], indent=1)[:2500])
print({k['kernel']:(round(k['avg_ms']*1e3,1), round(k['frac'],3)) for k in d['kernels']})
print('roofline', {k:d['roofline'][k] for k in ('frac','avg_ms')})
"
