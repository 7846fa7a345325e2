#!/bin/bash
# round 4, visit u: bench line fields (canonical launch; 16-bit roofline naming) + DCNv2 tests at HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4u
timeout 900 python3 -m pytest tests/test_hip_ops.py tests/test_hip_configs.py -q -x -m gpu -k "dcnv2 or launch_shape" > gpurun_out/r4u/pytest.log 2>&1
tail -2 gpurun_out/r4u/pytest.log
timeout 900 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r4u/bench.json 2> gpurun_out/r4u/bench.err
tail -1 gpurun_out/r4u/bench.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],2), {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if k in ('frac','avg_ms','canonical_launch','kernel')})
"
timeout 900 python3 bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4u/bench2.json 2> gpurun_out/r4u/bench2.err
tail -1 gpurun_out/r4u/bench2.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],2), d['share_of_step_in_16bit'], json.dumps(d['roofline'])[:900])
"
