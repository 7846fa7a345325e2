#!/bin/bash
# round 4, visit q: the whole GPU suite, smoke(), the default bench line, a 2-rank bench on the one GPU (per-rank fields), train bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4q
timeout 2400 python3 -m pytest tests -q -x -m gpu > gpurun_out/r4q/pytest_gpu.log 2>&1
tail -3 gpurun_out/r4q/pytest_gpu.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4q/smoke.log 2>&1; tail -2 gpurun_out/r4q/smoke.log
timeout 900 python3 bench.py > gpurun_out/r4q/bench.log 2> gpurun_out/r4q/bench.err
tail -1 gpurun_out/r4q/bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('value','ms_per_step','per_rank_ms','per_rank_device','host_threads_per_rank')})
print('dcn', d['kernels'][0]['avg_ms'], d['kernels'][0]['frac'], 'roofline', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'])
"
EAVSR_DIST_BACKEND=gloo timeout 900 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-profile > gpurun_out/r4q/bench2.log 2> gpurun_out/r4q/bench2.err
tail -1 gpurun_out/r4q/bench2.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('2 ranks on one GPU:', {k:d.get(k) for k in ('value','n_gpus','ms_per_step','per_rank_ms','per_rank_device','backend')})
"
timeout 900 python3 bench.py --mode train --steps 5 --warmup 2 > gpurun_out/r4q/train.log 2> gpurun_out/r4q/train.err
tail -1 gpurun_out/r4q/train.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('train:', d['value'], d['ms_per_step'], d['config']['launch'], d['loss'])
"
