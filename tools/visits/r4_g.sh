#!/bin/bash
# round 4, visit g: the whole GPU suite + the default bench line with eavsr_dcnv2_il2_f32 as the default DCNv2 schedule
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4g
timeout 1500 python3 -m pytest tests -q -x -m gpu > gpurun_out/r4g/pytest_gpu.log 2>&1
tail -4 gpurun_out/r4g/pytest_gpu.log
timeout 600 python3 bench.py --steps 10 --warmup 3 > gpurun_out/r4g/bench.log 2>&1
tail -1 gpurun_out/r4g/bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','timed_output_check') if k in d})
for k in d.get('kernels',[])[:12]: print(k)
"
