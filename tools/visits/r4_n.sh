#!/bin/bash
# round 4, visit n: the 16-bit configurations at HEAD -- full kernel breakdown of one sub-batch forward (which kernels are still fp32)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4n
for c in 2 4; do
  EAVSR_BREAKDOWN_N=40 timeout 900 python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4n/bench_c$c.log 2>&1
  tail -1 gpurun_out/r4n/bench_c$c.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('config', $c, d['value'], d['ms_per_step'], d.get('share_of_step_in_16bit'), d.get('psnr_vs_fp32',{}).get('psnr_db'))
tot=sum(d['step_breakdown_ms'].values())
for k,v in d['step_breakdown_ms'].items(): print(f'   {k:32s} {v:8.2f} ms {100*v/tot:5.1f}%')
"
done
