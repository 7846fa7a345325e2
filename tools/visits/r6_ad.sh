#!/bin/bash
# round 6: configs[2] as ONE graph over all 8 clips instead of two 4-clip graphs on two streams (in the 16-bit modes nothing of another
# stream runs beside the persistent convolution, and its per-tile cost falls with the tiles per workgroup)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6ad
for v in "--streams 2" "--streams 1 --graph" "--streams 2" "--streams 1 --graph" "--streams 4"; do
timeout 900 python bench.py --config 2 $v --steps 4 --warmup 1 --no-cpu-baseline --also '' 2>/dev/null | tail -1 | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('config 2 $v', round(d['ms_per_step'],2), d.get('timed_output_check',{}).get('bit_identical'), d.get('psnr_vs_fp32',{}).get('psnr_db'), d['config'].get('launch'))"
done
