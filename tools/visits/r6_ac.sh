#!/bin/bash
# round 6, last check at HEAD: the product suite, smoke, the default bench line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6ac
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/r6ac/default.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r6ac/smoke.txt
timeout 900 python bench.py > gpurun_out/r6ac/bench.json 2> gpurun_out/r6ac/bench.err; tail -c 400 gpurun_out/r6ac/bench.json
