#!/bin/bash
# full GPU suite on the product library, then on the lab library
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6t
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/r6t/default.txt
EAVSR_LIB_PATH=$PWD/eavsr_amd/lib/libeavsr_lab.so timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/r6t/lab.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r6t/smoke.txt
