#!/bin/bash
# Diagnostic libraries eavsr_amd/lib/libx6s_*.so: conv3_x6s.hip + conv_x6.hip (the weight pack) + capi.hip with -DEAVSR_X6S_* flags
# (tools/gpu_x6s_diag.py).   VARIANTS='stamps:-DEAVSR_X6S_STAMPS v_x:-DA,-DB' tools/build_x6s_diag.sh
set -e
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -ffp-contract=fast -fno-slp-vectorize -Iinclude -Ieavsr_amd/csrc -shared"
rm -f eavsr_amd/lib/libx6s_*.so
for v in full: ${VARIANTS:-stamps:-DEAVSR_X6S_STAMPS}; do
  name=${v%%:*}; flags=${v#*:}; flags=${flags//,/ }
  /opt/rocm/bin/hipcc $F $flags eavsr_amd/csrc/conv3_x6s.hip eavsr_amd/csrc/conv_x6.hip eavsr_amd/csrc/capi.hip -o eavsr_amd/lib/libx6s_$name.so 2>/dev/null &
done
wait
ls eavsr_amd/lib/libx6s_*.so
