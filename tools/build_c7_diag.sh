#!/bin/bash
# Diagnostic libraries eavsr_amd/lib/libc7_*.so: conv_x6.hip + capi.hip with -DEAVSR_C7_* flags (tools/gpu_conv7_diag.py).
#   VARIANTS='stamps:-DEAVSR_C7_STAMPS v_x:-DA,-DB' tools/build_c7_diag.sh
set -e
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -ffp-contract=fast -Iinclude -Ieavsr_amd/csrc -shared"
rm -f eavsr_amd/lib/libc7_*.so
for v in full: ${VARIANTS:-stamps:-DEAVSR_C7_STAMPS}; do
  name=${v%%:*}; flags=${v#*:}; flags=${flags//,/ }
  /opt/rocm/bin/hipcc $F $flags eavsr_amd/csrc/conv_x6.hip eavsr_amd/csrc/capi.hip -o eavsr_amd/lib/libc7_$name.so 2>/dev/null &
done
wait
ls eavsr_amd/lib/libc7_*.so
