#!/bin/bash
# Diagnostic libraries eavsr_amd/lib/libwg_*.so: conv_wgrad.hip + capi.hip with -DEAVSR_WX6_* flags (tools/gpu_wgrad_diag.py).
#   VARIANTS='v_prefetch:-DEAVSR_WX6_PREFETCH' tools/build_wgrad_diag.sh
set -e
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -ffp-contract=fast -fno-slp-vectorize -Iinclude -Ieavsr_amd/csrc -shared"
rm -f eavsr_amd/lib/libwg_*.so
for v in full: ${VARIANTS:-v_prefetch:-DEAVSR_WX6_PREFETCH}; do
  name=${v%%:*}; flags=${v#*:}; flags=${flags//,/ }
  /opt/rocm/bin/hipcc $F $flags eavsr_amd/csrc/conv_wgrad.hip eavsr_amd/csrc/capi.hip -o eavsr_amd/lib/libwg_$name.so 2>/dev/null &
done
wait
ls eavsr_amd/lib/libwg_*.so
