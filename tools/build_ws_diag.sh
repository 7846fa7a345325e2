#!/bin/bash
# Diagnostic libraries eavsr_amd/lib/libws_*.so: dcnv2_ws.hip + capi.hip with -DEAVSR_WS_EXP_* (timing ablations, results
# wrong by construction).  Built here (hipcc cross-compiles); they travel to the GPU box with the snapshot.
set -e
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -ffp-contract=fast -Iinclude -Ieavsr_amd/csrc -shared"
rm -f eavsr_amd/lib/libws_*.so
for v in full:"" nodma:-DEAVSR_WS_EXP_NO_DMA nomfma:-DEAVSR_WS_EXP_NO_MFMA nocontract:-DEAVSR_WS_EXP_NO_CONTRACT nosample:-DEAVSR_WS_EXP_NO_SAMPLE \
         nogather:-DEAVSR_WS_EXP_NO_GATHER nosplit:-DEAVSR_WS_EXP_NO_SPLIT nobwrite:-DEAVSR_WS_EXP_NO_BWRITE \
         barriers:"-DEAVSR_WS_EXP_NO_SAMPLE -DEAVSR_WS_EXP_NO_CONTRACT -DEAVSR_WS_EXP_NO_DMA" \
         sampleonly:"-DEAVSR_WS_EXP_NO_CONTRACT -DEAVSR_WS_EXP_NO_DMA" contractonly:"-DEAVSR_WS_EXP_NO_SAMPLE -DEAVSR_WS_EXP_NO_DMA" $EXTRA_VARIANTS; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc $F $flags eavsr_amd/csrc/dcnv2_ws.hip eavsr_amd/csrc/capi.hip -o eavsr_amd/lib/libws_$name.so 2>/dev/null &
done
wait
ls eavsr_amd/lib/libws_*.so
