#!/bin/bash
# Diagnostic libraries eavsr_amd/lib/libh16_*.so: conv_h16.hip + capi.hip with -DEAVSR_H16_EXP_* (timing ablations, results wrong).
set -e
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -ffp-contract=fast -fno-slp-vectorize -Iinclude -Ieavsr_amd/csrc -shared"
rm -f eavsr_amd/lib/libh16_*.so
for v in full:"" nomfma:-DEAVSR_H16_EXP_NO_MFMA nostore:-DEAVSR_H16_EXP_NO_STORE nodma:-DEAVSR_H16_EXP_NO_DMA nowdma:-DEAVSR_H16_EXP_NO_WDMA stamps:-DEAVSR_H16_STAMPS \
         nomem:"-DEAVSR_H16_EXP_NO_STORE -DEAVSR_H16_EXP_NO_DMA -DEAVSR_H16_EXP_NO_WDMA" \
         nothing:"-DEAVSR_H16_EXP_NO_STORE -DEAVSR_H16_EXP_NO_DMA -DEAVSR_H16_EXP_NO_WDMA -DEAVSR_H16_EXP_NO_MFMA" $EXTRA_VARIANTS; do
  name=${v%%:*}; flags=${v#*:}; flags=${flags//|/ }
  /opt/rocm/bin/hipcc $F $flags eavsr_amd/csrc/conv_h16.hip eavsr_amd/csrc/capi.hip -o eavsr_amd/lib/libh16_$name.so 2>/dev/null &
done
wait
ls eavsr_amd/lib/libh16_*.so
