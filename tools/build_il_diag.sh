#!/bin/bash
# Diagnostic libraries eavsr_amd/lib/libil_*.so: dcnv2_il.hip + capi.hip with -DEAVSR_IL_EXP_* (timing ablations, results
# wrong by construction).  Built here (hipcc cross-compiles); they travel to the GPU box with the snapshot.
set -e
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -ffp-contract=fast -Iinclude -Ieavsr_amd/csrc -shared"
rm -f eavsr_amd/lib/libil_*.so
for v in full:"" nodma:-DEAVSR_IL_EXP_NO_DMA noparams:-DEAVSR_IL_EXP_NO_PARAMS nogather:-DEAVSR_IL_EXP_NO_GATHER nosplit:-DEAVSR_IL_EXP_NO_SPLIT \
         nomfma:-DEAVSR_IL_EXP_NO_MFMA nofixup:-DEAVSR_IL_EXP_NO_FIXUP nostore:-DEAVSR_IL_EXP_NO_STORE stamps:-DEAVSR_IL_STAMPS \
         nomem:"-DEAVSR_IL_EXP_NO_DMA -DEAVSR_IL_EXP_NO_PARAMS -DEAVSR_IL_EXP_NO_STORE -DEAVSR_IL_EXP_NO_FIXUP" \
         onlymfma:"-DEAVSR_IL_EXP_NO_DMA -DEAVSR_IL_EXP_NO_PARAMS -DEAVSR_IL_EXP_NO_STORE -DEAVSR_IL_EXP_NO_FIXUP -DEAVSR_IL_EXP_NO_GATHER -DEAVSR_IL_EXP_NO_SPLIT" $EXTRA_VARIANTS; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc $F $flags eavsr_amd/csrc/dcnv2_il.hip eavsr_amd/csrc/capi.hip -o eavsr_amd/lib/libil_$name.so 2>/dev/null &
done
wait
ls eavsr_amd/lib/libil_*.so
