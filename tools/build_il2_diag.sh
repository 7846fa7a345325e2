#!/bin/bash
# Diagnostic libraries eavsr_amd/lib/libil2_*.so: dcnv2_il2.hip + dcnv2_il.hip + dcnv2_x9.hip + capi.hip with -DEAVSR_IL2_EXP_*
# (timing ablations, results wrong by construction) or -DEAVSR_IL2_STAMPS.  Built here (hipcc cross-compiles); they travel to
# the GPU box with the snapshot.  Loaded by tools/gpu_il2_ablate.py through ctypes.
set -e
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -ffp-contract=fast -Iinclude -Ieavsr_amd/csrc -shared -fno-slp-vectorize $IL2_EXTRA"
rm -f eavsr_amd/lib/libil2_*.so
for v in full:"" stamps:-DEAVSR_IL2_STAMPS nodma:-DEAVSR_IL2_EXP_NO_DMA noparams:-DEAVSR_IL2_EXP_NO_PARAMS nogather:-DEAVSR_IL2_EXP_NO_GATHER \
         nosplit:-DEAVSR_IL2_EXP_NO_SPLIT nomfma:-DEAVSR_IL2_EXP_NO_MFMA nofixup:-DEAVSR_IL2_EXP_NO_FIXUP nostore:-DEAVSR_IL2_EXP_NO_STORE \
         nobarrier:-DEAVSR_IL2_EXP_NO_BARRIER nosetup:-DEAVSR_IL2_EXP_NO_SETUP noblend:-DEAVSR_IL2_EXP_NO_BLEND \
         nomem:"-DEAVSR_IL2_EXP_NO_DMA -DEAVSR_IL2_EXP_NO_PARAMS -DEAVSR_IL2_EXP_NO_STORE -DEAVSR_IL2_EXP_NO_FIXUP" \
         nomembar:"-DEAVSR_IL2_EXP_NO_DMA -DEAVSR_IL2_EXP_NO_PARAMS -DEAVSR_IL2_EXP_NO_STORE -DEAVSR_IL2_EXP_NO_FIXUP -DEAVSR_IL2_EXP_NO_BARRIER" \
         onlymfma:"-DEAVSR_IL2_EXP_NO_DMA -DEAVSR_IL2_EXP_NO_PARAMS -DEAVSR_IL2_EXP_NO_STORE -DEAVSR_IL2_EXP_NO_FIXUP -DEAVSR_IL2_EXP_NO_GATHER -DEAVSR_IL2_EXP_NO_BLEND -DEAVSR_IL2_EXP_NO_SETUP" \
         nosampler:"-DEAVSR_IL2_EXP_NO_GATHER -DEAVSR_IL2_EXP_NO_BLEND -DEAVSR_IL2_EXP_NO_SETUP" $EXTRA_VARIANTS; do
  name=${v%%:*}; flags=${v#*:}; flags=${flags//|/ }
  /opt/rocm/bin/hipcc $F $flags eavsr_amd/csrc/dcnv2_il2.hip eavsr_amd/csrc/capi.hip -o eavsr_amd/lib/libil2_$name.so 2>/dev/null &
  if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
done
wait
ls eavsr_amd/lib/libil2_*.so
