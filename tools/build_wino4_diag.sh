#!/bin/bash
# Diagnostic libraries eavsr_amd/lib/libwino4_*.so: conv_wino6.hip + capi.hip with -DEAVSR_WINO_EXP_* (timing ablations,
# results wrong) or -DEAVSR_W4_* (schedule experiments, results right).  Built here (hipcc cross-compiles), they travel to the
# GPU box with the snapshot.  VARIANTS="name:flags ..." overrides the list.
set -e
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -ffp-contract=fast -Iinclude -Ieavsr_amd/csrc -shared"
rm -f eavsr_amd/lib/libwino4_*.so
DEFAULT='base: timeline:-DEAVSR_W4_TIMELINE nodma:-DEAVSR_WINO_EXP_NODMA notransform:-DEAVSR_WINO_EXP_NOTRANSFORM nomfma:-DEAVSR_WINO_EXP_NOMFMA nostore:-DEAVSR_WINO_EXP_NOSTORE'
for v in ${VARIANTS:-$DEFAULT} $EXTRA_VARIANTS; do
  name=${v%%:*}; flags=${v#*:}; flags=${flags//,/ }
  /opt/rocm/bin/hipcc $F $flags eavsr_amd/csrc/conv_wino6.hip eavsr_amd/csrc/capi.hip -o eavsr_amd/lib/libwino4_$name.so 2>/dev/null &
done
wait
ls eavsr_amd/lib/libwino4_*.so
