#!/bin/bash
# A/B build of the whole library with extra compiler flags: tools/build_ab_lib.sh NAME "-DFLAG ..." -> eavsr_amd/lib/libeavsr_NAME.so
# (selected at run time with EAVSR_LIB_PATH; same ABI as the default build)
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -ffp-contract=fast -fno-slp-vectorize -Wno-unused-function -DEAVSR_LAB=${LAB:-0}"
mkdir -p eavsr_amd/lib/obj_$NAME
LABSRC=" dcnv2_ws.hip conv_x9.hip conv_wino.hip rcab_h16.hip "      # eavsr_amd/build.py LAB_SOURCES: LAB=1 builds the lab flavour
for f in eavsr_amd/csrc/*.hip; do
  if [ "${LAB:-0}" != 1 ] && [[ "$LABSRC" == *" $(basename $f) "* ]]; then continue; fi
  ( /opt/rocm/bin/hipcc $F $FLAGS -c $f -o eavsr_amd/lib/obj_$NAME/$(basename ${f%.hip}).o 2>/dev/null ) &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.2; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o eavsr_amd/lib/libeavsr_$NAME.so eavsr_amd/lib/obj_$NAME/*.o
rm -rf eavsr_amd/lib/obj_$NAME
ls -la eavsr_amd/lib/libeavsr_$NAME.so
