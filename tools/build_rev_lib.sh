#!/bin/bash
# A/B build of the whole library with some sources taken from an older commit:
#   tools/build_rev_lib.sh NAME REV file.hip [file2.hip ..]  ->  eavsr_amd/lib/libeavsr_NAME.so  (EAVSR_LIB_PATH selects it)
# Everything else (header, the other sources, flags) is the working tree's, so the ABI is the current one.
set -e
cd "$(dirname "$0")/.."
NAME=$1; REV=$2; shift 2
T=$(mktemp -d)
mkdir -p $T/eavsr_amd/csrc $T/include
cp eavsr_amd/csrc/*.hip eavsr_amd/csrc/*.h $T/eavsr_amd/csrc/
cp include/*.h $T/include/
for f in "$@"; do git show $REV:eavsr_amd/csrc/$f > $T/eavsr_amd/csrc/$f; done
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -ffp-contract=fast -fno-slp-vectorize -Wno-unused-function -DEAVSR_LAB=${LAB:-0} $EXTRA_FLAGS"
mkdir -p $T/obj
LABSRC=" dcnv2_ws.hip conv_x9.hip conv_wino.hip rcab_h16.hip "      # eavsr_amd/build.py LAB_SOURCES: LAB=1 builds the lab flavour
for f in $T/eavsr_amd/csrc/*.hip; do
  if [ "${LAB:-0}" != 1 ] && [[ "$LABSRC" == *" $(basename $f) "* ]]; then continue; fi
  ( /opt/rocm/bin/hipcc $F -c $f -o $T/obj/$(basename ${f%.hip}).o 2>/dev/null ) &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.2; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o eavsr_amd/lib/libeavsr_$NAME.so $T/obj/*.o
rm -rf $T
ls -la eavsr_amd/lib/libeavsr_$NAME.so
