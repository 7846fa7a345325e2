#!/bin/bash
# Diagnostic build: libeavsr_hip_diag.so = the library with in-kernel cycle stamps in the pipelined DCNv2 kernel.
set -e
cd "$(dirname "$0")/.."
OUT=eavsr_amd/lib/libeavsr_hip_diag${SUFFIX}.so
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -ffp-contract=fast -DEAVSR_DCN_STAMPS $EXTRA \
  -shared -o $OUT eavsr_amd/csrc/dcnv2.hip eavsr_amd/csrc/capi.hip
echo $OUT
