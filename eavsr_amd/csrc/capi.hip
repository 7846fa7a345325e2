// Error plumbing, version and the MFMA layout self-test of libeavsr_hip.so.
#include "common.h"

#include <stdarg.h>
#include <stdio.h>

namespace eavsr {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

void clear_error() { g_err[0] = 0; }

int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

}  // namespace eavsr

extern "C" int eavsr_abi_version(void) { return EAVSR_ABI_VERSION; }
extern "C" const char* eavsr_version(void) { return "eavsr-hip 0.1.0 gfx950"; }
extern "C" const char* eavsr_last_error(void) { return eavsr::g_err; }
extern "C" size_t eavsr_conv2d_desc_size(void) { return sizeof(eavsr_conv2d_desc); }
extern "C" int eavsr_lab_build(void) { return EAVSR_LAB; }

// ---------------------------------------------------------------------------------------------
// MFMA layout self-test.  One wave computes D(32x32) = A(32x2) . B(2x32) with the operand map
// lane l -> A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31] and the accumulator map
// D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31], on asymmetric small-integer data, and
// counts mismatches against the scalar product.
// ---------------------------------------------------------------------------------------------
__global__ void selftest_mfma_kernel(float* scratch) {
  const int l = threadIdx.x;
  const int i = l & 31, k = l >> 5;
  // A[i][k] = 3 i + 7 k + 1 ; B[k][j] = 5 j - 11 k + 2   (asymmetric, exact in fp32)
  const float a = (float)(3 * i + 7 * k + 1);
  const float b = (float)(5 * i - 11 * k + 2);
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  int bad = 0;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    const int col = l & 31;
    float ref = 0.f;
    for (int kk = 0; kk < 2; ++kk) ref += (float)(3 * row + 7 * kk + 1) * (float)(5 * col - 11 * kk + 2);
    if (acc[r] != ref) ++bad;
    scratch[1 + row * 32 + col] = acc[r];
  }
  // wave-wide sum of mismatches
  for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off);
  if (l == 0) scratch[0] = (float)bad;
}

extern "C" int eavsr_selftest_mfma_f32(float* scratch, void* stream) {
  EAVSR_REQUIRE(scratch != nullptr, -1, "selftest: scratch is NULL");
  hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(64), 0, eavsr::as_stream(stream), scratch);
  return eavsr::launch_status("selftest_mfma");
}
