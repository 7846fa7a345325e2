// Floor of the 16-bit MFMA pipe at the clock the chip holds: 256 workgroups x 8 waves, each wave issues NM dependent-chain
// v_mfma_f32_32x32x16_bf16 on three accumulators from register operands (random bits), nothing else.  Prints us per launch and
// the cycles per MFMA and SIMD that implies at 2.4 GHz.  hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512, 2) void k(const unsigned* in, float* out, int nm) {
  const int t = threadIdx.x + blockIdx.x * 512;
  u32x4 a = {in[t & 4095], in[(t + 1) & 4095], in[(t + 2) & 4095], in[(t + 3) & 4095]};
  u32x4 b = {in[(t + 5) & 4095], in[(t + 6) & 4095], in[(t + 7) & 4095], in[(t + 8) & 4095]};
  f32x16 c0 = {0}, c1 = {0}, c2 = {0};
  for (int i = 0; i < nm; i += 3) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, a), c2, 0, 0, 0);
  }
  float s = 0;
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e];
  out[t] = s;
}
int main(int argc, char** argv) {
  const int nm = argc > 1 ? atoi(argv[1]) : 720;      // MFMAs per wave (two waves per SIMD)
  const int zero = argc > 2 ? atoi(argv[2]) : 0;
  unsigned* in; float* out;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 512 * 4);
  unsigned h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = zero ? 0u : (((unsigned)rand() & 0x3fff3fffu) | 0x3c003c00u);   // bf16 pairs in [0.0078, 2)
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, in, out, nm);
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, in, out, nm);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 20;
    printf("nm %d per wave (%d per SIMD)%s: %.1f us per launch = %.1f ns per MFMA and SIMD = %.2f PFLOP/s\n", nm, 2 * nm, zero ? " zeros" : "", us,
           us * 1e3 / (2.0 * nm), 256.0 * 8 * nm * 32768.0 / (us * 1e-6) / 1e15);
  }
  return 0;
}
