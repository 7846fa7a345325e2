// Shared host/device helpers of libeavsr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include <mutex>

#include "../../include/eavsr_hip.h"

// 0: the product build (default).  1: `python -m eavsr_amd.build --lab` -- also the retired schedules kept for A/B measurements
// (eavsr_amd/build.py LAB_SOURCES and the `#if EAVSR_LAB` pieces of predictor.hip / conv_wino6.hip; the header's experimental section)
#ifndef EAVSR_LAB
#define EAVSR_LAB 0
#endif

namespace eavsr {

// thread-local error string behind eavsr_last_error()
void set_error(const char* fmt, ...);
void clear_error();
// hipGetLastError() after a launch: 0 or the hipError_t (also records the message)
int launch_status(const char* what);

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// per-device one-time initialisation (hipFuncSetAttribute applies to the device that is current when it is called;
// ops._DeviceOf launches on whichever device owns the tensors)
constexpr int kMaxDevices = 64;
struct PerDeviceOnce {
  std::once_flag flag[kMaxDevices];
};
static inline int current_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return (d >= 0 && d < kMaxDevices) ? d : 0;
}

}  // namespace eavsr

#define EAVSR_REQUIRE(cond, code, ...)      \
  do {                                      \
    if (!(cond)) {                          \
      ::eavsr::set_error(__VA_ARGS__);      \
      return (code);                        \
    }                                       \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Bijective XCD-aware remap of a linear workgroup id.  Workgroups are dealt round-robin over the 8 XCDs
// (b and b+8 share an XCD and its private L2), so consecutive ids -- neighbouring tiles that share halo
// rows -- would land on 8 different L2s.  The remap gives every XCD a contiguous run of ids.  Placement
// is a speed hint only; nothing depends on it for correctness.
#ifdef __HIPCC__
__device__ __forceinline__ int eavsr_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, pos = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + pos;
}
#endif

// Static priority for the second-dispatched half of an 8-wave workgroup.  The two waves of a SIMD (w and w + 4) run the same
// program between the same barriers and otherwise contend cycle by cycle, in lockstep; with one of them preferred it runs
// ahead and the pair settles into a stagger (MI355X_MICROARCH.md, "Two waves per SIMD", items 4 and 9).  Measured on the
// F(4x4,3x3) kernel: 47.8 -> 43.9 us per 2 x 64 x 180 x 320 convolution (the priority level and which half gets it do not
// matter: all within 1 us).  -DEAVSR_NO_WAVE_PRIO builds the A-B reference.
#ifdef __HIPCC__
__device__ __forceinline__ void eavsr_stagger_priority(int wave) {
#if defined(EAVSR_PRIO_LOW_HALF)      // A/B: the first-dispatched half instead (what the round-4 DCNv2 kernel prefers)
  if (wave < 4) __builtin_amdgcn_s_setprio(3);
#elif !defined(EAVSR_NO_WAVE_PRIO)
  if (wave >= 4) __builtin_amdgcn_s_setprio(3);
#else
  (void)wave;
#endif
}
#endif

// Activation of the convolution epilogues, branch-free: v > 0 ? v : v (*) s with s = 1 (none), 0 (ReLU) or the leaky slope, where (*)
// is v_mul_legacy_f32 (0 times anything, infinities and NaNs included, is +0).  With the plain product ReLU(-inf) was -inf
// (-inf * 0 = NaN, and max returns its other operand) and ReLU of a negative value -0.0; with the legacy product both
// are +0, as torch.relu gives.  Infinities pass through the leaky / identity forms.  A NaN passes through EVERY form, as it does
// through torch.relu / F.leaky_relu (networks.py:149-150 builds them): v_max_f32 returns its other operand for a NaN, so the
// max form would turn ReLU(NaN) into +0 -- eavsr_act() is a SELECT instead: v itself where v > 0 or v is unordered, the legacy
// product elsewhere (a diverged training run must show in the loss through these epilogues too).

#ifdef __HIPCC__
// sigmoid as 1 / (1 + 2^(-x log2 e)) with v_exp_f32 and v_rcp_f32 (<= 1 ulp each): the mask activation of AdaptBlockOffset
// (networks.py:314), evaluated by the same instructions wherever it runs (the DCNv2 sampler or the epilogue of the 5x5 heads)
__device__ __forceinline__ float eavsr_sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

__device__ __forceinline__ float eavsr_mul_legacy(float a, float b) {
  float r;
  asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// act(v) for act_s = 1 (none) | 0 (ReLU) | slope in [0, 1] (leaky ReLU): see the note above
// Three instructions per value (v_mul_legacy, v_cmp_nle, v_cndmask): v where v > 0 or v is unordered, v (*) s elsewhere.  (The
// fmaxf form needed two canonicalising v_max + the max + an unordered compare + a select beside the product: six.)
__device__ __forceinline__ float eavsr_act(float v, float act_s) {
  const float z = eavsr_mul_legacy(v, act_s);
  return !(v <= 0.f) ? v : z;
}
#endif

// spatial tile of the implicit-GEMM conv kernels: (32, 16 or 8) rows x 32 columns per 512-thread workgroup,
// wave w owns NT = 4, 2 or 1 consecutive rows, one 32-pixel MFMA N-tile per row.
#define EAVSR_CONV_TW 32
