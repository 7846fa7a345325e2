// Micro-benchmark (gfx950): how many vector instructions hide in the gap behind an MFMA, in the SAME wave's stream and in the
// partner wave's stream of a SIMD -- for the fp32 MFMA the Winograd kernels use and the bf16 MFMA of the DCNv2 contraction.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_fill mfma_fill.hip ; run on the GPU box, prints one table.
//   own  : every wave runs [1 MFMA + F v_fma_f32] x GAPS           (1 or 2 waves per SIMD)
//   split: waves 0-3 run MFMAs only, waves 4-7 run 1 + F... vector instructions only (2 waves per SIMD), each timed on its own,
//          against the same streams alone (the partner half exits at once).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int GAPS = 16;    // MFMAs per loop trip (4 accumulators x 4)
constexpr int TRIPS = 256;

template <int F>
__device__ __forceinline__ void fillers(float& x0, float& x1, float& x2, float& x3, float m) {
#pragma unroll
    for (int i = 0; i < F; ++i) {
        if ((i & 3) == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(m));
        if ((i & 3) == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x1) : "v"(m));
        if ((i & 3) == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x2) : "v"(m));
        if ((i & 3) == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x3) : "v"(m));
    }
}

// KIND 0: v_mfma_f32_16x16x4_f32 (8 passes), 1: v_mfma_f32_32x32x16_bf16 (8 passes), 2: v_mfma_f32_16x16x32_bf16 (4 passes),
// 3: v_mfma_f32_32x32x2_f32 (16 passes)
// ROLE 0: MFMA + F fillers per gap; 1: MFMAs only in waves 0-3 and fillers only in waves 4-7; 2: as 1, waves 4-7 exit; 3: as 1, waves 0-3 exit
template <int KIND>
__device__ __forceinline__ void one_mfma(f4* a4, f16v* a16, float fa, float fb, s8 ha, s8 hb, int g) {
    if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a4[g & 3]) : "v"(fa), "v"(fb));
    if (KIND == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a16[g & 1]) : "v"(ha), "v"(hb));
    if (KIND == 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a4[g & 3]) : "v"(ha), "v"(hb));
    if (KIND == 3) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a16[g & 1]) : "v"(fa), "v"(fb));
}

// GROUP: MFMAs issued back to back in groups of GROUP, each group followed by its GROUP x F fillers (1 = one MFMA, F fillers, ...)
template <int KIND, int F, int ROLE, int GROUP>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, float seed) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3;
    const float m = 0.999f;
    f4 a4[4];
    f16v a16[2];
    for (int i = 0; i < 4; ++i) a4[i] = f4{seed, seed, seed, seed};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 16; ++j) a16[i][j] = seed;
    float fa = seed * 0.5f, fb = seed * 0.25f;
    s8 ha, hb;
    for (int j = 0; j < 8; ++j) { ha[j] = (short)(0x3f80 + j); hb[j] = (short)(0x3f00 + j); }
    const bool do_mfma = ROLE == 0 || ((ROLE == 1 || ROLE == 2) && wave < 4);
    const bool do_fill = ROLE == 0 || ((ROLE == 1 || ROLE == 3) && wave >= 4);
    if (!do_mfma && !do_fill) return;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (do_mfma && do_fill) {
        for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
            for (int g = 0; g < GAPS; g += GROUP) {
#pragma unroll
                for (int j = 0; j < GROUP; ++j) one_mfma<KIND>(a4, a16, fa, fb, ha, hb, g + j);
#pragma unroll
                for (int j = 0; j < GROUP; ++j) fillers<F>(x0, x1, x2, x3, m);
            }
        }
    } else if (do_mfma) {
        for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
            for (int g = 0; g < GAPS; ++g) one_mfma<KIND>(a4, a16, fa, fb, ha, hb, g);
        }
    } else {
        for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
            for (int g = 0; g < GAPS; ++g) fillers<F>(x0, x1, x2, x3, m);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    float s = x0 + x1 + x2 + x3;
    for (int i = 0; i < 4; ++i) s += a4[i][0] + a4[i][3];
    for (int i = 0; i < 2; ++i) s += a16[i][0] + a16[i][15];
    if (s == 12345.678f) sink[threadIdx.x] = s;
}

static unsigned long long* d_out;
static float* d_sink;

template <int KIND, int F, int ROLE, int GROUP = 1>
static void run(int threads, const char* what) {
    const int blocks = 256;
    CHECK(hipMemset(d_out, 0, blocks * 8 * sizeof(unsigned long long)));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, F, ROLE, GROUP>), dim3(blocks), dim3(threads), 0, 0, d_out, d_sink, 1.0f);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 8);
    CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double lo = 0, hi = 0;
    int nlo = 0, nhi = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < threads / 64; ++w) {
            if (!h[b * 8 + w]) continue;
            if (w < 4) { lo += (double)h[b * 8 + w]; ++nlo; } else { hi += (double)h[b * 8 + w]; ++nhi; }
        }
    const double per = (double)TRIPS * GAPS;
    printf("kind %d  F %2d  group %2d  %-44s waves/SIMD %d   waves 0-3: %7.1f cyc/gap   waves 4-7: %7.1f cyc/gap\n", KIND, F, GROUP, what, threads / 256,
           nlo ? lo / nlo / per : 0.0, nhi ? hi / nhi / per : 0.0);
}

template <int KIND>
static void sweep() {
    run<KIND, 0, 0>(256, "own stream");
    run<KIND, 2, 0>(256, "own stream");
    run<KIND, 4, 0>(256, "own stream");
    run<KIND, 6, 0>(256, "own stream");
    run<KIND, 8, 0>(256, "own stream");
    run<KIND, 12, 0>(256, "own stream");
    run<KIND, 0, 0>(512, "own stream");
    run<KIND, 2, 0>(512, "own stream");
    run<KIND, 4, 0>(512, "own stream");
    run<KIND, 6, 0>(512, "own stream");
    run<KIND, 8, 0>(512, "own stream");
    run<KIND, 12, 0>(512, "own stream");
    run<KIND, 4, 0, 4>(256, "own stream, grouped");
    run<KIND, 8, 0, 4>(256, "own stream, grouped");
    run<KIND, 4, 0, 16>(256, "own stream, grouped");
    run<KIND, 8, 0, 16>(256, "own stream, grouped");
    run<KIND, 4, 0, 4>(512, "own stream, grouped");
    run<KIND, 8, 0, 4>(512, "own stream, grouped");
    run<KIND, 12, 0, 4>(512, "own stream, grouped");
    run<KIND, 4, 0, 16>(512, "own stream, grouped");
    run<KIND, 8, 0, 16>(512, "own stream, grouped");
    run<KIND, 12, 0, 16>(512, "own stream, grouped");
    run<KIND, 4, 2>(512, "MFMA half alone");
    run<KIND, 4, 3>(512, "vector half alone (4 per gap)");
    run<KIND, 4, 1>(512, "split: MFMA half | vector half (4 per gap)");
    run<KIND, 8, 3>(512, "vector half alone (8 per gap)");
    run<KIND, 8, 1>(512, "split: MFMA half | vector half (8 per gap)");
    run<KIND, 12, 3>(512, "vector half alone (12 per gap)");
    run<KIND, 12, 1>(512, "split: MFMA half | vector half (12 per gap)");
}

int main() {
    CHECK(hipMalloc(&d_out, 256 * 8 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&d_sink, 4096));
    sweep<0>();
    sweep<3>();
    sweep<1>();
    sweep<2>();
    return 0;
}
