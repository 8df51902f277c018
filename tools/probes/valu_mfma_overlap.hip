// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/valu_mfma_overlap.hip -o tools/probes/valu_mfma_overlap ; run it on the GPU box.
// How much VALU work hides under bf16 MFMAs on gfx950?  (Round 5, the bf16x3 `extra`: the three-way split of fp32 operands costs
// 5.5 VALU instructions per element next to v_mfma_f32_32x32x16_bf16.)  One workgroup per CU slot, W waves per SIMD; every wave loops
//   [V split-like VALU instructions (cvt_pk / shift / and / sub chains on independent registers)] [24 MFMAs on 4 accumulators]
// and the host reports cycles per iteration per SIMD against the MFMA-only (V = 0) and VALU-only (no MFMA) runs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r; }
__device__ __forceinline__ float subf(float a, float b) { float r; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <int FRAGS, bool MFMA>
__global__ void __launch_bounds__(256) probe(float* out, const float* in, int iters) {
    float x[8];
    for (int e = 0; e < 8; ++e) x[e] = in[threadIdx.x * 8 + e];
    f32x16 acc[4] = {};
    u32x4 H = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, M = H, L = H;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < FRAGS; ++f) {          // one fragment = 8 elements = 44 VALU instructions, as split_bf16x3
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const unsigned h = cvt_pk(x[2 * p], x[2 * p + 1]);
                const float r0 = subf(x[2 * p], __uint_as_float(h << 16)), r1 = subf(x[2 * p + 1], __uint_as_float(h & 0xffff0000u));
                const unsigned m = cvt_pk(r0, r1);
                const float s0 = subf(r0, __uint_as_float(m << 16)), s1 = subf(r1, __uint_as_float(m & 0xffff0000u));
                H[p] ^= h; M[p] ^= m; L[p] ^= cvt_pk(s0, s1);
                x[2 * p] += 1e-3f; x[2 * p + 1] -= 1e-3f;
            }
        }
        if (MFMA) {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, t & 1 ? H : M), __builtin_bit_cast(bf16x8, t & 2 ? L : H), acc[a], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[blockIdx.x * 256 + threadIdx.x] = s + __uint_as_float(H[0] ^ M[1] ^ L[2]);
}

// Second experiment: the fused Winograd kernels' situation -- fp32 MFMAs (v_mfma_f32_16x16x4_f32, 32 cycles) with plain v_fma_f32 work (the
// output transform's T / Y updates) from the same wave.  NV independent FMAs + 32 MFMAs (1 024 cycles) per iteration.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NV, bool MFMA>
__global__ void __launch_bounds__(256) probe32(float* out, const float* in, int iters) {
    float y[32];
    for (int e = 0; e < 32; ++e) y[e] = in[(threadIdx.x * 8 + e) & 2047];
    f32x4 acc[2] = {};
    const float a = in[threadIdx.x & 63], b = in[(threadIdx.x + 7) & 63], c = 1.0009765625f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(y[v & 31]) : "v"(c), "v"(a));
        if (MFMA) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[1], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int e = 0; e < 32; ++e) s += y[e];
    out[blockIdx.x * 256 + threadIdx.x] = s + acc[0][0] + acc[1][3];
}
// Third experiment: TWO waves per SIMD from one 512-thread workgroup (waves w and w + 4 share a SIMD), the second one half an iteration out of
// phase (it starts with its MFMA block while the first one starts with its VALU block): the best case for cross-wave overlap.
template <int NV>
__global__ void __launch_bounds__(512) probe32_antiphase(float* out, const float* in, int iters, int antiphase) {
    float y[32];
    for (int e = 0; e < 32; ++e) y[e] = in[(threadIdx.x * 8 + e) & 2047];
    f32x4 acc[2] = {};
    const float a = in[threadIdx.x & 63], b = in[(threadIdx.x + 7) & 63], c = 1.0009765625f;
    const bool second = antiphase && ((threadIdx.x >> 8) & 1);
    auto valu = [&]() {
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(y[v & 31]) : "v"(c), "v"(a));
    };
    auto mfma = [&]() {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[1], 0, 0, 0);
        }
    };
    if (second) mfma();
    for (int it = 0; it < iters; ++it) { valu(); mfma(); }
    float s = 0.f;
    for (int e = 0; e < 32; ++e) s += y[e];
    out[blockIdx.x * 512 + threadIdx.x] = s + acc[0][0] + acc[1][3];
}
template <int NV>
static double run_anti(int iters, float* out, float* in, int antiphase) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe32_antiphase<NV><<<256, 512>>>(out, in, 10, antiphase);
    hipEventRecord(e0);
    probe32_antiphase<NV><<<256, 512>>>(out, in, iters, antiphase);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / iters;
}

// Fourth experiment: the fillers placed BETWEEN the MFMAs of one wave's stream: 32 x [v_mfma_f32_16x16x4_f32, F x v_fma_f32].
template <int F>
__global__ void __launch_bounds__(256) probe32_interleaved(float* out, const float* in, int iters) {
    float y[32];
    for (int e = 0; e < 32; ++e) y[e] = in[(threadIdx.x * 8 + e) & 2047];
    f32x4 acc[2] = {};
    const float a = in[threadIdx.x & 63], b = in[(threadIdx.x + 7) & 63], c = 1.0009765625f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            acc[t & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < F; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(y[(t * F + v) & 31]) : "v"(c), "v"(a));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int e = 0; e < 32; ++e) s += y[e];
    out[blockIdx.x * 256 + threadIdx.x] = s + acc[0][0] + acc[1][3];
}
template <int F>
static double run_inter(int w, int iters, float* out, float* in) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe32_interleaved<F><<<256 * w, 256>>>(out, in, 10);
    hipEventRecord(e0);
    probe32_interleaved<F><<<256 * w, 256>>>(out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / iters;
}

// Fifth experiment: the same FMAs as packed instructions: 32 x [MFMA, F x v_pk_fma_f32] does 2 F FMAs per gap -- against 2 F x v_fma_f32.
typedef float f32x2p __attribute__((ext_vector_type(2)));
template <int F>
__global__ void __launch_bounds__(256) probe32_interleaved_pk(float* out, const float* in, int iters) {
    f32x2p y[16];
    for (int e = 0; e < 16; ++e) y[e] = f32x2p{in[(threadIdx.x * 8 + e) & 2047], in[(threadIdx.x * 8 + e + 16) & 2047]};
    f32x4 acc[2] = {};
    const float a = in[threadIdx.x & 63], b = in[(threadIdx.x + 7) & 63];
    const f32x2p c = {1.0009765625f, 1.0009765625f}, av = {a, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            acc[t & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < F; ++v) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y[(t * F + v) & 15]) : "v"(c), "v"(av));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += y[e][0] + y[e][1];
    out[blockIdx.x * 256 + threadIdx.x] = s + acc[0][0] + acc[1][3];
}
template <int F>
static double run_inter_pk(int w, int iters, float* out, float* in) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe32_interleaved_pk<F><<<256 * w, 256>>>(out, in, 10);
    hipEventRecord(e0);
    probe32_interleaved_pk<F><<<256 * w, 256>>>(out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / iters;
}

template <int NV, bool MFMA>
static double run32(int wg_per_cu, int iters, float* out, float* in) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    probe32<NV, MFMA><<<grid, 256>>>(out, in, 10);
    hipEventRecord(e0);
    probe32<NV, MFMA><<<grid, 256>>>(out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / iters;
}

template <int FRAGS, bool MFMA>
static double run(int wg_per_cu, int iters, float* out, float* in) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    probe<FRAGS, MFMA><<<grid, 256>>>(out, in, 10);
    hipEventRecord(e0);
    probe<FRAGS, MFMA><<<grid, 256>>>(out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / iters;      // seconds per iteration (every SIMD runs wg_per_cu waves)
}

int main() {
    float *out, *in; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&in, 256 * 8 * 4);
    std::vector<float> h(2048); for (int i = 0; i < 2048; ++i) h[i] = 1.0f + i * 1e-4f;
    hipMemcpy(in, h.data(), 8192, hipMemcpyHostToDevice);
    const int iters = 20000;
    printf("per iteration and wave: FRAGS x 44 VALU instructions (+ 8 adds), 24 x v_mfma_f32_32x32x16_bf16 (768 cycles at 32 cycles each)\n");
    printf("waves/SIMD  frags  VALU-only ns   MFMA-only ns   both ns   both/(MFMA-only)  (sum would be %%)\n");
    for (int w = 1; w <= 4; w *= 2) {
        const double m0 = run<0, true>(w, iters, out, in);
#define ROW(F) { const double v = run<F, false>(w, iters, out, in), b = run<F, true>(w, iters, out, in); \
                 printf("%5d      %5d  %10.1f  %12.1f  %9.1f  %10.2f        (%.2f)\n", w, F, v * 1e9, m0 * 1e9, b * 1e9, b / m0, (v + m0) / m0); }
        ROW(1) ROW(2) ROW(4) ROW(6)
    }
    printf("\nfp32: per iteration and wave NV x v_fma_f32 (independent over 32 registers) + 32 x v_mfma_f32_16x16x4_f32 (1 024 cycles)\n");
    printf("waves/SIMD   NV   VALU-only ns   MFMA-only ns   both ns   both/(MFMA-only)  (sum would be)\n");
    for (int w = 1; w <= 2; ++w) {
        const double m0 = run32<0, true>(w, iters, out, in);
#define ROW32(N) { const double v = run32<N, false>(w, iters, out, in), b = run32<N, true>(w, iters, out, in); \
                   printf("%5d      %4d  %10.1f  %12.1f  %9.1f  %10.2f        (%.2f)\n", w, N, v * 1e9, m0 * 1e9, b * 1e9, b / m0, (v + m0) / m0); }
        ROW32(32) ROW32(64) ROW32(128) ROW32(256)
    }
    printf("\nfp32, two waves per SIMD from ONE 512-thread workgroup: in phase vs the second wave half an iteration ahead (ns per iteration)\n");
    printf("   NV   in phase   anti-phase   (MFMA-only 2 waves: see above)\n");
#define ROWA(N) printf("%5d  %9.1f  %10.1f\n", N, run_anti<N>(iters, out, in, 0) * 1e9, run_anti<N>(iters, out, in, 1) * 1e9);
    ROWA(64) ROWA(128) ROWA(256)
    printf("\nfp32, fillers BETWEEN the MFMAs of one stream: 32 x [MFMA, F x v_fma_f32] per iteration (ns per iteration; 32 MFMAs alone: see MFMA-only)\n");
    printf("waves/SIMD    F=0      F=2      F=4      F=6      F=8\n");
    for (int w = 1; w <= 2; ++w)
        printf("%5d    %7.1f  %7.1f  %7.1f  %7.1f  %7.1f\n", w, run_inter<0>(w, iters, out, in) * 1e9, run_inter<2>(w, iters, out, in) * 1e9,
               run_inter<4>(w, iters, out, in) * 1e9, run_inter<6>(w, iters, out, in) * 1e9, run_inter<8>(w, iters, out, in) * 1e9);
    printf("\nfp32, the same with PACKED fillers: 32 x [MFMA, F x v_pk_fma_f32] (2 F FMAs per gap)\n");
    printf("waves/SIMD    F=0      F=1      F=2      F=3      F=4\n");
    for (int w = 1; w <= 2; ++w)
        printf("%5d    %7.1f  %7.1f  %7.1f  %7.1f  %7.1f\n", w, run_inter_pk<0>(w, iters, out, in) * 1e9, run_inter_pk<1>(w, iters, out, in) * 1e9,
               run_inter_pk<2>(w, iters, out, in) * 1e9, run_inter_pk<3>(w, iters, out, in) * 1e9, run_inter_pk<4>(w, iters, out, in) * 1e9);
    return 0;
}
