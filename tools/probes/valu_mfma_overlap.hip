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
    return 0;
}
