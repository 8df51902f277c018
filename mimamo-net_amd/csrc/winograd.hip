// Winograd F(2x2,3x3) / F(4x4,3x3) transforms for the stride-1 3x3 convolutions of the ResNet50 trunk (conv2_x..conv5_x, 16
// layers, 42 % of the network's direct-form FLOPs): 2.25x / 4x fewer multiply-adds than the direct form for the same result up
// to fp32 rounding (the weight transform is done once on the host in float64).
//   V = B^T d B  per (m+2)x(m+2) input patch (stride m, pad 1)     -- this file, memory-bound
//   M_xi = V_xi * U_xi, xi = 0..(m+2)^2-1                           -- one BATCHED launch of the fp32 MFMA GEMM engine
//   Y = A^T M A + bias, ReLU  per m x m output patch                -- this file, memory-bound
// The default path of conv2_x..conv4_x replaces the last two steps by wino_fused.hip (output transform inside the GEMM kernel).
// The reference computes these layers with torch's direct fp32 conv (third-party ResNet50, api/resnet50_extractor.py:
// 74-83); parity is checked against the oracle's direct convolution.
#include "conv.h"

namespace mm {

// streaming (non-temporal) 16-byte store: planes and activations are consumed by a later kernel from HBM, not from L2
__device__ __forceinline__ float4 nt_load(const float4* p) {
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const f32x4_t x = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
    return float4{x.x, x.y, x.z, x.w};
}
__device__ __forceinline__ void nt_store(float4* p, const float4& v) {
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const f32x4_t x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<f32x4_t*>(p));
}


// one thread: one tile, four consecutive channels
__global__ void __launch_bounds__(256)
wino_in_kernel(const float* __restrict__ x, float* __restrict__ V, int B, int H, int W, int C4, int TH, int TW, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    const int64_t tile = i / C4;
    const int tx = (int)(tile % TW);
    const int ty = (int)((tile / TW) % TH);
    const int64_t b = tile / ((int64_t)TW * TH);
    const float4* src = reinterpret_cast<const float4*>(x);
    float4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int hy = 2 * ty - 1 + r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int wx = 2 * tx - 1 + q;
            const bool ok = (unsigned)hy < (unsigned)H && (unsigned)wx < (unsigned)W;
            d[r][q] = ok ? src[((b * H + hy) * W + wx) * C4 + c4] : float4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]:  t = B^T d (rows), V = t B (columns)
    auto sub = [](float4 a, float4 b) { return float4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; };
    auto add = [](float4 a, float4 b) { return float4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; };
    float4 t[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        t[0][q] = sub(d[0][q], d[2][q]);
        t[1][q] = add(d[1][q], d[2][q]);
        t[2][q] = sub(d[2][q], d[1][q]);
        t[3][q] = sub(d[1][q], d[3][q]);
    }
    float4* dst = reinterpret_cast<float4*>(V);
    const int64_t ntile = (int64_t)B * TH * TW;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float4 v0 = sub(t[r][0], t[r][2]), v1 = add(t[r][1], t[r][2]), v2 = sub(t[r][2], t[r][1]), v3 = sub(t[r][1], t[r][3]);
        dst[((int64_t)(r * 4 + 0) * ntile + tile) * C4 + c4] = v0;
        dst[((int64_t)(r * 4 + 1) * ntile + tile) * C4 + c4] = v1;
        dst[((int64_t)(r * 4 + 2) * ntile + tile) * C4 + c4] = v2;
        dst[((int64_t)(r * 4 + 3) * ntile + tile) * C4 + c4] = v3;
    }
}

__global__ void __launch_bounds__(256)
wino_out_kernel(const float* __restrict__ M, const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W, int C4,
                int TH, int TW, int relu, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    const int64_t tile = i / C4;
    const int tx = (int)(tile % TW);
    const int ty = (int)((tile / TW) % TH);
    const int64_t b = tile / ((int64_t)TW * TH);
    const int64_t ntile = (int64_t)B * TH * TW;
    const float4* src = reinterpret_cast<const float4*>(M);
    float4 m[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) m[r][q] = src[((int64_t)(r * 4 + q) * ntile + tile) * C4 + c4];
    // A^T = [1 1 1 0; 0 1 -1 -1]
    auto add3 = [](float4 a, float4 b, float4 c) { return float4{a.x + b.x + c.x, a.y + b.y + c.y, a.z + b.z + c.z, a.w + b.w + c.w}; };
    auto sub3 = [](float4 a, float4 b, float4 c) { return float4{a.x - b.x - c.x, a.y - b.y - c.y, a.z - b.z - c.z, a.w - b.w - c.w}; };
    float4 t[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        t[0][q] = add3(m[0][q], m[1][q], m[2][q]);
        t[1][q] = sub3(m[1][q], m[2][q], m[3][q]);
    }
    const float4 bs = bias ? reinterpret_cast<const float4*>(bias)[c4] : float4{0.f, 0.f, 0.f, 0.f};
    float4* dst = reinterpret_cast<float4*>(y);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        float4 o[2];
        o[0] = add3(t[r][0], t[r][1], t[r][2]);
        o[1] = sub3(t[r][1], t[r][2], t[r][3]);
        const int hy = 2 * ty + r;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int wx = 2 * tx + q;
            if (hy < H && wx < W) {
                float4 v = {o[q].x + bs.x, o[q].y + bs.y, o[q].z + bs.z, o[q].w + bs.w};
                if (relu) v = float4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
                dst[((b * H + hy) * W + wx) * C4 + c4] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// F(4x4, 3x3): 6x6 input patches (stride 4, pad 1), 36 positions, 4x4 outputs per tile (Lavin & Gray matrices).
// 4x fewer multiply-adds than the direct form (2.25 per output instead of 9) and 44 % less transform traffic than
// F(2x2,3x3); the transforms now contain small integer / dyadic constants, fp32 error per layer ~5x the direct
// form's (measured 1.9e-7 vs 3.7e-8 mean relative), far inside the pool5 / output tolerances.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 f4ax(float a, float4 x) { return float4{a * x.x, a * x.y, a * x.z, a * x.w}; }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return float4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return float4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }

// t = B^T d for one 6-vector:  B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void bt6(const float4 (&d)[6], float4 (&t)[6]) {
    const float4 d42 = f4sub(d[4], f4ax(4.f, d[2]));          // d4 - 4 d2
    const float4 d31 = f4sub(d[3], f4ax(4.f, d[1]));          // d3 - 4 d1
    const float4 e42 = f4sub(d[4], d[2]);                     // d4 - d2
    const float4 e31 = f4ax(2.f, f4sub(d[3], d[1]));          // 2 (d3 - d1)
    t[0] = f4add(f4sub(f4ax(4.f, d[0]), f4ax(5.f, d[2])), d[4]);
    t[1] = f4add(d42, d31);
    t[2] = f4sub(d42, d31);
    t[3] = f4add(e42, e31);
    t[4] = f4sub(e42, e31);
    t[5] = f4add(f4sub(f4ax(4.f, d[1]), f4ax(5.f, d[3])), d[5]);
}

// y = A^T m for one 6-vector:  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void at6(const float4 (&m)[6], float4 (&y)[4]) {
    const float4 s12 = f4add(m[1], m[2]), d12 = f4sub(m[1], m[2]);
    const float4 s34 = f4add(m[3], m[4]), d34 = f4sub(m[3], m[4]);
    y[0] = f4add(f4add(m[0], s12), s34);
    y[1] = f4add(d12, f4ax(2.f, d34));
    y[2] = f4add(s12, f4ax(4.f, s34));
    y[3] = f4add(f4add(d12, f4ax(8.f, d34)), m[5]);
}

__global__ void __launch_bounds__(256)
wino_in6_kernel(const float* __restrict__ x, float* __restrict__ V, int B, int H, int W, int C4, int X4, int TH, int TW, int64_t total) {
    // C4: channel quads of V; X4 <= C4: channel quads of x (its pixel stride): quads [X4, C4) of V are zero (K padded for the GEMMs)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    const int64_t tile = i / C4;
    const int tx = (int)(tile % TW);
    const int ty = (int)((tile / TW) % TH);
    const int64_t b = tile / ((int64_t)TW * TH);
    const float4* src = reinterpret_cast<const float4*>(x);
    float4 t[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {   // transform along the columns of every input row
        const int hy = 4 * ty - 1 + r;
        float4 d[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int wx = 4 * tx - 1 + q;
            const bool ok = (unsigned)hy < (unsigned)H && (unsigned)wx < (unsigned)W && c4 < X4;
            d[q] = ok ? src[((b * H + hy) * W + wx) * X4 + c4] : float4{0.f, 0.f, 0.f, 0.f};
        }
        bt6(d, t[r]);
    }
    float4* dst = reinterpret_cast<float4*>(V);
    const int64_t ntile = (int64_t)B * TH * TW;
#pragma unroll
    for (int q = 0; q < 6; ++q) {   // then along the rows
        const float4 col[6] = {t[0][q], t[1][q], t[2][q], t[3][q], t[4][q], t[5][q]};
        float4 v[6];
        bt6(col, v);
#pragma unroll
        for (int r = 0; r < 6; ++r) nt_store(&dst[((int64_t)(r * 6 + q) * ntile + tile) * C4 + c4], v[r]);
    }
}

__global__ void __launch_bounds__(256)
wino_out6_kernel(const float* __restrict__ M, const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W, int C4,
                 int TH, int TW, int relu, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    const int64_t tile = i / C4;
    const int tx = (int)(tile % TW);
    const int ty = (int)((tile / TW) % TH);
    const int64_t b = tile / ((int64_t)TW * TH);
    const int64_t ntile = (int64_t)B * TH * TW;
    const float4* src = reinterpret_cast<const float4*>(M);
    float4 t[4][6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {   // A^T along the rows of every column
        float4 col[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) col[r] = nt_load(&src[((int64_t)(r * 6 + q) * ntile + tile) * C4 + c4]);  // read exactly once
        float4 o[4];
        at6(col, o);
#pragma unroll
        for (int p = 0; p < 4; ++p) t[p][q] = o[p];
    }
    const float4 bs = bias ? reinterpret_cast<const float4*>(bias)[c4] : float4{0.f, 0.f, 0.f, 0.f};
    float4* dst = reinterpret_cast<float4*>(y);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float4 o[4];
        at6(t[p], o);
        const int hy = 4 * ty + p;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int wx = 4 * tx + q;
            if (hy < H && wx < W) {
                float4 v = f4add(o[q], bs);
                if (relu) v = float4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
                nt_store(&dst[((b * H + hy) * W + wx) * C4 + c4], v);
            }
        }
    }
}

int wino_input_transform(const float* x, float* V, int B, int H, int W, int C, int m, hipStream_t s, int x_channels) {
    if (x_channels <= 0) x_channels = C;
    if (C % 4 || (m != 2 && m != 4) || x_channels % 4 || x_channels > C || (m != 4 && x_channels != C)) return MM_ERR_INVALID_ARG;
    const int TH = (H + m - 1) / m, TW = (W + m - 1) / m;
    const int64_t total = (int64_t)B * TH * TW * (C / 4);
    if (total <= 0) return MM_OK;
    prof_before(3, (double)B * C * 4.0 * ((double)H * W + (double)((m + 2) * (m + 2)) * TH * TW), s, m == 4 ? "wino_in6" : "wino_in");   // read x once, write the planes
    if (m == 4)
        hipLaunchKernelGGL(wino_in6_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, V, B, H, W, C / 4, x_channels / 4, TH,
                           TW, total);
    else
        hipLaunchKernelGGL(wino_in_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, V, B, H, W, C / 4, TH, TW, total);
    prof_after(3, s);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

int wino_output_transform(const float* M, const float* bias, float* y, int B, int H, int W, int Cout, int relu, int m, hipStream_t s) {
    if (Cout % 4 || (m != 2 && m != 4)) return MM_ERR_INVALID_ARG;
    const int TH = (H + m - 1) / m, TW = (W + m - 1) / m;
    const int64_t total = (int64_t)B * TH * TW * (Cout / 4);
    if (total <= 0) return MM_OK;
    prof_before(3, (double)B * Cout * 4.0 * ((double)H * W + (double)((m + 2) * (m + 2)) * TH * TW), s, m == 4 ? "wino_out6" : "wino_out");
    if (m == 4)
        hipLaunchKernelGGL(wino_out6_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, M, bias, y, B, H, W, Cout / 4, TH,
                           TW, relu, total);
    else
        hipLaunchKernelGGL(wino_out_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, M, bias, y, B, H, W, Cout / 4, TH, TW,
                           relu, total);
    prof_after(3, s);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

}  // namespace mm
