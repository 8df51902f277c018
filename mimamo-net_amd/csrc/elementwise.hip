// Memory-bound helpers around the conv engine (layout change, pooling, GRU gates).  All are
// coalesced along the channel dimension of NHWC tensors (float4 where the shape allows).
#include "conv.h"

namespace mm {

// ---- NCHW -> NHWC through an LDS tile (reads coalesced along HW, writes along C) ------------------
__global__ void __launch_bounds__(256)
nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW, int cstride, int coff, int cpad) {
    __shared__ float tile[32][33];
    const int64_t n = blockIdx.z;
    const int hw0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, hw = hw0 + tx;
        tile[r][tx] = (c < C && hw < HW) ? in[(n * C + c) * HW + hw] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int hw = hw0 + r, c = c0 + tx;
        if (hw < HW && c < cpad) out[(n * HW + hw) * cstride + coff + c] = tile[tx][r];
    }
}

// 3-channel planes -> channels-last padded to 4 (the ResNet input): three coalesced plane reads, one float4 store
__global__ void __launch_bounds__(256)
nchw3_to_nhwc4_kernel(const float* __restrict__ in, float4* __restrict__ out, int64_t total, int HW) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t n = i / HW;
    const int hw = (int)(i - n * HW);
    const float* p = in + n * 3 * HW + hw;
    out[i] = float4{p[0], p[HW], p[2 * (int64_t)HW], 0.f};
}

// 3-channel planes [N,3,S,S] -> packed channels-last rows with a zero border of `pad` pixels [N,S+2pad,S+2pad,3] (the stem's input
// layout, conv_mfma.hip KMODE 5): one thread per output pixel, three coalesced plane reads
__global__ void __launch_bounds__(256)
nchw3_to_bordered_nhwc3_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t total, int S, int pad) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int SP = S + 2 * pad;
    const int64_t n = i / ((int64_t)SP * SP);
    const int rem = (int)(i - n * (int64_t)SP * SP);
    const int y = rem / SP - pad, x = rem - (rem / SP) * SP - pad;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if ((unsigned)y < (unsigned)S && (unsigned)x < (unsigned)S) {
        const float* p = in + (n * 3 * S + y) * (int64_t)S + x;
        v0 = p[0]; v1 = p[(int64_t)S * S]; v2 = p[2 * (int64_t)S * S];
    }
    float* o = out + i * 3;
    o[0] = v0; o[1] = v1; o[2] = v2;
}

int nchw3_to_bordered_nhwc3(const float* in, float* out, int64_t N, int S, int pad, hipStream_t s) {
    if (N <= 0) return MM_OK;
    const int64_t total = N * (int64_t)(S + 2 * pad) * (S + 2 * pad);
    prof_before(4, (double)N * 4.0 * (3.0 * S * S + 3.0 * (S + 2 * pad) * (S + 2 * pad)), s, "nchw3_to_bordered_nhwc3");
    hipLaunchKernelGGL(nchw3_to_bordered_nhwc3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, total, S, pad);
    prof_after(4, s);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

int nchw_to_nhwc(const float* in, float* out, int64_t N, int C, int HW, int cstride, int coff, int cpad, hipStream_t s) {
    if (N <= 0) return MM_OK;
    if (cpad < C) cpad = C;
    if (C == 3 && cpad == 4 && cstride == 4 && coff == 0) {
        const int64_t total = N * HW;
        prof_before(4, (double)total * 4.0 * 7.0, s, "nchw3_to_nhwc4");
        hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in,
                           reinterpret_cast<float4*>(out), total, HW);
        prof_after(4, s);
        MM_LAUNCH_CHECK();
        return MM_OK;
    }
    dim3 grid((HW + 31) / 32, (cpad + 31) / 32, (unsigned)N);
    prof_before(4, (double)N * HW * 4.0 * (C + cpad), s, "nchw_to_nhwc");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, s, in, out, C, HW, cstride, coff, cpad);
    prof_after(4, s);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

// ---- MaxPool 3x3 stride 2 pad 0, ceil_mode windows clipped at the border ---------------------------
__global__ void __launch_bounds__(256)
maxpool_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t total4, int H, int W, int C4, int Ho, int Wo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c4 = (int)(i % C4);
    int64_t t = i / C4;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const int64_t n = t / Ho;
    const float4* src = reinterpret_cast<const float4*>(in);
    float4 m = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
    for (int r = 0; r < 3; ++r) {
        const int hi = ho * 2 + r;
        if (hi >= H) break;
        for (int q = 0; q < 3; ++q) {
            const int wi = wo * 2 + q;
            if (wi >= W) break;
            const float4 v = src[((n * H + hi) * W + wi) * C4 + c4];
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    reinterpret_cast<float4*>(out)[i] = m;
}

int maxpool3x3s2(const float* in, float* out, int64_t N, int H, int W, int C, int Ho, int Wo, hipStream_t s) {
    if (C % 4) return MM_ERR_INVALID_ARG;
    const int64_t total4 = N * Ho * Wo * (C / 4);
    if (total4 <= 0) return MM_OK;
    prof_before(4, (double)N * C * 4.0 * ((double)H * W + (double)Ho * Wo), s, "maxpool3x3s2");   // every input read once, output written
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, in, out, total4, H, W, C / 4, Ho, Wo);
    prof_after(4, s);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

// ---- global average pool ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
avgpool_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t total, int HW, int C, int out_cstride, int out_coff, int relu) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int64_t n = i / C;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += in[(n * HW + p) * C + c];
    s *= 1.0f / HW;
    if (relu) s = fmaxf(s, 0.f);
    out[n * out_cstride + out_coff + c] = s;
}

// four channels per lane, seven pixels' loads in flight (round 5: the scalar form above is latency-bound -- 3.8 TB/s on the 7x7x2048 pool5 map);
// every channel is still summed pixel by pixel in the same order: bit-identical
__global__ void __launch_bounds__(256)
avgpool4_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t total4, int HW, int C4, int out_cstride, int out_coff, int relu) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c4 = (int)(i % C4);
    const int64_t n = i / C4;
    const float4* src = reinterpret_cast<const float4*>(in) + n * HW * C4 + c4;
    float4 s = {0.f, 0.f, 0.f, 0.f};
    int p = 0;
    for (; p + 7 <= HW; p += 7) {
        float4 v[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) v[u] = src[(int64_t)(p + u) * C4];
#pragma unroll
        for (int u = 0; u < 7; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; p < HW; ++p) {
        const float4 v = src[(int64_t)p * C4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float r = 1.0f / HW;
    s.x *= r; s.y *= r; s.z *= r; s.w *= r;
    if (relu) s = float4{fmaxf(s.x, 0.f), fmaxf(s.y, 0.f), fmaxf(s.z, 0.f), fmaxf(s.w, 0.f)};
    *reinterpret_cast<float4*>(out + n * out_cstride + out_coff + 4 * c4) = s;
}

int avgpool_hw(const float* in, float* out, int64_t N, int HW, int C, int out_cstride, int out_coff, int relu, hipStream_t s) {
    const int64_t total = N * C;
    if (total <= 0) return MM_OK;
    prof_before(4, (double)total * 4.0 * (HW + 1), s, "avgpool");
    if (((C | out_cstride | out_coff) & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
        hipLaunchKernelGGL(avgpool4_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, in, out, total / 4, HW, C / 4, out_cstride, out_coff, relu);
    else
    hipLaunchKernelGGL(avgpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, total, HW, C, out_cstride, out_coff, relu);
    prof_after(4, s);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

// ---- GRU cell (nn.GRU gate order r, z, n; mimamo_net.py:119) ----------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void __launch_bounds__(256)
gru_gates_kernel(const float* __restrict__ gi, int gi_stride, int gi_off, const float* __restrict__ gh,
                 const float* __restrict__ bhh, const float* __restrict__ h_prev, int hp_stride, int hp_off,
                 float* __restrict__ h_out, int out_stride, int out_off, int64_t total, int H) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % H);
    const int64_t b = i / H;
    const float* g = gi + b * gi_stride + gi_off;
    float hr, hz, hn, hp;
    if (gh) {
        const float* q = gh + b * 3 * H;
        hr = q[j]; hz = q[H + j]; hn = q[2 * H + j];
        hp = h_prev[b * hp_stride + hp_off + j];
    } else {  // first step: h == 0  ->  W_hh h + b_hh = b_hh
        hr = bhh[j]; hz = bhh[H + j]; hn = bhh[2 * H + j];
        hp = 0.f;
    }
    const float r = sigmoidf_(g[j] + hr);
    const float z = sigmoidf_(g[H + j] + hz);
    const float n = tanhf(g[2 * H + j] + r * hn);
    h_out[b * out_stride + out_off + j] = (1.f - z) * n + z * hp;
}

int gru_gates(const float* gi, int gi_stride, int gi_off, const float* gh, const float* bhh, const float* h_prev,
              int hp_stride, int hp_off, float* h_out, int out_stride, int out_off, int64_t Bt, int H, hipStream_t s) {
    const int64_t total = Bt * H;
    if (total <= 0) return MM_OK;
    prof_before(4, (double)total * 4.0 * (gh ? 8.0 : 4.0), s, "gru_gates");
    hipLaunchKernelGGL(gru_gates_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, gi, gi_stride, gi_off, gh, bhh,
                       h_prev, hp_stride, hp_off, h_out, out_stride, out_off, total, H);
    prof_after(4, s);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

}  // namespace mm
