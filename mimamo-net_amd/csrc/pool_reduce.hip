// MaxPool2d(3, 2, pad 0, ceil_mode) of the stem output AND conv2_1's 1x1 reduce conv (64 -> 64, BN folded, ReLU) in one kernel.
//
// pool1_3x3_s2 is HBM-bound (reads the 6.6 GB stem output of a 2 048-frame step once, 1.6 ms); the reduce conv that follows reads
// the pooled tensor back (1.6 GB) for 52.6 GFLOP of work.  Here a wave pools 16 pixels and multiplies them by the 64 x 64 reduce
// matrix while the next pixels' loads are in flight: lane (l16, lg) computes the max over the 3x3 window for pixel l16, channels
// 16 cb + 4 lg .. + 3 -- which is exactly the B-operand layout of v_mfma_f32_16x16x4_f32 for the transposed product
// y[n][pixel] = sum_c W[n][c] x[c][pixel] (k = 4 lg + e inside a 16-channel block; the A operand uses the same order) -- so the
// pooled values could go from the max straight into the MFMA -- built first and measured: 3.17 ms against 1.60 + 0.66 for the two
// kernels, the 64-byte-per-pixel loads of that lane order cost more than the fusion saves.  The kernel below pools in the coalesced layout and goes through a wave-private LDS stage.  Outputs: x (the pooled tensor: the projection needs it)
// and y1 = relu(W x + b), both NHWC, 16-byte stores.
// Reference: the third-party ResNet50's pool1_3x3_s2 + conv2_1_1x1_reduce[_bn] + ReLU (api/resnet50_extractor.py:74-83 runs the
// whole net); x is bit-identical to maxpool3x3s2 (elementwise.hip), y1 differs from the conv engine's by summation order only.
#include "conv.h"

namespace mm {

typedef float f32x4p __attribute__((ext_vector_type(4)));

// Workgroup = 4 waves (32 KB of LDS, five workgroups per CU; eight waves / 48 KB measured slower: 2.17 vs 1.84 ms); LDS: the 64 x 64 matrix (16 KB, rows XOR-swizzled by 16-byte slot so that the A-fragment reads -- 16 rows, same
// k-quad -- are conflict free) + a 4 KB stage per wave ([16 pixels][64 channels], same swizzle).  Per group of 16 pixels a wave
//   1. pools in the layout of maxpool_kernel (lane = pixel % 4, channel quad: 256 contiguous bytes per pixel and load instruction),
//      stores x from those registers and drops the pooled float4 into its stage,
//   2. reads the stage back as the B operand (lane (l16, lg): pixel l16, channels 16 cb + 4 lg ..), the matrix as the A operand:
//      64 MFMAs for y[n][pixel] = sum_c W[n][c] x[c][pixel],
//   3. transposes relu(y + bias) through the stage so that the stores are 256 contiguous bytes per pixel again.
// A wave's LDS operations execute in issue order and the stage is wave-private: no workgroup barrier after the matrix load.
constexpr int PR_WAVES = 4, PR_GPW = 8;     // waves per workgroup, pixel groups per wave (measured 8 / 16 / 32 / 64: 1.82 / 1.84 / 1.91 / 2.04 ms per 2 048 frames)
// HP (round 5): `in` is the stem output already pooled HORIZONTALLY by the stem's own epilogue (conv_mfma.hip hpool: [N, H, WP, 64] with
// WP = W / 2 columns, column j = max over x in {2j, 2j+1, 2j+2}): a pixel takes three vertical taps of ITS column instead of nine pixels --
// a third of the load instructions, and 4.9 instead of 10.1 GB read per 2 048 frames (PMC: every input row but the even ones is shared by
// two output rows that different workgroups pool at different times; with nine taps that re-read hit HBM at full width).
template <bool HP>
__global__ void __launch_bounds__(PR_WAVES * 64)
maxpool_reduce64_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ x_out,
                        float* __restrict__ y_out, int64_t M, int H, int W, int Ho, int Wo, int relu, int64_t groups) {
    constexpr int C = 64;
    __shared__ __attribute__((aligned(16))) float wl[C * C];
    __shared__ __attribute__((aligned(16))) float stage_all[PR_WAVES][16 * C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, lg = lane >> 4;              // MFMA view: pixel / k group
    const int c4 = lane & 15, pp = lane >> 4;               // pooling view: channel quad / pixel inside a round of four
    for (int i = tid; i < C * C / 4; i += PR_WAVES * 64) {            // row r, quad q -> slot q ^ (r & 15)
        const int r = i >> 4, q = i & 15;
        *reinterpret_cast<f32x4p*>(wl + r * C + ((q ^ (r & 15)) << 2)) = *reinterpret_cast<const f32x4p*>(w + r * C + (q << 2));
    }
    __syncthreads();
    float* stage = stage_all[wave];
    f32x4p b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b4[j] = bias ? *reinterpret_cast<const f32x4p*>(bias + 16 * j + 4 * lg) : f32x4p{0.f, 0.f, 0.f, 0.f};
    const int64_t wave_id = (int64_t)blockIdx.x * PR_WAVES + wave, n_waves = (int64_t)gridDim.x * PR_WAVES;
    for (int64_t g = wave_id; g < groups; g += n_waves) {
        // ---- 1. pool four rounds of four pixels
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int pl = rr * 4 + pp;                     // pixel of the group
            const int64_t m = g * 16 + pl;
            const bool ok = m < M;
            const int64_t mm_ = ok ? m : M - 1;
            const int wo = (int)(mm_ % Wo);
            const int64_t t = mm_ / Wo;
            const int ho = (int)(t % Ho);
            const int64_t n = t / Ho;
            f32x4p v = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
            // ceil-mode windows are clipped at the border: a tap past it re-reads the last row / column of the window, which leaves
            // the maximum unchanged -- no branches, the nine loads of a round (36 per group) are issued as one batch
            if constexpr (HP) {
                f32x4p a[3];
                const int WP = W >> 1;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int hi = min(ho * 2 + r, H - 1);
                    a[r] = *reinterpret_cast<const f32x4p*>(in + ((n * H + hi) * WP + wo) * C + 4 * c4);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], a[k][e]);
            } else {
            f32x4p a[9];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int hi = min(ho * 2 + r, H - 1);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int wi = min(wo * 2 + q, W - 1);
                    a[r * 3 + q] = *reinterpret_cast<const f32x4p*>(in + ((n * H + hi) * W + wi) * C + 4 * c4);
                }
            }
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], a[k][e]);
            }
            if (ok) __builtin_nontemporal_store(v, reinterpret_cast<f32x4p*>(x_out + m * C + 4 * c4));
            *reinterpret_cast<f32x4p*>(stage + pl * C + ((c4 ^ pl) << 2)) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- 2. y[n][pixel] on the matrix cores
        f32x4p acc[4], bv[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) bv[cb] = *reinterpret_cast<const f32x4p*>(stage + l16 * C + (((4 * cb + lg) ^ l16) << 2));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = b4[j];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            f32x4p wf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const f32x4p*>(wl + (16 * j + l16) * C + (((4 * cb + lg) ^ l16) << 2));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][e], bv[cb][e], acc[j], 0, 0, 0);
        }
        // ---- 3. relu, transpose through the stage (the B reads above are complete: their values fed the MFMAs), coalesced stores
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4p o = acc[j];
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
            }
            *reinterpret_cast<f32x4p*>(stage + l16 * C + (((4 * j + lg) ^ l16) << 2)) = o;      // channels 16 j + 4 lg .. of pixel l16
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int pl = rr * 4 + pp;
            const int64_t m = g * 16 + pl;
            const f32x4p o = *reinterpret_cast<const f32x4p*>(stage + pl * C + ((c4 ^ pl) << 2));
            if (m < M) __builtin_nontemporal_store(o, reinterpret_cast<f32x4p*>(y_out + m * C + 4 * c4));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // the next group's stage writes come after these reads
    }
}

// in NHWC [N,H,W,64] -> x NHWC [N,Ho,Wo,64] = MaxPool2d(3, 2, 0, ceil_mode as given by Ho / Wo), y = relu?(w x + bias), w [64][64]
int maxpool_reduce64(const float* in, const float* w, const float* bias, float* x_out, float* y_out, int64_t N, int H, int W, int Ho, int Wo,
                     int relu, hipStream_t s, int hp) {
    if (!in || !w || !x_out || !y_out || N < 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return MM_ERR_INVALID_ARG;
    if (hp && (W % 2 || Wo > W / 2)) return MM_ERR_INVALID_ARG;
    const int64_t M = N * Ho * Wo;
    if (M <= 0) return MM_OK;
    const int64_t groups = (M + 15) / 16;
    // a workgroup loads the matrix once (16 KB from L2) and its waves walk PR_GPW pixel groups each: enough workgroups to fill the chip
    // several times over, the matrix load ~1.5 % of the traffic
    int64_t blocks = (groups + PR_WAVES * PR_GPW - 1) / (PR_WAVES * PR_GPW);
    if (blocks < 1) blocks = 1;
    if (blocks > 0x7fffffff) return MM_ERR_INVALID_ARG;
    // algorithmic bytes: the (horizontally pooled) stem output read once, x and y1 written
    prof_before(4, (double)N * 64 * 4.0 * ((double)H * (hp ? W / 2 : W) + 2.0 * Ho * Wo), s, hp ? "vpool+reduce64" : "maxpool+reduce64");
    if (hp)
        hipLaunchKernelGGL(maxpool_reduce64_kernel<true>, dim3((unsigned)blocks), dim3(PR_WAVES * 64), 0, s, in, w, bias, x_out, y_out, M, H, W, Ho, Wo, relu, groups);
    else
        hipLaunchKernelGGL(maxpool_reduce64_kernel<false>, dim3((unsigned)blocks), dim3(PR_WAVES * 64), 0, s, in, w, bias, x_out, y_out, M, H, W, Ho, Wo, relu, groups);
    prof_after(4, s);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

}  // namespace mm
