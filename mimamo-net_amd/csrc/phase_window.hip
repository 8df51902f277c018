// Inter-frame phase difference of a 13-frame window of band coefficients, fused on gfx950.
//
// Replaces Phase_Difference_Extractor.extract (api/phase_difference_extractor.py:93-134):
//   phase = atan2(im, re); mag = sqrt(im^2 + re^2) + 1e-10                      (:100-104)
//   temporal unwrap over the window, torch.fmod semantics                         (phase_utils.py:5-20)
//   amplitude-weighted 11x11 Gaussian blur, zero padding: conv(mag*phase)/conv(mag) (phase_utils.py:78-90)
//   temporal difference, spatial-mean subtraction, clamp to +-5 pi                (:115-116,130-133)
// One workgroup owns one (window, band): every thread keeps 4 horizontally adjacent pixels of the
// W x W plane in registers across the 13 frames (unwrap state, previous blurred phase, the 12
// differences), the two planes to blur go through LDS as a separable row/column stencil, and the
// 12 spatial means are reduced wavefront-first.  ~25 full-tensor elementwise launches and one
// device->host sync of the reference (:105) become one launch.
//
// Reproduced on purpose: fmod-based unwrap that only corrects positive jumps (quirk Q2), the
// un-normalised Gaussian and zero padding (Q5).  Not reproduced: the mag<=0 host assert (Q11)
// and the dead mean-centring of the denoised phase (Q12).
#include "mm_common.h"
#include "phase_math.h"

namespace mm {

constexpr int P = 13;          // window length (num_phase + 1)
constexpr int TAP = 11, R = 5; // gaussian_kernel(std=2, tap=11), phase_utils.py:108-115
constexpr int PX = 4;          // pixels per thread (one float4)
constexpr int PADX = 8;        // the row pass reads [x0-8, x0+12): two float4 slots of halo on each side

// (float)exp(-d^2/8), d = -5..5 (exact decimal expansions of the fp32 values); the 2-D kernel is its outer product.
// Compile-time constants: no per-device symbol upload, the taps fold into the FMAs as literals.
__device__ constexpr float c_gauss[TAP] = {0.043936934322118759f, 0.1353352814912796f, 0.32465246319770813f,
                                           0.60653066635131836f,  0.88249689340591431f, 1.0f,
                                           0.88249689340591431f,  0.60653066635131836f, 0.32465246319770813f,
                                           0.1353352814912796f,   0.043936934322118759f};

template <int W>
struct WinCfg {
    static constexpr int STRIPS = W / PX;          // strips per row
    static constexpr int ACTIVE = STRIPS * W;      // 576 (W=48) / 144 (W=24)
    static constexpr int NTHREADS = (ACTIVE + 63) / 64 * 64;
    // input planes are stored WITHOUT row padding (lane l owns 16-byte slot l, so consecutive lanes hit consecutive
    // banks on both the write and the row-pass reads; a padded 64-float row stride made 55 % of the LDS cycles bank
    // conflicts).  Halo slots that fall outside the row are zeroed by a select instead of by padding.
    static constexpr int LDI = W;
    static constexpr int IN_PLANE = W * LDI + 2 * PADX;   // + slack so clamped halo reads stay in bounds
    static constexpr int TMP_ROWS = W + 2 * R;     // zero rows above/below
    static constexpr int TMP_PLANE = TMP_ROWS * W;
    static constexpr int LDS_FLOATS = 2 * IN_PLANE + 2 * TMP_PLANE + 64 * (P - 1);
};

// POLAR: the planes already hold (phase, magnitude) pairs -- written by the pyramid kernel on the fused path, which
// computes atan2/sqrt once per unique frame instead of once per window containing it (13x).
template <int W, bool POLAR>
__global__ void __launch_bounds__(WinCfg<W>::NTHREADS)
phase_window_kernel(const float* __restrict__ coeff, const int32_t* __restrict__ ids, int64_t img_stride,
                    int64_t band_stride, float* __restrict__ out, int out_nhwc, int out_cstride, int out_coffset) {
    using C = WinCfg<W>;
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    float* in_num = lds;
    float* in_den = in_num + C::IN_PLANE;
    float* tmp_num = in_den + C::IN_PLANE;
    float* tmp_den = tmp_num + C::TMP_PLANE;
    float* red = tmp_den + C::TMP_PLANE;  // [nwaves][P-1]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t j = blockIdx.x >> 1;
    const int band = blockIdx.x & 1;
    const bool active = tid < C::ACTIVE;
    const int y = active ? tid / C::STRIPS : 0;
    const int x0 = active ? (tid - y * C::STRIPS) * PX : 0;

    // zero the pads once (interior is rewritten every frame)
    for (int i = tid; i < 2 * C::IN_PLANE + 2 * C::TMP_PLANE; i += C::NTHREADS) lds[i] = 0.f;
    __syncthreads();

    const float PI_F = 3.14159265358979323846f;       // float(math.pi)
    const float TWO_PI_F = 6.28318530717958647692f;   // float(2*math.pi)
    float prev_phase[PX], cum[PX], prev_blur[PX], d[P - 1][PX], part[P - 1];
#pragma unroll
    for (int k = 0; k < P - 1; ++k) part[k] = 0.f;

#pragma unroll
    for (int i = 0; i < P; ++i) {
        if (active) {
            const float* plane = coeff + (int64_t)ids[j * P + i] * img_stride + band * band_stride;
            const float4* src = reinterpret_cast<const float4*>(plane + (y * W + x0) * 2);
            const float4 a = src[0], b = src[1];
            const float re[PX] = {a.x, a.z, b.x, b.z}, im[PX] = {a.y, a.w, b.y, b.w};
            float num[PX], den[PX];
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                float ph, mag;
                if (POLAR) {
                    ph = re[p];
                    mag = im[p];
                } else {
                    to_polar(re[p], im[p], ph, mag);
                }
                float up = ph;
                if (i == 0) {
                    cum[p] = 0.f;
                } else {
                    // torch_unwrap: ddmod = fmod(dd + pi, 2 pi) - pi; (ddmod == -pi & dd > 0) -> pi;
                    // corr = ddmod - dd, zeroed where |dd| < pi; up = p + cumsum(corr)
                    // fmod(x, 2 pi) for x = dd + pi in [-pi, 3 pi] (dd is a difference of two atan2 values) is x
                    // when x < 2 pi (C fmod keeps the sign of x) and x - 2 pi otherwise, an exact subtraction
                    // (Sterbenz) -- bit-identical to fmodf without its ~25-instruction expansion.
                    const float dd = ph - prev_phase[p];
                    const float xs = dd + PI_F;
                    float ddmod = (xs >= TWO_PI_F ? xs - TWO_PI_F : xs) - PI_F;
                    if (ddmod == -PI_F && dd > 0.f) ddmod = PI_F;
                    float corr = ddmod - dd;
                    if (fabsf(dd) < PI_F) corr = 0.f;
                    cum[p] += corr;
                    up = ph + cum[p];
                }
                prev_phase[p] = ph;
                num[p] = mag * up;
                den[p] = mag;
            }
            *reinterpret_cast<float4*>(in_num + y * C::LDI + x0) = float4{num[0], num[1], num[2], num[3]};
            *reinterpret_cast<float4*>(in_den + y * C::LDI + x0) = float4{den[0], den[1], den[2], den[3]};
        }
        __syncthreads();
        // ---- row pass: tmp[y][x] = sum_d g[d] in[y][x+d]
        if (active) {
            float vn[PX + 2 * PADX], vd[PX + 2 * PADX];  // [x0-8, x0+12)
            const float4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < (PX + 2 * PADX) / 4; ++q) {
                const int xs = x0 - PADX + 4 * q;                      // first float of this slot
                const bool in_row = xs >= 0 && xs < W;                 // zero padding of the 11-tap blur (Q5)
                const int off = y * C::LDI + (in_row ? xs : x0);
                float4 a = *reinterpret_cast<const float4*>(in_num + off);
                float4 b = *reinterpret_cast<const float4*>(in_den + off);
                if (!in_row) { a = zero4; b = zero4; }
                vn[4 * q] = a.x; vn[4 * q + 1] = a.y; vn[4 * q + 2] = a.z; vn[4 * q + 3] = a.w;
                vd[4 * q] = b.x; vd[4 * q + 1] = b.y; vd[4 * q + 2] = b.z; vd[4 * q + 3] = b.w;
            }
            float hn[PX], hd[PX];
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                float sn = 0.f, sd = 0.f;
#pragma unroll
                for (int t = 0; t < TAP; ++t) {
                    sn = fmaf(c_gauss[t], vn[PADX - R + p + t], sn);
                    sd = fmaf(c_gauss[t], vd[PADX - R + p + t], sd);
                }
                hn[p] = sn;
                hd[p] = sd;
            }
            *reinterpret_cast<float4*>(tmp_num + (y + R) * W + x0) = float4{hn[0], hn[1], hn[2], hn[3]};
            *reinterpret_cast<float4*>(tmp_den + (y + R) * W + x0) = float4{hd[0], hd[1], hd[2], hd[3]};
        }
        __syncthreads();
        // ---- column pass + ratio + temporal difference
        if (active) {
            float sn[PX] = {0.f, 0.f, 0.f, 0.f}, sd[PX] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < TAP; ++t) {
                const float4 a = *reinterpret_cast<const float4*>(tmp_num + (y + t) * W + x0);
                const float4 b = *reinterpret_cast<const float4*>(tmp_den + (y + t) * W + x0);
                const float gk = c_gauss[t];
                sn[0] = fmaf(gk, a.x, sn[0]); sn[1] = fmaf(gk, a.y, sn[1]); sn[2] = fmaf(gk, a.z, sn[2]); sn[3] = fmaf(gk, a.w, sn[3]);
                sd[0] = fmaf(gk, b.x, sd[0]); sd[1] = fmaf(gk, b.y, sd[1]); sd[2] = fmaf(gk, b.z, sd[2]); sd[3] = fmaf(gk, b.w, sd[3]);
            }
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                const float blur = sn[p] / sd[p];
                if (i > 0) {
                    d[i - 1][p] = blur - prev_blur[p];
                    part[i - 1] += d[i - 1][p];
                }
                prev_blur[p] = blur;
            }
        }
        // (the next frame's writes to in_* are separated from this frame's row-pass reads by the
        //  barrier above; its tmp_* writes from these column reads by the barrier after its own
        //  in_* stores)
    }

    // ---- spatial means of the 12 difference planes: wave shuffle reduce, then across waves
#pragma unroll
    for (int k = 0; k < P - 1; ++k) {
        float v = active ? part[k] : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave * (P - 1) + k] = v;
    }
    __syncthreads();
    constexpr int NWAVES = C::NTHREADS / 64;
    const float LIM = 5.f * PI_F;
    if (active) {
        float mean[P - 1];
#pragma unroll
        for (int k = 0; k < P - 1; ++k) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) s += red[w * (P - 1) + k];
            mean[k] = s * (1.0f / (W * W));
        }
        if (!out_nhwc) {
            // reference layout [J, band*12 + k, y, x]: one float4 (4 adjacent pixels) per channel plane
#pragma unroll
            for (int k = 0; k < P - 1; ++k) {
                float o[PX];
#pragma unroll
                for (int p = 0; p < PX; ++p) o[p] = fminf(fmaxf(d[k][p] - mean[k], -LIM), LIM);
                const int c = band * (P - 1) + k;
                float* dst = out + ((j * (2 * (P - 1)) + c) * W + y) * W + x0;
                *reinterpret_cast<float4*>(dst) = float4{o[0], o[1], o[2], o[3]};
            }
        } else {
            // channels-last: this band's 12 channels of a pixel are 48 contiguous bytes -> three 16-byte stores
            // per pixel (scalar channel-strided stores amplified HBM writes 9x: every 4-byte store dirtied a sector)
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                float* dst = out + ((j * W + y) * W + x0 + p) * out_cstride + out_coffset + band * (P - 1);
#pragma unroll
                for (int q = 0; q < (P - 1) / 4; ++q) {
                    float4 v;
                    v.x = fminf(fmaxf(d[4 * q][p] - mean[4 * q], -LIM), LIM);
                    v.y = fminf(fmaxf(d[4 * q + 1][p] - mean[4 * q + 1], -LIM), LIM);
                    v.z = fminf(fmaxf(d[4 * q + 2][p] - mean[4 * q + 2], -LIM), LIM);
                    v.w = fminf(fmaxf(d[4 * q + 3][p] - mean[4 * q + 3], -LIM), LIM);
                    reinterpret_cast<float4*>(dst)[q] = v;
                }
            }
        }
    }
}

int launch_phase_window(const float* coeff, const int32_t* ids, int64_t img_stride, int64_t band_stride, int64_t J,
                        int W, float* out, int out_nhwc, int out_cstride, int out_coffset, int polar, hipStream_t stream) {
    if (J <= 0) return MM_OK;
    const dim3 grid((unsigned)(2 * J));
    prof_before(2, (double)J * 2 * (P - 1) * W * W * 4, stream, "phase_window");  // algorithmic write: 24 phase-difference planes
#define MM_WIN(WW, PP)                                                                                                  \
    hipLaunchKernelGGL((phase_window_kernel<WW, PP>), grid, dim3(WinCfg<WW>::NTHREADS), 0, stream, coeff, ids, img_stride, \
                       band_stride, out, out_nhwc, out_cstride, out_coffset)
    if (W == 48 && polar) MM_WIN(48, true);
    else if (W == 48) MM_WIN(48, false);
    else if (W == 24 && polar) MM_WIN(24, true);
    else if (W == 24) MM_WIN(24, false);
    else return MM_ERR_UNSUPPORTED;
#undef MM_WIN
    prof_after(2, stream);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

}  // namespace mm
