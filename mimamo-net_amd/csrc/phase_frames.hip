// Phase difference of 13-frame windows from de-duplicated per-frame planes (the fused path behind mm_phase_diff_frames).
//
// Reference arithmetic (Phase_Difference_Extractor.extract, api/phase_difference_extractor.py:93-134, per window and band):
//   up_i  = phase_i + acc_i,  acc_i = cumulative torch_unwrap correction since the window's first frame (phase_utils.py:5-20)
//   out_i = blur(mag_i * up_i) / blur(mag_i)          un-normalised 11x11 Gaussian, zero padding (phase_utils.py:78-90)
//   d_i   = out_{i+1} - out_i, minus its spatial mean, clamped to +-5 pi                                (:115-116,130-133)
// phase_window.hip evaluates that literally: two 11x11 blurs per (window, frame) = 13 x per unique frame, because acc differs
// between the up-to-13 windows that contain a frame.  But the blur is linear and acc is a multiple of 2 pi:
//   blur(mag (phase + acc)) / blur(mag) = blur(mag phase) / blur(mag)  +  blur(mag acc) / blur(mag)
//                                       = B_i                          +  blur(mag_i * acc_i) * R_i
// with B_i, R_i = 1 / blur(mag_i) per UNIQUE frame (phase_frame_kernel, once) and ONE blur per (window, frame) left
// (phase_window2_kernel); where no pixel of a window has wrapped yet (always true for its first frame, and for whole windows
// of slowly moving faces) that blur is skipped.  torch_unwrap only corrects positive jumps (fmod keeps the dividend's sign,
// quirk Q2): corr_t = -2 pi where dd_t = phase_t - phase_{t-1} >= pi (decided on the same fp32 expression as the reference,
// (dd + pi) >= 2 pi), else 0 up to a rounding residue of (dd + pi) - pi - dd <= 2.4e-7 that the reference accumulates and this
// form drops -- far inside the phase tolerance (tests: 1e-3 max / 3e-4 p99.99).  So a frame carries one wrap FLAG per pixel
// (w_t, vs its predecessor in the stack) and a window sums the flags of its frames: acc = -2 pi * k.  Window ids are clamped
// inside a video (snippet_sampler.py:144-152), so consecutive ids are equal (repeated edge frame: dd = 0, no wrap) or
// consecutive frames.
#include "mm_common.h"
#include "phase_math.h"

namespace mm {

namespace {
constexpr int P = 13, TAP = 11, R = 5, PX = 4, PADX = 8;
__device__ constexpr float c_g[TAP] = {0.043936934322118759f, 0.1353352814912796f, 0.32465246319770813f,
                                       0.60653066635131836f,  0.88249689340591431f, 1.0f,
                                       0.88249689340591431f,  0.60653066635131836f, 0.32465246319770813f,
                                       0.1353352814912796f,   0.043936934322118759f};

template <int W>
struct Cfg {
    static constexpr int STRIPS = W / PX;
    static constexpr int ACTIVE = STRIPS * W;                 // 576 (W = 48) / 144 (W = 24)
    static constexpr int NTHREADS = (ACTIVE + 63) / 64 * 64;
    static constexpr int IN_PLANE = W * W + 2 * PADX;         // un-padded rows (lane-linear, conflict free) + slack for clamped halo reads
    static constexpr int TMP_PLANE = (W + 2 * R) * W;         // zero rows above / below
    static constexpr int PLANE = W * W;
    // per (frame, band) planes in the workspace, floats: mag, B, R, then W*W wrap-flag bytes
    static constexpr int FRAME_FLOATS = 3 * PLANE + PLANE / 4;
};

// separable 11-tap pass over rows: in[y][x0-8 .. x0+12) -> h[4]; slots outside the row are zero (Q5 zero padding)
template <int W>
__device__ __forceinline__ void row_pass(const float* in, int y, int x0, float (&h)[PX]) {
    float v[PX + 2 * PADX];
#pragma unroll
    for (int q = 0; q < (PX + 2 * PADX) / 4; ++q) {
        const int xs = x0 - PADX + 4 * q;
        const bool in_row = xs >= 0 && xs < W;
        float4 a = *reinterpret_cast<const float4*>(in + y * W + (in_row ? xs : x0));
        if (!in_row) a = float4{0.f, 0.f, 0.f, 0.f};
        v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
    }
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < TAP; ++t) s = fmaf(c_g[t], v[PADX - R + p + t], s);
        h[p] = s;
    }
}

template <int W>
__device__ __forceinline__ void col_pass(const float* tmp, int y, int x0, float (&s)[PX]) {
    s[0] = s[1] = s[2] = s[3] = 0.f;
#pragma unroll
    for (int t = 0; t < TAP; ++t) {
        const float4 a = *reinterpret_cast<const float4*>(tmp + (y + t) * W + x0);
        const float gk = c_g[t];
        s[0] = fmaf(gk, a.x, s[0]); s[1] = fmaf(gk, a.y, s[1]); s[2] = fmaf(gk, a.z, s[2]); s[3] = fmaf(gk, a.w, s[3]);
    }
}
}  // namespace

// ---- once per unique (frame, band): B = blur(mag phase) / blur(mag), R = 1 / blur(mag), mag, wrap flag vs the previous frame
template <int W>
__global__ void __launch_bounds__(Cfg<W>::NTHREADS)
phase_frame_kernel(const float* __restrict__ polar, int64_t img_stride, int64_t band_stride, float* __restrict__ fr, int64_t n) {
    using C = Cfg<W>;
    __shared__ __attribute__((aligned(16))) float lds[2 * C::IN_PLANE + 2 * C::TMP_PLANE];
    float* in_num = lds;
    float* in_den = in_num + C::IN_PLANE;
    float* tmp_num = in_den + C::IN_PLANE;
    float* tmp_den = tmp_num + C::TMP_PLANE;
    const int tid = threadIdx.x;
    const int64_t f = blockIdx.x >> 1;
    const int band = blockIdx.x & 1;
    const bool active = tid < C::ACTIVE;
    const int y = active ? tid / C::STRIPS : 0;
    const int x0 = active ? (tid - y * C::STRIPS) * PX : 0;
    for (int i = tid; i < 2 * C::IN_PLANE + 2 * C::TMP_PLANE; i += C::NTHREADS) lds[i] = 0.f;
    __syncthreads();
    const float PI_F = 3.14159265358979323846f, TWO_PI_F = 6.28318530717958647692f;
    float mag[PX], ph[PX];
    unsigned wbits = 0;
    if (active) {
        const float4* src = reinterpret_cast<const float4*>(polar + f * img_stride + band * band_stride + (y * W + x0) * 2);
        const float4 a = src[0], b = src[1];
        ph[0] = a.x; ph[1] = a.z; ph[2] = b.x; ph[3] = b.z;
        mag[0] = a.y; mag[1] = a.w; mag[2] = b.y; mag[3] = b.w;
        if (f > 0) {
            const float4* prv = reinterpret_cast<const float4*>(polar + (f - 1) * img_stride + band * band_stride + (y * W + x0) * 2);
            const float4 c = prv[0], d = prv[1];
            const float pp[PX] = {c.x, c.z, d.x, d.z};
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                const float dd = ph[p] - pp[p];
                if (dd + PI_F >= TWO_PI_F) wbits |= 1u << (8 * p);   // the jump torch_unwrap corrects by -2 pi
            }
        }
        *reinterpret_cast<float4*>(in_num + y * W + x0) = float4{mag[0] * ph[0], mag[1] * ph[1], mag[2] * ph[2], mag[3] * ph[3]};
        *reinterpret_cast<float4*>(in_den + y * W + x0) = float4{mag[0], mag[1], mag[2], mag[3]};
    }
    __syncthreads();
    if (active) {
        float hn[PX], hd[PX];
        row_pass<W>(in_num, y, x0, hn);
        row_pass<W>(in_den, y, x0, hd);
        *reinterpret_cast<float4*>(tmp_num + (y + R) * W + x0) = float4{hn[0], hn[1], hn[2], hn[3]};
        *reinterpret_cast<float4*>(tmp_den + (y + R) * W + x0) = float4{hd[0], hd[1], hd[2], hd[3]};
    }
    __syncthreads();
    if (active) {
        float sn[PX], sd[PX];
        col_pass<W>(tmp_num, y, x0, sn);
        col_pass<W>(tmp_den, y, x0, sd);
        float* o = fr + (f * 2 + band) * C::FRAME_FLOATS;
        const int px = y * W + x0;
        *reinterpret_cast<float4*>(o + px) = float4{mag[0], mag[1], mag[2], mag[3]};
        *reinterpret_cast<float4*>(o + C::PLANE + px) = float4{sn[0] / sd[0], sn[1] / sd[1], sn[2] / sd[2], sn[3] / sd[3]};
        *reinterpret_cast<float4*>(o + 2 * C::PLANE + px) = float4{1.0f / sd[0], 1.0f / sd[1], 1.0f / sd[2], 1.0f / sd[3]};
        reinterpret_cast<unsigned*>(o + 3 * C::PLANE)[px / 4] = wbits;
    }
}

// ---- per (window, band): 12 phase-difference planes from the frame planes.
//   A  every frame's B plane and wrap flags are requested up front (26 independent loads per thread, one latency instead of one
//      per barrier round); the flags are summed bytewise in one register per frame (k <= 12 per pixel), the difference planes start
//      as B_i - B_{i-1}, and the workgroup agrees on the first frame in which ANY of its pixels has wrapped (one LDS min, one barrier)
//   B  from that frame on, F frames per barrier round: blur(mag * (-2 pi k)) * R, whose frame-to-frame change is added to the planes.
//      Frames before it have blur(mag * 0) = 0: rounds that end before the first wrap are skipped (always the window's first frame,
//      whole windows of slowly moving faces)
//   C  spatial means, clamp, store.  NHWC pixels take 48 of their 96 bytes from this band: the rows go through LDS so that three
//      consecutive lanes write the 48 contiguous bytes of a pixel (a lane-per-pixel-strip store issues 64 separate 16-byte requests
//      per instruction: 0.16 of the 0.61 ms these kernels took)
#ifndef MM_PHASE_FPR
#define MM_PHASE_FPR 3      // frames per barrier round
#endif
template <int W>
__global__ void __launch_bounds__(Cfg<W>::NTHREADS)
phase_window2_kernel(const float* __restrict__ fr, const int32_t* __restrict__ ids, float* __restrict__ out, int out_nhwc,
                     int out_cstride, int out_coffset) {
    using C = Cfg<W>;
    constexpr int F = MM_PHASE_FPR;
    constexpr int RPG = W == 48 ? 16 : 12;                    // rows per store group: RPG * W * 12 floats staged at a time
    constexpr int G = W / RPG;
    constexpr int WORK = F * (C::IN_PLANE + C::TMP_PLANE);
    static_assert(RPG * W * (P - 1) <= WORK, "store staging fits the blur planes");
    __shared__ __attribute__((aligned(16))) float lds[WORK + 64 * (P - 1)];
    __shared__ int first_wrap;
    float* in_x = lds;                              // [F][IN_PLANE]
    float* tmp_x = in_x + F * C::IN_PLANE;          // [F][TMP_PLANE]
    float* red = lds + WORK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // consecutive windows share 12 of their 13 frames: keep them on one XCD (one L2) instead of spreading them over all eight
    const int logical = xcd_contiguous(blockIdx.x, gridDim.x);
    const int64_t j = logical >> 1;
    const int band = logical & 1;
    const bool active = tid < C::ACTIVE;
    const int y = active ? tid / C::STRIPS : 0;
    const int x0 = active ? (tid - y * C::STRIPS) * PX : 0;
    const int px = y * W + x0;
    for (int i = tid; i < WORK; i += C::NTHREADS) lds[i] = 0.f;    // the zero rows above / below tmp_x
    if (tid == 0) first_wrap = P;
    const float TWO_PI_F = 6.28318530717958647692f, PI_F = 3.14159265358979323846f;
    // ---- A
    const float* fo[P];
    float d[P - 1][PX];
    unsigned kb[P];
    {
        float4 bprev = {0.f, 0.f, 0.f, 0.f};
        int prev_id = -1;
        unsigned run = 0;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int id = ids[j * P + i];
            fo[i] = fr + ((int64_t)id * 2 + band) * C::FRAME_FLOATS;
            float4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (active) {
                b4 = *reinterpret_cast<const float4*>(fo[i] + C::PLANE + px);
                // a new frame: its wrap flags refer to the frame before it, which is prev_id (a repeated edge frame has dd = 0)
                if (i > 0 && id != prev_id) run += reinterpret_cast<const unsigned*>(fo[i] + 3 * C::PLANE)[px / 4];
            }
            kb[i] = run;
            if (i > 0) {
                d[i - 1][0] = b4.x - bprev.x; d[i - 1][1] = b4.y - bprev.y; d[i - 1][2] = b4.z - bprev.z; d[i - 1][3] = b4.w - bprev.w;
            }
            bprev = b4;
            prev_id = id;
        }
    }
    {
        int mine = P;                     // first frame in which one of this thread's pixels has wrapped (counts only grow)
#pragma unroll
        for (int i = P - 1; i >= 0; --i) mine = kb[i] != 0u ? i : mine;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mine = min(mine, __shfl_xor(mine, off, 64));
        __syncthreads();                  // first_wrap initialised, LDS zeroed
        if (lane == 0 && mine < P) atomicMin(&first_wrap, mine);
        __syncthreads();
    }
    const int first = first_wrap;
    // ---- B
    float cprev[PX] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int base = 0; base < P; base += F) {
        if (base + F <= first) continue;             // block-uniform: no pixel of the window has wrapped up to the round's last frame
        float4 r4[F];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int i = base + f;
            r4[f] = float4{0.f, 0.f, 0.f, 0.f};
            if (i >= P || !active) continue;
            const float4 m4 = *reinterpret_cast<const float4*>(fo[i] + px);
            r4[f] = *reinterpret_cast<const float4*>(fo[i] + 2 * C::PLANE + px);    // a pixel's blur also sums its neighbours' wraps
            const unsigned k = kb[i];
            *reinterpret_cast<float4*>(in_x + f * C::IN_PLANE + px) =
                float4{m4.x * (-TWO_PI_F * (float)(k & 255u)), m4.y * (-TWO_PI_F * (float)((k >> 8) & 255u)),
                       m4.z * (-TWO_PI_F * (float)((k >> 16) & 255u)), m4.w * (-TWO_PI_F * (float)(k >> 24))};
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int f = 0; f < F; ++f) {
                if (base + f >= P) continue;
                float h[PX];
                row_pass<W>(in_x + f * C::IN_PLANE, y, x0, h);
                *reinterpret_cast<float4*>(tmp_x + f * C::TMP_PLANE + (y + R) * W + x0) = float4{h[0], h[1], h[2], h[3]};
            }
        }
        __syncthreads();
        // (the next round's in_x stores are separated from this round's row-pass reads by the barrier above, its tmp_x stores from
        //  the column reads below by its own first barrier)
        if (active) {
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const int i = base + f;
                if (i >= P) continue;
                float sb[PX];
                col_pass<W>(tmp_x + f * C::TMP_PLANE, y, x0, sb);
                const float c[PX] = {sb[0] * r4[f].x, sb[1] * r4[f].y, sb[2] * r4[f].z, sb[3] * r4[f].w};
#pragma unroll
                for (int p = 0; p < PX; ++p) {
                    if (i > 0) d[i - 1][p] += c[p] - cprev[p];
                    cprev[p] = c[p];
                }
            }
        }
    }
    // ---- C: spatial means of the 12 difference planes: wave shuffle reduce, then across waves
#pragma unroll
    for (int k = 0; k < P - 1; ++k) {
        float v = active ? (d[k][0] + d[k][1]) + (d[k][2] + d[k][3]) : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave * (P - 1) + k] = v;
    }
    __syncthreads();                      // also: every blur read of the planes is done, they become the store staging
    constexpr int NWAVES = C::NTHREADS / 64;
    const float LIM = 5.f * PI_F;
    float mean[P - 1];
#pragma unroll
    for (int k = 0; k < P - 1; ++k) {
        float sm = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) sm += red[w * (P - 1) + k];
        mean[k] = sm * (1.0f / (W * W));
    }
    if (!out_nhwc) {
        if (active) {
#pragma unroll
            for (int k = 0; k < P - 1; ++k) {
                float ov[PX];
#pragma unroll
                for (int p = 0; p < PX; ++p) ov[p] = fminf(fmaxf(d[k][p] - mean[k], -LIM), LIM);
                float* dst = out + ((j * (2 * (P - 1)) + band * (P - 1) + k) * W + y) * W + x0;
                *reinterpret_cast<float4*>(dst) = float4{ov[0], ov[1], ov[2], ov[3]};
            }
        }
        return;
    }
    float4* stage = reinterpret_cast<float4*>(lds);      // [RPG * W pixels][3 float4]
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (active && y / RPG == g) {
#pragma unroll
            for (int p = 0; p < PX; ++p)
#pragma unroll
                for (int q = 0; q < (P - 1) / 4; ++q) {
                    float4 v;
                    v.x = fminf(fmaxf(d[4 * q][p] - mean[4 * q], -LIM), LIM);
                    v.y = fminf(fmaxf(d[4 * q + 1][p] - mean[4 * q + 1], -LIM), LIM);
                    v.z = fminf(fmaxf(d[4 * q + 2][p] - mean[4 * q + 2], -LIM), LIM);
                    v.w = fminf(fmaxf(d[4 * q + 3][p] - mean[4 * q + 3], -LIM), LIM);
                    stage[((y - g * RPG) * W + x0 + p) * 3 + q] = v;
                }
        }
        __syncthreads();
        for (int idx = tid; idx < RPG * W * 3; idx += C::NTHREADS) {
            const int pix = idx / 3, part = idx - pix * 3;
            float* dst = out + ((j * W + g * RPG) * W + pix) * out_cstride + out_coffset + band * (P - 1) + part * 4;
            *reinterpret_cast<float4*>(dst) = stage[idx];
        }
        if (g + 1 < G) __syncthreads();
    }
}

int64_t phase_frames_floats(int W, int64_t n) { return n * 2 * (W == 48 ? Cfg<48>::FRAME_FLOATS : Cfg<24>::FRAME_FLOATS); }

// polar planes [n][2][W][W][2] (phase, magnitude) -> frame planes; then windows -> out
int launch_phase_frames(const float* polar, int64_t img_stride, int64_t band_stride, float* fr, int64_t n, int W, hipStream_t s) {
    if (n <= 0) return MM_OK;
    const dim3 grid((unsigned)(2 * n));
    if (W == 48) hipLaunchKernelGGL(phase_frame_kernel<48>, grid, dim3(Cfg<48>::NTHREADS), 0, s, polar, img_stride, band_stride, fr, n);
    else if (W == 24) hipLaunchKernelGGL(phase_frame_kernel<24>, grid, dim3(Cfg<24>::NTHREADS), 0, s, polar, img_stride, band_stride, fr, n);
    else return MM_ERR_UNSUPPORTED;
    MM_LAUNCH_CHECK();
    return MM_OK;
}

int launch_phase_window2(const float* fr, const int32_t* ids, int64_t J, int W, float* out, int out_nhwc, int out_cstride,
                         int out_coffset, hipStream_t s) {
    if (J <= 0) return MM_OK;
    const dim3 grid((unsigned)(2 * J));
    if (W == 48)
        hipLaunchKernelGGL(phase_window2_kernel<48>, grid, dim3(Cfg<48>::NTHREADS), 0, s, fr, ids, out, out_nhwc, out_cstride, out_coffset);
    else if (W == 24)
        hipLaunchKernelGGL(phase_window2_kernel<24>, grid, dim3(Cfg<24>::NTHREADS), 0, s, fr, ids, out, out_nhwc, out_cstride, out_coffset);
    else return MM_ERR_UNSUPPORTED;
    MM_LAUNCH_CHECK();
    return MM_OK;
}

}  // namespace mm
