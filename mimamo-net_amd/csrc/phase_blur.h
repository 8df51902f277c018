// Pieces shared by the two kernels of the fused phase stage (pyramid_frames.hip: once per unique frame; phase_frames.hip: once per
// window): plane geometry, the layout of a frame's planes in the workspace, and the separable 11-tap Gaussian passes
// (amplitude_based_gaussian_blur, api/utils/phase_utils.py:78-90 -- un-normalised exp(-(x^2+y^2)/8), zero padding 5, quirk Q5).
#pragma once
#include <hip/hip_runtime.h>

namespace mm {
namespace blur {

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
    static constexpr int IN_PLANE = W * W + 2 * PADX;         // un-padded rows (lane-linear, conflict free) + slack
    static constexpr int TMP_PLANE = (W + 2 * R) * W;         // zero rows above / below
    static constexpr int PLANE = W * W;
    // per (frame, band) planes in the workspace, floats: mag, B = blur(mag phase) / blur(mag), R = 1 / blur(mag), phase
    static constexpr int FRAME_FLOATS = 4 * PLANE;
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

// The same pass with the five 16-byte chunk offsets (floats, relative to the plane) made ONCE per thread by row_chunk_offsets: a chunk outside
// the row points at the plane's slack [W * W, W * W + 16), which the caller keeps ZERO -- no select on the address or on the data in the frame
// loop (round 6: 20 v_cndmask per frame and thread in the window kernels, and vector instructions are what those kernels are bound by).  Same
// values into the same fused multiply-adds in the same order as row_pass: bit-identical.
template <int W>
__device__ __forceinline__ void row_chunk_offsets(int y, int x0, int (&off)[(PX + 2 * PADX) / 4]) {
#pragma unroll
    for (int q = 0; q < (PX + 2 * PADX) / 4; ++q) {
        const int xs = x0 - PADX + 4 * q;
        off[q] = (xs >= 0 && xs < W) ? y * W + xs : W * W + 4 * (q & 3);
    }
}
template <int W>
__device__ __forceinline__ void row_pass_pre(const float* in, const int (&off)[(PX + 2 * PADX) / 4], float (&h)[PX]) {
    float v[PX + 2 * PADX];
#pragma unroll
    for (int q = 0; q < (PX + 2 * PADX) / 4; ++q) {
        // (every offset is a multiple of four floats and the planes are 16-byte aligned: say so, or hipcc splits the read into b32 / b64 pieces)
        const float4 a = *reinterpret_cast<const float4*>(__builtin_assume_aligned(in + off[q], 16));
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

}  // namespace blur
}  // namespace mm
