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
// with B_i, R_i = 1 / blur(mag_i) per UNIQUE frame (pyramid_frame_kernel, pyramid_frames.hip: the pyramid's epilogue) and ONE blur
// per (window, frame) left (phase_window2_kernel, here); where no pixel of a window has wrapped yet (always true for its first
// frame, and for whole windows of slowly moving faces) that blur is skipped.  torch_unwrap only corrects positive jumps (fmod keeps
// the dividend's sign, quirk Q2): corr_t = -2 pi where dd_t = phase_t - phase_{t-1} exceeds pi -- decided on the reference's own
// fp32 expression, (dd + pi) > 2 pi: at exact equality fmod gives 0, ddmod = -pi is reset to +pi (dd > 0) and the correction is
// ~0, not -2 pi -- else 0 up to a rounding residue of (dd + pi) - pi - dd <= 2.4e-7 that the reference accumulates and this form
// drops (far inside the phase tolerance; tests: 1e-3 max / 3e-4 p99.99 and a tight regression bound).  A window counts, per pixel,
// the corrected steps between ITS consecutive frames from the frames' phase planes: acc = -2 pi * k.  Any id pattern is therefore
// handled like the reference handles it (a repeated edge frame of a clamped window, snippet_sampler.py:144-152, has dd = 0).
#include <cstdlib>
#include "mm_common.h"
#include "phase_math.h"
#include "phase_blur.h"

#ifndef MM_PW_ABLATE
#define MM_PW_ABLATE 0           // measurement builds only (tools/): 1 no output stores, 2 no blur rounds, 4 every window reads window 0's frames
#endif
#ifndef MM_PW_SPLIT_F
#define MM_PW_SPLIT_F 2         // frames per barrier round of the time-split kernel: two padded planes = 65 KB of LDS, two workgroups per CU
#endif
#ifndef MM_PW_CHUNK
#define MM_PW_CHUNK 4   // frames whose plane loads the split kernel keeps in flight at once
#endif
#ifndef MM_PW_WAVES
#define MM_PW_WAVES 5   // waves per SIMD the split kernel is held to (HIP's second launch-bound): 5 -> 96 registers -> two nine-wave workgroups per CU
#endif

namespace mm {

using namespace blur;

// Sum over the 64 lanes of a wave, result in LANE 63, on the VALU's DPP path (row_shr 1 / 2 / 4 / 8, then row_bcast:15 and row_bcast:31 -- the
// wave64 reduction of the GFX9 family): six v_add_f32 with a DPP source instead of six ds_bpermute_b32 + s_waitcnt round trips through the
// LDS pipe per plane (round 6: the window kernels are bound by VALU + LDS issue, and 78 of their 420 LDS instructions were these shuffles).
#ifndef MM_PW_ROW_PRE
#define MM_PW_ROW_PRE 0         // 1: row_pass_pre -- the five chunk offsets of a thread made once, out-of-row chunks pointing at a zeroed slack, no
#endif                          // select left in the frame loop (260 fewer vector instructions per thread).  Measured SLOWER (0.550 vs 0.510 ms,
                                // profiles/r06_ab_phase_window_micro.txt): the lanes that read the shared zero slot conflict with the row reads
                                // of their neighbours; the selects are cheaper than that.  Bit-identical either way.
#ifndef MM_PW_MEAN_LANES
#define MM_PW_MEAN_LANES 1      // 0: every thread adds all 9 x 12 partial sums itself (round-3 form), for the A/B
#endif
#ifndef MM_PW_DPP_REDUCE
#define MM_PW_DPP_REDUCE 1      // 0: the round-3 __shfl_down tree (result in lane 0), for the A/B
#endif
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_min(int v) {
    return min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ int wave_min_lane63(int v) {      // minimum over the wave, in lane 63 (same path as wave_sum_lane63)
    v = dpp_min<0x111, 0xf>(v);
    v = dpp_min<0x112, 0xf>(v);
    v = dpp_min<0x114, 0xf>(v);
    v = dpp_min<0x118, 0xf>(v);
    v = dpp_min<0x142, 0xa>(v);
    v = dpp_min<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v = dpp_add<0x111, 0xf>(v);      // row_shr:1
    v = dpp_add<0x112, 0xf>(v);      // row_shr:2
    v = dpp_add<0x114, 0xf>(v);      // row_shr:4
    v = dpp_add<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of every row holds its row's sum
    v = dpp_add<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
    return v;
}

// ---- per (window, band): 12 phase-difference planes from the frame planes.
//   A  every frame's B and phase planes are requested up front (26 independent loads per thread, one latency instead of one
//      per barrier round); the wrapped steps are counted bytewise in one register per frame (k <= 12 per pixel), the difference planes start
//      as B_i - B_{i-1}, and the workgroup agrees on the first frame in which ANY of its pixels has wrapped (one LDS min, one barrier)
//   B  from that frame on, F frames per barrier round: blur(mag * (-2 pi k)) * R, whose frame-to-frame change is added to the planes.
//      Frames before it have blur(mag * 0) = 0: rounds that end before the first wrap are skipped (always the window's first frame,
//      whole windows of slowly moving faces)
//   C  spatial means, clamp, store.  NHWC pixels take 48 of their 96 bytes from this band: the rows go through LDS so that three
//      consecutive lanes write the 48 contiguous bytes of a pixel (a lane-per-pixel-strip store issues 64 separate 16-byte requests
//      per instruction: 0.16 of the 0.61 ms these kernels took).  A thread's four pixels are 12 float4 slots; its block in the staging
//      is 13 slots long (round 5): with 12, eight consecutive lanes of a ds_write_b128 fall on two of the eight 16-byte bank groups
//      (192-byte lane stride: 35 % of the kernel's LDS-active cycles were bank conflicts, PMC SQ_LDS_BANK_CONFLICT), with the odd
//      stride on all eight
// F = frames per barrier round (three barriers each).  A 9-wave workgroup at ~160 registers is alone on its CU, so the whole
// 160 KB of LDS is its to use: F planes of blur input + row-pass output.
// Row-pass input planes of phase_window2_kernel (round 6): every row carries eight zero floats on both sides -- the blur's zero padding is IN
// the plane and the row pass is five unconditional 16-byte reads.  Rounds 3-5: 20 v_cndmask per frame and thread on the chunk addresses and
// values, and hipcc narrowed the outer chunks to b32 / b96 reads whose lane groups conflict (PMC SQ_LDS_BANK_CONFLICT 2.7e7 -> 1.8e6 per
// launch of <48>, window kernels -9.5 % same-box: profiles/r06_ab_phase_window_pad.txt).  Row stride 112 (W = 48) / 88 (W = 24) floats = the
// dense stride mod 64: the sixteen lanes of a ds_read_b128 group still fall on sixteen different 16-byte bank groups, as in the dense plane.
#ifndef MM_PW_PAD_ROWS
#define MM_PW_PAD_ROWS 1        // 0: dense planes + selects (rounds 3-5), for the A/B
#endif
#ifndef MM_PW_PAD_ROWS_24
#define MM_PW_PAD_ROWS_24 1     // 0: W = 24 stays dense (measured equal either way), for the A/B
#endif
template <int W> struct PwIn {
    static constexpr bool PADDED = MM_PW_PAD_ROWS && (W == 48 || MM_PW_PAD_ROWS_24);
    static constexpr int STRIDE = PADDED ? (W == 48 ? 112 : 88) : W;      // 4 x (12 | 6 strips + 16 k)
    static constexpr int LEFT = PADDED ? PADX : 0;
    static constexpr int PLANE = PADDED ? (W - 1) * STRIDE + W + 2 * PADX : Cfg<W>::IN_PLANE;
};
// Workgroup barrier that orders LDS traffic only (the LDS queue drained, then s_barrier) -- hipcc's __syncthreads() puts s_waitcnt vmcnt(0) in front
// of the barrier as well, i.e. waits for every global load AND STORE in flight.  Used for the blur loop and the store groups with
// -DMM_PW_LDS_BARRIER=1: measured null (0.437 vs 0.441 ms, profiles/r06_ab_phase_window_barrier.txt), so the default stays __syncthreads().
#ifndef MM_PW_LDS_BARRIER
#define MM_PW_LDS_BARRIER 0
#endif
__device__ __forceinline__ void lds_barrier() {
#if MM_PW_LDS_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}
template <int W> struct PwDense {
    static constexpr bool PADDED = false;
    static constexpr int STRIDE = W, LEFT = 0, PLANE = Cfg<W>::IN_PLANE;
};
template <int W>
__device__ __forceinline__ void row_pass_padded(const float* row_x0, float (&h)[PX]) {      // row_x0: the row's padded start + x0, i.e. pixel x0 - 8
    float v[PX + 2 * PADX];
    float4 a[(PX + 2 * PADX) / 4];
#pragma unroll
    for (int q = 0; q < (PX + 2 * PADX) / 4; ++q) a[q] = *reinterpret_cast<const float4*>(__builtin_assume_aligned(row_x0 + 4 * q, 16));
    // (only v[3..16] are used: left alone, hipcc shrinks the outer chunks and re-forms the 14 floats as seven ds_read2_b32, whose 32-lane
    //  groups conflict on this row stride -- one empty asm over all five values keeps the five 16-byte reads and lets them be in flight together)
    asm volatile("" : "+v"(a[0].x), "+v"(a[0].y), "+v"(a[0].z), "+v"(a[0].w), "+v"(a[1].x), "+v"(a[1].y), "+v"(a[1].z), "+v"(a[1].w),
                      "+v"(a[2].x), "+v"(a[2].y), "+v"(a[2].z), "+v"(a[2].w), "+v"(a[3].x), "+v"(a[3].y), "+v"(a[3].z), "+v"(a[3].w),
                      "+v"(a[4].x), "+v"(a[4].y), "+v"(a[4].z), "+v"(a[4].w));
#pragma unroll
    for (int q = 0; q < (PX + 2 * PADX) / 4; ++q) {
        v[4 * q] = a[q].x; v[4 * q + 1] = a[q].y; v[4 * q + 2] = a[q].z; v[4 * q + 3] = a[q].w;
    }
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < TAP; ++t) s = fmaf(c_g[t], v[PADX - R + p + t], s);
        h[p] = s;
    }
}

// Floats of LDS the body below uses for plane side W (its working planes + the per-wave partial sums); the agreed first-wrap frame is one more word.
template <int W, int F> constexpr int pw2_lds_floats() { return F * (PwIn<W>::PLANE + Cfg<W>::TMP_PLANE) + 64 * (P - 1); }
constexpr int pw2_store_groups(int W) { return W / (W == 48 ? 16 : 12); }

// The body of phase_window2_kernel for a team of Cfg<W>::NTHREADS threads (tid = the thread's index in its team, lds = the team's region,
// first_wrap_p = the word the workgroup agrees on).  PAIR: the team shares its workgroup with the team of the other level (phase_window2_kernel_pair
// below): every team then executes the SAME number of barriers -- the blur rounds follow the shared first-wrap frame, and the level-2 team
// adds the barriers of the level-1 team's third store group at the end.
template <int W, int F, bool PAIR>
__device__ __forceinline__ void pw2_body(const float* __restrict__ fr, const int32_t* __restrict__ ids, int n_frames, float* __restrict__ out,
                                         int out_nhwc, int out_cstride, int out_coffset, float* lds, int* first_wrap_p, const int tid) {
    using C = Cfg<W>;
    constexpr int RPG = W == 48 ? 16 : 12;                    // rows per store group: RPG * W * 12 floats staged at a time
    constexpr int G = W / RPG;
    using PW = PwIn<W>;
    constexpr int IN_PLANE_ = PW::PLANE;
    constexpr int WORK = F * (IN_PLANE_ + C::TMP_PLANE);
#ifndef MM_PW_STAGE_PAD
#define MM_PW_STAGE_PAD 1                                     // 0: the round-4 staging (12-slot thread blocks), for the A/B
#endif
    constexpr int SLOTS = 3 * PX + MM_PW_STAGE_PAD;           // float4 slots per thread block in the store staging: 12 used + 1 pad
    static_assert(RPG * C::STRIPS * SLOTS * 4 <= WORK, "store staging fits the blur planes");
    static_assert(WORK + 64 * (P - 1) == pw2_lds_floats<W, F>(), "the launchers size the region with this");
    int& first_wrap = *first_wrap_p;
    float* in_x = lds;                              // [F][IN_PLANE]
    float* tmp_x = in_x + F * IN_PLANE_;            // [F][TMP_PLANE]
    float* red = lds + WORK;
    const int lane = tid & 63, wave = tid >> 6;
    // consecutive windows share 12 of their 13 frames: keep them on one XCD (one L2) instead of spreading them over all eight
    const int logical = xcd_contiguous(blockIdx.x, gridDim.x);
    const int64_t j = logical >> 1;
    const int band = logical & 1;
    const bool active = C::ACTIVE == C::NTHREADS || tid < C::ACTIVE;      // (W = 48: every thread -- no exec-mask branches)
    const int y = active ? tid / C::STRIPS : 0;
    const int x0 = active ? (tid - y * C::STRIPS) * PX : 0;
    const int px = y * W + x0;
    const int ipx = y * PW::STRIDE + PW::LEFT + x0;      // the pixel strip in a row-pass input plane
    // the zero rows above / below each tmp_x plane (column-pass halo; everything else is written before it is read -- round 6: the whole 61 KB
    // region used to be cleared, 27 ds_write_b32 per thread)
#ifndef MM_PW_CLEAR_ALL
#define MM_PW_CLEAR_ALL 0       // 1: clear the whole working region (round-3 form), for the A/B
#endif
    if (MM_PW_CLEAR_ALL) {
        for (int i = tid; i < WORK; i += C::NTHREADS) lds[i] = 0.f;
    } else {
        for (int i = tid; i < F * 2 * R * W; i += C::NTHREADS) {
            const int f = i / (2 * R * W), r = i - f * (2 * R * W);
            tmp_x[f * C::TMP_PLANE + (r < R * W ? r : (W + R) * W + (r - R * W))] = 0.f;
        }
        if (!PW::PADDED && tid < F * 2 * PADX) in_x[(tid / (2 * PADX)) * IN_PLANE_ + W * W + tid % (2 * PADX)] = 0.f;   // the slack row_pass_pre reads as zero padding
        if (PW::PADDED) {                   // the eight zero floats left and right of every row of the F input planes: one 16-byte store per thread
            for (int i = tid; i < F * W * 4; i += C::NTHREADS) {
                const int f = i / (W * 4), r = (i >> 2) % W, q = i & 3;
                *reinterpret_cast<float4*>(in_x + f * IN_PLANE_ + r * PW::STRIDE + (q < 2 ? 4 * q : W + PADX + 4 * (q - 2))) = float4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    int roff[(PX + 2 * PADX) / 4];                  // this thread's five row-pass chunks: loop constants
    row_chunk_offsets<W>(y, x0, roff);
    if (tid == 0 && (!PAIR || W == 48)) first_wrap = P;
    const float TWO_PI_F = 6.28318530717958647692f, PI_F = 3.14159265358979323846f;
    // ---- A
    const float* fo[P];
    float d[P - 1][PX];
    unsigned kb[P];
    {
        float4 bprev = {0.f, 0.f, 0.f, 0.f}, pprev = {0.f, 0.f, 0.f, 0.f};
        unsigned run = 0;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int id = min(max(ids[(MM_PW_ABLATE & 4 ? 0 : j) * P + i], 0), n_frames - 1);    // memory-safe whatever the table holds (the shim range-checks it)
            fo[i] = fr + ((int64_t)id * 2 + band) * C::FRAME_FLOATS;
            float4 b4 = {0.f, 0.f, 0.f, 0.f}, p4 = {0.f, 0.f, 0.f, 0.f};
            if (active) {
                b4 = *reinterpret_cast<const float4*>(fo[i] + C::PLANE + px);
                p4 = *reinterpret_cast<const float4*>(fo[i] + 3 * C::PLANE + px);
            }
            if (i > 0) {
                // the steps torch_unwrap corrects by -2 pi (phase_utils.py:9-17), on its own fp32 expression
                if ((p4.x - pprev.x) + PI_F > TWO_PI_F) run += 1u;
                if ((p4.y - pprev.y) + PI_F > TWO_PI_F) run += 1u << 8;
                if ((p4.z - pprev.z) + PI_F > TWO_PI_F) run += 1u << 16;
                if ((p4.w - pprev.w) + PI_F > TWO_PI_F) run += 1u << 24;
                d[i - 1][0] = b4.x - bprev.x; d[i - 1][1] = b4.y - bprev.y; d[i - 1][2] = b4.z - bprev.z; d[i - 1][3] = b4.w - bprev.w;
            }
            kb[i] = run;
            bprev = b4;
            pprev = p4;
        }
    }
    {
        int mine = P;                     // first frame in which one of this thread's pixels has wrapped (counts only grow)
#pragma unroll
        for (int i = P - 1; i >= 0; --i) mine = kb[i] != 0u ? i : mine;
#if MM_PW_DPP_REDUCE
        mine = wave_min_lane63(mine);
        __syncthreads();                  // first_wrap initialised, LDS zeroed
        if (lane == 63 && mine < P) atomicMin(&first_wrap, mine);
#else
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mine = min(mine, __shfl_xor(mine, off, 64));
        __syncthreads();                  // first_wrap initialised, LDS zeroed
        if (lane == 0 && mine < P) atomicMin(&first_wrap, mine);
#endif
        __syncthreads();
    }
#if MM_PW_ABLATE & 2
    const int first = P;               // measurement build: no blur round
#else
    const int first = first_wrap;
#endif
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
            *reinterpret_cast<float4*>(in_x + f * IN_PLANE_ + ipx) =
                float4{m4.x * (-TWO_PI_F * (float)(k & 255u)), m4.y * (-TWO_PI_F * (float)((k >> 8) & 255u)),
                       m4.z * (-TWO_PI_F * (float)((k >> 16) & 255u)), m4.w * (-TWO_PI_F * (float)(k >> 24))};
        }
        lds_barrier();
        if (active) {
#pragma unroll
            for (int f = 0; f < F; ++f) {
                if (base + f >= P) continue;
                float h[PX];
#if MM_PW_ROW_PRE
                row_pass_pre<W>(in_x + f * IN_PLANE_, roff, h);
#else
                if (PW::PADDED) row_pass_padded<W>(in_x + f * IN_PLANE_ + y * PW::STRIDE + x0, h);
                else row_pass<W>(in_x + f * IN_PLANE_, y, x0, h);
#endif
                *reinterpret_cast<float4*>(tmp_x + f * C::TMP_PLANE + (y + R) * W + x0) = float4{h[0], h[1], h[2], h[3]};
            }
        }
        lds_barrier();
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
#if MM_PW_DPP_REDUCE
        v = wave_sum_lane63(v);
        if (lane == 63) red[wave * (P - 1) + k] = v;
#else
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave * (P - 1) + k] = v;
#endif
    }
    __syncthreads();                      // also: every blur read of the planes is done, they become the store staging
    constexpr int NWAVES = C::NTHREADS / 64;
    const float LIM = 5.f * PI_F;
    float mean[P - 1];
#if MM_PW_MEAN_LANES
    {   // lane k of every wave adds the NWAVES partial sums of plane k (same order as below), the planes' means then reach all lanes as
        // scalars (v_readlane): 9 ds_read_b32 + 8 adds + 12 readlanes per thread instead of 27 broadcast ds_read_b128 + 96 adds
        float part = 0.f;
        if (lane < P - 1) {
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) part += red[w * (P - 1) + lane];
        }
        part *= (1.0f / (W * W));
#pragma unroll
        for (int k = 0; k < P - 1; ++k) mean[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(part), k));
    }
#else
#pragma unroll
    for (int k = 0; k < P - 1; ++k) {
        float sm = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) sm += red[w * (P - 1) + k];
        mean[k] = sm * (1.0f / (W * W));
    }
#endif
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
    float4* stage = reinterpret_cast<float4*>(lds);      // [RPG * STRIPS thread blocks][SLOTS float4]: pixel-major, 3 float4 per pixel
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (active && y / RPG == g) {
            float4* mine = stage + (tid - g * RPG * C::STRIPS) * SLOTS;
#pragma unroll
            for (int p = 0; p < PX; ++p)
#pragma unroll
                for (int q = 0; q < (P - 1) / 4; ++q) {
                    float4 v;
                    v.x = fminf(fmaxf(d[4 * q][p] - mean[4 * q], -LIM), LIM);
                    v.y = fminf(fmaxf(d[4 * q + 1][p] - mean[4 * q + 1], -LIM), LIM);
                    v.z = fminf(fmaxf(d[4 * q + 2][p] - mean[4 * q + 2], -LIM), LIM);
                    v.w = fminf(fmaxf(d[4 * q + 3][p] - mean[4 * q + 3], -LIM), LIM);
                    mine[p * 3 + q] = v;
                }
        }
        lds_barrier();
        for (int idx = tid; idx < RPG * W * 3; idx += C::NTHREADS) {
            const int pix = idx / 3, part = idx - pix * 3;
            float* dst = out + ((j * W + g * RPG) * W + pix) * out_cstride + out_coffset + band * (P - 1) + part * 4;
#if MM_PW_ABLATE & 1
            if (stage[idx].x == 12345.678f)      // measurement build: no output stores (results wrong by construction)
#endif
            *reinterpret_cast<float4*>(dst) = stage[idx + MM_PW_STAGE_PAD * (idx / (3 * PX))];      // skip the pad slot of every thread block
        }
        if (g + 1 < G) lds_barrier();      // (the stores of this group stay in flight: their data left LDS with the reads above)
    }
    if (PAIR) {   // the barriers of the store groups the OTHER team has and this one has not (level 1: three groups, level 2: two)
        constexpr int G_MAX = pw2_store_groups(48);
        static_assert(pw2_store_groups(48) >= pw2_store_groups(24), "level 1 has the most store groups");
#pragma unroll
        for (int g = G; g < G_MAX; ++g) {
            lds_barrier();
            lds_barrier();
        }
    }
}

template <int W, int F>
__global__ void __launch_bounds__(Cfg<W>::NTHREADS)
phase_window2_kernel(const float* __restrict__ fr, const int32_t* __restrict__ ids, int n_frames, float* __restrict__ out, int out_nhwc,
                     int out_cstride, int out_coffset) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // pw2_lds_floats + first_wrap
    pw2_body<W, F, false>(fr, ids, n_frames, out, out_nhwc, out_cstride, out_coffset, lds, reinterpret_cast<int*>(lds + pw2_lds_floats<W, F>()),
                          (int)threadIdx.x);
}

// Round 6: BOTH levels of a (window, band) in one workgroup of twelve waves -- nine (level 1, 48 x 48) + three (level 2, 24 x 24).  The
// level-1 kernel alone puts nine waves on a CU's four SIMDs (3 + 2 + 2 + 2: the waves of the two-wave SIMDs wait at every barrier for the
// three-wave one; PMC: 54 % of the wave time at s_waitcnt / s_barrier) and its 97 KB of LDS keep every other workgroup off the CU; the level-2
// kernel then runs as a launch of its own (0.098 of the 0.41 ms).  Three waves of level-2 work are exactly what the three lighter SIMDs
// have room for: twelve waves, three per SIMD, the level-2 work inside the time the level-1 work takes anyway.  Each team runs the
// unchanged arithmetic of its level in its own LDS region; they share the barriers (same count in both teams: see pw2_body) and the
// first-wrap word -- a blur round one team would have skipped runs on an all-zero wrap count there and adds exact zeros.
constexpr int PAIR_THREADS = Cfg<48>::NTHREADS + Cfg<24>::NTHREADS;
static_assert(Cfg<48>::NTHREADS % 64 == 0 && PAIR_THREADS == 768, "whole waves per team; twelve waves");
template <int F>
__global__ void __launch_bounds__(PAIR_THREADS)
phase_window2_kernel_pair(const float* __restrict__ fr1, const float* __restrict__ fr2, const int32_t* __restrict__ ids, int n_frames,
                          float* __restrict__ out0, int out0_cstride, int out0_coffset, float* __restrict__ out1, int out1_cstride,
                          int out1_coffset, int out_nhwc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int L1 = pw2_lds_floats<48, F>(), L2 = pw2_lds_floats<24, F>();
    static_assert(L1 % 4 == 0 && L2 % 4 == 0, "16-byte aligned regions");
    int* first_wrap = reinterpret_cast<int*>(lds + L1 + L2);
    const int tid = threadIdx.x;
    if (tid < Cfg<48>::NTHREADS) pw2_body<48, F, true>(fr1, ids, n_frames, out0, out_nhwc, out0_cstride, out0_coffset, lds, first_wrap, tid);
    else pw2_body<24, F, true>(fr2, ids, n_frames, out1, out_nhwc, out1_cstride, out1_coffset, lds + L1, first_wrap, tid - Cfg<48>::NTHREADS);
}


// =====================================================================================================================================
// Round 6 (round-5 verdict item 3b): the same kernel split ALONG TIME over two workgroups per (window, band) -- BUILT, EQUAL UP TO ONE FMA CONTRACTION, MEASURED
// SLOWER (profiles/r06_ab_phase_window_split.txt: 0.65-0.66 ms against 0.545 ms for the two launches), so it is NOT the default:
// MM_PW_SPLIT=2 selects it (the GPU test test_time_split_window_kernel_equals_the_one_workgroup_form runs both: 22 of 24 channels bit-equal, the first
// difference plane of half 1 one rounding apart -- hipcc contracts blur * R into the subtraction in one kernel and not in the other).
//   NH = 2: half h owns differences [6 h, 6 h + 6), i.e. frames 6 h .. 6 h + 6.  The spatial mean is per difference plane, so nothing
//   crosses the workgroups (splitting a plane in SPACE would need a cross-workgroup mean); half 1 only has to know how often each pixel
//   wrapped before its first frame: six more phase planes, no B, no blur.  The steps torch_unwrap corrects are kept as ONE BIT per
//   (frame, pixel) (the count a frame needs is a popcount), the plane loads are requested in chunks of MM_PW_CHUNK frames and R only before
//   the column pass: d[6][4] + the wrap bits fit 96 registers -> TWO nine-wave workgroups per CU (18 waves, five on a SIMD), which is what
//   the verdict asked for (one 162-register workgroup per CU spends 58 % of its wave time at waitcnt / barrier).  Frame 6 is blurred by both
//   halves (14 instead of 13 blurs per window and band), half 1 re-reads six phase planes, every workgroup pays the fixed parts (LDS
//   clear, first-wrap agreement, mean reduction, store staging) for half the output, and the stores are 8-byte instead of 16-byte
//   channel groups: the added instructions cost more than the second workgroup's overlap buys -- the kernel is bound by VALU / LDS issue,
//   not by the latency a second workgroup hides.
template <int W, int F, int NH, int H>
__device__ __forceinline__ void phase_window2s_body(const float* __restrict__ fr, const int32_t* __restrict__ ids, int n_frames,
                                                   float* __restrict__ out, int out_nhwc, int out_cstride, int out_coffset, float* lds,
                                                   int64_t j, int band) {
    using C = Cfg<W>;
    constexpr int KN = (P - 1) / NH;                          // difference planes of this workgroup
    constexpr int K0 = H * KN;                                // ... starting at this one: frames K0 .. K0 + KN
    constexpr int EW = NH == 1 ? 4 : 2;                       // floats per staging slot / store: 16-byte stores for 12 channels, 8-byte for 6
    constexpr int PPX = KN / EW;                              // slots per pixel (3 either way)
    static_assert(KN * NH == P - 1 && PPX * EW == KN, "the difference planes split evenly");
    constexpr int RPG = W == 48 ? (NH == 1 ? 16 : 24) : (NH == 1 ? 12 : 24);   // rows per store group
    constexpr int G = W / RPG;
    using PW = PwIn<W>;                                       // padded row-pass planes, as in the one-workgroup kernel
    constexpr int IN_PLANE_ = PW::PLANE;
    constexpr int WORK = F * (IN_PLANE_ + C::TMP_PLANE);
#ifndef MM_PW_STAGE_PAD
#define MM_PW_STAGE_PAD 1                                     // 0: the round-4 staging (12-slot thread blocks), for the A/B
#endif
    constexpr int SLOTS = PPX * PX + MM_PW_STAGE_PAD * (4 / EW);   // slots per thread block in the store staging: 12 used + 16 bytes of pad
    static_assert(RPG * C::STRIPS * SLOTS * EW <= WORK, "store staging fits the blur planes");
    int& first_wrap = *reinterpret_cast<int*>(lds + WORK + 64 * (P - 1));
    float* in_x = lds;                              // [F][IN_PLANE]
    float* tmp_x = in_x + F * IN_PLANE_;            // [F][TMP_PLANE]
    float* red = lds + WORK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool active = tid < C::ACTIVE;
    const int y = active ? tid / C::STRIPS : 0;
    const int x0 = active ? (tid - y * C::STRIPS) * PX : 0;
    const int px = y * W + x0;
    const int ipx = y * PW::STRIDE + PW::LEFT + x0;
    // the zero rows above / below each tmp_x plane (column-pass halo; everything else is written before it is read -- round 6: the whole 61 KB
    // region used to be cleared, 27 ds_write_b32 per thread)
#ifndef MM_PW_CLEAR_ALL
#define MM_PW_CLEAR_ALL 0       // 1: clear the whole working region (round-3 form), for the A/B
#endif
    if (MM_PW_CLEAR_ALL) {
        for (int i = tid; i < WORK; i += C::NTHREADS) lds[i] = 0.f;
    } else {
        for (int i = tid; i < F * 2 * R * W; i += C::NTHREADS) {
            const int f = i / (2 * R * W), r = i - f * (2 * R * W);
            tmp_x[f * C::TMP_PLANE + (r < R * W ? r : (W + R) * W + (r - R * W))] = 0.f;
        }
        if (!PW::PADDED && tid < F * 2 * PADX) in_x[(tid / (2 * PADX)) * IN_PLANE_ + W * W + tid % (2 * PADX)] = 0.f;   // the slack row_pass_pre reads as zero padding
        if (PW::PADDED) {
            for (int i = tid; i < F * W * 4; i += C::NTHREADS) {
                const int f = i / (W * 4), r = (i >> 2) % W, q = i & 3;
                *reinterpret_cast<float4*>(in_x + f * IN_PLANE_ + r * PW::STRIDE + (q < 2 ? 4 * q : W + PADX + 4 * (q - 2))) = float4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    int roff[(PX + 2 * PADX) / 4];                  // this thread's five row-pass chunks: loop constants
    row_chunk_offsets<W>(y, x0, roff);
    if (tid == 0) first_wrap = P;
    const float TWO_PI_F = 6.28318530717958647692f, PI_F = 3.14159265358979323846f;
    // ---- A
    const float* fo[P];
    float d[KN][PX];
    // wrapped steps, one bit per (frame, pixel): bit i of wb[p] = the step INTO frame i was corrected; the count a frame needs is
    // popcount(wb[p] & ((2 << i) - 1)) (k <= 12)
    unsigned wb[PX] = {0u, 0u, 0u, 0u};
    {
        // NH = 1: all 26 loads of the window in flight at once (104 registers of landing space); the split form asks for them in chunks of
        // CH frames behind a scheduling barrier -- the other workgroup of the CU covers the latency, and the registers are what buys it
        constexpr int NF = K0 + KN + 1, CH = NH == 1 ? NF : MM_PW_CHUNK;
        float4 bprev = {0.f, 0.f, 0.f, 0.f}, pprev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c0 = 0; c0 < NF; c0 += CH) {
            float4 bq[CH], pq[CH];
#pragma unroll
            for (int t = 0; t < CH; ++t) {
                const int i = c0 + t;
                bq[t] = pq[t] = float4{0.f, 0.f, 0.f, 0.f};
                if (i >= NF) continue;
                const int id = min(max(ids[j * P + i], 0), n_frames - 1);    // memory-safe whatever the table holds (the shim range-checks it)
                fo[i] = fr + ((int64_t)id * 2 + band) * C::FRAME_FLOATS;
                if (active) {
                    if (i >= K0) bq[t] = *reinterpret_cast<const float4*>(fo[i] + C::PLANE + px);
                    pq[t] = *reinterpret_cast<const float4*>(fo[i] + 3 * C::PLANE + px);
                }
            }
#pragma unroll
            for (int t = 0; t < CH; ++t) {
                const int i = c0 + t;
                if (i >= NF) continue;
                const float4 b4 = bq[t], p4 = pq[t];
                if (i > 0) {
                    // the steps torch_unwrap corrects by -2 pi (phase_utils.py:9-17), on its own fp32 expression
                    if ((p4.x - pprev.x) + PI_F > TWO_PI_F) wb[0] |= 1u << i;
                    if ((p4.y - pprev.y) + PI_F > TWO_PI_F) wb[1] |= 1u << i;
                    if ((p4.z - pprev.z) + PI_F > TWO_PI_F) wb[2] |= 1u << i;
                    if ((p4.w - pprev.w) + PI_F > TWO_PI_F) wb[3] |= 1u << i;
                }
                if (i > K0) {
                    d[i - 1 - K0][0] = b4.x - bprev.x; d[i - 1 - K0][1] = b4.y - bprev.y;
                    d[i - 1 - K0][2] = b4.z - bprev.z; d[i - 1 - K0][3] = b4.w - bprev.w;
                }
                bprev = b4;
                pprev = p4;
            }
            if (NH != 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
    {
        // first frame OF THIS WORKGROUP'S RANGE in which one of this thread's pixels has a non-zero count (counts only grow): the first set
        // bit, but not before frame K0
        const unsigned any = wb[0] | wb[1] | wb[2] | wb[3];
        int mine = any ? max(__builtin_ctz(any), K0) : P;
#if MM_PW_DPP_REDUCE
        mine = wave_min_lane63(mine);
        __syncthreads();                  // first_wrap initialised, LDS zeroed
        if (lane == 63 && mine < P) atomicMin(&first_wrap, mine);
#else
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mine = min(mine, __shfl_xor(mine, off, 64));
        __syncthreads();                  // first_wrap initialised, LDS zeroed
        if (lane == 0 && mine < P) atomicMin(&first_wrap, mine);
#endif
        __syncthreads();
    }
    const int first = first_wrap;
    // ---- B
    float cprev[PX] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int base = K0; base <= K0 + KN; base += F) {
        if (base + F <= first) continue;             // block-uniform: no pixel of the window has wrapped up to the round's last frame
        float4 r4[F];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int i = base + f;
            r4[f] = float4{0.f, 0.f, 0.f, 0.f};
            if (i > K0 + KN || !active) continue;
            const float4 m4 = *reinterpret_cast<const float4*>(fo[i] + px);
            // (a pixel's blur also sums its neighbours' wraps: R is needed wherever the window has wrapped at all.  NH = 1 requests it with
            //  the magnitude; the split form only before the column pass -- 12 registers less across the two barriers of a round)
            if (NH == 1) r4[f] = *reinterpret_cast<const float4*>(fo[i] + 2 * C::PLANE + px);
            const unsigned msk = (2u << i) - 1u;
            *reinterpret_cast<float4*>(in_x + f * IN_PLANE_ + ipx) =
                float4{m4.x * (-TWO_PI_F * (float)__builtin_popcount(wb[0] & msk)), m4.y * (-TWO_PI_F * (float)__builtin_popcount(wb[1] & msk)),
                       m4.z * (-TWO_PI_F * (float)__builtin_popcount(wb[2] & msk)), m4.w * (-TWO_PI_F * (float)__builtin_popcount(wb[3] & msk))};
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int f = 0; f < F; ++f) {
                if (base + f > K0 + KN) continue;
                float h[PX];
#if MM_PW_ROW_PRE
                row_pass_pre<W>(in_x + f * IN_PLANE_, roff, h);
#else
                if (PW::PADDED) row_pass_padded<W>(in_x + f * IN_PLANE_ + y * PW::STRIDE + x0, h);
                else row_pass<W>(in_x + f * IN_PLANE_, y, x0, h);
#endif
                *reinterpret_cast<float4*>(tmp_x + f * C::TMP_PLANE + (y + R) * W + x0) = float4{h[0], h[1], h[2], h[3]};
                if (NH != 1) __builtin_amdgcn_sched_barrier(0);       // one frame's 20-float row window at a time (registers)
            }
        }
        __syncthreads();
        // (the next round's in_x stores are separated from this round's row-pass reads by the barrier above, its tmp_x stores from
        //  the column reads below by its own first barrier)
        if (active) {
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const int i = base + f;
                if (i > K0 + KN) continue;
                if (NH != 1) r4[f] = *reinterpret_cast<const float4*>(fo[i] + 2 * C::PLANE + px);
                float sb[PX];
                col_pass<W>(tmp_x + f * C::TMP_PLANE, y, x0, sb);
                const float c[PX] = {sb[0] * r4[f].x, sb[1] * r4[f].y, sb[2] * r4[f].z, sb[3] * r4[f].w};
#pragma unroll
                for (int p = 0; p < PX; ++p) {
                    if (i > K0) d[i - 1 - K0][p] += c[p] - cprev[p];
                    cprev[p] = c[p];
                }
                if (NH != 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // ---- C: spatial means of the difference planes: wave shuffle reduce, then across waves
#pragma unroll
    for (int k = 0; k < KN; ++k) {
        float v = active ? (d[k][0] + d[k][1]) + (d[k][2] + d[k][3]) : 0.f;
#if MM_PW_DPP_REDUCE
        v = wave_sum_lane63(v);
        if (lane == 63) red[wave * (P - 1) + k] = v;
#else
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave * (P - 1) + k] = v;
#endif
    }
    __syncthreads();                      // also: every blur read of the planes is done, they become the store staging
    constexpr int NWAVES = C::NTHREADS / 64;
    const float LIM = 5.f * PI_F;
    float mean[KN];
#if MM_PW_MEAN_LANES
    {
        float part = 0.f;
        if (lane < KN) {
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) part += red[w * (P - 1) + lane];
        }
        part *= (1.0f / (W * W));
#pragma unroll
        for (int k = 0; k < KN; ++k) mean[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(part), k));
    }
#else
#pragma unroll
    for (int k = 0; k < KN; ++k) {
        float sm = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) sm += red[w * (P - 1) + k];
        mean[k] = sm * (1.0f / (W * W));
    }
#endif
    if (!out_nhwc) {
        if (active) {
#pragma unroll
            for (int k = 0; k < KN; ++k) {
                float ov[PX];
#pragma unroll
                for (int p = 0; p < PX; ++p) ov[p] = fminf(fmaxf(d[k][p] - mean[k], -LIM), LIM);
                float* dst = out + ((j * (2 * (P - 1)) + band * (P - 1) + K0 + k) * W + y) * W + x0;
                *reinterpret_cast<float4*>(dst) = float4{ov[0], ov[1], ov[2], ov[3]};
            }
        }
        return;
    }
    typedef float slot_t __attribute__((ext_vector_type(EW)));
    slot_t* stage = reinterpret_cast<slot_t*>(lds);           // [RPG * STRIPS thread blocks][SLOTS]: pixel-major, PPX slots per pixel
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (active && y / RPG == g) {
            slot_t* mine = stage + (tid - g * RPG * C::STRIPS) * SLOTS;
#pragma unroll
            for (int p = 0; p < PX; ++p)
#pragma unroll
                for (int q = 0; q < PPX; ++q) {
                    slot_t v;
#pragma unroll
                    for (int e = 0; e < EW; ++e) v[e] = fminf(fmaxf(d[EW * q + e][p] - mean[EW * q + e], -LIM), LIM);
                    mine[p * PPX + q] = v;
                }
        }
        __syncthreads();
        for (int idx = tid; idx < RPG * W * PPX; idx += C::NTHREADS) {
            const int pix = idx / PPX, part = idx - pix * PPX;
            float* dst = out + ((j * W + g * RPG) * W + pix) * out_cstride + out_coffset + band * (P - 1) + K0 + part * EW;
            *reinterpret_cast<slot_t*>(dst) = stage[idx + MM_PW_STAGE_PAD * (4 / EW) * (idx / (PPX * PX))];      // skip the pad of every thread block
        }
        if (g + 1 < G) __syncthreads();
    }
}

// registers: NH = 1 as the compiler likes it (162: one nine-wave workgroup per CU); NH = 2 held to 96 so that two nine-wave workgroups
// (five waves on one SIMD) share a CU
template <int W, int F, int NH>
__global__ void __launch_bounds__(Cfg<W>::NTHREADS, NH == 1 ? 1 : MM_PW_WAVES)
phase_window2s_kernel(const float* __restrict__ fr, const int32_t* __restrict__ ids, int n_frames, float* __restrict__ out, int out_nhwc,
                     int out_cstride, int out_coffset) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // WORK + 64 * (P - 1) floats + first_wrap
    // consecutive windows share 12 of their 13 frames: keep them on one XCD (one L2) instead of spreading them over all eight
    const int logical = xcd_contiguous(blockIdx.x, gridDim.x);
    const int64_t j = logical / (2 * NH);
    const int rem = logical - (int)j * (2 * NH);
    const int band = rem / NH;
    if (NH == 1 || rem - band * NH == 0) phase_window2s_body<W, F, NH, 0>(fr, ids, n_frames, out, out_nhwc, out_cstride, out_coffset, lds, j, band);
    else phase_window2s_body<W, F, NH, NH - 1>(fr, ids, n_frames, out, out_nhwc, out_cstride, out_coffset, lds, j, band);
}

template <int W, int F>
static int launch_w2(const float* fr, const int32_t* ids, int n, int64_t J, float* out, int out_nhwc, int out_cstride, int out_coffset,
                     hipStream_t s) {
    using C = Cfg<W>;
    const int lds_bytes = (pw2_lds_floats<W, F>() + 4) * 4;
    MM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(phase_window2_kernel<W, F>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               lds_bytes));
    hipLaunchKernelGGL((phase_window2_kernel<W, F>), dim3((unsigned)(2 * J)), dim3(C::NTHREADS), lds_bytes, s, fr, ids, n, out, out_nhwc,
                       out_cstride, out_coffset);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

template <int W, int F>
static int launch_w2s(const float* fr, const int32_t* ids, int n, int64_t J, float* out, int out_nhwc, int out_cstride, int out_coffset,
                      hipStream_t s) {
    using C = Cfg<W>;
    const int lds_bytes = (F * (PwIn<W>::PLANE + C::TMP_PLANE) + 64 * (P - 1) + 4) * 4;
    MM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(phase_window2s_kernel<W, F, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               lds_bytes));
    hipLaunchKernelGGL((phase_window2s_kernel<W, F, 2>), dim3((unsigned)(4 * J)), dim3(C::NTHREADS), lds_bytes, s, fr, ids, n, out, out_nhwc,
                       out_cstride, out_coffset);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

// Both levels in one launch (phase_window2_kernel_pair): opt-in, MM_PW_PAIR=1 (read per call: a test switches it), and only when the two
// outputs share a layout (else the teams would not meet the same barriers) and MM_PW_SPLIT is not 2.  Measured: one launch fewer pays on small
// batches (64 frames: 0.041 -> 0.028 ms, 192: 0.064 -> 0.052), is within +-2 % at 2 048 frames and 3 % SLOWER at 16 384
// (profiles/r06_ab_phase_window_barrier.txt) -- the level-2 team's work does not hide in the level-1 kernel's idle SIMD slots.
bool phase_window2_pair_applies(int out0_nhwc, int out1_nhwc) {
    const char* e = getenv("MM_PW_PAIR");
    const char* sp = getenv("MM_PW_SPLIT");
    return e && atoi(e) == 1 && !(sp && atoi(sp) == 2) && (out0_nhwc != 0) == (out1_nhwc != 0);
}
int launch_phase_window2_pair(const float* fr1, const float* fr2, const int32_t* ids, int64_t n, int64_t J, float* out0, int out0_nhwc,
                              int out0_cstride, int out0_coffset, float* out1, int out1_nhwc, int out1_cstride, int out1_coffset,
                              hipStream_t s) {
    if (J <= 0) return MM_OK;
    constexpr int F = 3;
    const int lds_bytes = (pw2_lds_floats<48, F>() + pw2_lds_floats<24, F>() + 4) * 4;
    MM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(phase_window2_kernel_pair<F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL((phase_window2_kernel_pair<F>), dim3((unsigned)(2 * J)), dim3(PAIR_THREADS), lds_bytes, s, fr1, fr2, ids, (int)n, out0,
                       out0_cstride, out0_coffset, out1, out1_cstride, out1_coffset, out0_nhwc);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

int launch_phase_window2(const float* fr, const int32_t* ids, int64_t n, int64_t J, int W, float* out, int out_nhwc, int out_cstride,
                         int out_coffset, hipStream_t s) {
    if (J <= 0) return MM_OK;
    if (n <= 0 || n > 0x7fffffff || 4 * J > 0x7fffffff) return MM_ERR_INVALID_ARG;
    // MM_PW_SPLIT=2 (read per call: a test switches it): the time-split form above -- measured slower, default off
    const char* e = getenv("MM_PW_SPLIT");
    if (e && atoi(e) == 2) {
        if (W == 48) return launch_w2s<48, MM_PW_SPLIT_F>(fr, ids, (int)n, J, out, out_nhwc, out_cstride, out_coffset, s);
        if (W == 24) return launch_w2s<24, MM_PW_SPLIT_F>(fr, ids, (int)n, J, out, out_nhwc, out_cstride, out_coffset, s);
        return MM_ERR_UNSUPPORTED;
    }
    // F = 3 frames per barrier round; 4 / 5 / 7 measured equal or slower (0.547 / 0.554 / 0.554 / 0.593 ms per 2 048 windows):
    // the kernel is not waiting at its barriers
    if (W == 48) return launch_w2<48, 3>(fr, ids, (int)n, J, out, out_nhwc, out_cstride, out_coffset, s);
    if (W == 24) return launch_w2<24, 3>(fr, ids, (int)n, J, out, out_nhwc, out_cstride, out_coffset, s);
    return MM_ERR_UNSUPPORTED;
}

}  // namespace mm
