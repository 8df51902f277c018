// ONE kernel per unique frame for the fused phase stage: complex steerable pyramid (levels 1 and 2, kept quadrant) AND everything
// of Phase_Difference_Extractor.extract that does not depend on the window a frame is seen through.
//
// Replaces, per 48x48 frame (api/tester.py:122-139 with the published configuration):
//   symmetric_extension_batch              api/utils/phase_utils.py:116-129        (folded into the DCT identity below)
//   SCFpyr_PyTorch.build / _build_levels   api/steerable/SCFpyr_PyTorch.py:70-208
//   build_pyramid's quadrant keep          api/phase_difference_extractor.py:76-92
//   extract: atan2 / magnitude, and the two Gaussian blurs of amplitude_based_gaussian_blur that are linear in the frame
//                                          api/phase_difference_extractor.py:100-104, api/utils/phase_utils.py:78-90
// and writes, per (frame, band, level), four W x W planes: mag, B = blur(mag phase) / blur(mag), R = 1 / blur(mag), phase.
// phase_window2_kernel (phase_frames.hip) turns those into the 12 phase differences of every 13-frame window.
//
// Pyramid math as in pyramid.hip (the drop-in build_pyramid keeps that kernel): the mirrored image's DFT is
// exp(i pi (fu+fv)/96) G[|fu|,|fv|] with G = D x D^T a 48x48 DCT-II; every band spectrum is G times a constant complex table with a
// zero half plane; only a quadrant of each inverse transform is kept, so each band is two small complex products with the twiddle
// table E.  Same operand values, same MFMA k order per output element => the coefficients are bit-identical to pyramid.hip's.
//
// What is different here (round 3; pyramid.hip: one 4-wave workgroup per CU holding 117 KB of LDS, every v_mfma fed by two
// ds_read_b32 of its own, 0.40 ms + 0.07 ms of phase_frame_kernel per 2 048 frames):
//   * 3 waves per workgroup, a wave owns one 16-row TILE ROW of every product (48 = 3 x 16): no tile-count imbalance (9 and 18 tiles
//     over 4 waves left a quarter of the wave-slots empty), the A fragment of a k-step is read once for the 3 column tiles
//     (8 LDS reads per 12 MFMAs instead of 12), 6 independent accumulator chains per wave
//   * the level-1 products run in two halves of the contraction / row range (the inverse product accumulates in registers across
//     the halves), band 0 writes T over the S rows it was computed from (a wave's T rows depend on its own S rows only), band 1's
//     first product feeds its second from the wave's own rows: 78 KB of LDS instead of 117 => TWO workgroups per CU, whose
//     barrier-separated phases interleave
//   * the band plane never leaves the CU before it is blurred: the epilogue converts it to (phase, magnitude) in LDS, runs the two
//     separable 11-tap blurs there and stores the four planes the window kernel needs (no 46 KB/frame polar round trip through HBM,
//     two launches fewer)
#include <cstdlib>
#include "mm_common.h"
#include "phase_math.h"
#include "phase_blur.h"
#include "pyramid_tables.h"

namespace mm {

namespace pf {

using namespace pyr;
using namespace blur;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 192;      // 3 waves
constexpr int LD = 49;       // odd stride: conflict-free as A operand (16 rows x 4 k) and nearly so as B operand

// LDS carve (floats)
constexpr int L_EC = 0;                        // [48][48]  persistent
constexpr int L_ES = L_EC + S * S;             // [48][48]  persistent
constexpr int L_DCT = L_ES + S * S;            // [48][49]  persistent
constexpr int L_G = L_DCT + S * LD;            // [48][49]  per frame
constexpr int L_W = L_G + S * LD;              // working region, re-carved per phase
constexpr int W_FLOATS = 2 * Cfg<48>::IN_PLANE + 2 * Cfg<48>::TMP_PLANE;   // 10208: the blur's four planes are the largest tenant
constexpr int L_TOTAL = L_W + W_FLOATS;        // 19520 floats = 78080 B  (two workgroups per CU: 2 x 78080 <= 163840)
static_assert(L_TOTAL * 4 <= 80 * 1024, "two workgroups per CU");
// working-region tenants (offsets from L_W)
constexpr int WX = 0, WT1 = S * S;                                   // DCT phase: x [48][48], T1 [48][49]
constexpr int WS_RE = 0;                                             // spectrum halves / planes
// blur tenant: in_num (= stage re), in_den (= stage im), tmp_num, tmp_den
template <int W> struct BL {
    static constexpr int IN_NUM = 0, IN_DEN = Cfg<W>::IN_PLANE, TMP_NUM = 2 * Cfg<W>::IN_PLANE,
                         TMP_DEN = 2 * Cfg<W>::IN_PLANE + Cfg<W>::TMP_PLANE;
};
static_assert(WT1 + S * LD <= W_FLOATS, "DCT tenant");
static_assert(2 * S * S + 2 * S * LD <= W_FLOATS, "level-1 band-1 tenant");

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// One tile row (16 rows) x NTJ column tiles of a complex product, accumulated INTO cre / cim (the caller zeroes them):
// per k-step one (a_r, a_i) fragment and NTJ (b_r, b_i) fragments.  Per output element the four MFMAs of a k-step come in
// pyramid.hip's order (re += ar br; im += ar bi; re -= ai bi; im += ai br), so the sums are bit-identical to that kernel's.
template <int K, int NTJ, class FAR, class FAI, class FBR, class FBI>
__device__ __forceinline__ void row_cplx(int lane, FAR ar, FAI ai, FBR br, FBI bi, f32x4 (&cre)[NTJ], f32x4 (&cim)[NTJ]) {
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + lk;
        const float a_r = ar(li, k), a_i = ai(li, k);
        float b_r[NTJ], b_i[NTJ];
#pragma unroll
        for (int tj = 0; tj < NTJ; ++tj) {
            b_r[tj] = br(k, tj * 16 + li);
            b_i[tj] = bi(k, tj * 16 + li);
        }
#pragma unroll
        for (int tj = 0; tj < NTJ; ++tj) {
            cre[tj] = mfma4(a_r, b_r[tj], cre[tj]);
            cim[tj] = mfma4(a_r, b_i[tj], cim[tj]);
            cre[tj] = mfma4(-a_i, b_i[tj], cre[tj]);
            cim[tj] = mfma4(a_i, b_r[tj], cim[tj]);
        }
    }
}

template <int K, int NTJ, class FA, class FB>
__device__ __forceinline__ void row_real(int lane, FA a, FB b, f32x4 (&c)[NTJ]) {
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int k0 = 0; k0 < K; k0 += 4) {
        const float av = a(li, k0 + lk);
#pragma unroll
        for (int tj = 0; tj < NTJ; ++tj) c[tj] = mfma4(av, b(k0 + lk, tj * 16 + li), c[tj]);
    }
}

template <int NTJ>
__device__ __forceinline__ void zero(f32x4 (&c)[NTJ]) {
#pragma unroll
    for (int t = 0; t < NTJ; ++t) c[t] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// The band masks are the same for every frame, so hipcc hoists their loads out of the frame loop and keeps ~150 registers of
// mask values alive (113 spilled).  Passing the pointer through an empty asm per use makes the loads opaque: they stay where
// they are written, L2 hits issued as one batch per pass.
__device__ __forceinline__ const float2* opaque(const float2* p) {
    asm volatile("" : "+s"(p));
    return p;
}

// wave-local LDS hand-over (a wave re-reads rows it wrote itself): order the ds_writes before the ds_reads
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// MFMA C layout: element e of lane (li, lk) is row 4 lk + e, column li of the 16x16 tile
template <int NTJ, class F>
__device__ __forceinline__ void store_tiles(int lane, const f32x4 (&c)[NTJ], F put) {
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int tj = 0; tj < NTJ; ++tj)
#pragma unroll
        for (int e = 0; e < 4; ++e) put(4 * lk + e, tj * 16 + li, c[tj][e]);
}

// ---- epilogue of one band plane: stage (re, im planes in LDS) -> polar -> two separable blurs -> four planes in HBM
template <int W>
__device__ __forceinline__ void polar_blur_store(int tid, float* w, float* __restrict__ o, int ablate) {
    if (ablate & 1) return;      // measurement only (MM_PF_ABLATE): no epilogue
    using C = Cfg<W>;
    using B_ = BL<W>;
    constexpr int ROUNDS = (C::ACTIVE + NT - 1) / NT;      // strips of 4 pixels per thread: 3 (W = 48) / 1 (W = 24)
    float* in_num = w + B_::IN_NUM;
    float* in_den = w + B_::IN_DEN;
    float* tmp_num = w + B_::TMP_NUM;
    float* tmp_den = w + B_::TMP_DEN;
    // zero rows above / below the row-pass planes (the region was a spectrum tenant a moment ago)
    for (int i = tid; i < R * W; i += NT) {
        tmp_num[i] = 0.f; tmp_num[(W + R) * W + i] = 0.f;
        tmp_den[i] = 0.f; tmp_den[(W + R) * W + i] = 0.f;
    }
    // (re, im) -> (mag * phase, mag) in place; phase and mag are final: stored now
#pragma unroll
    for (int u = 0; u < ROUNDS; ++u) {
        const int s = tid + u * NT;
        if (s >= C::ACTIVE) break;
        const int px = s * PX;                              // strips are row-major: px = y * W + x0
        const float4 re = *reinterpret_cast<const float4*>(in_num + px), im = *reinterpret_cast<const float4*>(in_den + px);
        float4 ph, mg;
        to_polar(re.x, im.x, ph.x, mg.x);
        to_polar(re.y, im.y, ph.y, mg.y);
        to_polar(re.z, im.z, ph.z, mg.z);
        to_polar(re.w, im.w, ph.w, mg.w);
        *reinterpret_cast<float4*>(in_num + px) = float4{mg.x * ph.x, mg.y * ph.y, mg.z * ph.z, mg.w * ph.w};
        *reinterpret_cast<float4*>(in_den + px) = mg;
        *reinterpret_cast<float4*>(o + px) = mg;
        *reinterpret_cast<float4*>(o + 3 * C::PLANE + px) = ph;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < ROUNDS; ++u) {
        const int s = tid + u * NT;
        if (s >= C::ACTIVE) break;
        const int y = s / C::STRIPS, x0 = (s - y * C::STRIPS) * PX;
        float hn[PX], hd[PX];
        row_pass<W>(in_num, y, x0, hn);
        row_pass<W>(in_den, y, x0, hd);
        *reinterpret_cast<float4*>(tmp_num + (y + R) * W + x0) = float4{hn[0], hn[1], hn[2], hn[3]};
        *reinterpret_cast<float4*>(tmp_den + (y + R) * W + x0) = float4{hd[0], hd[1], hd[2], hd[3]};
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < ROUNDS; ++u) {
        const int s = tid + u * NT;
        if (s >= C::ACTIVE) break;
        const int y = s / C::STRIPS, x0 = (s - y * C::STRIPS) * PX;
        float sn[PX], sd[PX];
        col_pass<W>(tmp_num, y, x0, sn);
        col_pass<W>(tmp_den, y, x0, sd);
        const int px = s * PX;
        *reinterpret_cast<float4*>(o + C::PLANE + px) = float4{sn[0] / sd[0], sn[1] / sd[1], sn[2] / sd[2], sn[3] / sd[3]};
        *reinterpret_cast<float4*>(o + 2 * C::PLANE + px) = float4{1.0f / sd[0], 1.0f / sd[1], 1.0f / sd[2], 1.0f / sd[3]};
    }
    __syncthreads();      // the working region changes tenant
}

// signed-frequency helpers of the inverse transform along the full axis (index k of the 2H-long axis <-> frequency k - H)
template <int H> __device__ __forceinline__ int af_of(int k) { int f = k - H; f = f < 0 ? -f : f; return f > S - 1 ? S - 1 : f; }

// ---- BAND 0 of a level: spectrum S[r][fv] on the half plane fv in [0,H), rows r = fu + H in [0,2H).
//      T[r][q] = sum_fv S[r][fv] E[fv][STEP q];  out[p][q] = sum_r F[p][r] T[r][q], F[p][r] = exp(2 pi i (r-H) p / 2H)
template <int H>
__device__ __forceinline__ void band0(int tid, float* lds, const float* __restrict__ mask, float* __restrict__ o, int ablate) {
    constexpr int N2 = 2 * H, STEP = S / H, MT = (H + 15) / 16;
    constexpr int HALVES = H == 48 ? 2 : 1, ROWS = N2 / HALVES;      // rows of r per pass: 48
    constexpr int LDS_ = H == 48 ? LD : 25;                          // S rows as A operand: odd stride
    constexpr int LDT = H == 48 ? LD : 48;                           // T rows as B operand (level 1: in place over S)
    constexpr int WT_RE = H == 48 ? 0 : 2 * ROWS * LDS_;             // level 2: T next to S
    static_assert(ROWS == 48, "three tile rows per pass");
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* ec = lds + L_EC;
    const float* es = lds + L_ES;
    const float* g = lds + L_G;
    float* w = lds + L_W;
    float* sre = w + WS_RE;
    float* sim = sre + ROWS * LDS_;
    float* tre = w + WT_RE;
    float* tim = tre + ROWS * LDT;
    f32x4 ore[MT], oim[MT];
    zero(ore); zero(oim);
#pragma unroll 1
    for (int hb = 0; hb < HALVES; ++hb) {
        // spectrum rows [hb * 48, hb * 48 + 48): all of a thread's mask loads first (L2 latency once, not once per element)
        if (!(ablate & 4)) {
            constexpr int PER = ROWS * H / NT;
            static_assert(ROWS * H % NT == 0, "whole elements per thread");
            const float2* m2 = opaque(reinterpret_cast<const float2*>(mask) + hb * ROWS * H);      // r * H + fv = hb * ROWS * H + idx
            float2 m[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) m[u] = m2[tid + u * NT];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int idx = tid + u * NT;
                const int rr = idx / H, fv = idx - rr * H;
                const int fu = rr + hb * ROWS - H, af = fu < 0 ? -fu : fu;
                const float gv = af >= S ? 0.f : g[af * LD + fv];
                sre[rr * LDS_ + fv] = gv * m[u].x;
                sim[rr * LDS_ + fv] = gv * m[u].y;
            }
        }
        __syncthreads();
        if (!(ablate & 2)) {   // T rows of this wave's tile row, all column tiles
            f32x4 cre[MT], cim[MT];
            zero(cre); zero(cim);
            row_cplx<H, MT>(lane,
                [&](int i, int k) { return sre[(wave * 16 + i) * LDS_ + k]; },
                [&](int i, int k) { return sim[(wave * 16 + i) * LDS_ + k]; },
                [&](int k, int j) { return ec[k * S + STEP * j]; },
                [&](int k, int j) { return es[k * S + STEP * j]; }, cre, cim);
            if (H == 48) wave_lds_fence();      // in place: this wave's S rows are dead, nobody else reads them
            store_tiles<MT>(lane, cre, [&](int row, int col, float v) { tre[(wave * 16 + row) * LDT + col] = v; });
            store_tiles<MT>(lane, cim, [&](int row, int col, float v) { tim[(wave * 16 + row) * LDT + col] = v; });
        }
        __syncthreads();
        if (wave < MT && !(ablate & 2)) {   // out rows p of this wave's tile row, contraction over this pass's 48 values of r
            row_cplx<ROWS, MT>(lane,
                [&](int i, int k) { return ec[af_of<H>(k + hb * ROWS) * S + STEP * (wave * 16 + i)]; },
                [&](int i, int k) { const float v = es[af_of<H>(k + hb * ROWS) * S + STEP * (wave * 16 + i)]; return k + hb * ROWS < H ? -v : v; },
                [&](int k, int j) { return tre[k * LDT + j]; },
                [&](int k, int j) { return tim[k * LDT + j]; }, ore, oim);
        }
        __syncthreads();   // T is dead: next pass / the stage may overwrite it
    }
    if (wave < MT) {
        float* st_re = w + BL<H>::IN_NUM;
        float* st_im = w + BL<H>::IN_DEN;
        store_tiles<MT>(lane, ore, [&](int row, int col, float v) { const int p = wave * 16 + row; if (p < H && col < H) st_re[p * H + col] = v; });
        store_tiles<MT>(lane, oim, [&](int row, int col, float v) { const int p = wave * 16 + row; if (p < H && col < H) st_im[p * H + col] = v; });
    }
    __syncthreads();
    polar_blur_store<H>(tid, w, o, ablate);
}

// ---- BAND 1: spectrum S[fu][c] on the half plane fu in [0,H), columns c = fv + H in [0,2H).
//      T'[p][c] = sum_fu E[fu][STEP p] S[fu][c];  out[p][q] = sum_c T'[p][c] F[q][c].  A wave's T' rows are the A operand of ITS OWN
//      rows of the second product: no workgroup barrier between the two.
template <int H>
__device__ __forceinline__ void band1(int tid, float* lds, const float* __restrict__ mask, float* __restrict__ o, int ablate) {
    constexpr int N2 = 2 * H, STEP = S / H, MT = (H + 15) / 16;
    constexpr int HALVES = H == 48 ? 2 : 1, COLS = N2 / HALVES;      // columns c per pass: 48
    constexpr int LDS1 = 48;                                         // S rows as B operand: stride = 16 mod 32
    static_assert(COLS == 48, "three column tiles per pass");
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* ec = lds + L_EC;
    const float* es = lds + L_ES;
    const float* g = lds + L_G;
    float* w = lds + L_W;
    float* sre = w + WS_RE;                       // [H][48]
    float* sim = sre + H * LDS1;
    float* tre = sim + H * LDS1;                  // [16 MT][49]
    float* tim = tre + 16 * MT * LD;
    f32x4 ore[MT], oim[MT];
    zero(ore); zero(oim);
#pragma unroll 1
    for (int hb = 0; hb < HALVES; ++hb) {
        if (!(ablate & 4)) {
            constexpr int PER = H * COLS / NT;
            static_assert(H * COLS % NT == 0, "whole elements per thread");
            const float2* m2 = opaque(reinterpret_cast<const float2*>(mask));
            float2 m[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int idx = tid + u * NT;
                const int fu = idx / COLS, cc = idx - fu * COLS;
                m[u] = m2[fu * N2 + cc + hb * COLS];
            }
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int idx = tid + u * NT;
                const int fu = idx / COLS, cc = idx - fu * COLS;
                const int fv = cc + hb * COLS - H, af = fv < 0 ? -fv : fv;
                const float gv = af >= S ? 0.f : g[fu * LD + af];
                sre[fu * LDS1 + cc] = gv * m[u].x;
                sim[fu * LDS1 + cc] = gv * m[u].y;
            }
        }
        __syncthreads();
        if (wave < MT && !(ablate & 2)) {
            f32x4 cre[3], cim[3];
            zero(cre); zero(cim);
            row_cplx<H, 3>(lane,
                [&](int i, int k) { return ec[k * S + STEP * (wave * 16 + i)]; },
                [&](int i, int k) { return es[k * S + STEP * (wave * 16 + i)]; },
                [&](int k, int j) { return sre[k * LDS1 + j]; },
                [&](int k, int j) { return sim[k * LDS1 + j]; }, cre, cim);
            store_tiles<3>(lane, cre, [&](int row, int col, float v) { tre[(wave * 16 + row) * LD + col] = v; });
            store_tiles<3>(lane, cim, [&](int row, int col, float v) { tim[(wave * 16 + row) * LD + col] = v; });
            wave_lds_fence();
            row_cplx<COLS, MT>(lane,
                [&](int i, int k) { return tre[(wave * 16 + i) * LD + k]; },
                [&](int i, int k) { return tim[(wave * 16 + i) * LD + k]; },
                [&](int k, int j) { return ec[af_of<H>(k + hb * COLS) * S + STEP * j]; },
                [&](int k, int j) { const float v = es[af_of<H>(k + hb * COLS) * S + STEP * j]; return k + hb * COLS < H ? -v : v; },
                ore, oim);
        }
        __syncthreads();   // S is dead (every wave read all of it): next pass / the stage may overwrite it
    }
    if (wave < MT) {
        float* st_re = w + BL<H>::IN_NUM;
        float* st_im = w + BL<H>::IN_DEN;
        store_tiles<MT>(lane, ore, [&](int row, int col, float v) { const int p = wave * 16 + row; if (p < H && col < H) st_re[p * H + col] = v; });
        store_tiles<MT>(lane, oim, [&](int row, int col, float v) { const int p = wave * 16 + row; if (p < H && col < H) st_im[p * H + col] = v; });
    }
    __syncthreads();
    polar_blur_store<H>(tid, w, o, ablate);
}

// grid-stride over frames; tables are loaded into LDS once per workgroup.  f1 / f2: frame planes of level 1 / 2,
// [n][band]{mag, B, R, phase: W*W floats each}.
__global__ void __launch_bounds__(NT, 2)      // two waves per SIMD: 6 waves of two workgroups on a CU's four SIMDs
pyramid_frame_kernel(const float* __restrict__ tables, const float* __restrict__ frames, int64_t n, float* __restrict__ f1,
                     float* __restrict__ f2, int ablate) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int tid = threadIdx.x;
    for (int i = tid; i < S * S; i += NT) {
        const int f = i / S, m = i - f * S;
        lds[L_DCT + f * LD + m] = tables[OFF_DCT + i];
        lds[L_EC + i] = tables[OFF_EC + i];
        lds[L_ES + i] = tables[OFF_ES + i];
    }
    const float* dct = lds + L_DCT;
    float* w = lds + L_W;
    // measurement knob (MM_PF_ABLATE >> 8): the second workgroup of a CU starts late by that many x 3.4 us, so that the two do not
    // walk through their MFMA / VALU / memory phases in lock step
    if ((blockIdx.x >> 8) & 1)
        for (int i = 0; i < (ablate >> 8); ++i) __builtin_amdgcn_s_sleep(127);
    for (int64_t img = blockIdx.x; img < n; img += gridDim.x) {
        // Every LDS address below is a function of the thread id and compile-time constants, i.e. invariant across frames: hipcc
        // hoists ~150 of them out of this loop and spills 90 to scratch.  An opaque thread id per frame keeps the address
        // arithmetic (a few hundred VALU instructions per frame) next to its use.
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        {   // x -> LDS
            const float4* src = reinterpret_cast<const float4*>(frames + img * (S * S));
            float4* dst = reinterpret_cast<float4*>(w + WX);
            for (int i = tid; i < S * S / 4; i += NT) dst[i] = src[i];
        }
        __syncthreads();   // also covers the table load on the first pass
        {   // T1 = D x (this wave's 16 rows), then G = T1 D^T for the same rows: T1 rows are private to the wave
            const float* x = w + WX;
            float* t1 = w + WT1;
            float* g = lds + L_G;
            f32x4 c[3];
            zero(c);
            row_real<S, 3>(lane, [&](int i, int k) { return dct[(wave * 16 + i) * LD + k]; }, [&](int k, int j) { return x[k * S + j]; }, c);
            store_tiles<3>(lane, c, [&](int row, int col, float v) { t1[(wave * 16 + row) * LD + col] = v; });
            wave_lds_fence();
            zero(c);
            row_real<S, 3>(lane, [&](int i, int k) { return t1[(wave * 16 + i) * LD + k]; }, [&](int k, int j) { return dct[j * LD + k]; }, c);
            store_tiles<3>(lane, c, [&](int row, int col, float v) { g[(wave * 16 + row) * LD + col] = v; });
        }
        __syncthreads();
        float* o1 = f1 + img * (2 * Cfg<48>::FRAME_FLOATS);
        float* o2 = f2 + img * (2 * Cfg<24>::FRAME_FLOATS);
        band0<48>(tid, lds, tables + OFF_M1B0, o1, ablate);
        band0<24>(tid, lds, tables + OFF_M2B0, o2, ablate);
        band1<48>(tid, lds, tables + OFF_M1B1, o1 + Cfg<48>::FRAME_FLOATS, ablate);
        band1<24>(tid, lds, tables + OFF_M2B1, o2 + Cfg<24>::FRAME_FLOATS, ablate);
    }
}

}  // namespace pf

int64_t phase_frames_floats(int W, int64_t n) {
    return n * 2 * (W == 48 ? blur::Cfg<48>::FRAME_FLOATS : blur::Cfg<24>::FRAME_FLOATS);
}

int launch_pyramid_waves(const mm_pyramid* h, const float* frames, int64_t n, float* f1, float* f2, hipStream_t stream);   // pyramid_wave.hip

// frames [n][48][48] -> frame planes of both levels (the whole per-unique-frame part of the fused phase stage)
int launch_pyramid_frames(const mm_pyramid* h, const float* frames, int64_t n, float* f1, float* f2, hipStream_t stream) {
    if (n <= 0) return MM_OK;
    // round 6: one wave per frame (pyramid_wave.hip) for every whole round of 2 048 frames and for a remainder above 512 frames, the
    // three-wave-workgroup kernel below for what is left (finer grained: 0.058 vs 0.120 ms on 64 frames).  The two kernels' planes are
    // bit-identical, so the split changes no result.  MM_PF_WAVE (read per call: a test switches it): 0 = this kernel only, 2 = the wave
    // kernel only.
    const char* pf_wave = getenv("MM_PF_WAVE");
    const int mode = pf_wave ? atoi(pf_wave) : 1;
    if (mode != 0) {
        // whole rounds first, then the remainder as a launch of its own (its waves per workgroup follow ITS size)
        const int64_t rest = n % 2048;
        const int64_t part[2] = {n - rest, mode == 2 || rest > 512 ? rest : 0};
        for (int i = 0; i < 2; ++i) {
            if (part[i] <= 0) continue;
            prof_before(1, (double)part[i] * (pyr::S * pyr::S * 4), stream, "pyramid_frame");
            const int rc = launch_pyramid_waves(h, frames, part[i], f1, f2, stream);
            prof_after(1, stream);
            if (rc != MM_OK) return rc;
            MM_LAUNCH_CHECK();
            frames += part[i] * (pyr::S * pyr::S);
            f1 += part[i] * (2 * blur::Cfg<48>::FRAME_FLOATS);
            f2 += part[i] * (2 * blur::Cfg<24>::FRAME_FLOATS);
            n -= part[i];
        }
        if (n <= 0) return MM_OK;
    }
    // MM_PF_LDS_PAD / MM_PF_ABLATE (attribution of the kernel's time to its parts; results wrong by construction) exist only in a
    // library built with -DMM_MEASURE (tools/ scripts); the default build ignores the variables and always runs the full kernel
#ifdef MM_MEASURE
    static const int lds_pad = getenv("MM_PF_LDS_PAD") ? atoi(getenv("MM_PF_LDS_PAD")) : 0;   // measurement knob: force one workgroup per CU
    static const int ablate = getenv("MM_PF_ABLATE") ? atoi(getenv("MM_PF_ABLATE")) : 0;     // measurement knob: skip parts of the kernel
#else
    constexpr int lds_pad = 0, ablate = 0;
#endif
    const int lds_bytes = pf::L_TOTAL * 4 + lds_pad;
    MM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pf::pyramid_frame_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               lds_bytes));
#ifdef MM_MEASURE
    static const int grid_cap = getenv("MM_PF_GRID") ? atoi(getenv("MM_PF_GRID")) : 2048;
#else
    constexpr int grid_cap = 2048;
#endif
    int64_t grid = n;
    if (grid > grid_cap) grid = grid_cap;   // 256 CUs x 2 resident workgroups x 4 rounds; the rest grid-strides (tables stay in LDS)
    prof_before(1, (double)n * (pyr::S * pyr::S * 4), stream, "pyramid_frame");   // algorithmic read of the stage: one fp32 frame
    hipLaunchKernelGGL(pf::pyramid_frame_kernel, dim3((unsigned)grid), dim3(pf::NT), lds_bytes, stream, h->d_tables, frames, n, f1, f2, ablate);
    prof_after(1, stream);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

}  // namespace mm
