// ONE WAVE per unique frame (round 6): the whole per-frame part of the fused phase stage -- complex steerable pyramid (levels 1 and 2,
// kept quadrant), atan2 / magnitude and the two frame-linear Gaussian blurs -- with no workgroup barrier after the table load.
//
// Replaces, per 48x48 frame (api/tester.py:122-139 with the published configuration), exactly what pyramid_frames.hip replaces:
//   symmetric_extension_batch              api/utils/phase_utils.py:116-129        (folded into the DCT identity)
//   SCFpyr_PyTorch.build / _build_levels   api/steerable/SCFpyr_PyTorch.py:70-208
//   build_pyramid's quadrant keep          api/phase_difference_extractor.py:76-92
//   extract: atan2 / magnitude and the two blurs of amplitude_based_gaussian_blur that are linear in the frame
//                                          api/phase_difference_extractor.py:100-104, api/utils/phase_utils.py:78-90
// and writes the same four W x W planes per (frame, band, level): mag, B = blur(mag phase) / blur(mag), R = 1 / blur(mag), phase.
//
// Why (pyramid_frames.hip, round 3: 0.315 ms per 2 048 frames against an MFMA floor of 0.116): three-wave workgroups put six waves on a
// CU's four SIMDs (two SIMDs carry twice the matrix work of the other two), every product's result goes through LDS to the waves that
// need it, and the phases of a frame (mask loads, products, polar, blur) are separated by workgroup barriers.  Here:
//   * a wave owns a frame end to end; eight waves (eight frames) per workgroup share the twiddle / DCT tables in LDS, two waves on
//     every SIMD, each at its own place in its frame: nothing waits for anything but its own data
//   * the accumulator of v_mfma_f32_16x16x4_f32 IS an operand of the next product: register e of lane (li, lk) holds row 4 lk + e, a
//     k-slot layout of the following MFMA's A or B operand.  T = S E feeds out = F T from registers, D x feeds G the same way, and the
//     band spectrum S = G x mask is made in registers from a coalesced fragment-ordered copy of the mask (pyramid_tables.h): no
//     spectrum, no T and no T1 in LDS (15.9 KB of LDS per frame instead of 78)
//   * rows are dealt to lanes through frag_row() so that every such chained product still contracts in ascending k: operand values
//     and MFMA k order per output element are those of pyramid_frames.hip / pyramid.hip, i.e. the coefficients -- and with them the
//     phase and magnitude planes -- are BIT-IDENTICAL to the round-3 kernel's
//   * the two separable 11-tap blurs are banded products on the matrix pipe as well (K in K^T, K the 48x48 Toeplitz matrix of the
//     taps; all-zero 16x4 blocks skipped): rows first, then columns, ascending taps -- the fused-multiply-add chain of
//     phase_blur.h's row_pass / col_pass with exact zero terms added, so B and R come out identical too (tests/test_phase_gpu.py)
#include <cstdlib>
#include "mm_common.h"
#include "phase_math.h"
#include "phase_blur.h"
#include "pyramid_tables.h"

namespace mm {

namespace pw {

using namespace pyr;
using blur::Cfg;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LD = 49;                          // odd row stride of everything read as an MFMA operand
constexpr int L_EC = 0;                         // [48][48]
constexpr int L_ES = L_EC + S * S;              // [48][48]
constexpr int L_DCT = L_ES + S * S;             // [48][49]
constexpr int L_TAB = L_DCT + S * LD;           // 6 960 floats of tables per workgroup
constexpr int G_FLOATS = 2404;                  // per wave: G [49][49] -- row 48 and column 48 stay zero (frequency 48 of the mirrored axis)
constexpr int SC_FLOATS = 2 * 16 * LD;          // per wave: the blur's hand-over of one 16-row strip, two planes
constexpr int WAVE_FLOATS = G_FLOATS + SC_FLOATS;
constexpr int lds_bytes(int waves) { return (L_TAB + waves * WAVE_FLOATS) * 4; }
static_assert(G_FLOATS >= 49 * LD && G_FLOATS % 4 == 0, "G with its zero border");
static_assert(lds_bytes(8) <= 160 * 1024, "eight frames per CU");

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int A, int B>
__device__ __forceinline__ void zero(f32x4 (&c)[A][B]) {
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
        for (int j = 0; j < B; ++j) c[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}
template <int A>
__device__ __forceinline__ void zero(f32x4 (&c)[A]) {
#pragma unroll
    for (int i = 0; i < A; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// a wave re-reads LDS it wrote itself: keep the compiler from moving the reads above the writes (the LDS queue of a wave is in order)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float tap(int t) { return (unsigned)t < (unsigned)blur::TAP ? blur::c_g[t] : 0.f; }

// frame-invariant per-lane values
struct Lane {
    int li, lk, pli;
    float kf1[3][4];      // row pass:    K[16 tx + li][16 ti + 4 e + lk], [tx - ti + 1][e]
    float kf2[8];         // column pass: K[16 pt + li][16 pt - 8 + 4 s + lk], [s]
};

// ---- x -> G = D x D^T into this wave's G[f][f2] (LDS).  T1^T = x^T D^T with x straight from HBM as the A operand (rows dealt by
//      frag_row), then G^T = D T1^T with the accumulators of the first product as the B operand of the second.
__device__ __forceinline__ void dct_stage(const float* __restrict__ x, const float* dct, float* g, const Lane& L) {
    float xa[S / 4][3];
#pragma unroll
    for (int ks = 0; ks < S / 4; ++ks)
#pragma unroll
        for (int t = 0; t < 3; ++t) xa[ks][t] = x[(4 * ks + L.lk) * S + 16 * t + L.pli];
    f32x4 t1[3][3];
    zero(t1);
#pragma unroll
    for (int ks = 0; ks < S / 4; ++ks) {
        float b[3];
#pragma unroll
        for (int tj = 0; tj < 3; ++tj) b[tj] = dct[(16 * tj + L.li) * LD + 4 * ks + L.lk];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int tj = 0; tj < 3; ++tj) t1[t][tj] = mfma4(xa[ks][t], b[tj], t1[t][tj]);
    }
    f32x4 gz[3][3];
    zero(gz);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a[3];
#pragma unroll
            for (int pt = 0; pt < 3; ++pt) a[pt] = dct[(16 * pt + L.li) * LD + 16 * t + 4 * e + L.lk];
#pragma unroll
            for (int pt = 0; pt < 3; ++pt)
#pragma unroll
                for (int tj = 0; tj < 3; ++tj) gz[pt][tj] = mfma4(a[pt], t1[t][tj][e], gz[pt][tj]);
        }
#pragma unroll
    for (int pt = 0; pt < 3; ++pt)
#pragma unroll
        for (int tj = 0; tj < 3; ++tj)
#pragma unroll
            for (int e = 0; e < 4; ++e) g[(16 * tj + L.li) * LD + 16 * pt + 4 * L.lk + e] = gz[pt][tj][e];
    wave_lds_fence();
}

template <int H> __device__ __forceinline__ int af_of(int k) { int f = k - H; f = f < 0 ? -f : f; return f > S - 1 ? S - 1 : f; }

// ---- one band of one level.  With r the full (2H-long) frequency axis of the band's half plane and k the half one:
//        T[r][.] = sum_k S[r][k] E[k][.]      S = G x mask: band 0 rows r = fu + H, columns fv; band 1 rows r = fv + H, columns fu
//        Z[x][y] = sum_r F[.][r] T[r][.]      the kept quadrant, held TRANSPOSED (rows x): the row pass of the blur contracts over x
//      16 rows of r at a time: T of the tile (registers) is consumed by the second product at once.  The MFMA order inside a complex
//      step is pyramid_frames.hip's for the same output element (band 1 is that kernel's band 1 with the operand roles swapped).
template <int H>
__device__ __forceinline__ void load_masks(const float2* __restrict__ mfrag, int tr, int lane, float2 (&m)[H / 4]) {
    const float2* mp = mfrag + (size_t)(tr * (H / 4)) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < H / 4; ++ks) m[ks] = mp[ks * 64];
}

// m: the mask fragments of tile row 0, loaded by the caller (during the previous band's epilogue); the next tile row's are fetched under
// this one's second product.  prefetch_next() is called between the products and the epilogue (the next band's first fragments).
template <int H, int BAND, class Prefetch>
__device__ __forceinline__ void band(const float* ec, const float* es, const float* g, float* sc, const float2* __restrict__ mfrag,
                                     float* __restrict__ o, int lane, const Lane& L, float2 (&m)[H / 4], Prefetch&& prefetch_next) {
    constexpr int STEP = S / H, MT = (H + 15) / 16, NTR = 2 * H / 16, KS = H / 4, PLANE = H * H;
    constexpr int KS_ZERO = H == 48 ? EDGE_ZERO_KSTEPS : 0;
    const int li = L.li, lk = L.lk, pli = L.pli;
    f32x4 zre[MT][MT], zim[MT][MT];      // [x tile][y tile]; register e of lane (li, lk): x = 16 tx + 4 e + lk, y = 16 ty + li
    zero(zre);
    zero(zim);
    const int ebase = lk * S + STEP * (BAND == 0 ? pli : li);      // E[4 ks + lk][STEP (16 tj + column)]
#pragma unroll 1
    for (int tr = 0; tr < NTR; ++tr) {
        // A fragments of the spectrum tile: rows 16 tr + frag_row(li)
        int fa = 16 * tr + pli - H;
        fa = fa < 0 ? -fa : fa;                                    // |frequency| <= 48; 48 reads G's zero border
        const int gbase = BAND == 0 ? fa * LD + lk : lk * LD + fa;
        constexpr int GSTEP = BAND == 0 ? 4 : 4 * LD;
        f32x4 tre[MT], tim[MT];
        zero(tre);
        zero(tim);
        // level 1: the spectrum is zero beyond radius 48, i.e. in the last KS_ZERO k-steps of the first and the last tile row (checked when
        // the tables are packed): exact zeros into a fused multiply-add chain, skipped
        const bool edge = KS_ZERO > 0 && (tr == 0 || tr == NTR - 1);
        auto kstep = [&](int ks) {
            const float gv = g[gbase + ks * GSTEP];
            const float a_r = gv * m[ks].x, a_i = gv * m[ks].y, na_i = -a_i;
#pragma unroll
            for (int tj = 0; tj < MT; ++tj) {
                const float b_r = ec[ebase + ks * 4 * S + tj * 16 * STEP], b_i = es[ebase + ks * 4 * S + tj * 16 * STEP];
                if (BAND == 0) {
                    tre[tj] = mfma4(a_r, b_r, tre[tj]);
                    tim[tj] = mfma4(a_r, b_i, tim[tj]);
                    tre[tj] = mfma4(na_i, b_i, tre[tj]);
                    tim[tj] = mfma4(a_i, b_r, tim[tj]);
                } else {
                    tre[tj] = mfma4(a_r, b_r, tre[tj]);
                    tim[tj] = mfma4(a_i, b_r, tim[tj]);
                    tre[tj] = mfma4(na_i, b_i, tre[tj]);
                    tim[tj] = mfma4(a_r, b_i, tim[tj]);
                }
            }
        };
#pragma unroll
        for (int ks = 0; ks < KS - KS_ZERO; ++ks) kstep(ks);
        if (KS_ZERO > 0 && !edge) {
#pragma unroll
            for (int ks = KS - KS_ZERO; ks < KS; ++ks) kstep(ks);
        }
        load_masks<H>(mfrag, tr + 1 < NTR ? tr + 1 : tr, lane, m);      // (the last one again: no branch)
        // second product over this tile's 16 values of r: step e covers r = 16 tr + 4 e + lk
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int fr = af_of<H>(16 * tr + 4 * e + lk);
            const float sg = 16 * tr + 4 * e < H ? -1.0f : 1.0f;   // F = exp(2 pi i (r - H) . / 2H): negative frequencies, conjugate twiddle
            if (BAND == 0) {
                // A = T (accumulators: rows = T's columns x), B = F[y = 16 pt + li][r]
#pragma unroll
                for (int pt = 0; pt < MT; ++pt) {
                    const int fo = fr * S + STEP * (16 * pt + li);
                    const float f_r = ec[fo], f_i = es[fo] * sg, nf_i = -f_i;
#pragma unroll
                    for (int qt = 0; qt < MT; ++qt) {
                        zre[qt][pt] = mfma4(tre[qt][e], f_r, zre[qt][pt]);
                        zim[qt][pt] = mfma4(tim[qt][e], f_r, zim[qt][pt]);
                        zre[qt][pt] = mfma4(tim[qt][e], nf_i, zre[qt][pt]);
                        zim[qt][pt] = mfma4(tre[qt][e], f_i, zim[qt][pt]);
                    }
                }
            } else {
                // A = F[x = 16 qt + frag_row(li)][r], B = T (accumulators: columns y)
#pragma unroll
                for (int qt = 0; qt < MT; ++qt) {
                    const int fo = fr * S + STEP * (16 * qt + pli);
                    const float f_r = ec[fo], f_i = es[fo] * sg, nf_i = -f_i;
#pragma unroll
                    for (int pt = 0; pt < MT; ++pt) {
                        zre[qt][pt] = mfma4(f_r, tre[pt][e], zre[qt][pt]);
                        zim[qt][pt] = mfma4(f_i, tre[pt][e], zim[qt][pt]);
                        zre[qt][pt] = mfma4(nf_i, tim[pt][e], zre[qt][pt]);
                        zim[qt][pt] = mfma4(f_r, tim[pt][e], zim[qt][pt]);
                    }
                }
            }
        }
    }
    prefetch_next();
    // ---- polar: phase and magnitude are final (stored); (re, im) -> (mag phase, mag) in place.  Level 2 computes 32 x 32: columns x >= 24
    //      (whole registers) are skipped, rows y >= 24 (lanes) are computed and not stored.
    // (store addresses: one opaque per-lane pointer per band + constants, or hipcc keeps a 64-bit index pair per pixel alive across the bands)
    int sb_in = li * H + lk, sb_out = 4 * lk * H + li;
    asm volatile("" : "+v"(sb_in), "+v"(sb_out));
    float* const o_in = o + sb_in;
    float* const o_out = o + sb_out;
#pragma unroll
    for (int tx = 0; tx < MT; ++tx)
#pragma unroll
        for (int ty = 0; ty < MT; ++ty) {
            const bool rows_ok = 16 * ty + 16 <= H || 16 * ty + li < H;
            float ph[4], mg[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (16 * tx + 4 * e >= H) continue;
                to_polar(zre[tx][ty][e], zim[tx][ty][e], ph[e], mg[e]);
                zre[tx][ty][e] = mg[e] * ph[e];
                zim[tx][ty][e] = mg[e];
            }
            if (rows_ok) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (16 * tx + 4 * e >= H) continue;
                    o_in[16 * ty * H + 16 * tx + 4 * e] = mg[e];
                    o_in[3 * PLANE + 16 * ty * H + 16 * tx + 4 * e] = ph[e];
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // four pixels at a time: the scheduler otherwise interleaves all 36
        }
    // ---- blur, one 16-column strip of the output at a time: rows (contract x', from the registers), hand over through LDS,
    //      columns (contract y'), divide, store
    float* sc_n = sc;
    float* sc_d = sc + 16 * LD;
#pragma unroll
    for (int tx = 0; tx < MT; ++tx) {
        f32x4 n1[MT], d1[MT];
        zero(n1);
        zero(d1);
#pragma unroll
        for (int ti = (tx > 0 ? tx - 1 : 0); ti <= (tx + 1 < MT ? tx + 1 : MT - 1); ++ti)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (16 * ti + 4 * e >= H) continue;                // beyond the plane: zero padding, nothing to add
                if ((ti < tx && e < 2) || (ti > tx && e >= 2)) continue;   // more than five columns away: K is zero on the whole block
                const float kf = L.kf1[tx - ti + 1][e];
#pragma unroll
                for (int ty = 0; ty < MT; ++ty) {
                    n1[ty] = mfma4(kf, zre[ti][ty][e], n1[ty]);
                    d1[ty] = mfma4(kf, zim[ti][ty][e], d1[ty]);
                }
            }
#pragma unroll
        for (int ty = 0; ty < MT; ++ty)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sc_n[(4 * lk + e) * LD + 16 * ty + li] = n1[ty][e];
                sc_d[(4 * lk + e) * LD + 16 * ty + li] = d1[ty][e];
            }
        wave_lds_fence();
        f32x4 n2[MT], d2[MT];
        zero(n2);
        zero(d2);
#pragma unroll
        for (int pt = 0; pt < MT; ++pt)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int k0 = 16 * pt - 8 + 4 * s;
                if (k0 < 0 || k0 >= H) continue;
                n2[pt] = mfma4(L.kf2[s], sc_n[li * LD + k0 + lk], n2[pt]);
                d2[pt] = mfma4(L.kf2[s], sc_d[li * LD + k0 + lk], d2[pt]);
            }
#pragma unroll
        for (int pt = 0; pt < MT; ++pt) {
            // register e of lane (li, lk): y = 16 pt + 4 lk + e, x = 16 tx + li
            const bool ok = (16 * tx + 16 <= H || 16 * tx + li < H) && (16 * pt + 16 <= H || 16 * pt + 4 * lk + 3 < H);
            static_assert(H % 4 == 0, "a lane's four rows are inside or outside together");
            if (ok) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o_out[PLANE + (16 * pt + e) * H + 16 * tx] = n2[pt][e] / d2[pt][e];
                    o_out[2 * PLANE + (16 * pt + e) * H + 16 * tx] = 1.0f / d2[pt][e];
                }
            }
        }
        wave_lds_fence();
    }
}

template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES)
pyramid_wave_kernel(const float* __restrict__ tables, const float* __restrict__ frames, int64_t n, float* __restrict__ f1,
                    float* __restrict__ f2) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < S * S; i += 64 * WAVES) {
        const int f = i / S, m = i - f * S;
        lds[L_DCT + f * LD + m] = tables[OFF_DCT + i];
        lds[L_EC + i] = tables[OFF_EC + i];
        lds[L_ES + i] = tables[OFF_ES + i];
    }
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* g = lds + L_TAB + wave * WAVE_FLOATS;
    float* sc = g + G_FLOATS;
    if (lane < 49) {
        g[48 * LD + lane] = 0.f;
        g[lane * LD + 48] = 0.f;
    }
    Lane L;
    {
        const int li = lane & 15, lk = lane >> 4;
        L.li = li;
        L.lk = lk;
        L.pli = frag_row(li);
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int e = 0; e < 4; ++e) L.kf1[d][e] = tap(-16 * (d - 1) + 4 * e + L.lk - L.li + blur::R);
#pragma unroll
    for (int s = 0; s < 8; ++s) L.kf2[s] = tap(4 * s - 8 + L.lk - L.li + blur::R);
    __syncthreads();      // the only barrier: tables are in LDS
    const float* ec = lds + L_EC;
    const float* es = lds + L_ES;
    const float* dct = lds + L_DCT;
    const float2* t2 = reinterpret_cast<const float2*>(tables);
    for (int64_t img = (int64_t)blockIdx.x * WAVES + wave; img < n; img += (int64_t)gridDim.x * WAVES) {
        // every LDS / table address is a function of the lane id: opaque per frame, or hipcc hoists a few hundred of them out of this
        // loop and spills (pyramid_frames.hip met the same)
        int lane_ = lane;
        asm volatile("" : "+v"(lane_));
        L.li = lane_ & 15;
        L.lk = lane_ >> 4;
        L.pli = frag_row(L.li);
        float2 m48[12], m24[6];
        load_masks<48>(t2 + OFF_F1B0 / 2, 0, lane_, m48);
        dct_stage(frames + img * (S * S), dct, g, L);
        float* o1 = f1 + img * (2 * Cfg<48>::FRAME_FLOATS);
        float* o2 = f2 + img * (2 * Cfg<24>::FRAME_FLOATS);
        band<48, 0>(ec, es, g, sc, t2 + OFF_F1B0 / 2, o1, lane_, L, m48, [&]() { load_masks<24>(t2 + OFF_F2B0 / 2, 0, lane_, m24); });
        band<24, 0>(ec, es, g, sc, t2 + OFF_F2B0 / 2, o2, lane_, L, m24, [&]() { load_masks<48>(t2 + OFF_F1B1 / 2, 0, lane_, m48); });
        band<48, 1>(ec, es, g, sc, t2 + OFF_F1B1 / 2, o1 + Cfg<48>::FRAME_FLOATS, lane_, L, m48,
                    [&]() { load_masks<24>(t2 + OFF_F2B1 / 2, 0, lane_, m24); });
        band<24, 1>(ec, es, g, sc, t2 + OFF_F2B1 / 2, o2 + Cfg<24>::FRAME_FLOATS, lane_, L, m24, []() {});
    }
}

template <int WAVES>
int launch(const mm_pyramid* h, const float* frames, int64_t n, float* f1, float* f2, hipStream_t stream) {
    constexpr int bytes = lds_bytes(WAVES);
    MM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pyramid_wave_kernel<WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    int64_t grid = (n + WAVES - 1) / WAVES;
    if (grid > 4096) grid = 4096;      // the rest grid-strides
    hipLaunchKernelGGL(pyramid_wave_kernel<WAVES>, dim3((unsigned)grid), dim3(64 * WAVES), bytes, stream, h->d_tables, frames, n, f1, f2);
    return MM_OK;
}

}  // namespace pw

// frames [n][48][48] -> frame planes of both levels, one wave per frame.  Waves per workgroup by the number of frames, so that a small
// batch still spreads over the chip (a wave's arithmetic does not depend on it: same bits for every n).
int launch_pyramid_waves(const mm_pyramid* h, const float* frames, int64_t n, float* f1, float* f2, hipStream_t stream) {
    static_assert(pyr::OFF_F1B0 % 2 == 0 && pyr::OFF_F1B1 % 2 == 0 && pyr::OFF_F2B0 % 2 == 0 && pyr::OFF_F2B1 % 2 == 0, "float2 tables");
    if (n > 1024) return pw::launch<8>(h, frames, n, f1, f2, stream);
    if (n > 512) return pw::launch<4>(h, frames, n, f1, f2, stream);
    if (n > 256) return pw::launch<2>(h, frames, n, f1, f2, stream);
    return pw::launch<1>(h, frames, n, f1, f2, stream);
}

}  // namespace mm
