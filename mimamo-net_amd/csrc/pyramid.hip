// Complex steerable pyramid (levels 1 and 2, kept quadrant only) on gfx950.
//
// Replaces SCFpyr_PyTorch.build/_build_levels (api/steerable/SCFpyr_PyTorch.py:70-208) as it is
// driven by Phase_Difference_Extractor.build_pyramid (api/phase_difference_extractor.py:38-87)
// with symmetry=True, height=4, nbands=2.
//
// Math (one 48x48 frame x, mirrored to 96x96 by symmetric_extension_batch, phase_utils.py:116-129):
//   * the DFT of the mirrored image is separable and real up to a phase:
//       F[fu,fv] = exp(i pi (fu+fv)/96) * G[|fu|,|fv|],   G = D x D^T,  D[f,m] = 2 cos(pi f (2m+1)/96)
//     so the forward 96x96 complex FFT + fftshift collapses to two real 48x48x48 products.
//   * every band spectrum is G times a constant complex table M_b (lo0 * himask * anglemask *
//     (-i) * phase * 1/N^2, built in float64 on the host -- mm_masks.cpp).  The angular mask of
//     band 0 vanishes for fv < 0 and that of band 1 for fu < 0, so only a half plane is non-zero.
//   * only the [:N/2,:N/2] quadrant of each inverse transform is kept (build_pyramid :84-85), so
//     the inverse is two small complex products with the DFT twiddle table E[f,q] = exp(2 pi i f q/96)
//     (level 2 uses every second column: exp(2 pi i f q/48) = E[f,2q]).
//   * the hi-pass and low-pass residual IFFTs (SCFpyr_PyTorch.py:120-124,132-135) are never used by
//     inference and are not computed.
// All products run on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains), one
// 256-thread workgroup per frame (both bands off one DCT), operands staged in LDS with bank-conflict-free strides.
#include "mm_common.h"
#include "phase_math.h"
#include "pyramid_tables.h"

namespace mm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

using namespace pyr;          // S and the table offsets (pyramid_tables.h)
constexpr int LDD = 49;      // D as A/B operand: odd stride
constexpr int NT = 256;      // threads per workgroup (512 measured slower: 0.62 vs 0.46 ms per 2048 frames)
constexpr int NW = NT / 64;  // waves

// LDS carve (floats)
constexpr int L_DCT = 0;                       // [48][49]
constexpr int L_EC = L_DCT + S * LDD;          // [48][48]
constexpr int L_ES = L_EC + S * S;             // [48][48]
constexpr int L_G = L_ES + S * S;              // [48][48]
constexpr int L_S = L_G + S * S;               // 2 planes of 5376 (band-1 level-1: [48][112])
constexpr int S_PLANE = 5376;
constexpr int L_T = L_S + 2 * S_PLANE;         // 2 planes of 4656 (band-1 level-1: [48][97])
constexpr int T_PLANE = 4656;
constexpr int L_TOTAL = L_T + 2 * T_PLANE;     // 29232 floats = 116928 B
constexpr int L_X = L_T;                       // x [48][48]   (aliases T, dead before T is written)
constexpr int L_T1 = L_T + S * S;              // T1 [48][49]

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// One 16x16 complex output tile: C = A * B, A/B given by element functors (row, k) / (k, col).
template <int K, class FAR, class FAI, class FBR, class FBI>
__device__ __forceinline__ void tile_cplx(int lane, FAR ar, FAI ai, FBR br, FBI bi, f32x4& cre, f32x4& cim) {
    const int li = lane & 15, lk = lane >> 4;
    cre = f32x4{0.f, 0.f, 0.f, 0.f};
    cim = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + lk;
        const float a_r = ar(li, k), a_i = ai(li, k), b_r = br(k, li), b_i = bi(k, li);
        cre = mfma4(a_r, b_r, cre);
        cim = mfma4(a_r, b_i, cim);
        cre = mfma4(-a_i, b_i, cre);
        cim = mfma4(a_i, b_r, cim);
    }
}

template <int K, class FA, class FB>
__device__ __forceinline__ f32x4 tile_real(int lane, FA a, FB b) {
    const int li = lane & 15, lk = lane >> 4;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k0 = 0; k0 < K; k0 += 4) c = mfma4(a(li, k0 + lk), b(k0 + lk, li), c);
    return c;
}

// One band of one level.  H = side of the kept quadrant (48 level 1, 24 level 2); the level's grid
// is 2H x 2H, signed frequency f = index - H.  BAND 0: half plane fv in [0,H), rows r = fu + H.
// BAND 1: half plane fu in [0,H), columns c = fv + H.
template <int H, int BAND>
__device__ __forceinline__ void band_pass(float* lds, const float* __restrict__ mask, float* __restrict__ out_plane, int polar) {
    constexpr int N2 = 2 * H;
    constexpr int STEP = S / H;               // column step into E: exp(2 pi i f q / (2H)) = E[f][STEP*q]
    constexpr int MT = (H + 15) / 16;         // tiles covering the kept quadrant (3 / 2)
    constexpr int FT = N2 / 16;               // tiles covering the full side (6 / 3)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* ec = lds + L_EC;
    const float* es = lds + L_ES;
    const float* g = lds + L_G;
    float* sre = lds + L_S;
    float* sim = sre + S_PLANE;
    float* tre = lds + L_T;
    float* tim = tre + T_PLANE;

    if (BAND == 0) {
        constexpr int LDS0 = H + 1;   // S as A operand: odd stride
        constexpr int LDT0 = 48;      // T as B operand
        // ---- spectrum S[r][fv] = G[|fu|][fv] * M[r][fv]
        for (int idx = tid; idx < N2 * H; idx += NT) {
            const int r = idx / H, fv = idx - r * H;
            const int fu = r - H, af = fu < 0 ? -fu : fu;
            const float gv = af >= S ? 0.f : g[af * S + fv];
            const float2 m = reinterpret_cast<const float2*>(mask)[idx];
            sre[r * LDS0 + fv] = gv * m.x;
            sim[r * LDS0 + fv] = gv * m.y;
        }
        __syncthreads();
        // ---- T[r][q] = sum_fv S[r][fv] E[fv][STEP q]          (M = 2H, N = H, K = H)
        for (int t = wave; t < FT * MT; t += NW) {
            const int ti = t / MT, tj = t - ti * MT;
            f32x4 cre, cim;
            tile_cplx<H>(lane,
                [&](int i, int k) { return sre[(ti * 16 + i) * LDS0 + k]; },
                [&](int i, int k) { return sim[(ti * 16 + i) * LDS0 + k]; },
                [&](int k, int j) { return ec[k * S + STEP * (tj * 16 + j)]; },
                [&](int k, int j) { return es[k * S + STEP * (tj * 16 + j)]; }, cre, cim);
            const int col = tj * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = ti * 16 + (lane >> 4) * 4 + e;
                tre[row * LDT0 + col] = cre[e];
                tim[row * LDT0 + col] = cim[e];
            }
        }
        __syncthreads();
        // ---- out[p][q] = sum_r F[p][r] T[r][q],  F[p][r] = exp(2 pi i (r-H) p / 2H)   (M = H, N = H, K = 2H)
        float* stage = lds + L_S;  // S is dead
        for (int t = wave; t < MT * MT; t += NW) {
            const int ti = t / MT, tj = t - ti * MT;
            f32x4 cre, cim;
            auto af_of = [](int k) { int fu = k - H; fu = fu < 0 ? -fu : fu; return fu > S - 1 ? S - 1 : fu; };
            tile_cplx<N2>(lane,
                [&](int i, int k) { return ec[af_of(k) * S + STEP * (ti * 16 + i)]; },
                [&](int i, int k) { const float v = es[af_of(k) * S + STEP * (ti * 16 + i)]; return k < H ? -v : v; },
                [&](int k, int j) { return tre[k * LDT0 + tj * 16 + j]; },
                [&](int k, int j) { return tim[k * LDT0 + tj * 16 + j]; }, cre, cim);
            const int col = tj * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = ti * 16 + (lane >> 4) * 4 + e;
                if (row < H && col < H) reinterpret_cast<float2*>(stage)[row * H + col] = float2{cre[e], cim[e]};
            }
        }
    } else {
        constexpr int LDS1 = (H == 48) ? 112 : 48;  // S as B operand: stride = 16 mod 32
        constexpr int LDT1 = N2 + 1;                // T' as A operand: odd stride
        // ---- spectrum S[fu][c] = G[fu][|fv|] * M[fu][c]
        for (int idx = tid; idx < H * N2; idx += NT) {
            const int fu = idx / N2, c = idx - fu * N2;
            const int fv = c - H, af = fv < 0 ? -fv : fv;
            const float gv = af >= S ? 0.f : g[fu * S + af];
            const float2 m = reinterpret_cast<const float2*>(mask)[idx];
            sre[fu * LDS1 + c] = gv * m.x;
            sim[fu * LDS1 + c] = gv * m.y;
        }
        __syncthreads();
        // ---- T'[p][c] = sum_fu E[fu][STEP p] S[fu][c]          (M = H, N = 2H, K = H)
        for (int t = wave; t < MT * FT; t += NW) {
            const int ti = t / FT, tj = t - ti * FT;
            f32x4 cre, cim;
            tile_cplx<H>(lane,
                [&](int i, int k) { return ec[k * S + STEP * (ti * 16 + i)]; },
                [&](int i, int k) { return es[k * S + STEP * (ti * 16 + i)]; },
                [&](int k, int j) { return sre[k * LDS1 + tj * 16 + j]; },
                [&](int k, int j) { return sim[k * LDS1 + tj * 16 + j]; }, cre, cim);
            const int col = tj * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = ti * 16 + (lane >> 4) * 4 + e;
                if (row < H) {
                    tre[row * LDT1 + col] = cre[e];
                    tim[row * LDT1 + col] = cim[e];
                }
            }
        }
        __syncthreads();
        // ---- out[p][q] = sum_c T'[p][c] F[q][c],  F[q][c] = exp(2 pi i (c-H) q / 2H)   (M = H, N = H, K = 2H)
        float* stage = lds + L_S;
        for (int t = wave; t < MT * MT; t += NW) {
            const int ti = t / MT, tj = t - ti * MT;
            f32x4 cre, cim;
            auto af_of = [](int k) { int fv = k - H; fv = fv < 0 ? -fv : fv; return fv > S - 1 ? S - 1 : fv; };
            // rows >= H of T' (level-2 padding) are never written: clamp the row so reads stay defined
            auto row_of = [&](int i) { const int r = ti * 16 + i; return r < H ? r : H - 1; };
            tile_cplx<N2>(lane,
                [&](int i, int k) { return tre[row_of(i) * LDT1 + k]; },
                [&](int i, int k) { return tim[row_of(i) * LDT1 + k]; },
                [&](int k, int j) { return ec[af_of(k) * S + STEP * (tj * 16 + j)]; },
                [&](int k, int j) { const float v = es[af_of(k) * S + STEP * (tj * 16 + j)]; return k < H ? -v : v; },
                cre, cim);
            const int col = tj * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = ti * 16 + (lane >> 4) * 4 + e;
                if (row < H && col < H) reinterpret_cast<float2*>(stage)[row * H + col] = float2{cre[e], cim[e]};
            }
        }
    }
    __syncthreads();
    // ---- coalesced store of the [H][H][2] plane: (re, im) pairs, or (phase, magnitude) on the fused path
    {
        const float4* src = reinterpret_cast<const float4*>(lds + L_S);
        float4* dst = reinterpret_cast<float4*>(out_plane);
        for (int i = tid; i < H * H / 2; i += NT) {
            float4 v = src[i];
            if (polar) {
                float4 q;
                to_polar(v.x, v.y, q.x, q.y);
                to_polar(v.z, v.w, q.z, q.w);
                v = q;
            }
            dst[i] = v;
        }
    }
    __syncthreads();
}

// grid-stride over frames; both bands of a frame share its DCT (x -> T1 -> G); tables are loaded into LDS once per workgroup.
__global__ void __launch_bounds__(NT)
pyramid_kernel(const float* __restrict__ tables, const float* __restrict__ frames, int64_t n, int64_t group,
               float* __restrict__ c1, int64_t group_stride1, int64_t img_stride1, int64_t band_stride1,
               float* __restrict__ c2, int64_t group_stride2, int64_t img_stride2, int64_t band_stride2, int polar) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < S * S; i += NT) {
        const int f = i / S, m = i - f * S;
        lds[L_DCT + f * LDD + m] = tables[OFF_DCT + i];
        lds[L_EC + i] = tables[OFF_EC + i];
        lds[L_ES + i] = tables[OFF_ES + i];
    }
    const float* dct = lds + L_DCT;
    for (int64_t img = blockIdx.x; img < n; img += gridDim.x) {
        // ---- x -> LDS
        {
            const float4* src = reinterpret_cast<const float4*>(frames + img * (S * S));
            float4* dst = reinterpret_cast<float4*>(lds + L_X);
            for (int i = tid; i < S * S / 4; i += NT) dst[i] = src[i];
        }
        __syncthreads();
        // ---- T1[f][n] = sum_m D[f][m] x[m][n]
        {
            const float* x = lds + L_X;
            float* t1 = lds + L_T1;
            for (int t = wave; t < 9; t += NW) {
                const int ti = t / 3, tj = t - ti * 3;
                f32x4 c = tile_real<S>(lane, [&](int i, int k) { return dct[(ti * 16 + i) * LDD + k]; },
                                       [&](int k, int j) { return x[k * S + tj * 16 + j]; });
#pragma unroll
                for (int e = 0; e < 4; ++e) t1[(ti * 16 + (lane >> 4) * 4 + e) * LDD + tj * 16 + (lane & 15)] = c[e];
            }
        }
        __syncthreads();
        // ---- G[f][g] = sum_n T1[f][n] D[g][n]
        {
            const float* t1 = lds + L_T1;
            float* g = lds + L_G;
            for (int t = wave; t < 9; t += NW) {
                const int ti = t / 3, tj = t - ti * 3;
                f32x4 c = tile_real<S>(lane, [&](int i, int k) { return t1[(ti * 16 + i) * LDD + k]; },
                                       [&](int k, int j) { return dct[(tj * 16 + j) * LDD + k]; });
#pragma unroll
                for (int e = 0; e < 4; ++e) g[(ti * 16 + (lane >> 4) * 4 + e) * S + tj * 16 + (lane & 15)] = c[e];
            }
        }
        __syncthreads();
        // output plane of (img, band): images come in groups of `group` (one window on the drop-in path)
        const int64_t grp = img / group, pos = img - grp * group;
        float* o1 = c1 + grp * group_stride1 + pos * img_stride1;
        float* o2 = c2 + grp * group_stride2 + pos * img_stride2;
        band_pass<48, 0>(lds, tables + OFF_M1B0, o1, polar);
        band_pass<24, 0>(lds, tables + OFF_M2B0, o2, polar);
        band_pass<48, 1>(lds, tables + OFF_M1B1, o1 + band_stride1, polar);
        band_pass<24, 1>(lds, tables + OFF_M2B1, o2 + band_stride2, polar);
    }
}

int launch_pyramid(const mm_pyramid* h, const float* frames, int64_t n, int64_t group, float* c1, int64_t gs1,
                   int64_t is1, int64_t bs1, float* c2, int64_t gs2, int64_t is2, int64_t bs2, int polar, hipStream_t stream) {
    if (n <= 0) return MM_OK;
    static_assert(L_TOTAL * 4 <= 160 * 1024, "LDS budget");
    const int lds_bytes = L_TOTAL * 4;
    // per launch (microseconds): the attribute belongs to the current device's copy of the kernel
    MM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pyramid_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               lds_bytes));
    int64_t grid = n;
    if (grid > 1024) grid = 1024;  // 256 CUs x 1 resident workgroup; the rest grid-strides
    prof_before(1, (double)n * (S * S * 4), stream, "pyramid");  // algorithmic read of the stage: one fp32 frame
    hipLaunchKernelGGL(pyramid_kernel, dim3((unsigned)grid), dim3(NT), lds_bytes, stream, h->d_tables, frames, n,
                       group > 0 ? group : n, c1, gs1, is1, bs1, c2, gs2, is2, bs2, polar);
    prof_after(1, stream);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

int pyramid_table_floats() { return TABLE_FLOATS; }

int pack_pyramid_tables(const PyramidTables& t, std::vector<float>& packed) {
    packed.assign(TABLE_FLOATS, 0.f);
    std::copy(t.dct.begin(), t.dct.end(), packed.begin() + OFF_DCT);
    std::copy(t.ec.begin(), t.ec.end(), packed.begin() + OFF_EC);
    std::copy(t.es.begin(), t.es.end(), packed.begin() + OFF_ES);
    if (t.m1[0].size() != 96 * 48 * 2 || t.m1[1].size() != 96 * 48 * 2 || t.m2[0].size() != 48 * 24 * 2 ||
        t.m2[1].size() != 48 * 24 * 2)
        return MM_ERR_INVALID_ARG;
    std::copy(t.m1[0].begin(), t.m1[0].end(), packed.begin() + OFF_M1B0);
    std::copy(t.m1[1].begin(), t.m1[1].end(), packed.begin() + OFF_M1B1);
    std::copy(t.m2[0].begin(), t.m2[0].end(), packed.begin() + OFF_M2B0);
    std::copy(t.m2[1].begin(), t.m2[1].end(), packed.begin() + OFF_M2B1);
    // fragment order for pyramid_wave.hip (pyramid_tables.h): band 0 [r][fv] as is, band 1 [fu][c] read transposed
    for (int level = 1; level <= 2; ++level) {
        const int H = level == 1 ? 48 : 24, KS = H / 4, NTR = 2 * H / 16;
        for (int band = 0; band < 2; ++band) {
            const std::vector<float>& m = level == 1 ? t.m1[band] : t.m2[band];
            float* dst = packed.data() + (level == 1 ? (band ? OFF_F1B1 : OFF_F1B0) : (band ? OFF_F2B1 : OFF_F2B0));
            for (int tr = 0; tr < NTR; ++tr)
                for (int ks = 0; ks < KS; ++ks)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int row = 16 * tr + frag_row(lane & 15), col = 4 * ks + (lane >> 4);
                        const size_t src = band == 0 ? (size_t)row * H + col : (size_t)col * 2 * H + row;
                        dst[((tr * KS + ks) * 64 + lane) * 2] = m[2 * src];
                        dst[((tr * KS + ks) * 64 + lane) * 2 + 1] = m[2 * src + 1];
                        if (level == 1 && (tr == 0 || tr == NTR - 1) && ks >= KS - EDGE_ZERO_KSTEPS && (m[2 * src] != 0.f || m[2 * src + 1] != 0.f))
                            return MM_ERR_INVALID_ARG;      // pyramid_wave.hip skips these k-steps
                    }
        }
    }
    return MM_OK;
}

}  // namespace mm
