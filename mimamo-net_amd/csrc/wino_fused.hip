// Winograd F(4x4,3x3): the 36 position GEMMs AND the output transform in one kernel -- M never goes to HBM.
//
// The three-kernel form (winograd.hip + the batched launch of conv_mfma.hip) writes M = [36][tiles][Cout] and reads it
// back in the output transform: 2 x 2.25 x the layer's output bytes, which is what bounds the conv2_x / conv3_x layers
// (their position GEMMs are HBM-bound in fp32: 32..64 FLOP per byte of V).  Here a workgroup owns BM Winograd tiles x
// BN output channels and walks ALL 36 positions xi = (r, q) for them:
//     M_xi = V_xi[tiles, Cin] * U_xi[Cout, Cin]^T          fp32 MFMA (v_mfma_f32_32x32x2_f32), one 32x32 accumulator per wave
//     t[p]    += At[p][r] * M_xi        (r = 0..5, for the current column q)        VALU, in the shadow of the next MFMAs
//     Y[p][q'] += At[q'][q] * t[p]      (once per column q)                          16 output pixels x 16 registers
// and stores Y + bias (+ReLU) as the layer's NHWC output.  Y (256 registers), t (64) and M live in registers, so a wave
// owns its SIMD alone (512-register budget, one 256-thread workgroup per CU); operands stream global -> LDS directly
// (buffer_load ... lds, 64-deep K slabs of 32 KB, 3-deep ring, one s_barrier per slab = 32 MFMAs per wave).
// Same arithmetic as wino_out6_kernel (A^T M A evaluated column-first), same per-position GEMM summation order as the
// batched launch up to the k order inside a slab, so results agree with the three-kernel form to fp32 rounding.
// The reference computes these layers with torch's direct fp32 conv (third-party ResNet50, api/resnet50_extractor.py:
// 74-83); parity is checked against the oracle's direct convolution.
#include <cstdlib>
#include <type_traits>
#include "conv.h"
#include "wino_fused_slab.inc"

#ifndef MM_WF_SCHED
#define MM_WF_SCHED 1      // scheduling hints on (0: leave the order to hipcc; measurement knob)
#endif
#ifndef MM_WF_VALU
#define MM_WF_VALU 6       // VALU instructions placed behind each MFMA of a slab
#endif

namespace mm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WinoFusedParams {
    const float* V;      // [36][ntile][K]   (wino_in6_kernel)
    const float* U;      // [36][Cout][K]    (host, float64 -> fp32)
    const float* bias;   // [Cout] or null
    float* out;          // NHWC [B][H][W][Cout]
    int ntile, K, Cout, B, H, W, TH, TW, relu, tiles_n;
    float at_cols[24];   // At[q'][q] as [q][q']: the Y-update coefficients of column q (filled by the launcher)
};

// A^T of F(4x4,3x3): rows p = 0..3, columns r = 0..5 (Lavin & Gray)
static constexpr float kAtHost[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
__device__ constexpr float kAt[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};

template <int WGM, int WGN>
__global__ void __launch_bounds__(WGM * WGN * 64) __attribute__((amdgpu_waves_per_eu(1, 1)))
wino_fused_kernel(const WinoFusedParams p) {
    constexpr int NW = WGM * WGN;
    static_assert(WGM == 2 && WGN == 2, "the slab asm issues 4 + 4 operand pieces per wave: 64 tiles x 64 channels");
    constexpr int BM = 32 * WGM, BN = 32 * WGN, KS = 64, NBUF = 3;
    constexpr int ROWS = BM + BN;                    // operand rows of one slab (A rows then B rows), 256 B each
    constexpr int NIA = BM / (4 * NW), NIB = BN / (4 * NW);   // DMA instructions per wave per slab (4 rows = 1 KB each)
    constexpr int NL = NIA + NIB;
    constexpr int SLAB_BYTES = ROWS * KS * 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 31, lh = lane >> 5;
    const int tile_m = blockIdx.x / p.tiles_n, tile_n = blockIdx.x - tile_m * p.tiles_n;
    const int m_base = tile_m * BM, n_base = tile_n * BN;
    const int K = p.K, kslabs = K / KS;
    const int nslab = 36 * kslabs;

    // ---- DMA addressing.  A slab row is 64 floats = 16 slots of 16 B; slot s of row r holds k-quad s ^ (r & 15), so the
    //      lane-linear image a wave instruction writes (lane l -> row l >> 4, slot l & 15) is also conflict-free for the
    //      ds_read_b128 fragment reads (32 consecutive rows, one logical k-quad).
    const unsigned OOB = 0xFFFFFFFFu;
    unsigned va[NIA], vb[NIB];
#pragma unroll
    for (int it = 0; it < NIA; ++it) {
        const int row = (it * NW + wave) * 4 + (lane >> 4);
        const int t = m_base + row;
        const int kq = (lane & 15) ^ (row & 15);
        va[it] = t < p.ntile ? (unsigned)(t * K + kq * 4) * 4u : OOB;
    }
#pragma unroll
    for (int it = 0; it < NIB; ++it) {
        const int row = (it * NW + wave) * 4 + (lane >> 4);
        const int n = n_base + row;
        const int kq = (lane & 15) ^ (row & 15);
        vb[it] = n < p.Cout ? (unsigned)(n * K + kq * 4) * 4u : OOB;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    const unsigned plane_a = (unsigned)p.ntile * (unsigned)K * 4u;   // bytes of one V plane (host checks < 4 GB)
    const unsigned plane_b = (unsigned)p.Cout * (unsigned)K * 4u;

    auto dma1 = [&](const __amdgpu_buffer_rsrc_t& rsrc, unsigned voff, unsigned lds_byte) {
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_byte);
        unsigned keep;  // M0 is compiler-reserved: save it, point it at the 1 KB slab piece, restore it, all in one statement
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(m0v) : "memory");
    };
    // slab sequence: column q outer, row r middle, k slab inner; position plane xi = r * 6 + q
    int d_q = 0, d_r = 0, d_ks = 0, d_buf = 0;
    struct Next {
        __amdgpu_buffer_rsrc_t ra, rb;
        unsigned voff[NL];
        unsigned m0base;
    };
    // addresses of the next slab in the sequence (branch-free; past the last slab the offsets are out of range: the loads
    // keep the vmcnt bookkeeping uniform and write zeros into a slot nobody reads)
    auto dma_prepare = [&]() {
        Next nx;
        const bool live = d_q < 6;
        const int xi = live ? d_r * 6 + d_q : 0;
        nx.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.V) + (int64_t)xi * p.ntile * K, 0, plane_a, 0x00020000);
        nx.rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.U) + (int64_t)xi * p.Cout * K, 0, plane_b, 0x00020000);
        const unsigned koff = (unsigned)d_ks * (KS * 4u);
#pragma unroll
        for (int it = 0; it < NIA; ++it) nx.voff[it] = (va[it] == OOB || !live) ? OOB : va[it] + koff;
#pragma unroll
        for (int it = 0; it < NIB; ++it) nx.voff[NIA + it] = (vb[it] == OOB || !live) ? OOB : vb[it] + koff;
        nx.m0base = lds0 + (unsigned)d_buf * SLAB_BYTES + (unsigned)wave * 1024u;   // piece i of A at + i * NW KB, of B at + BM * 256 + i * NW KB
        const bool wrap_k = d_ks + 1 == kslabs;
        d_ks = wrap_k ? 0 : d_ks + 1;
        const bool wrap_r = wrap_k && d_r == 5;
        d_r = wrap_k ? (wrap_r ? 0 : d_r + 1) : d_r;
        d_q += wrap_r ? 1 : 0;
        d_buf = d_buf == NBUF - 1 ? 0 : d_buf + 1;
        return nx;
    };
    auto dma_issue = [&](const Next& nx) {   // prologue only: inside the loop the slab asm issues the pieces itself
#pragma unroll
        for (int it = 0; it < NIA; ++it) dma1(nx.ra, nx.voff[it], nx.m0base + it * (NW * 1024));
#pragma unroll
        for (int it = 0; it < NIB; ++it) dma1(nx.rb, nx.voff[NIA + it], nx.m0base + BM * (KS * 4) + it * (NW * 1024));
    };

    // fragment addressing inside a slab (bytes): logical k-quad 2t + lh of row `row` sits in slot (2t + lh) ^ (row & 15), i.e.
    // at row * 256 + ((lh ^ row) & 1) * 16 + ((t ^ ((row & 15) >> 1)) << 5)
    const unsigned c0a = (unsigned)((wm * 32 + lr) * (KS * 4) + (((lh ^ lr) & 1) << 4));
    const unsigned c0b = (unsigned)((BM + wn * 32 + lr) * (KS * 4) + (((lh ^ lr) & 1) << 4));
    const unsigned ysw = (unsigned)((lr & 15) >> 1);

    f32x16 Y[4][4], T[4], M[2];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int e = 0; e < 16; ++e) T[a][e] = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) Y[a][b][e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) M[0][e] = M[1][e] = 0.f;

    // t[p] += At[p][r] * M  (compile-time r: zero coefficients vanish, unit coefficients are adds)
    auto update_t = [&](const f32x16& Mx, auto rtag) {
        constexpr int r = decltype(rtag)::value;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const float c = kAt[pp][r];   // folds after unrolling
            if (c == 0.f) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) T[pp][e] = c == 1.f ? T[pp][e] + Mx[e] : fmaf(c, Mx[e], T[pp][e]);
        }
    };
    // Y[p][q'] += At[q'][qc] * t[p], t = 0; qc is wave-uniform at run time: the four coefficients sit in SGPRs (a zero
    // coefficient is an FMA with 0, not a branch: the block stays one scheduling region)
    auto update_y = [&](int qc) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const float c = p.at_cols[qc * 4 + qq];   // kernel argument: a scalar load, no branches
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                for (int e = 0; e < 16; ++e) Y[pp][qq][e] = fmaf(c, T[pp][e], Y[pp][qq][e]);
        }
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
            for (int e = 0; e < 16; ++e) T[pp][e] = 0.f;
    };

    // One slab = 32 MFMAs on one accumulator; `extra` is the VALU work deferred from the previous position (it reads the
    // OTHER accumulator, so it does not wait for this slab's MFMAs to drain).
    int buf = 0;
    auto slab = [&](f32x16& Mc, auto first_tag, auto extra) {
        constexpr bool first = decltype(first_tag)::value;
        Next nx = dma_prepare();   // address arithmetic of the slab after next
        unsigned m0b = __builtin_amdgcn_readfirstlane(nx.m0base);
        // slab landed (the next one may still be in flight); every wave is done reading the slot the slab after next goes
        // into.  lgkmcnt(0): hipcc may sink ds_reads / MFMAs of the previous slab below this point.  The offsets are tied to
        // the statement as operands so that their arithmetic is done BEFORE the wave parks at the barrier, not after it.
        asm volatile("s_waitcnt vmcnt(%9) lgkmcnt(0)\n\ts_barrier"
                     : "+v"(nx.voff[0]), "+v"(nx.voff[1]), "+v"(nx.voff[2]), "+v"(nx.voff[3]), "+v"(nx.voff[4]), "+v"(nx.voff[5]),
                       "+v"(nx.voff[6]), "+v"(nx.voff[7]), "+s"(m0b)
                     : "n"(NL) : "memory");
        // The 16 fragment reads, 32 MFMAs and 8 direct-to-LDS loads of the slab are ONE hand-scheduled asm statement
        // (wino_fused_slab.inc): hipcc serialises "ds_read, s_waitcnt lgkmcnt(0), 4 MFMAs" per k-step whatever the source
        // order or scheduling hints say, and issues the loads in front of the MFMAs -- with the SIMD to itself the wave
        // would idle the matrix pipe for an LDS round trip every 256 cycles and for the ~500 cycles the loads take to issue.
        const unsigned sb = lds0 + (unsigned)buf * (unsigned)SLAB_BYTES;
        unsigned keep;
        if constexpr (first)
            asm volatile(MM_WF_SLAB_FIRST : "+v"(Mc), "=&s"(keep) : "v"(c0a + sb), "v"(c0b + sb), "v"(ysw), "v"(nx.voff[0]), "v"(nx.voff[1]),
                         "v"(nx.voff[2]), "v"(nx.voff[3]), "v"(nx.voff[4]), "v"(nx.voff[5]), "v"(nx.voff[6]), "v"(nx.voff[7]),
                         "s"(nx.ra), "s"(nx.rb), "s"(m0b) : "memory", "scc", MM_WF_SLAB_CLOBBERS);
        else
            asm volatile(MM_WF_SLAB_NEXT : "+v"(Mc), "=&s"(keep) : "v"(c0a + sb), "v"(c0b + sb), "v"(ysw), "v"(nx.voff[0]), "v"(nx.voff[1]),
                         "v"(nx.voff[2]), "v"(nx.voff[3]), "v"(nx.voff[4]), "v"(nx.voff[5]), "v"(nx.voff[6]), "v"(nx.voff[7]),
                         "s"(nx.ra), "s"(nx.rb), "s"(m0b) : "memory", "scc", MM_WF_SLAB_CLOBBERS);
        extra();
        buf = buf == NBUF - 1 ? 0 : buf + 1;
    };
    auto nothing = [] {};

    dma_issue(dma_prepare());
    dma_issue(dma_prepare());
    for (int q = 0; q < 6; ++q) {
        // r = 0: the deferred work is the last row of the previous column, then that column's Y update (q = 0: M[1] and t
        // are zero, so both are no-ops on zeros)
        slab(M[0], std::true_type(), [&] { update_t(M[1], std::integral_constant<int, 5>()); update_y(q == 0 ? 0 : q - 1); });
        for (int ks = 1; ks < kslabs; ++ks) slab(M[0], std::false_type(), nothing);
        slab(M[1], std::true_type(), [&] { update_t(M[0], std::integral_constant<int, 0>()); });
        for (int ks = 1; ks < kslabs; ++ks) slab(M[1], std::false_type(), nothing);
        slab(M[0], std::true_type(), [&] { update_t(M[1], std::integral_constant<int, 1>()); });
        for (int ks = 1; ks < kslabs; ++ks) slab(M[0], std::false_type(), nothing);
        slab(M[1], std::true_type(), [&] { update_t(M[0], std::integral_constant<int, 2>()); });
        for (int ks = 1; ks < kslabs; ++ks) slab(M[1], std::false_type(), nothing);
        slab(M[0], std::true_type(), [&] { update_t(M[1], std::integral_constant<int, 3>()); });
        for (int ks = 1; ks < kslabs; ++ks) slab(M[0], std::false_type(), nothing);
        slab(M[1], std::true_type(), [&] { update_t(M[0], std::integral_constant<int, 4>()); });
        for (int ks = 1; ks < kslabs; ++ks) slab(M[1], std::false_type(), nothing);
    }
    update_t(M[1], std::integral_constant<int, 5>());
    update_y(5);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the over-issued tail loads

    // ---- epilogue.  MFMA C layout: column (output channel) = lane & 31, row (Winograd tile) = (e & 3) + 8 (e >> 2) + 4 lh, i.e.
    //      a lane owns ONE channel of 16 tiles: direct stores would be 4-byte, 256 per wave, and store-issue bound (measured:
    //      half of the K = 64 kernel's time).  Each wave transposes one output pixel position at a time through its private
    //      4 KB of the (dead) operand ring -- [tile][channel], rows padded to 36 floats -- and then owns 4 consecutive
    //      channels of a tile per lane: 16-byte stores, 128 contiguous bytes per tile, 64 store instructions per wave.
    __builtin_amdgcn_s_barrier();                     // every wave is done with the last slab before the ring is re-used
    constexpr int SLD = 36;
    float* st = lds + wave * (32 * SLD);
    const int cq = lane & 7, tr = lane >> 3;          // read side: channel quad, tile row within a group of 8
    const int n4 = n_base + wn * 32 + cq * 4;
    const bool nok = n4 < p.Cout;                     // Cout % 32 == 0 is required by the launcher, so a quad is all-in or all-out
    float4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (nok && p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + n4);
    const int tpi = p.TH * p.TW;
    int64_t obase[4];
    int ty4[4], tx4[4];
    bool tok[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int t = m_base + wm * 32 + g * 8 + tr;
        tok[g] = nok && t < p.ntile;
        const int tt = tok[g] ? t : 0;
        const int b = tt / tpi, rem = tt - b * tpi;
        const int ty = rem / p.TW, tx = rem - ty * p.TW;
        ty4[g] = 4 * ty;
        tx4[g] = 4 * tx;
        obase[g] = (((int64_t)b * p.H + 4 * ty) * p.W + 4 * tx) * p.Cout + n4;
    }
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
#pragma unroll
            for (int e = 0; e < 16; ++e) st[((e & 3) + 8 * (e >> 2) + 4 * lh) * SLD + lr] = Y[pp][qq][e];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            float4 v[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) v[g] = *reinterpret_cast<const float4*>(st + (g * 8 + tr) * SLD + cq * 4);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // the next position re-writes the region these reads came from
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (!tok[g] || ty4[g] + pp >= p.H || tx4[g] + qq >= p.W) continue;
                float4 o = {v[g].x + bias4.x, v[g].y + bias4.y, v[g].z + bias4.z, v[g].w + bias4.w};
                if (p.relu) o = float4{fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f)};
                typedef float f32x4_t __attribute__((ext_vector_type(4)));
                const f32x4_t ov = {o.x, o.y, o.z, o.w};
                __builtin_nontemporal_store(ov, reinterpret_cast<f32x4_t*>(p.out + obase[g] + ((int64_t)pp * p.W + qq) * p.Cout));
            }
        }
    }
}

template <int WGM, int WGN>
static int launch_fused(WinoFusedParams p, hipStream_t s) {
    constexpr int BM = 32 * WGM, BN = 32 * WGN;
    constexpr int LDS_BYTES = 3 * (BM + BN) * 64 * 4;
    static bool attr_set[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 16 && !attr_set[dev]) {
        MM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_fused_kernel<WGM, WGN>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set[dev] = true;
    }
    p.tiles_n = (p.Cout + BN - 1) / BN;
    for (int q = 0; q < 6; ++q)
        for (int qq = 0; qq < 4; ++qq) p.at_cols[q * 4 + qq] = kAtHost[qq][q];
    const int64_t blocks = (int64_t)((p.ntile + BM - 1) / BM) * p.tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffff) return MM_ERR_INVALID_ARG;
    if (prof_enabled()) {
        char tag[64];
        snprintf(tag, sizeof(tag), "wino-fused M=%d K=%d N=%d t%dx%d b36", p.ntile, p.K, p.Cout, BM, BN);
        prof_before(0, 2.0 * 36.0 * (double)p.ntile * (double)p.K * (double)p.Cout, s, tag);
    }
    hipLaunchKernelGGL((wino_fused_kernel<WGM, WGN>), dim3((unsigned)blocks), dim3(WGM * WGN * 64), LDS_BYTES, s, p);
    prof_after(0, s);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

// V [36][ntile][Cin] (from wino_input_transform, m = 4), U [36][Cout][Cin] -> y NHWC [B,H,W,Cout] (+bias, ReLU)
int wino_gemm_output_fused(const float* V, const float* U, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                           int relu, int shape, hipStream_t s) {
    if (Cin % 64 || Cout % 32) return MM_ERR_UNSUPPORTED;
    WinoFusedParams p;
    p.V = V; p.U = U; p.bias = bias; p.out = y;
    p.TH = (H + 3) / 4; p.TW = (W + 3) / 4;
    const int64_t ntile = (int64_t)B * p.TH * p.TW;
    if (ntile <= 0) return MM_OK;
    if (ntile * Cin * 4 >= 0xFFFFF000ll || (int64_t)Cout * Cin * 4 >= 0xFFFFF000ll) return MM_ERR_INVALID_ARG;  // 32-bit offsets in a plane
    p.ntile = (int)ntile; p.K = Cin; p.Cout = Cout; p.B = B; p.H = H; p.W = W; p.relu = relu; p.tiles_n = 0;
    (void)shape;
    return launch_fused<2, 2>(p, s);   // 64 tiles x 64 channels per workgroup
}

}  // namespace mm
