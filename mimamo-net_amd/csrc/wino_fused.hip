// Winograd F(4x4,3x3): the 36 position GEMMs AND the output transform in one kernel -- M never goes to HBM.
//
// The three-kernel form (winograd.hip + the batched launch of conv_mfma.hip) writes M = [36][tiles][Cout] and reads it back in
// the output transform: 2 x 2.25 x the layer's output bytes, which is what bounds the conv2_x / conv3_x layers (their position
// GEMMs are HBM-bound in fp32: 32..64 FLOP per byte of V).  Here a workgroup owns 64 Winograd tiles x 64 output channels and
// walks ALL 36 positions xi = (r, q) for them:
//     M_xi = V_xi[tiles, Cin] * U_xi[Cout, Cin]^T          fp32 MFMA (v_mfma_f32_16x16x4_f32)
//     t[p]    += At[p][r] * M_xi        (r = 0..5, for the current column q)        VALU
//     Y[p][q'] += At[q'][q] * t[p]      (once per column q)                          16 output pixels per tile
// and stores Y + bias (+ReLU) as the layer's NHWC output.  A wave owns 16 tiles x 32 channels: Y = 128 registers, t = 32, so
// two waves share a SIMD (default: two four-wave workgroups of 32 tiles x 64 channels per CU) and one wave's loads, transform
// VALU and barrier waits run under the other's MFMAs.  Operands stream global -> LDS directly (buffer_load ... lds) in 64-deep K slabs
// of 32 KB through a 3-deep ring, one s_barrier per slab.
// Tried first and dropped (round 2, measured): 32 tiles x 32 channels per wave on v_mfma_f32_32x32x2_f32 -- 256 Y registers,
// hence ONE wave per SIMD, and an in-order wave alone on its SIMD exposes every instruction that is not an MFMA: matrix pipes
// 42 % busy at Cin = 64 and 69 % at Cin = 512 (rocprofv3 PMC), each direct-to-LDS piece stalling the MFMA stream ~100 cycles,
// the transform VALU (3 instructions per Y FMA through the accumulation registers) filling the issue slots; hand-scheduled
// assembly for the whole loop (reads two k-steps ahead, loads and VALU placed in the MFMA shadows, two partial accumulators)
// did not change the rate, a wave-skewed start made it worse.  Same arithmetic as wino_out6_kernel (A^T M A evaluated
// column-first); results agree with the three-kernel form to fp32 rounding.
// The reference computes these layers with torch's direct fp32 conv (third-party ResNet50, api/resnet50_extractor.py:74-83);
// parity is checked against the oracle's direct convolution.
//
// mm-hipcc-flags: -mllvm -amdgpu-promote-alloca-to-vector-limit=4096 -mllvm -pragma-unroll-threshold=131072
// (build.py passes this to hipcc for this file.  The eight-wave INC 3 instantiation has more per-thread arrays than the backend's
//  default promote-to-vector budget for a 512-thread workgroup: without it half of the 128-register Y array stays in scratch --
//  272 bytes per lane, element-wise scratch stores in the transform -- although only 190 VGPRs are in use.  The other
//  instantiations compile to the same registers / scratch with and without the option.)
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "conv.h"

namespace mm {

typedef float f32x4v __attribute__((ext_vector_type(4)));
#ifndef MM_WF_NT_STORES
#define MM_WF_NT_STORES 1   // 0: the INC epilogues' 64-byte-per-pixel stores WITHOUT the non-temporal hint (A/B knob, round 6)
#endif
// the INC epilogues' output stores: four consecutive channels of one pixel per lane, 64 contiguous bytes per pixel and instruction
__device__ __forceinline__ void inc_store(const f32x4v& v, float* dst) {
    if (MM_WF_NT_STORES) __builtin_nontemporal_store(v, reinterpret_cast<f32x4v*>(dst));
    else *reinterpret_cast<f32x4v*>(dst) = v;
}

struct WinoFusedParams {
    const float* V;      // [36][ntile][K]
    const float* U;      // [36][Cout][K]
    const float* bias;
    float* out;          // NHWC [B][H][W][Cout]
    int ntile, K, Cout, B, H, W, TH, TW, relu, tiles_n;
    float at_cols[24];   // At[q'][q] as [q][q']
    // INC instantiation only: the residual block's 1x1 increase conv applied to this layer's output before it leaves the CU
    const float* w2;     // [C2][Cout] BN-folded increase weights (Cout == 64: the workgroup owns every channel of its pixels)
    const float* bias2;  // [C2]
    const float* res;    // INC 1: residual, NHWC [B][H][W][C2].  INC 2: the block input x, NHWC [B][H][W][Cout] -- the second K source
    int C2;              // 256
    // NEXT instantiation only (round 6): the NEXT residual block's 1x1 reduce conv (C2 -> 64, BN folded, ReLU) applied to the block output
    // this kernel has just computed -- pointwise, no halo: out3 [B][H][W][64] = relu(W3 out + bias3) beside `out`
    const float* w3;     // [64][C2]
    const float* bias3;  // [64]
    float* out3;         // NHWC [B][H][W][64]
    int generic_loop;    // 1: the runtime-scheduled main loop whatever K is (the parity twin of the compile-time-scheduled one)
    int ablate;          // -DMM_MEASURE builds only (results wrong by construction): bit 0 = every workgroup reads the V rows of the first
                         // 1 024 tiles (L2-resident: the kernel without its V traffic), bit 1 = residual rows from the first 4 096 pixels,
                         // 4 = no output-transform updates in the main loop, 8 = INC 1 epilogue without residual loads, 16 = without output
                         // stores, 32 = without its exchange barriers, 64 = without its MFMAs, 128 = main loop without the operand DMA
};

#ifndef MM_INC3_PAIR
#define MM_INC3_PAIR 0      // 1: the conv3_x fused kernel exchanges two output positions per barrier (A/B variant of round 5)
#endif
// dynamic LDS of the eight-wave INC 3 instantiation: the operand ring, or the epilogue's tenants when those are larger
__host__ __device__ constexpr int inc3_lds_bytes(int ring_bytes) {
    return MM_INC3_PAIR && ring_bytes < (128 * 128 + 4 * 8 * 512 + 512) * 4 ? (128 * 128 + 4 * 8 * 512 + 512) * 4 : ring_bytes;
}

__device__ constexpr float kAt[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
static constexpr float kAtHost[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};

// INC (round 4, conv2_x blocks 2 and 3: Cin = Cout = 64, increase 64 -> 256): the position GEMMs run TRANSPOSED (M^T = U V^T, so a
// lane holds 4 consecutive channels of ONE tile) and the epilogue contracts relu(Y + bias) with the 64 KB increase matrix straight
// from the accumulator registers -- in that layout they ARE the B operand of the second MFMA -- adds the residual, applies the ReLU
// and writes the block's 256-channel output: the 64-channel tensor between the 3x3 and the increase conv (1.6 MB per frame written
// and re-read) never exists, and the 49 %-busy K = 64 GEMM launch disappears.
// WGN = 4 (INC 3, conv3_x blocks 2-4: Cin = Cout = 128, increase 128 -> 512): eight waves as 2 x 4, 32 tiles x 128 channels -- the
// workgroup again owns every channel of its pixels -- one workgroup per CU (120 KB ring); the 256 KB increase matrix passes through
// LDS in four row quarters.
// KSL (round 5): K / 64 as a compile-time constant (1, 2, 4; 0 = any K, the round-2 loop).  tools/probes/valu_mfma_overlap.hip showed that VALU
// instructions do not hide under MFMAs on gfx950 -- their issue time ADDS to the matrix time, in block form, between the MFMAs of a stream,
// and across the waves of a SIMD alike -- so the per-slab address arithmetic of the loop is paid in matrix time: 12 v_lshl_add_u32 for the
// fragment addresses (ring slot base + swizzled row offset) and 5-6 v_add_u32 for the DMA offsets (+ k offset) per 32 MFMAs.  With the slab
// sequence of a column unrolled (6 positions x KSL slabs, a multiple of the 3 ring slots) the ring slot is a compile-time constant that
// folds into the ds_read_b128 offset field and into the M0 immediates of the DMA, and the k offset rides in the buffer instruction's scalar
// offset: no VALU instruction is left in a slab besides the MFMAs and the output transform's updates.  Same operations on the same values in
// the same order: bit-identical results (the KSL = 0 loop stays as the twin for any other K).
template <int NBUF, int WGM, int INC = 0, int WGN = 2, int KSL = 0, int NEXT = 0>
__global__ void __launch_bounds__(WGM * WGN * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
wino_fused_kernel(const WinoFusedParams p) {
    // WGM = 4: eight waves, 64 tiles x 64 channels, one workgroup per CU; WGM = 2: four waves, 32 tiles x 64 channels, two
    // independent workgroups per CU (no common barrier between the two waves of a SIMD)
    constexpr int NW = WGN * WGM, BM = 16 * WGM, BN = 32 * WGN, KS = 64;
    constexpr int ROWS = BM + BN;
    constexpr int NIA = BM / (4 * NW), NIB = BN / (4 * NW);   // 2 + 2 pieces of 1 KB (4 rows) per wave per slab
    constexpr int NL = NIA + NIB;
    constexpr int SLAB_BYTES = ROWS * KS * 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;            // 16 tiles x 32 channels per wave
    const int l16 = lane & 15, lg = lane >> 4;             // MFMA operand row / column, k index inside a k-step of 4
    // the n-tiles of an m-tile are neighbours in the logical order and stay on one XCD: its L2 serves the V rows they share
    const int logical = xcd_contiguous(blockIdx.x, gridDim.x);
    const int tile_m = logical / p.tiles_n, tile_n = logical - tile_m * p.tiles_n;
    const int m_base = tile_m * BM, n_base = tile_n * BN;
#ifdef MM_MEASURE
    const int m_load = (p.ablate & 1) ? (m_base & 1023) : m_base;     // cost-sheet ablation: V rows from an L2-resident subset
#else
    const int m_load = m_base;
#endif
    static_assert(BM % (4 * NW) == 0 && BN % (4 * NW) == 0, "whole 1 KB pieces per wave");
    const int K = p.K, kslabs = K / KS;

    // ---- DMA (as wino_fused.hip): slot s of slab row r holds k-quad s ^ (r & 15); rows past the plane read as zeros
    unsigned va[NIA], vb[NIB];
#pragma unroll
    for (int it = 0; it < NIA; ++it) {
        const int row = (it * NW + wave) * 4 + (lane >> 4);
        va[it] = (unsigned)((m_load + row) * K + (((lane & 15) ^ (row & 15)) << 2)) * 4u;
    }
#pragma unroll
    for (int it = 0; it < NIB; ++it) {
        const int row = (it * NW + wave) * 4 + (lane >> 4);
        vb[it] = (unsigned)((n_base + row) * K + (((lane & 15) ^ (row & 15)) << 2)) * 4u;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    const unsigned plane_a = (unsigned)p.ntile * (unsigned)K * 4u;
    const unsigned plane_b = (unsigned)p.Cout * (unsigned)K * 4u;
    auto dma1 = [&](const __amdgpu_buffer_rsrc_t& rsrc, unsigned voff, unsigned lds_byte) {
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_byte);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(m0v) : "memory");
    };
    // The DMA stream runs NBUF-1 slabs ahead of the MFMAs, through the slab sequence (column q outer, row r, k slab inner; position
    // plane xi = r * 6 + q).  Its state advances incrementally -- plane pointers by +6 planes per position and -29 at a column
    // change, the k offset by 256 bytes -- because with two waves per SIMD the scalar bookkeeping of a slab is paid in issue slots
    // (rocprofv3: 2.6 SALU instructions per MFMA with per-slab 64-bit multiplies; the MFMAs of a slab take 1 024 cycles).
    const char* pv = reinterpret_cast<const char*>(p.V);
    const char* pu = reinterpret_cast<const char*>(p.U);
    const int64_t step_a = 6 * (int64_t)plane_a, step_b = 6 * (int64_t)plane_b;
    const int64_t back_a = 35 * (int64_t)plane_a, back_b = 35 * (int64_t)plane_b;
    unsigned rec_a = plane_a, rec_b = plane_b;     // num_records: 0 past the last slab (loads return zeros into a dead slot)
    unsigned koff = 0;
    int d_q = 0, d_r = 0, d_ks = 0, d_buf = 0;
    auto dma_next = [&]() {
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pv), 0, rec_a, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pu), 0, rec_b, 0x00020000);
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)d_buf * SLAB_BYTES + (unsigned)wave * 1024u);
        // all pieces of the slab in ONE statement: M0 (compiler-reserved) is saved and restored once, and each piece is
        // s_add_u32 m0 / s_nop / buffer_load (three instructions instead of six)
        unsigned keep;
        if constexpr (NIA == 2 && NIB == 4) {
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_add_u32 m0, %9, %10\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %7, 0 offen lds\n\t"
                         "s_add_u32 m0, %9, %11\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %7, 0 offen lds\n\t"
                         "s_add_u32 m0, %9, %12\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %8, 0 offen lds\n\t"
                         "s_add_u32 m0, %9, %13\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %8, 0 offen lds\n\t"
                         "s_add_u32 m0, %9, %14\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %8, 0 offen lds\n\t"
                         "s_add_u32 m0, %9, %15\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %8, 0 offen lds\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(va[0] + koff), "v"(va[1] + koff), "v"(vb[0] + koff), "v"(vb[1] + koff), "v"(vb[2] + koff), "v"(vb[3] + koff),
                           "s"(ra), "s"(rb), "s"(base), "n"(0), "n"(NW * 1024), "n"(BM * KS * 4), "n"(BM * KS * 4 + NW * 1024),
                           "n"(BM * KS * 4 + 2 * NW * 1024), "n"(BM * KS * 4 + 3 * NW * 1024)
                         : "memory", "scc");
        } else if constexpr (NIA == 1 && NIB == 4) {
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_add_u32 m0, %8, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %6, 0 offen lds\n\t"
                         "s_add_u32 m0, %8, %10\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %7, 0 offen lds\n\t"
                         "s_add_u32 m0, %8, %11\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %7, 0 offen lds\n\t"
                         "s_add_u32 m0, %8, %12\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %7, 0 offen lds\n\t"
                         "s_add_u32 m0, %8, %13\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %7, 0 offen lds\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(va[0] + koff), "v"(vb[0] + koff), "v"(vb[1] + koff), "v"(vb[2] + koff), "v"(vb[3] + koff),
                           "s"(ra), "s"(rb), "s"(base), "n"(0), "n"(BM * KS * 4), "n"(BM * KS * 4 + NW * 1024),
                           "n"(BM * KS * 4 + 2 * NW * 1024), "n"(BM * KS * 4 + 3 * NW * 1024)
                         : "memory", "scc");
        } else {
            static_assert(NIA == 2 && NIB == 2, "piece counts of the workgroup shapes");
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_add_u32 m0, %7, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %5, 0 offen lds\n\t"
                         "s_add_u32 m0, %7, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %5, 0 offen lds\n\t"
                         "s_add_u32 m0, %7, %10\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %6, 0 offen lds\n\t"
                         "s_add_u32 m0, %7, %11\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, 0 offen lds\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(va[0] + koff), "v"(va[1] + koff), "v"(vb[0] + koff), "v"(vb[1] + koff), "s"(ra), "s"(rb), "s"(base), "n"(0),
                           "n"(NW * 1024), "n"(BM * KS * 4), "n"(BM * KS * 4 + NW * 1024)
                         : "memory", "scc");
        }
        koff += KS * 4u;
        if (++d_ks == kslabs) {            // next position
            d_ks = 0;
            koff = 0;
            pv += step_a;
            pu += step_b;
            if (++d_r == 6) {              // next column
                d_r = 0;
                pv -= back_a;
                pu -= back_b;
                if (++d_q == 6) rec_a = rec_b = 0;
            }
        }
        d_buf = d_buf == NBUF - 1 ? 0 : d_buf + 1;
    };

    // ---- fragments: lane (row l16, k group lg) reads k-quads lg, lg + 4, lg + 8, lg + 12 of its row; element c of quad j feeds
    //      the MFMA of k-step 4 j + c (every lane group then supplies a different k of that step, the same on the A and B side)
    int fa[4], fb0[4], fb1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int sw = ((lg + 4 * j) ^ l16) << 2;          // (row & 15) == l16 for all three rows
        fa[j] = (wm * 16 + l16) * KS + sw;
        fb0[j] = (BM + wn * 32 + l16) * KS + sw;
        fb1[j] = (BM + wn * 32 + 16 + l16) * KS + sw;
    }

    f32x4v Y[4][4][2], T[4][2], M[2][2];   // [..][channel block]; M[position parity][channel block]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            T[a][c] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 4; ++b) Y[a][b][c] = f32x4v{0.f, 0.f, 0.f, 0.f};
        }
    M[0][0] = M[0][1] = M[1][0] = M[1][1] = f32x4v{0.f, 0.f, 0.f, 0.f};

    auto update_t = [&](const f32x4v (&Mx)[2], auto rtag) {
        constexpr int r = decltype(rtag)::value;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const float c = kAt[pp][r];   // folds after unrolling
            if (c == 0.f) continue;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int e = 0; e < 4; ++e) T[pp][cb][e] = c == 1.f ? T[pp][cb][e] + Mx[cb][e] : fmaf(c, Mx[cb][e], T[pp][cb][e]);
        }
    };
    auto update_y = [&](int qc) {
        // (A^T's columns 0 and 5 are unit vectors -- three of their four FMAs per element multiply by zero -- but a workgroup-uniform branch
        //  around them, like skipping the all-zero update of the first column, breaks the register allocation of the unrolled column body:
        //  86-146 spilled registers in the KSL instantiations; built and dropped in round 5)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const float c = p.at_cols[qc * 4 + qq];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) Y[pp][qq][cb][e] = fmaf(c, T[pp][cb][e], Y[pp][qq][cb][e]);
        }
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) T[pp][cb] = f32x4v{0.f, 0.f, 0.f, 0.f};
    };

    int buf = 0;
    auto slab = [&](f32x4v (&Mc)[2], auto first_tag) {
        constexpr bool first = decltype(first_tag)::value;
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"((NBUF - 2) * NL) : "memory");
        dma_next();
        const float* sl = lds + buf * (ROWS * KS);
        float4 a[4], b0[4], b1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = *reinterpret_cast<const float4*>(sl + fa[j]);
            b0[j] = *reinterpret_cast<const float4*>(sl + fb0[j]);
            b1[j] = *reinterpret_cast<const float4*>(sl + fb1[j]);
        }
        if (first) Mc[0] = Mc[1] = f32x4v{0.f, 0.f, 0.f, 0.f};
        if constexpr (INC) {
            // transposed product: rows = channels (U), columns = tiles (V) -> C row 4 lg + e = channel, column l16 = tile
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[j].x, a[j].x, Mc[0], 0, 0, 0);
                Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[j].x, a[j].x, Mc[1], 0, 0, 0);
                Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[j].y, a[j].y, Mc[0], 0, 0, 0);
                Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[j].y, a[j].y, Mc[1], 0, 0, 0);
                Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[j].z, a[j].z, Mc[0], 0, 0, 0);
                Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[j].z, a[j].z, Mc[1], 0, 0, 0);
                Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[j].w, a[j].w, Mc[0], 0, 0, 0);
                Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[j].w, a[j].w, Mc[1], 0, 0, 0);
            }
        } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b0[j].x, Mc[0], 0, 0, 0);
            Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b1[j].x, Mc[1], 0, 0, 0);
            Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b0[j].y, Mc[0], 0, 0, 0);
            Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b1[j].y, Mc[1], 0, 0, 0);
            Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b0[j].z, Mc[0], 0, 0, 0);
            Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b1[j].z, Mc[1], 0, 0, 0);
            Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b0[j].w, Mc[0], 0, 0, 0);
            Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b1[j].w, Mc[1], 0, 0, 0);
        }
        }
        buf = buf == NBUF - 1 ? 0 : buf + 1;
    };

    if constexpr (KSL > 0) {
        static_assert(NBUF == 3 && (6 * KSL) % NBUF == 0, "a column's slabs cover the ring a whole number of times");
        constexpr int NSL = 6 * KSL;                         // slabs per column q
        typedef __attribute__((address_space(3))) const f32x4v* lds_f4;      // (ext-vector type: HIP's float4 struct has no address-space-3 assignment on the host pass)
        // LDS byte addresses of the fragment rows in ring slot 0; slot s adds s * SLAB_BYTES as an instruction offset
        // (eight-wave shapes: 40 KB slabs put ring slot 2 at byte 81 920, beyond the 16-bit offset field -- hipcc would add the slot base per read
        //  again and, no longer seeing the 16-byte alignment, split every ds_read_b128 into two ds_read2_b32.  Slot 2 gets its own base registers.)
        constexpr bool FAR2 = (NBUF - 1) * SLAB_BYTES > 65535;
        unsigned ab[4], bb0[4], bb1[4], ab2[4], bb0_2[4], bb1_2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ab[j] = lds0 + (unsigned)fa[j] * 4u;
            bb0[j] = lds0 + (unsigned)fb0[j] * 4u;
            bb1[j] = lds0 + (unsigned)fb1[j] * 4u;
            if constexpr (FAR2) {
                ab2[j] = ab[j] + (NBUF - 1) * SLAB_BYTES;
                bb0_2[j] = bb0[j] + (NBUF - 1) * SLAB_BYTES;
                bb1_2[j] = bb1[j] + (NBUF - 1) * SLAB_BYTES;
                asm volatile("" : "+v"(ab2[j]), "+v"(bb0_2[j]), "+v"(bb1_2[j]));      // opaque: kept in registers, not re-derived per read
            }
        }
        // fragment row j of ring slot BUF: base register + slot offset in the instruction (slot 2 of the 40 KB slabs: its own base)
        auto frag = [&](const unsigned (&lo)[4], const unsigned (&hi)[4], int j, auto buf_tag) -> f32x4v {
            constexpr int BUF = decltype(buf_tag)::value;
            if constexpr (FAR2 && BUF == NBUF - 1) return *(lds_f4)(uintptr_t)(hi[j]);
            else return *(lds_f4)(uintptr_t)(lo[j] + BUF * SLAB_BYTES);
        };
        const unsigned dbase = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);   // this wave's first piece in slot 0
        int dq = 0;                                           // columns the DMA stream has finished
        // DMA of the slab that is D slabs into the current column (D may run past the column's end: the stream is NBUF - 1 ahead)
        auto dma_c = [&](auto d_tag) {
            constexpr int D = decltype(d_tag)::value % NSL;
            constexpr int DBUF = D % NBUF, DKS = D % KSL, DR = D / KSL;
            constexpr unsigned SB = DBUF * SLAB_BYTES;
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pv), 0, rec_a, 0x00020000);
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pu), 0, rec_b, 0x00020000);
            const unsigned ko = __builtin_amdgcn_readfirstlane(DKS * KS * 4);       // scalar offset of the buffer instruction
            unsigned keep;
#ifdef MM_MEASURE
            if (p.ablate & 128) {   // measurement only: the pieces are requested with zero records (no memory traffic, same instruction stream)
                rec_a = rec_b = 0;
            }
#endif
            if constexpr (NIA == 2 && NIB == 4) {
                asm volatile("s_mov_b32 %0, m0\n\t"
                             "s_add_u32 m0, %9, %10\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %7, %16 offen lds\n\t"
                             "s_add_u32 m0, %9, %11\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %7, %16 offen lds\n\t"
                             "s_add_u32 m0, %9, %12\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %8, %16 offen lds\n\t"
                             "s_add_u32 m0, %9, %13\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %8, %16 offen lds\n\t"
                             "s_add_u32 m0, %9, %14\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %8, %16 offen lds\n\t"
                             "s_add_u32 m0, %9, %15\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %8, %16 offen lds\n\t"
                             "s_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(va[0]), "v"(va[1]), "v"(vb[0]), "v"(vb[1]), "v"(vb[2]), "v"(vb[3]),
                               "s"(ra), "s"(rb), "s"(dbase), "n"(SB), "n"(SB + NW * 1024), "n"(SB + BM * KS * 4), "n"(SB + BM * KS * 4 + NW * 1024),
                               "n"(SB + BM * KS * 4 + 2 * NW * 1024), "n"(SB + BM * KS * 4 + 3 * NW * 1024), "s"(ko)
                             : "memory", "scc");
            } else if constexpr (NIA == 1 && NIB == 4) {
                asm volatile("s_mov_b32 %0, m0\n\t"
                             "s_add_u32 m0, %8, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %6, %14 offen lds\n\t"
                             "s_add_u32 m0, %8, %10\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %7, %14 offen lds\n\t"
                             "s_add_u32 m0, %8, %11\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %7, %14 offen lds\n\t"
                             "s_add_u32 m0, %8, %12\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %7, %14 offen lds\n\t"
                             "s_add_u32 m0, %8, %13\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %7, %14 offen lds\n\t"
                             "s_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(va[0]), "v"(vb[0]), "v"(vb[1]), "v"(vb[2]), "v"(vb[3]),
                               "s"(ra), "s"(rb), "s"(dbase), "n"(SB), "n"(SB + BM * KS * 4), "n"(SB + BM * KS * 4 + NW * 1024),
                               "n"(SB + BM * KS * 4 + 2 * NW * 1024), "n"(SB + BM * KS * 4 + 3 * NW * 1024), "s"(ko)
                             : "memory", "scc");
            } else {
                static_assert(NIA == 2 && NIB == 2, "piece counts of the workgroup shapes");
                asm volatile("s_mov_b32 %0, m0\n\t"
                             "s_add_u32 m0, %7, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %5, %12 offen lds\n\t"
                             "s_add_u32 m0, %7, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %5, %12 offen lds\n\t"
                             "s_add_u32 m0, %7, %10\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %6, %12 offen lds\n\t"
                             "s_add_u32 m0, %7, %11\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, %12 offen lds\n\t"
                             "s_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(va[0]), "v"(va[1]), "v"(vb[0]), "v"(vb[1]), "s"(ra), "s"(rb), "s"(dbase), "n"(SB),
                               "n"(SB + NW * 1024), "n"(SB + BM * KS * 4), "n"(SB + BM * KS * 4 + NW * 1024), "s"(ko)
                             : "memory", "scc");
            }
            if constexpr (DKS == KSL - 1) {                    // that was the position's last slab: next position
                pv += step_a;
                pu += step_b;
                if constexpr (DR == 5) {                       // ... and the column's last position: next column
                    pv -= back_a;
                    pu -= back_b;
                    if (++dq == 6) rec_a = rec_b = 0;          // past the last slab: loads return zeros into a dead slot
                }
            }
        };
#ifndef MM_WF_HOOK
#define MM_WF_HOOK 1      // 1: the pending output-transform updates run as ONE block between a slab's fragment reads and its MFMA burst (below)
#endif
        // `hook` (first slab of a position): the output-transform updates of the PREVIOUS position, placed between this slab's fragment
        // reads and its MFMAs and pinned there by scheduling barriers -- one VALU block under the LDS latency, then 32 MFMAs back to back.
        // Left to itself hipcc spreads those ~25-70 VALU instructions over the gaps of the MFMA burst; on gfx950 every MFMA -> VALU -> MFMA
        // transition costs ~2.5 ns on top of the instructions themselves (tools/probes/valu_mfma_overlap.hip, interleaved vs block form).
        auto slab_c = [&](auto n_tag, auto&& hook) {
            constexpr int N = decltype(n_tag)::value;          // slab index inside the column
            constexpr int BUF = N % NBUF;
            constexpr bool first = N % KSL == 0;
            f32x4v (&Mc)[2] = M[(N / KSL) & 1];
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"((NBUF - 2) * NL) : "memory");
            dma_c(std::integral_constant<int, N + NBUF - 1>());
            if (first) Mc[0] = Mc[1] = f32x4v{0.f, 0.f, 0.f, 0.f};
            if constexpr (INC) {
                // the INC instantiations carry the epilogue's state through the loop: fragments in two halves (six reads, sixteen MFMAs
                // each, pinned by a scheduling barrier) instead of all twelve up front -- with every address arithmetic gone hipcc would
                // otherwise hoist all reads and spill 30 registers INSIDE the loop (scratch traffic would also break the counted vmcnt waits)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    f32x4v a[2], b0[2], b1[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        a[j] = frag(ab, ab2, 2 * hh + j, std::integral_constant<int, BUF>());
                        b0[j] = frag(bb0, bb0_2, 2 * hh + j, std::integral_constant<int, BUF>());
                        b1[j] = frag(bb1, bb1_2, 2 * hh + j, std::integral_constant<int, BUF>());
                    }
                    // (no hook here: the INC kernels measured 0.5-1 % slower with the transform block pinned, r05_ab_wino_loop.txt)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[j].x, a[j].x, Mc[0], 0, 0, 0);
                        Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[j].x, a[j].x, Mc[1], 0, 0, 0);
                        Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[j].y, a[j].y, Mc[0], 0, 0, 0);
                        Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[j].y, a[j].y, Mc[1], 0, 0, 0);
                        Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[j].z, a[j].z, Mc[0], 0, 0, 0);
                        Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[j].z, a[j].z, Mc[1], 0, 0, 0);
                        Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[j].w, a[j].w, Mc[0], 0, 0, 0);
                        Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[j].w, a[j].w, Mc[1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
            f32x4v a[4], b0[4], b1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = frag(ab, ab2, j, std::integral_constant<int, BUF>());
                b0[j] = frag(bb0, bb0_2, j, std::integral_constant<int, BUF>());
                b1[j] = frag(bb1, bb1_2, j, std::integral_constant<int, BUF>());
            }
            if (MM_WF_HOOK) {
                __builtin_amdgcn_sched_barrier(0);
                hook();
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (false) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[j].x, a[j].x, Mc[0], 0, 0, 0);
                    Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[j].x, a[j].x, Mc[1], 0, 0, 0);
                    Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[j].y, a[j].y, Mc[0], 0, 0, 0);
                    Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[j].y, a[j].y, Mc[1], 0, 0, 0);
                    Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[j].z, a[j].z, Mc[0], 0, 0, 0);
                    Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[j].z, a[j].z, Mc[1], 0, 0, 0);
                    Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[j].w, a[j].w, Mc[0], 0, 0, 0);
                    Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[j].w, a[j].w, Mc[1], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b0[j].x, Mc[0], 0, 0, 0);
                    Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b1[j].x, Mc[1], 0, 0, 0);
                    Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b0[j].y, Mc[0], 0, 0, 0);
                    Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b1[j].y, Mc[1], 0, 0, 0);
                    Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b0[j].z, Mc[0], 0, 0, 0);
                    Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b1[j].z, Mc[1], 0, 0, 0);
                    Mc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b0[j].w, Mc[0], 0, 0, 0);
                    Mc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b1[j].w, Mc[1], 0, 0, 0);
                }
            }
            }
        };
        // the slabs of position r of a column (compile-time slab indices r * KSL .. r * KSL + KSL - 1)
        auto nothing = []() {};
        auto position = [&](auto r_tag, auto&& hook) {
            constexpr int r = decltype(r_tag)::value;
            slab_c(std::integral_constant<int, r * KSL>(), hook);
            if constexpr (KSL > 1) slab_c(std::integral_constant<int, r * KSL + 1>(), nothing);
            if constexpr (KSL > 2) {
                slab_c(std::integral_constant<int, r * KSL + 2>(), nothing);
                slab_c(std::integral_constant<int, r * KSL + 3>(), nothing);
            }
            static_assert(KSL == 1 || KSL == 2 || KSL == 4, "K = 64, 128 or 256");
        };
        dma_c(std::integral_constant<int, 0>());
        dma_c(std::integral_constant<int, 1>());
        for (int q = 0; q < 6; ++q) {
            if constexpr (MM_WF_HOOK && INC == 0) {
            // position r's first slab carries the transform of position r - 1 (position 0: the previous column's last row, then that column
            // into Y; q = 0: zeros)
            position(std::integral_constant<int, 0>(), [&]() { update_t(M[1], std::integral_constant<int, 5>()); update_y(q == 0 ? 0 : q - 1); });
            position(std::integral_constant<int, 1>(), [&]() { update_t(M[0], std::integral_constant<int, 0>()); });
            position(std::integral_constant<int, 2>(), [&]() { update_t(M[1], std::integral_constant<int, 1>()); });
            position(std::integral_constant<int, 3>(), [&]() { update_t(M[0], std::integral_constant<int, 2>()); });
            position(std::integral_constant<int, 4>(), [&]() { update_t(M[1], std::integral_constant<int, 3>()); });
            position(std::integral_constant<int, 5>(), [&]() { update_t(M[0], std::integral_constant<int, 4>()); });
            } else {
#ifdef MM_MEASURE
            if (p.ablate & 4) {      // measurement only: the column without its transform updates
                position(std::integral_constant<int, 0>(), nothing);
                position(std::integral_constant<int, 1>(), nothing);
                position(std::integral_constant<int, 2>(), nothing);
                position(std::integral_constant<int, 3>(), nothing);
                position(std::integral_constant<int, 4>(), nothing);
                position(std::integral_constant<int, 5>(), nothing);
                Y[0][0][0] += M[0][0] + M[1][0]; Y[0][0][1] += M[0][1] + M[1][1];      // (keeps the MFMAs alive)
                continue;
            }
#endif
            position(std::integral_constant<int, 0>(), nothing);
            update_t(M[1], std::integral_constant<int, 5>());   // last row of the previous column (q = 0: zeros)
            update_y(q == 0 ? 0 : q - 1);
            position(std::integral_constant<int, 1>(), nothing);
            update_t(M[0], std::integral_constant<int, 0>());
            position(std::integral_constant<int, 2>(), nothing);
            update_t(M[1], std::integral_constant<int, 1>());
            position(std::integral_constant<int, 3>(), nothing);
            update_t(M[0], std::integral_constant<int, 2>());
            position(std::integral_constant<int, 4>(), nothing);
            update_t(M[1], std::integral_constant<int, 3>());
            position(std::integral_constant<int, 5>(), nothing);
            update_t(M[0], std::integral_constant<int, 4>());
            }
        }
    } else {
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) dma_next();
    for (int q = 0; q < 6; ++q) {
        // position (r, q) accumulates into M[r & 1]; the transform of the previous position reads the other pair, so hipcc /
        // the hardware may run it under this position's MFMAs
        slab(M[0], std::true_type());
        for (int ks = 1; ks < kslabs; ++ks) slab(M[0], std::false_type());
        update_t(M[1], std::integral_constant<int, 5>());   // last row of the previous column (q = 0: zeros)
        update_y(q == 0 ? 0 : q - 1);
        slab(M[1], std::true_type());
        for (int ks = 1; ks < kslabs; ++ks) slab(M[1], std::false_type());
        update_t(M[0], std::integral_constant<int, 0>());
        slab(M[0], std::true_type());
        for (int ks = 1; ks < kslabs; ++ks) slab(M[0], std::false_type());
        update_t(M[1], std::integral_constant<int, 1>());
        slab(M[1], std::true_type());
        for (int ks = 1; ks < kslabs; ++ks) slab(M[1], std::false_type());
        update_t(M[0], std::integral_constant<int, 2>());
        slab(M[0], std::true_type());
        for (int ks = 1; ks < kslabs; ++ks) slab(M[0], std::false_type());
        update_t(M[1], std::integral_constant<int, 3>());
        slab(M[1], std::true_type());
        for (int ks = 1; ks < kslabs; ++ks) slab(M[1], std::false_type());
        update_t(M[0], std::integral_constant<int, 4>());
    }
    }
    update_t(M[1], std::integral_constant<int, 5>());
    update_y(5);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tail loads drained, everyone done with the ring

    if constexpr (INC == 3) {
        // ---- conv3_x blocks 2-4: out = relu( W2 relu(Y + bias) + bias2 + res ), W2 [512][128].  A wave holds 32 of the 128 channels of
        // its 16 tiles; the three waves with the other channels of the same tiles (same wm) hand over lane-for-lane register copies
        // through a double-buffered exchange area (one barrier per position).  The matrix goes through LDS in four passes of 128
        // rows (64 KB); wave (wm, wn) produces channels [128 pass + 32 wn, + 32) of its 16 tiles: 64 MFMAs per position and pass.
        static_assert(WGM == 2 && WGN == 4 && NBUF == 3, "INC 3 is built for the 2 x 4 wave workgroup");
        constexpr int K2 = 128, ROWS2 = 128;
        constexpr int W2_FLOATS = ROWS2 * K2;                          // 64 KB
        constexpr int XB_FLOATS = NW * 512;                            // one position's exchange area: 8 waves x 2 KB
        constexpr int XPOS = MM_INC3_PAIR ? 2 : 1;                     // positions exchanged per barrier
        static_assert((W2_FLOATS + 2 * XPOS * XB_FLOATS + 512) * 4 <= inc3_lds_bytes(NBUF * ROWS * KS * 4), "epilogue tenants fit");
        float* xb = lds + W2_FLOATS;
        float* b2s = xb + 2 * XPOS * XB_FLOATS;
        // (The exchange only couples the four waves of one tile half (same wm).  A per-half barrier on an LDS counter -- ds_add to
        //  arrive, poll to wait -- instead of the workgroup-wide s_barrier below was built and measured: 103.64-103.84 vs
        //  103.30-103.66 ms per step on one box, profiles/r04_ab_inc3_barrier.txt: the polling costs what the decoupling gains.)
        f32x4v b1[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const float4 t = p.bias ? *reinterpret_cast<const float4*>(p.bias + wn * 32 + cb * 16 + 4 * lg) : float4{0.f, 0.f, 0.f, 0.f};
            b1[cb] = f32x4v{t.x, t.y, t.z, t.w};
        }
        const int tpi = p.TH * p.TW;
        const int t = m_base + wm * 16 + l16;
        const bool tok = t < p.ntile;
        const int tt = tok ? t : 0;
        const int bimg = tt / tpi, rem = tt - bimg * tpi;
        const int ty = rem / p.TW, tx = rem - ty * p.TW;
        const int64_t pix0 = ((int64_t)bimg * p.H + 4 * ty) * p.W + 4 * tx;
        // A-operand offsets inside a pass's 128 x 128 block: row 32 wn + 16 j + l16, k-quad (8 src + 4 cb + lg) ^ l16; sidx 0-1 = own
        // channel blocks, 2-7 = the three partners' (source wave column (wn + 1 + (sidx - 2) / 2) % 4)
        int wa[8], xr[3];
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx) {
            const int src = sidx < 2 ? wn : (wn + 1 + (sidx - 2) / 2) & 3;
            const int qd = (src * 8 + (sidx & 1) * 4 + lg) ^ l16;
            wa[sidx] = (wn * 32 + l16) * K2 + (qd << 2);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) xr[k] = ((wm * 4 + ((wn + 1 + k) & 3)) * 8 + lg) * 64 + l16 * 4;
        const int x_own = (wave * 8 + lg) * 64 + l16 * 4;
        for (int i = tid; i < 512; i += NW * 64) b2s[i] = p.bias2[i];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = Y[pp][qq][cb][e] + b1[cb][e];
                        Y[pp][qq][cb][e] = p.relu ? fmaxf(v, 0.f) : v;
                    }
        for (int pass = 0; pass < 4; ++pass) {
            // (later passes: every wave is past its last read of the previous quarter)
            if (pass) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            {
                const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w2) + (int64_t)pass * W2_FLOATS, 0,
                                                                                    (unsigned)(W2_FLOATS * 4), 0x00020000);
#pragma unroll
                for (int it = 0; it < ROWS2 / (2 * NW); ++it) {
                    const int row = (it * NW + wave) * 2 + (lane >> 5);
                    const unsigned voff = (unsigned)(row * K2 + (((lane & 31) ^ (row & 15)) << 2)) * 4u;
                    dma1(rw, voff, lds0 + (unsigned)((it * NW + wave) * 1024));
                }
            }
            const int c0 = pass * 128 + wn * 32 + 4 * lg;               // first output channel of this lane in this pass
            const float* rbase = p.res + pix0 * p.C2 + c0;
            float* obase = p.out + pix0 * p.C2 + c0;
            auto res_rows = [&](f32x4v (&r)[2], int pos) {
                const int pp = pos >> 2, qq = pos & 3;
                const bool pok = tok && 4 * ty + pp < p.H && 4 * tx + qq < p.W;
                const float* rp = pok ? rbase + ((int64_t)pp * p.W + qq) * p.C2 : p.res + c0;
#pragma unroll
                for (int j = 0; j < 2; ++j) r[j] = *reinterpret_cast<const f32x4v*>(rp + j * 16);
            };
            f32x4v rs[2][2], Pr[3][2];
            res_rows(rs[0], 0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the matrix quarter (and position 0's residual rows) landed
#if MM_INC3_PAIR
            // Two positions per exchange round (round 5): the eight waves of the workgroup meet at 8 instead of 16 barriers per pass,
            // 128 MFMAs per wave between them; the exchange area holds 2 x 2 positions (32 KB more than the dead ring: 130 KB)
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) {
                float* xbuf = xb + (pr & 1) * 2 * XB_FLOATS;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        *reinterpret_cast<f32x4v*>(xbuf + h * XB_FLOATS + x_own + cb * 256) = Y[(2 * pr + h) >> 2][(2 * pr + h) & 3][cb];
                // buffer (pr & 1) was last read in round pr - 2; every wave has passed the barrier of round pr - 1, i.e. finished those reads
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int pos = 2 * pr + h, pp = pos >> 2, qq = pos & 3;
                    // (the partners' copies of ONE position at a time: both positions' at once spilled 64 registers, 12.4 vs 9.8 ms;
                    //  the scheduling barrier keeps hipcc from hoisting the second position's reads above the first one's MFMAs)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) Pr[k][cb] = *reinterpret_cast<const f32x4v*>(xbuf + h * XB_FLOATS + xr[k] + cb * 256);
                    if (pos + 1 < 16) res_rows(rs[(pos + 1) & 1], pos + 1);
                    f32x4v acc[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j] = *reinterpret_cast<const f32x4v*>(b2s + c0 + j * 16);
#pragma unroll
                    for (int sidx = 0; sidx < 8; ++sidx) {
                        const f32x4v Bv = sidx < 2 ? Y[pp][qq][sidx & 1] : Pr[(sidx - 2) >> 1][sidx & 1];
                        float4 w4[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j) w4[j] = *reinterpret_cast<const float4*>(lds + wa[sidx] + (j * 16) * K2);
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].x, Bv[0], acc[j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].y, Bv[1], acc[j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].z, Bv[2], acc[j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].w, Bv[3], acc[j], 0, 0, 0);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[j] = acc[j] + rs[pos & 1][j];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[j][e] = fmaxf(acc[j][e], 0.f);
                    }
                    if (tok && 4 * ty + pp < p.H && 4 * tx + qq < p.W) {
                        float* op = obase + ((int64_t)pp * p.W + qq) * p.C2;
#pragma unroll
                        for (int j = 0; j < 2; ++j) inc_store(acc[j], op + j * 16);
                    }
                }
            }
            continue;
#endif
#pragma unroll
            for (int pos = 0; pos < 16; ++pos) {
                const int pp = pos >> 2, qq = pos & 3;
                float* xbuf = xb + (pos & 1) * XB_FLOATS;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) *reinterpret_cast<f32x4v*>(xbuf + x_own + cb * 256) = Y[pp][qq][cb];
                // one barrier per position: buffer (pos & 1) was last read at position pos - 2, and every wave has passed the
                // barrier of position pos - 1 -- i.e. finished those reads (lgkmcnt(0) before it) -- before anyone writes it again
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) Pr[k][cb] = *reinterpret_cast<const f32x4v*>(xbuf + xr[k] + cb * 256);
                if (pos + 1 < 16) res_rows(rs[(pos + 1) & 1], pos + 1);
                f32x4v acc[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = *reinterpret_cast<const f32x4v*>(b2s + c0 + j * 16);
#pragma unroll
                for (int sidx = 0; sidx < 8; ++sidx) {
                    const f32x4v Bv = sidx < 2 ? Y[pp][qq][sidx & 1] : Pr[(sidx - 2) >> 1][sidx & 1];
                    float4 w4[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) w4[j] = *reinterpret_cast<const float4*>(lds + wa[sidx] + (j * 16) * K2);
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].x, Bv[0], acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].y, Bv[1], acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].z, Bv[2], acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].w, Bv[3], acc[j], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[j] = acc[j] + rs[pos & 1][j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][e] = fmaxf(acc[j][e], 0.f);
                }
                if (tok && 4 * ty + pp < p.H && 4 * tx + qq < p.W) {
                    float* op = obase + ((int64_t)pp * p.W + qq) * p.C2;
#pragma unroll
                    for (int j = 0; j < 2; ++j) inc_store(acc[j], op + j * 16);
                }
            }
        }
        return;
    }
    if constexpr (INC == 2) {
        // ---- first block of conv2_x: increase conv + projection shortcut as ONE contraction over K = 64 (this layer's output) + 64 (the
        // block input x at the same pixels): out = relu( [W2a | W2b] [relu(Y + bias); x] + bias2 ), W2 = [256][128] (make_layer_dual).
        // The matrix is 128 KB: two passes over its row halves (128 output channels each, 64 KB in the dead ring); wave (wm, wn)
        // produces channels [128 pass + 64 wn, + 64) of its 16 tiles.  The x operand needs no staging at all: in the transposed
        // layout a lane's B values are 4 consecutive channels of its tile's pixel -- one 16-byte global load.
        static_assert(WGM == 2 && NBUF == 3, "INC is built for the four-wave workgroup");
        constexpr int K2 = 2 * KS;                                     // 128
        constexpr int W2_FLOATS = 128 * K2;                            // one row half
        float* xb = lds + W2_FLOATS;
        float* b2s = xb + 2048;
        const int x_own = (wave * 8 + lg) * 64 + l16 * 4, x_par = ((wave ^ 1) * 8 + lg) * 64 + l16 * 4;
        f32x4v b1[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const float4 t = p.bias ? *reinterpret_cast<const float4*>(p.bias + wn * 32 + cb * 16 + 4 * lg) : float4{0.f, 0.f, 0.f, 0.f};
            b1[cb] = f32x4v{t.x, t.y, t.z, t.w};
        }
        const int tpi = p.TH * p.TW;
        const int t = m_base + wm * 16 + l16;
        const bool tok = t < p.ntile;
        const int tt = tok ? t : 0;
        const int bimg = tt / tpi, rem = tt - bimg * tpi;
        const int ty = rem / p.TW, tx = rem - ty * p.TW;
        const int64_t pix0 = ((int64_t)bimg * p.H + 4 * ty) * p.W + 4 * tx;
        const float* xbase = p.res + pix0 * KS + 4 * lg;               // x: 64 channels per pixel
        // A-operand offsets (row stride 128 floats, 32 k-quads per row, quad q of row r in slot q ^ (r & 15)): sidx 0-1 own channel
        // blocks, 2-3 the partner's, 4-7 the four 16-channel blocks of x
        int wa[8];
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx) {
            const int qd = sidx < 4 ? ((sidx < 2 ? wn : 1 - wn) * 8 + (sidx & 1) * 4 + lg) : 16 + (sidx - 4) * 4 + lg;
            wa[sidx] = (wn * 64 + l16) * K2 + ((qd ^ l16) << 2);
        }
        b2s[tid] = p.bias2[tid];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = Y[pp][qq][cb][e] + b1[cb][e];
                        Y[pp][qq][cb][e] = p.relu ? fmaxf(v, 0.f) : v;
                    }
        auto x_rows = [&](f32x4v (&r)[4], int pos) {
            const int pp = pos >> 2, qq = pos & 3;
            const bool pok = tok && 4 * ty + pp < p.H && 4 * tx + qq < p.W;
            const float* xp = pok ? xbase + ((int64_t)pp * p.W + qq) * KS : p.res + 4 * lg;
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = *reinterpret_cast<const f32x4v*>(xp + j * 16);
        };
        for (int pass = 0; pass < 2; ++pass) {
            // (second pass: every wave is past its last read of the first half)
            // (second pass: every wave is past its last read of the first half)
            if (pass) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            {
                const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w2) + (int64_t)pass * W2_FLOATS, 0,
                                                                                    (unsigned)(W2_FLOATS * 4), 0x00020000);
#pragma unroll
                for (int it = 0; it < 128 / (2 * NW); ++it) {
                    const int row = (it * NW + wave) * 2 + (lane >> 5);
                    const unsigned voff = (unsigned)(row * K2 + (((lane & 31) ^ (row & 15)) << 2)) * 4u;
                    dma1(rw, voff, lds0 + (unsigned)((it * NW + wave) * 1024));
                }
            }
            const int c0 = pass * 128 + wn * 64 + 4 * lg;               // first output channel of this lane in this pass
            float* obase = p.out + pix0 * p.C2 + c0;
            f32x4v xs[2][4], Pr[2];
            x_rows(xs[0], 0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the matrix half (and position 0's x rows) landed
#pragma unroll
            for (int pos = 0; pos < 16; ++pos) {
                const int pp = pos >> 2, qq = pos & 3;
                if (pos) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) *reinterpret_cast<f32x4v*>(xb + x_own + cb * 256) = Y[pp][qq][cb];
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) Pr[cb] = *reinterpret_cast<const f32x4v*>(xb + x_par + cb * 256);
                if (pos + 1 < 16) x_rows(xs[(pos + 1) & 1], pos + 1);
                f32x4v acc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = *reinterpret_cast<const f32x4v*>(b2s + c0 + j * 16);
#pragma unroll
                for (int sidx = 0; sidx < 8; ++sidx) {
                    const f32x4v Bv = sidx < 2 ? Y[pp][qq][sidx & 1] : sidx < 4 ? Pr[sidx & 1] : xs[pos & 1][sidx - 4];
                    float4 w4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) w4[j] = *reinterpret_cast<const float4*>(lds + wa[sidx] + (j * 16) * K2);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].x, Bv[0], acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].y, Bv[1], acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].z, Bv[2], acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].w, Bv[3], acc[j], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][e] = fmaxf(acc[j][e], 0.f);
                if (tok && 4 * ty + pp < p.H && 4 * tx + qq < p.W) {
                    float* op = obase + ((int64_t)pp * p.W + qq) * p.C2;
#pragma unroll
                    for (int j = 0; j < 4; ++j) inc_store(acc[j], op + j * 16);
                }
            }
        }
        return;
    }
    if constexpr (INC == 1) {
        // ---- second GEMM: out[n2][pixel] = relu( sum_c W2[n2][c] * relu(Y + bias)[c][pixel] + bias2[n2] + res[pixel][n2] ), C2 = 256.
        // The dead operand ring takes the whole increase matrix (256 rows x 256 B, same XOR-swizzled rows as the slabs) plus an 8 KB
        // exchange area.  A wave holds 32 of the 64 channels of its 16 tiles; the wave with the other 32 (same tiles: wave ^ 1) gets a
        // lane-for-lane copy of the registers through LDS -- in the transposed layout the accumulator image of a position IS the B
        // operand (k = 4 lg + e inside a 16-channel block, any k order is a valid contraction order as long as A uses the same).
        // Wave (wm, wn) then produces channels [128 wn, 128 wn + 128) of its 16 tiles: lane (l16, lg) ends up with 4 consecutive
        // output channels of tile l16 per 16-row block -> 16-byte stores and residual loads, 64 contiguous bytes per tile per
        // instruction, no transpose.
        // NEXT (round 6): ... and a THIRD GEMM chained from those accumulators -- again the B operand of a 16x16x4 MFMA as they are --
        // against the next block's 64 x 256 reduce matrix W3 (64 KB, 1 KB rows, same XOR swizzle): each wave contracts its own 128
        // channels into partial sums D for all 64 outputs (128 MFMAs per position, as many as the increase conv), hands the half the
        // partner finishes over through the exchange area, adds the partner's half to its own, + bias3, ReLU, 16-byte stores of the
        // 64-channel tensor.  Both matrices + exchange = 145 KB: an EIGHT-wave workgroup (WGM = 4: 64 tiles x 64 channels, one per
        // CU, still two waves per SIMD); the four-wave NEXT instantiation exists in -DMM_MEASURE builds only and reads its "W3"
        // fragments from the W2 area (results wrong by construction: the cost of the third GEMM with two workgroups per CU kept).
        static_assert((WGM == 2 || WGM == 4) && WGN == 2 && NBUF == 3, "INC 1 is built for the four- and the eight-wave workgroup");
        constexpr int W2_FLOATS = 256 * KS;
        constexpr bool W3_REAL = NEXT && WGM == 4;                        // W3 has its own 64 KB of LDS
        {
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w2), 0, (unsigned)(W2_FLOATS * 4), 0x00020000);
#pragma unroll
            for (int it = 0; it < 256 / (4 * NW); ++it) {
                const int row = (it * NW + wave) * 4 + (lane >> 4);
                const unsigned voff = (unsigned)(row * KS + (((lane & 15) ^ (row & 15)) << 2)) * 4u;
                dma1(rw, voff, lds0 + (unsigned)((it * NW + wave) * 1024));
            }
        }
        float* xb = lds + W2_FLOATS;                                   // [NW waves][2 channel blocks][4 lg][16 l16] float4
        const int x_own = (wave * 8 + lg) * 64 + l16 * 4, x_par = ((wave ^ 1) * 8 + lg) * 64 + l16 * 4;
        // bias2 -> LDS behind the exchange area (1 KB): read back per block as the accumulators' initial value
        float* b2s = xb + NW * 512;
        float* b3s = b2s + 256;                                        // NEXT: bias3 [64]
        float* w3s = b3s + 64;                                         // NEXT: [64][256] floats, quad q of row r in slot q ^ (r & 15)
        if constexpr (W3_REAL) {
            const __amdgpu_buffer_rsrc_t rw3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w3), 0, (unsigned)(64 * 256 * 4), 0x00020000);
            const unsigned w3b = lds0 + (unsigned)(W2_FLOATS + NW * 512 + 256 + 64) * 4u;
#pragma unroll
            for (int it = 0; it < 64 / NW; ++it) {
                const int row = it * NW + wave;                            // one 1 KB row per wave instruction
                const unsigned voff = (unsigned)(row * 256 + ((lane ^ (row & 15)) << 2)) * 4u;
                dma1(rw3, voff, w3b + (unsigned)(row * 1024));
            }
        }
        f32x4v b1[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const float4 t = p.bias ? *reinterpret_cast<const float4*>(p.bias + wn * 32 + cb * 16 + 4 * lg) : float4{0.f, 0.f, 0.f, 0.f};
            b1[cb] = f32x4v{t.x, t.y, t.z, t.w};
        }
        const int tpi = p.TH * p.TW;
        const int t = m_base + wm * 16 + l16;
        const bool tok = t < p.ntile;
        const int tt = tok ? t : 0;
        const int bimg = tt / tpi, rem = tt - bimg * tpi;
        const int ty = rem / p.TW, tx = rem - ty * p.TW;
        const int64_t pix0 = ((int64_t)bimg * p.H + 4 * ty) * p.W + 4 * tx;
        const int c0 = wn * 128 + 4 * lg;                             // first output channel of this lane (block 0)
        const float* rbase = p.res + pix0 * p.C2 + c0;
        float* obase = p.out + pix0 * p.C2 + c0;
        // A-operand fragment offsets: row n2 = 128 wn + 16 blk + l16, k-quad (8 src_wn + 4 cb + lg) ^ l16
        int wa[4];
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int qd = ((sidx < 2 ? wn : 1 - wn) * 8 + (sidx & 1) * 4 + lg) ^ l16;
            wa[sidx] = (wn * 128 + l16) * KS + (qd << 2);
        }
        // NEXT: W3 fragment of output block nb, half h, 16-channel block j: row 16 nb + l16, k-quad (32 wn + 16 h + 4 j + lg) ^ l16 (the XOR only
        // touches the low four bits of the quad index: one lane offset per j; nb and h fold into the instruction's offset field)
        int w3a[4] = {0, 0, 0, 0};
        float* o3base = nullptr;
        if constexpr (NEXT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (W3_REAL) w3a[j] = l16 * 256 + (((32 * wn + 4 * j + lg) ^ l16) << 2);
                else w3a[j] = l16 * KS + (((4 * j + lg) ^ l16) << 2);    // measurement proxy: valid, conflict-free addresses inside W2
            }
            o3base = p.out3 + pix0 * 64 + 32 * wn + 4 * lg;
        }
        if (tid < 256) b2s[tid] = p.bias2[tid];                          // C2 == 256
        if (NEXT && tid < 64) b3s[tid] = p.bias3[tid];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // W2 (W3) and bias2 landed
        // 32 half-position steps (position = (pp, qq), half = 64 of this wave's 128 output channels).  The residual rows of step
        // s + 1 are requested before the MFMAs of step s (their HBM latency runs under 64 MFMAs and, across a position change,
        // under the exchange barriers).
        auto res_rows = [&](f32x4v (&r)[4], int step) {
            const int pp = step >> 3, qq = (step >> 1) & 3, h = step & 1;
            const bool pok = tok && 4 * ty + pp < p.H && 4 * tx + qq < p.W;
#ifdef MM_MEASURE
            const float* rp = !pok ? p.res + c0
                              : (p.ablate & 2) ? p.res + ((pix0 + (int64_t)pp * p.W + qq) & 4095) * p.C2 + c0
                                               : rbase + ((int64_t)pp * p.W + qq) * p.C2;
#else
            const float* rp = pok ? rbase + ((int64_t)pp * p.W + qq) * p.C2 : p.res + c0;   // clipped pixel: a valid address, nothing stored
#endif
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = *reinterpret_cast<const f32x4v*>(rp + h * 64 + j * 16);
        };
        // (round 5) the residual rows of up to MM_INC1_RES_DEPTH steps ahead are in flight: the kernel's time is the SUM of its MFMA time and
        // its memory time (profiles/r05_inc1_ablation.txt: without residual loads / stores / operand DMA -1.0 / -1.5 / -1.1 ms of 7.8, without 64 %
        // of the MFMAs -0.65) -- eight waves per CU with one step (4 KB per wave) of loads in flight do not cover the HBM latency.  The depth
        // grows as the Y registers of finished positions die (8 per position): 1 for positions 0-1, 2 for 2-3, 3 from position 4 on
        // (NEXT, which also carries the 16 partial-sum registers D: two positions later).
#ifndef MM_INC1_RES_DEPTH
#define MM_INC1_RES_DEPTH 3
#endif
        auto res_ahead = [](int step) {      // last step whose residual rows have been requested once step `step` has issued its loads
            if (step < 0) return 0;
            int d = 1 + ((NEXT ? (step < 4 ? 0 : step - 4) : step) >> 2);
            d = d > MM_INC1_RES_DEPTH ? MM_INC1_RES_DEPTH : d;
            const int t = step + d;
            return t > 31 ? 31 : t;
        };
        if constexpr (NEXT) {      // bias + ReLU of all 16 positions up front: the 8 bias registers are not carried through the steps
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = Y[pp][qq][cb][e] + b1[cb][e];
                            Y[pp][qq][cb][e] = p.relu ? fmaxf(v, 0.f) : v;
                        }
        }
        f32x4v rs[4][4], Pr[2], D[4];
#ifdef MM_MEASURE
        const int abl = p.ablate;
#else
        constexpr int abl = 0;
#endif
        if (abl & 8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) rs[0][j] = rs[1][j] = rs[2][j] = rs[3][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        } else
        res_rows(rs[0], 0);
#pragma unroll
        for (int step = 0; step < 32; ++step) {
            const int pp = step >> 3, qq = (step >> 1) & 3, h = step & 1;
            if (h == 0) {
                if constexpr (!NEXT) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = Y[pp][qq][cb][e] + b1[cb][e];
                        Y[pp][qq][cb][e] = p.relu ? fmaxf(v, 0.f) : v;
                    }
                }
                // the previous position's copies have been read.  (NEXT: no barrier here -- a wave's area is last read BY ITSELF, the partial
                // sums its partner left there, before it writes the next position's copy: program order)
                if (!NEXT && step && !(abl & 32)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) *reinterpret_cast<f32x4v*>(xb + x_own + cb * 256) = Y[pp][qq][cb];
                if (!(abl & 32)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) Pr[cb] = *reinterpret_cast<const f32x4v*>(xb + x_par + cb * 256);
                if constexpr (NEXT) D[0] = D[1] = D[2] = D[3] = f32x4v{0.f, 0.f, 0.f, 0.f};
            }
            if (!(abl & 8)) {
#pragma unroll
                for (int t = res_ahead(step - 1) + 1; t <= res_ahead(step); ++t) res_rows(rs[t & 3], t);
            }
            f32x4v acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = *reinterpret_cast<const f32x4v*>(b2s + c0 + h * 64 + j * 16);
            if (!(abl & 64))
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const f32x4v Bv = sidx < 2 ? Y[pp][qq][sidx & 1] : Pr[sidx & 1];
                float4 w4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w4[j] = *reinterpret_cast<const float4*>(lds + wa[sidx] + (h * 64 + j * 16) * KS);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].x, Bv[0], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].y, Bv[1], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].z, Bv[2], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j].w, Bv[3], acc[j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = acc[j] + rs[step & 3][j];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][e] = fmaxf(acc[j][e], 0.f);
            }
            const bool pix_ok = tok && 4 * ty + pp < p.H && 4 * tx + qq < p.W;
            if (pix_ok && !(abl & 16)) {
                float* op = obase + ((int64_t)pp * p.W + qq) * p.C2 + h * 64;
#pragma unroll
                for (int j = 0; j < 4; ++j) inc_store(acc[j], op + j * 16);
            }
            if constexpr (NEXT) {
                // third GEMM: D[nb] += W3[16 nb + l16][these 64 channels] * x[these 64 channels][tile l16]; x = acc as it stands
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4 w4[4];
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
                        w4[nb] = *reinterpret_cast<const float4*>((W3_REAL ? w3s + nb * 16 * 256 + 16 * h * 4 : lds + (nb * 16 + h * 64) * KS) + w3a[j]);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) D[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[nb].x, acc[j][0], D[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) D[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[nb].y, acc[j][1], D[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) D[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[nb].z, acc[j][2], D[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) D[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[nb].w, acc[j][3], D[nb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);      // one block's four fragments at a time (hoisted, the reads of all four blocks spill)
                }
                if (h == 1) {
                    // wave wn finishes output blocks 2 wn, 2 wn + 1: the other two go to the partner through ITS area (which this wave has finished
                    // reading the partner's Y copy from), the partner's arrive in this wave's own
                    f32x4v give[2], keep[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float lo = D[i][e], hi = D[2 + i][e];
                            give[i][e] = wn ? lo : hi;
                            keep[i][e] = wn ? hi : lo;
                        }
#pragma unroll
                    for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4v*>(xb + x_par + i * 256) = give[i];
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const f32x4v got = *reinterpret_cast<const f32x4v*>(xb + x_own + i * 256);
                        f32x4v d = keep[i] + got + *reinterpret_cast<const f32x4v*>(b3s + 32 * wn + 16 * i + 4 * lg);
#pragma unroll
                        for (int e = 0; e < 4; ++e) d[e] = fmaxf(d[e], 0.f);
                        if (pix_ok && !(abl & 16)) __builtin_nontemporal_store(d, reinterpret_cast<f32x4v*>(o3base + ((int64_t)pp * p.W + qq) * 64 + i * 16));
                    }
                }
            }
        }
        return;
    }
    // ---- epilogue.  MFMA C layout: column (channel) = l16 (+ 16 per channel block), row (tile) = 4 lg + e.  Each wave transposes
    //      one output pixel position at a time through its private 16 x 36-float staging rows and then owns 4 consecutive channels
    //      of a tile per lane: 16-byte stores, 128 contiguous bytes per tile.
    constexpr int SLD = 36;
    float* st = lds + wave * (16 * SLD);
    const int cq = lane & 7, tr = lane >> 3;
    const int n4 = n_base + wn * 32 + cq * 4;
    const bool nok = n4 < p.Cout;
    float4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (nok && p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + n4);
    const int tpi = p.TH * p.TW;
    int64_t obase[2];
    int ty4[2], tx4[2];
    bool tok[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int t = m_base + wm * 16 + g * 8 + tr;
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
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int e = 0; e < 4; ++e) st[(4 * lg + e) * SLD + cb * 16 + l16] = Y[pp][qq][cb][e];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            float4 v[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) v[g] = *reinterpret_cast<const float4*>(st + (g * 8 + tr) * SLD + cq * 4);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                if (!tok[g] || ty4[g] + pp >= p.H || tx4[g] + qq >= p.W) continue;
                float4 o = {v[g].x + bias4.x, v[g].y + bias4.y, v[g].z + bias4.z, v[g].w + bias4.w};
                if (p.relu) o = float4{fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f)};
                const f32x4v ov = {o.x, o.y, o.z, o.w};
                __builtin_nontemporal_store(ov, reinterpret_cast<f32x4v*>(p.out + obase[g] + ((int64_t)pp * p.W + qq) * p.Cout));
            }
        }
    }
}

template <int NBUF, int WGM, int INC = 0, int WGN = 2, int KSL = 0, int NEXT = 0>
static int launch_fused_k(WinoFusedParams p, hipStream_t s) {
    constexpr int BM = 16 * WGM, BN = 32 * WGN;
    // INC: the increase matrix (64 KB) + exchange area (8 KB) take the dead operand ring, bias2 (1 KB) sits behind it: 73 KB, two
    // workgroups per CU still fit the 160 KB.  Eight waves (WGM = 4): 16 KB exchange area, and with NEXT the 64 KB reduce matrix: 145 KB
    constexpr int RING_BYTES = NBUF * (BM + BN) * 64 * 4;
    constexpr int EPI_BYTES = (INC == 1 || INC == 2) ? (256 * 64 + WGM * WGN * 512 + 256 + (NEXT ? 64 : 0) + (NEXT && WGM == 4 ? 64 * 256 : 0)) * 4 : 0;
    constexpr int LDS_BYTES = (INC == 1 || INC == 2) ? (EPI_BYTES > RING_BYTES ? EPI_BYTES : RING_BYTES)
                              : INC == 3 ? inc3_lds_bytes(RING_BYTES) : RING_BYTES;
    static_assert(LDS_BYTES >= RING_BYTES && LDS_BYTES <= 160 * 1024, "the ring must fit too");
    static bool attr_set[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 16 && !attr_set[dev]) {
        MM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_fused_kernel<NBUF, WGM, INC, WGN, KSL, NEXT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   LDS_BYTES));
        attr_set[dev] = true;
    }
#ifdef MM_MEASURE
    // cost-side measurements (tools/ scripts, never the shipped library): MM_WF_LDS_PAD = extra dynamic LDS bytes (forces one workgroup
    // per CU), MM_WF_ABLATE = see WinoFusedParams::ablate; both apply to the INC instantiations only
    static const int lds_pad = getenv("MM_WF_LDS_PAD") ? atoi(getenv("MM_WF_LDS_PAD")) : 0;
    static const int ablate = getenv("MM_WF_ABLATE") ? atoi(getenv("MM_WF_ABLATE")) : 0;
    const int lds_bytes = LDS_BYTES + (INC ? lds_pad : 0);
    p.ablate = INC ? ablate : 0;
    if (INC && lds_pad)
        MM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_fused_kernel<NBUF, WGM, INC, WGN, KSL, NEXT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   lds_bytes));
#else
    constexpr int lds_bytes = LDS_BYTES;
    p.ablate = 0;
#endif
    p.tiles_n = (p.Cout + BN - 1) / BN;
    for (int q = 0; q < 6; ++q)
        for (int qq = 0; qq < 4; ++qq) p.at_cols[q * 4 + qq] = kAtHost[qq][q];
    const int64_t blocks = (int64_t)((p.ntile + BM - 1) / BM) * p.tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffff) return MM_ERR_INVALID_ARG;
    if (prof_enabled()) {
        char tag[64];
        snprintf(tag, sizeof(tag), "wino-fused%s%s M=%d K=%d N=%d t%dx%d b36", INC == 3 ? "+inc512" : INC == 2 ? "+incproj256" : INC ? "+inc256" : "",
                 NEXT ? "+red64" : "", p.ntile, p.K, p.Cout, BM, BN);
        double fl = 2.0 * 36.0 * (double)p.ntile * (double)p.K * (double)p.Cout;
        if (INC) fl += 2.0 * (double)p.B * p.H * p.W * (double)(INC == 2 ? 2 * p.Cout : p.Cout) * (double)p.C2;
        if (NEXT) fl += 2.0 * (double)p.B * p.H * p.W * (double)p.C2 * 64.0;
        prof_before(0, fl, s, tag);
    }
    hipLaunchKernelGGL((wino_fused_kernel<NBUF, WGM, INC, WGN, KSL, NEXT>), dim3((unsigned)blocks), dim3(WGM * WGN * 64), lds_bytes, s, p);
    prof_after(0, s);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

// K = 64 / 128 / 256 (every ResNet50 / PhaseNet layer that comes here) take the compile-time-scheduled loop, any other multiple of 64 -- and
// every K when the caller asks for the twin (p.generic_loop: MM_WF_KSL=0 at create time, the parity twin) -- the generic one
template <int NBUF, int WGM, int INC = 0, int WGN = 2, int NEXT = 0>
static int launch_fused(const WinoFusedParams& p, hipStream_t s) {
    if constexpr (INC == 1 && (NEXT || WGM == 4)) {      // round 6: the eight-wave conv2_x kernel and its NEXT form exist for K = 64 on the compile-time loop,
        if (p.K != 64) return MM_ERR_UNSUPPORTED;        // the generic loop as the twin
        return p.generic_loop ? launch_fused_k<NBUF, WGM, INC, WGN, 0, NEXT>(p, s) : launch_fused_k<NBUF, WGM, INC, WGN, 1, NEXT>(p, s);
    } else {
    // (the eight-wave conv3_x kernel -- 2 x 4 waves, K = 128 -- takes it with a second set of fragment base registers for ring slot 2, see FAR2)
    if constexpr (NBUF == 3 && WGM == 2 && WGN == 4 && INC == 3) {
        if (!p.generic_loop && p.K == 128) return launch_fused_k<NBUF, WGM, INC, WGN, 2>(p, s);
    }
    if constexpr (NBUF == 3 && WGM * WGN == 4) {
        if (!p.generic_loop && p.K == 64) return launch_fused_k<NBUF, WGM, INC, WGN, 1>(p, s);
        if constexpr (INC == 0) {      // the INC kernels exist for K = 64 only
            if (!p.generic_loop && p.K == 128) return launch_fused_k<NBUF, WGM, INC, WGN, 2>(p, s);
            if (!p.generic_loop && p.K == 256) return launch_fused_k<NBUF, WGM, INC, WGN, 4>(p, s);
        }
    }
    return launch_fused_k<NBUF, WGM, INC, WGN, 0>(p, s);
    }
}

bool wino_fused_supported(int64_t ntile, int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 32) return false;
    // 32-bit buffer offsets inside a position plane: larger problems are not an error, the caller takes another form
    return (ntile + 64) * Cin * 4 < 0xFFFFF000ll && ((int64_t)Cout + 64) * Cin * 4 < 0xFFFFF000ll;
}

bool wino_fused_inc_supported(int64_t ntile, int Cin, int Cout, int C2) {
    return ((Cout == 64 && C2 == 256) || (Cout == 128 && C2 == 512)) && wino_fused_supported(ntile, Cin, Cout);
}

// V [36][ntile][Cin] (from wino_input_transform, m = 4), U [36][Cout][Cin] -> y NHWC [B,H,W,Cout] (+bias, ReLU)
int wino_gemm_output_fused(const float* V, const float* U, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                            int relu, int shape, hipStream_t s, int generic_loop) {
    WinoFusedParams p;
    p.generic_loop = generic_loop;
    p.V = V; p.U = U; p.bias = bias; p.out = y;
    p.w2 = nullptr; p.bias2 = nullptr; p.res = nullptr; p.C2 = 0;
    p.w3 = nullptr; p.bias3 = nullptr; p.out3 = nullptr;
    p.TH = (H + 3) / 4; p.TW = (W + 3) / 4;
    const int64_t ntile = (int64_t)B * p.TH * p.TW;
    if (!wino_fused_supported(ntile, Cin, Cout)) return MM_ERR_UNSUPPORTED;
    if (ntile <= 0) return MM_OK;
    p.ntile = (int)ntile; p.K = Cin; p.Cout = Cout; p.B = B; p.H = H; p.W = W; p.relu = relu; p.tiles_n = 0;
    // shape: 0 = four-wave workgroups of 32 tiles x 64 channels, two per CU (default: measured 3-6 % faster than one eight-wave
    // workgroup per CU -- no common barrier between the two waves of a SIMD, prologue / epilogue of one workgroup under the
    // other's main loop); 8 = eight waves (64 x 64), 3-deep ring; 4 = eight waves, 4-deep ring
    return shape == 4 ? launch_fused<4, 4>(p, s) : shape == 8 ? launch_fused<3, 4>(p, s) : launch_fused<3, 2>(p, s);
}

// The 3x3 layer AND the block's increase conv: V, U as above with Cout == 64;  out [B,H,W,C2] = relu( W2 relu(conv3x3 + bias) +
// bias2 + res ), W2 [C2][64] (BN folded), res NHWC [B,H,W,C2], C2 == 256.  MM_ERR_UNSUPPORTED for any other shape.
int wino_gemm_output_fused_inc(const float* V, const float* U, const float* bias, const float* W2, const float* bias2, const float* res,
                               float* out, int B, int H, int W, int Cin, int Cout, int C2, int relu, hipStream_t s, int generic_loop,
                               const float* next_w, const float* next_bias, float* next_out, int shape) {
    if (!V || !U || !W2 || !bias2 || !res || !out) return MM_ERR_INVALID_ARG;
    WinoFusedParams p;
    p.generic_loop = generic_loop;
    p.V = V; p.U = U; p.bias = bias; p.out = out;
    p.w2 = W2; p.bias2 = bias2; p.res = res; p.C2 = C2;
    p.w3 = nullptr; p.bias3 = nullptr; p.out3 = nullptr;
    p.TH = (H + 3) / 4; p.TW = (W + 3) / 4;
    const int64_t ntile = (int64_t)B * p.TH * p.TW;
    if (!wino_fused_inc_supported(ntile, Cin, Cout, C2)) return MM_ERR_UNSUPPORTED;
    if (ntile <= 0) return MM_OK;
    p.ntile = (int)ntile; p.K = Cin; p.Cout = Cout; p.B = B; p.H = H; p.W = W; p.relu = relu; p.tiles_n = 0;
    if (Cout == 128) return next_w ? MM_ERR_UNSUPPORTED : launch_fused<3, 2, 3, 4>(p, s);     // conv3_x: 2 x 4 waves, 32 tiles x 128 channels, 128 -> 512
    if (next_w) {
        // the next block's reduce conv in the same kernel (NEXT): eight-wave workgroups, W2 + W3 in LDS.  shape 2 (-DMM_MEASURE builds only): the
        // four-wave cost proxy whose third GEMM reads W2's LDS rows -- results wrong by construction
        if (!next_bias || !next_out || C2 != 256) return MM_ERR_INVALID_ARG;
        p.w3 = next_w; p.bias3 = next_bias; p.out3 = next_out;
#ifdef MM_MEASURE
        if (shape == 2) return launch_fused<3, 2, 1, 2, 1>(p, s);
#endif
        return launch_fused<3, 4, 1, 2, 1>(p, s);
    }
    if (shape == 8) return launch_fused<3, 4, 1>(p, s);          // the eight-wave shape without NEXT (A/B of the shape alone)
    return launch_fused<3, 2, 1>(p, s);
}

// First block of conv2_x: the 3x3 layer AND increase conv + projection shortcut as one contraction over two K sources.
// out [B,H,W,C2] = relu( W2[:, :64] relu(conv3x3 + bias) + W2[:, 64:] x + bias2 ), W2 [C2][128] (make_layer_dual), x NHWC [B,H,W,64]
// at the same pixels (stride-1 shortcut), C2 == 256.
int wino_gemm_output_fused_incproj(const float* V, const float* U, const float* bias, const float* W2, const float* bias2, const float* x,
                                   float* out, int B, int H, int W, int Cin, int Cout, int C2, int relu, hipStream_t s, int generic_loop) {
    if (!V || !U || !W2 || !bias2 || !x || !out) return MM_ERR_INVALID_ARG;
    WinoFusedParams p;
    p.generic_loop = generic_loop;
    p.V = V; p.U = U; p.bias = bias; p.out = out;
    p.w2 = W2; p.bias2 = bias2; p.res = x; p.C2 = C2;
    p.w3 = nullptr; p.bias3 = nullptr; p.out3 = nullptr;
    p.TH = (H + 3) / 4; p.TW = (W + 3) / 4;
    const int64_t ntile = (int64_t)B * p.TH * p.TW;
    if (Cout != 64 || !wino_fused_inc_supported(ntile, Cin, Cout, C2)) return MM_ERR_UNSUPPORTED;
    if (ntile <= 0) return MM_OK;
    p.ntile = (int)ntile; p.K = Cin; p.Cout = Cout; p.B = B; p.H = H; p.W = W; p.relu = relu; p.tiles_n = 0;
    return launch_fused<3, 2, 2>(p, s);
}

}  // namespace mm
