// fp32 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One engine serves every dense contraction of the hot path:
//   * the 53 conv+BN(+ReLU)(+residual) layers of the ResNet50 trunk behind
//     Resnet50_Extractor.get_vec (api/resnet50_extractor.py:74-83),
//   * PhaseNet's six 3x3 convs and every nn.Linear of Two_Stream_RNN (api/mimamo_net.py:14-26,
//     41-95,115-122), the GRU input/recurrent products (api/mimamo_net.py:119) -- a Linear is a
//     1x1 conv on a 1x1 image,
//   * the batched Winograd-domain GEMMs of the stride-1 3x3 ResNet layers (36 problems per launch, winograd.hip).
// fp32 in / fp32 accumulate: the MFMA is bit-for-bit an fmaf chain, so results differ from the
// reference's fp32 convs only by summation order (the 1e-4 output tolerance of north_star rules
// out bf16/fp16 operands).
//
// GEMM view:  out[m][n] = sum_k A[m][k] * Wt[n][k],  m = (b, ho, wo),  n = cout,  k = (r, s, c) or, for
//   multi-tap kernels with Cin % 16 == 0, the slice-major order (c/16, r, s, c%16) (see `korder` below);
//   activations NHWC (channel stride/offset allow channel-sliced reads and concat-writes),
//   weights packed [Cout][Kpad] with k contiguous (Kpad = K rounded up to 16, zero filled).
// Workgroup = 256 threads = 4 waves (2x2, or 4x1 for the 256x64 tile) or 512 threads = 8 waves (2x4, the 128x256
// tile of the N % 256 == 0 GEMM shapes); block tile BM x BN x 16; each wave
// owns (BM/WGM)x(BN/WGN) as 32x32 MFMA sub-tiles.  Operand tiles go global -> LDS directly
// (buffer_load ... lds from inline asm: no VGPR staging, no ds_write) into a 3-deep LDS ring tracked with
// counted `s_waitcnt vmcnt(N)`, one `s_barrier` per 16-deep chunk; the 16-byte slots of each LDS row are
// XOR-swizzled so the ds_read_b128 fragment reads are bank-conflict free without padding.  Inside a chunk
// lanes 0-31 take k = 0..7 and lanes 32-63 take k = 8..15 (any k order is a valid contraction order), so one
// b128 read feeds four MFMAs.  Zero padding of every kind comes from the buffer descriptors' range check.
// Epilogue (fused): + bias (BN folded on the host) [+ residual] [ReLU] [* post_scale + post_shift]
// (BN placed after ReLU, mimamo_net.py:54-62,115-117); the accumulator tile is transposed through LDS so
// bias/residual/output move as 16-byte accesses, 256 bytes per row (wave-private staging, no workgroup barrier;
// outputs are written with the non-temporal hint: the next layer reads them from HBM, not from L2).
#include "mm_common.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "conv.h"

namespace mm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// compile-time loop: f(std::integral_constant<int, 0>()) ... f(std::integral_constant<int, N - 1>())
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>(), f);
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- X3 (round 5, `extra` only -- never the headline): fp32 operands split three ways into bf16 (x = h + m + l, every part the
// round-to-nearest bf16 of what is left: 3 x 8 = 24 mantissa bits) and multiplied on the bf16 matrix pipes, which run at 16 x the
// fp32-MFMA rate: a b ~= al bh + ah bl + am bm + am bh + ah bm + ah bh (six v_mfma_f32_32x32x16_bf16, fp32 accumulate, the three dropped
// cross terms are < 2^-24 relative).  The split runs in registers on the fp32 fragments the fp32 loop reads (same LDS image, same DMA,
// same epilogue: the bf16 32x32x16 C layout is the fp32 32x32x2 one), 5.5 VALU instructions per element.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {      // v_cvt_pk_bf16_f32 (round to nearest even), lo in bits 0..15
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float sub_f32(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split_bf16x3(const float4& q0, const float4& q1, u32x4& H, u32x4& M, u32x4& L) {
    const float x[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        // (the subtractions go through inline asm so that hipcc's SLP vectoriser cannot pair them into v_pk_add_f32: packed fp32 VALU
        //  next to MFMAs is an anti-lever on this chip -- cdna_hip_programming.md, and 4.03 vs 3.37 ms measured on the 512 -> 128 layers)
        const unsigned h = cvt_pk_bf16(x[2 * p], x[2 * p + 1]);
        const float r0 = sub_f32(x[2 * p], __uint_as_float(h << 16)), r1 = sub_f32(x[2 * p + 1], __uint_as_float(h & 0xffff0000u));   // exact
        const unsigned m = cvt_pk_bf16(r0, r1);
        const float s0 = sub_f32(r0, __uint_as_float(m << 16)), s1 = sub_f32(r1, __uint_as_float(m & 0xffff0000u));                   // exact
        H[p] = h; M[p] = m; L[p] = cvt_pk_bf16(s0, s1);
    }
}

// m / d for 0 <= m < 2^31 with (mul, sh) from fastdiv_make(d): q = (m * mul) >> (31 + L), L = ceil(log2 d), mul = ceil(2^(31 + L) / d) < 2^32
// (Granlund-Montgomery for 31-bit dividends: exact).  d == 1: mul = 0 marks the identity.
__device__ __forceinline__ int fastdiv(int m, unsigned mul, unsigned sh) {
    return mul ? (int)(__umulhi((unsigned)m, mul) >> sh) : m;
}
static void fastdiv_make(int d, unsigned& mul, unsigned& sh) {
    if (d <= 1) { mul = 0; sh = 0; return; }
    int L = 0;
    while ((1ll << L) < d) ++L;
    const unsigned long long num = 1ull << (31 + L);
    mul = (unsigned)((num + (unsigned long long)d - 1) / (unsigned long long)d);      // < 2^32 because d > 2^(L - 1)
    sh = (unsigned)(L - 1);                                                            // (m * mul) >> 32 >> (L - 1)
}

constexpr int CBK = 16;   // k-chunk
constexpr int CLD = 16;   // LDS row = one 16-float chunk; the four 16-byte slots of a row are XOR-swizzled

// ABL: measurement-only instantiation whose loop stages can be switched off at run time (p.ablate bits:
// 1 no loads + no tap math, 128 no loads, 256 no tap math, 4 no barrier, 8 no fragment reads, 32/64 wave-priority
// experiments) to attribute time; its results are wrong by construction.  Compiled only with -DMM_MEASURE (the profiling
// scripts under tools/ build that library; __graft_entry__.build() never does) and reached there through tile >= 16; the
// default library has no such instantiation and answers tile > 5 with MM_ERR_INVALID_ARG.
// KMODE selects the tap iteration at compile time (straight-line VALU in the hot loop):
//   0  k = (r,s,c), any Cin % 4 == 0 (stem: Cin = 4)      2  k = (r,s,c), Cin >= 16 (one wrap per chunk at most)
//   1  slice-major k = (c/16, r, s, c%16), Cin % 16 == 0    3  1x1 kernel, pad 0: no taps, no border
//   4  Cin == 4 and kw >= 4 (the 7x7 stem on NHWC4): a k-quad is one tap, four taps per chunk
//   5  3-channel input packed NHWC3 with its zero border IN MEMORY (korder 2, the stem): k = r * RG + s * 3 + c with the kw * 3
//      floats of a kernel row padded to RG = a multiple of 4 (7 x 3 = 21 -> 24: K = 168 instead of 7 * 7 * 4 = 196 -> 208); a
//      k-quad is 4 consecutive floats of the input row -- 4-byte aligned only, which the LDS-DMA accepts (tools/probes/unaligned_dma.hip)
//      -- and the three floats past a row group meet zero weights.  No border logic: pad must be 0.
//   6  1x1 kernel over TWO inputs (ConvParams::in2): chunks [0, Cin/16) come from `in`, the rest from `in2` sampled at stride2 --
//      increase conv + projection shortcut of a residual block in one accumulation (the chunk -> source choice is wave-uniform)
//   7 / 8  (round 5) modes 3 / 6 for K % 16 == 0 with NO vector instruction left in the loop besides the MFMAs: the per-lane DMA offsets are
//      loop constants and the k offset rides in the buffer instructions' SCALAR offset; the chunk loop is unrolled by the ring depth, so the
//      ring slot is a compile-time constant that folds into the ds_read_b128 offset fields and into the M0 immediates of the DMA.  Why:
//      tools/probes/valu_mfma_overlap.hip -- VALU issue time ADDS to matrix time on gfx950 (22 VALU instructions per 32 MFMAs in mode 3:
//      fragment addresses, lane offsets + their out-of-range selects).  Same products in the same order: bit-identical to modes 3 / 6.
//   9 / 10  (round 5) modes 5 / 2 for a FIXED chunk count, fully unrolled -- 9: the 7x7 stem (K = 7 x 24 = 168 -> 176: eleven chunks), 10: a 3x3
//      layer on 24 channels (PhaseNet's first conv, K = 216 -> 224: fourteen chunks).  The tap offset of a lane's k-quad in chunk c is not
//      separable into a lane part and a chunk part, so the byte offsets of every chunk (11 or 14 chunks x 2 pieces, out-of-range taps folded
//      in) are computed once with the base mode's own tap walk and held in registers; ring slots, the weight rows' k offset (scalar offset)
//      and the counted waits are compile-time constants.  Mode 5's loop spent ~28 vector instructions per 16 MFMAs on them.
//   11 (round 5) mode 1 (slice-major k) for 3x3 kernels: a chunk is one tap of one 16-channel slice, so the loop is unrolled by the nine taps
//      (a multiple of the ring depth): the nine per-lane tap offsets (border taps folded in as out-of-range) are loop constants, the slice rides
//      in the scalar offset of the buffer instructions.  Same products in the same order as the base modes: bit-identical.
// (the scheduled 1x1 loop of the 128x128 tile would take 188 registers -- every fragment read of a step in flight at once -- where its mode-3
//  twin's 156 keep three workgroups on a CU: held to three waves per SIMD.  The bf16x3 instantiations are held to the occupancy of their fp32 twins: the eight-wave one needs 133 registers where 128 keep two
//  workgroups on a CU; the four-wave 128x256 one -- 64x128 wave tiles -- 280 where 256 keep two waves on a SIMD)
template <int BM, int BN, int WGM, int WGN, int KMODE, bool ABL = false, int X3 = 0>
__global__ void __launch_bounds__(WGM * WGN * 64)
    __attribute__((amdgpu_waves_per_eu(X3 == 2 ? 2 : X3 ? (WGM * WGN == 8 ? 4 : 2) : (KMODE >= 7 && WGM * WGN == 4 && BM * BN == 128 * 128) ? 3 : 1, 8)))
conv_mfma_kernel(const ConvParams p) {
    constexpr int NW = WGM * WGN;                     // waves per workgroup: 4, or 8 for the 128x256 tile
    constexpr int TMODE = KMODE == 9 ? 5 : KMODE == 10 ? 2 : KMODE == 11 ? 1 : KMODE;      // the tap walk a scheduled mode takes its offsets from
    static_assert(NW == 4 || NW == 8, "four or eight waves per workgroup");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RPR = NW * 16;                      // operand rows one round of DMA instructions covers
    static_assert(BM % RPR == 0 && BN % RPR == 0, "tile sides must be multiples of 16 rows per wave");
    constexpr int AIT = BM / RPR, BIT = BN / RPR;
    // Operand tiles land in LDS by direct-to-LDS loads (buffer_load ... lds): a wave instruction writes 64 lanes
    // x 16 B = 1 KB contiguous, i.e. 16 rows x 4 slots, lane l -> slot l.  Slot s of row m holds k-quad
    // s ^ ((m >> 2) & 3): with that swizzle the ds_read_b128 fragment reads (lanes = consecutive rows, same
    // k-quad) hit 16 distinct 16-byte bank groups per service group, so no padding is needed and no VGPRs or
    // ds_write instructions are spent on staging.  The staging epilogue re-uses the same bytes (SLD = WN + 4).
    constexpr int NBUF = 3;               // LDS ring: chunk kc is consumed while kc+1 and kc+2 are in flight
    constexpr int STAGE_FLOATS = NW * 32 * (BN / WGN + 4);
    // X3 == 2 (round 6): the weights arrive PRE-SPLIT as three bf16 planes (ConvParams::w3): a row of a ring slot is three times 16 bf16 = 96 B
    // instead of 16 fp32 = 64 B -- plane-major [slot][plane][BN rows][32 B], lane-linear for the DMA (lane l -> row l >> 1, half l & 1) and for the
    // fragment reads (lanes = consecutive rows x two 16-byte halves: conflict free without a swizzle)
    constexpr int BROW = X3 == 2 ? 24 : CLD;      // floats of LDS per B row and ring slot
    constexpr int OPER_FLOATS = NBUF * (BM * CLD + BN * BROW);
    __shared__ __attribute__((aligned(16))) float lds[OPER_FLOATS > STAGE_FLOATS ? OPER_FLOATS : STAGE_FLOATS];
    float* As = lds;                      // [NBUF][BM][16]
    float* Bs = lds + NBUF * BM * CLD;    // [NBUF][BN][16]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    // XCD-aware tile order: workgroup b runs on XCD b % 8; give each XCD a contiguous run of logical
    // tiles, n-tiles of one m-tile adjacent, so the activation rows they share stay in that XCD's L2.
    const int nblk = gridDim.x;
    int logical;
    {
        const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // batched launch (Winograd: 16 independent GEMMs of one shape): the grid holds `batch` copies of the tile grid
    const int per_batch = p.tiles_m * p.tiles_n;
    const int bz = logical / per_batch;
    logical -= bz * per_batch;
    const float* __restrict__ in_b = p.in + (int64_t)bz * p.in_bstride;
    const float* __restrict__ w_b = p.w + (int64_t)bz * p.w_bstride;
    float* __restrict__ out_b = p.out + (int64_t)bz * p.out_bstride;
    const int tile_m = logical / p.tiles_n, tile_n = logical - tile_m * p.tiles_n;
    // hpool (the stem, KMODE 5 on the 128x64 tile): tiles start every BM - 2 rows -- a tile's last two pixels are computed again by the
    // next one, so that every 3-pixel pooling window that starts at an even pixel lies inside ONE tile (1.6 % more MFMA work)
    const bool hpool = TMODE == 5 && BM == 128 && BN == 64 && p.hpool;
    const int m_base = p.m_off + tile_m * (hpool ? BM - 2 : BM), n_base = tile_n * BN;

    // ---- operand fetch through buffer descriptors: the hardware range check returns 0 for any offset
    //      >= num_records, which gives zero padding (image border taps, K tail, row/channel tails) without
    //      branches or selects on the data.  The A descriptor is rebased to the first image this block
    //      touches so 32-bit byte offsets always suffice (a block spans a handful of images).
    const int hw_out = p.Ho * p.Wo;
    const int img0 = fastdiv(m_base, p.div_hw_mul, p.div_hw_sh);       // wave-uniform
    const int64_t img_elems = (int64_t)p.H * p.W * p.in_cstride;
    const int64_t rem_elems = ((int64_t)p.B - img0) * img_elems;
    const unsigned a_bytes = rem_elems * 4 > 0xFFFFF000ll ? 0xFFFFF000u : (unsigned)(rem_elems * 4);
    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_b) + img0 * img_elems, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w_b), 0, (unsigned)((int64_t)p.Cout * p.Kpad * 4), 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFFFu;
    const int64_t img_elems2 = (KMODE == 6 || KMODE == 8) ? (int64_t)p.H2 * p.W2 * p.in2_cstride : 0;
    const int64_t rem_elems2 = ((int64_t)p.B - img0) * img_elems2;
    const unsigned a2_bytes = rem_elems2 * 4 > 0xFFFFF000ll ? 0xFFFFF000u : (unsigned)(rem_elems2 * 4);
    const __amdgpu_buffer_rsrc_t rsrc_a2 = (KMODE == 6 || KMODE == 8)
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in2) + img0 * img_elems2, 0, a2_bytes, 0x00020000) : rsrc_a;

    // DMA lane mapping: wave w, instruction `it` covers rows (it*4 + w)*16 .. +15; lane l -> row + (l >> 2), slot l & 3
    const int lrow = wave * 16 + (lane >> 2);            // + RPR * it
    const int kq = (lane & 3) ^ ((lrow >> 2) & 3);       // k-quad stored in this lane's slot (same for every it)
    int a_hi0[AIT], a_wi0[AIT], a_pix[AIT], a_pix2[AIT];
    bool a_ok[AIT];
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
        const int m = m_base + lrow + it * RPR;
        a_ok[it] = m < p.M;
        const int mm_ = a_ok[it] ? m : m_base;
        const int b = fastdiv(mm_, p.div_hw_mul, p.div_hw_sh), rem = mm_ - b * hw_out;
        const int ho = fastdiv(rem, p.div_wo_mul, p.div_wo_sh), wo = rem - ho * p.Wo;
        a_hi0[it] = ho * p.stride - p.pad;
        a_wi0[it] = wo * p.stride - p.pad;
        // element offset of tap (0,0), channel 0, relative to the descriptor base (may be negative: padding)
        a_pix[it] = (b - img0) * (int)img_elems + (a_hi0[it] * p.W + a_wi0[it]) * p.in_cstride + p.in_coff;
        a_pix2[it] = (KMODE == 6 || KMODE == 8) ? (b - img0) * (int)img_elems2 + (ho * p.stride2 * p.W2 + wo * p.stride2) * p.in2_cstride + p.in2_coff - p.Cin
                                : 0;   // - Cin: the second source is indexed with the running k
    }
    unsigned vb[BIT];
#pragma unroll
    for (int it = 0; it < BIT; ++it) {
        const int n = n_base + lrow + it * RPR;
        vb[it] = n < p.Cout ? (unsigned)(n * p.Kpad + kq * 4) * 4u : OOB;
    }
    // X3 == 2: one DMA instruction covers 32 rows x 32 B of ONE plane; wave w fetches rows [32 w, 32 w + 32) of each of the three planes
    unsigned vb3[3] = {OOB, OOB, OOB};
    __amdgpu_buffer_rsrc_t rsrc_b3 = rsrc_b;
    if constexpr (X3 == 2) {
        static_assert(BN == 32 * NW, "one 32-row piece per wave and plane");
        const int n = n_base + wave * 32 + (lane >> 1);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            vb3[pl] = n < p.Cout ? (unsigned)(((int64_t)pl * p.w3_plane + (int64_t)n * p.Kpad + (lane & 1) * 8) * 2) : OOB;
        rsrc_b3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w3) + (int64_t)bz * p.w_bstride, 0,
                                                    (unsigned)((2 * p.w3_plane + (int64_t)p.Cout * p.Kpad) * 2), 0x00020000);
    }
    // tap state of this lane's k-quad, advanced by one chunk (16) per iteration.  Two K orders:
    //   korder 0:  k = (r*kw + s)*Cin + c              (any Cin % 4 == 0)
    //   korder 1:  k = ((c/16 * kh + r)*kw + s)*16 + c%16   (Cin % 16 == 0): all taps of a 16-channel slice are
    //              consecutive chunks, so the ~kh*kw-fold re-use of every input pixel happens within a few
    //              chunks and is served by L1/L2 instead of the fabric (measured: conv4_x 3x3 fetched 8.5x its
    //              input with korder 0).
    int tk = kq * 4, tr, ts, tc;
    const int rgq = TMODE == 5 ? (p.kw * 3 + 3) / 4 : 1;   // k-quads per kernel row (KMODE 5)
    if (TMODE == 5) {
        tr = kq / rgq;
        ts = kq - tr * rgq;      // quad inside the row group
        tc = 0;
    } else if (TMODE != 1) {
        const int rs = tk / p.Cin;
        tc = tk - rs * p.Cin;
        tr = rs / p.kw;
        ts = rs - tr * p.kw;
    } else {
        tc = tk;  // chunk 0 = slice 0, tap (0,0)
        tr = 0;
        ts = 0;
    }
    unsigned va[AIT];
    auto tap_offsets = [&]() {
        const bool kok = tk < p.K;
        if (TMODE == 5) {
            const int tapoff = tr * p.W * 3 + ts * 4;
#pragma unroll
            for (int it = 0; it < AIT; ++it) va[it] = (kok && a_ok[it]) ? (unsigned)(a_pix[it] + tapoff) * 4u : OOB;
            return;
        }
        if (KMODE == 3) {
#pragma unroll
            for (int it = 0; it < AIT; ++it) va[it] = (kok && a_ok[it]) ? (unsigned)(a_pix[it] + tk) * 4u : OOB;
            return;
        }
        if (KMODE == 6) {
            const bool second = tk >= p.Cin;
#pragma unroll
            for (int it = 0; it < AIT; ++it) va[it] = (kok && a_ok[it]) ? (unsigned)((second ? a_pix2[it] : a_pix[it]) + tk) * 4u : OOB;
            return;
        }
        const int tapoff = (tr * p.W + ts) * p.in_cstride + tc;
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            const int hi = a_hi0[it] + tr, wi = a_wi0[it] + ts;
            const bool ok = kok && a_ok[it] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            va[it] = ok ? (unsigned)(a_pix[it] + tapoff) * 4u : OOB;
        }
    };
    auto tap_advance = [&]() {
        tk += CBK;
        if (TMODE == 0) {
            tc += CBK;
            while (tc >= p.Cin) {
                tc -= p.Cin;
                if (++ts == p.kw) { ts = 0; ++tr; }
            }
        } else if (TMODE == 2) {
            tc += CBK;
            if (tc >= p.Cin) {
                tc -= p.Cin;
                if (++ts == p.kw) { ts = 0; ++tr; }
            }
        } else if (TMODE == 1) {
            if (++ts == p.kw) {
                ts = 0;
                if (++tr == p.kh) { tr = 0; tc += CBK; }
            }
        } else if (KMODE == 4) {
            ts += CBK / 4;
            if (ts >= p.kw) { ts -= p.kw; ++tr; }
        } else if (TMODE == 5) {
            ts += CBK / 4;
            while (ts >= rgq) { ts -= rgq; ++tr; }
        }
#pragma unroll
        for (int it = 0; it < BIT; ++it) vb[it] = vb[it] == OOB ? OOB : vb[it] + CBK * 4u;
        if constexpr (X3 == 2) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) vb3[pl] = vb3[pl] == OOB ? OOB : vb3[pl] + CBK * 2u;
        }
    };

    // One chunk = AIT + BIT direct-to-LDS loads per wave, issued from inline asm: hipcc models an LDS-DMA builtin
    // as an LDS store and puts `s_waitcnt vmcnt(0)` in front of the next ds_read (observed in the .s), which
    // would expose the full memory latency every chunk.  Hidden in asm, the loads stay in flight across the
    // fragment reads, the MFMAs and the barrier; completion is tracked by hand with counted vmcnt waits (the
    // loop issues no other VMEM instruction).  M0 (LDS base of the 1 KB slab) is written in the same statement.
    constexpr int NL = AIT + (X3 == 2 ? 3 : BIT);  // loads per wave per chunk
    const unsigned lds_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)As;
    const unsigned lds_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)Bs;
    auto dma1 = [&](const __amdgpu_buffer_rsrc_t& rsrc, unsigned voff, unsigned lds_byte) {
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_byte);
        unsigned keep;  // M0 is compiler-reserved: save it, point it at the slab, restore it, all in one statement
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(m0v) : "memory");
    };
    int dchunk = 0;                       // chunk the next dma() fetches (KMODE 6: which source)
    const int nk1 = p.Cin / CBK;
    auto dma = [&](int buf) {
        if (KMODE == 6 && dchunk >= nk1) {
#pragma unroll
            for (int it = 0; it < AIT; ++it) dma1(rsrc_a2, va[it], lds_a + ((buf * BM + (it * NW + wave) * 16) * CLD) * 4);
        } else {
#pragma unroll
            for (int it = 0; it < AIT; ++it) dma1(rsrc_a, va[it], lds_a + ((buf * BM + (it * NW + wave) * 16) * CLD) * 4);
        }
        ++dchunk;
        if constexpr (X3 == 2) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) dma1(rsrc_b3, vb3[pl], lds_b + (((buf * 3 + pl) * BN + wave * 32) * 32));
        } else {
#pragma unroll
        for (int it = 0; it < BIT; ++it) dma1(rsrc_b, vb[it], lds_b + ((buf * BN + (it * NW + wave) * 16) * CLD) * 4);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    const int nk = p.Kpad / CBK;
    if constexpr (KMODE == 7 || KMODE == 8) {
        static_assert(!ABL && !X3, "the scheduled 1x1 loop has no measurement / bf16x3 form");
        // ---- loop-constant lane offsets (bytes; OOB = out of range for the hardware check, which looks at the vector offset only)
        unsigned vac[AIT], vac2[AIT], vbc[BIT];
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            vac[it] = a_ok[it] ? (unsigned)(a_pix[it] + kq * 4) * 4u : OOB;
            vac2[it] = (KMODE == 8 && a_ok[it]) ? (unsigned)(a_pix2[it] + p.Cin + kq * 4) * 4u : OOB;   // a_pix2 carries - Cin for the running k of mode 6
        }
#pragma unroll
        for (int it = 0; it < BIT; ++it) vbc[it] = vb[it];
        // this wave's first piece in ring slot 0 of each operand
        const unsigned wa0 = __builtin_amdgcn_readfirstlane(lds_a + (unsigned)(wave * 16 * CLD * 4));
        const unsigned wb0 = __builtin_amdgcn_readfirstlane(lds_b + (unsigned)(wave * 16 * CLD * 4));
        auto piece = [&](const __amdgpu_buffer_rsrc_t& rsrc, unsigned voff, unsigned soff, unsigned wbase, auto off_tag) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %4, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(wbase), "n"(decltype(off_tag)::value) : "memory", "scc");
        };
        int dch = 0;                                       // chunk the next DMA round fetches
        auto dma_f = [&](auto dbuf_tag) {
            constexpr int DBUF = decltype(dbuf_tag)::value;
            // past the last chunk the scalar offset would leave the tensors (the range check does not see it): zero records instead
            const unsigned live = dch < nk ? 0xFFFFFFFFu : 0u;
            const bool second = KMODE == 8 && dch >= nk1;
            const unsigned so_b = __builtin_amdgcn_readfirstlane((unsigned)dch * (CBK * 4u));
            const unsigned so_a = __builtin_amdgcn_readfirstlane(second ? (unsigned)(dch - nk1) * (CBK * 4u) : (unsigned)dch * (CBK * 4u));
            const __amdgpu_buffer_rsrc_t ra = second
                ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in2) + img0 * img_elems2, 0, a2_bytes & live, 0x00020000)
                : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_b) + img0 * img_elems, 0, a_bytes & live, 0x00020000);
            const __amdgpu_buffer_rsrc_t rb =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w_b), 0, (unsigned)((int64_t)p.Cout * p.Kpad * 4) & live, 0x00020000);
            static_for<AIT>([&](auto it_tag) {
                constexpr int IT = decltype(it_tag)::value;
                piece(ra, (KMODE == 8 && second) ? vac2[IT] : vac[IT], so_a, wa0, std::integral_constant<int, (DBUF * BM + IT * RPR) * CLD * 4>());
            });
            static_for<BIT>([&](auto it_tag) {
                constexpr int IT = decltype(it_tag)::value;
                piece(rb, vbc[IT], so_b, wb0, std::integral_constant<int, (DBUF * BN + IT * RPR) * CLD * 4>());
            });
            ++dch;
        };
        // fragment addresses in ring slot 0 (floats); slot s adds s * BM (BN) * CLD as an instruction offset
        int fao[TM][2], fbo[TN][2];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = wm * WM + i * 32 + lr, g = (r >> 2) & 3;
            fao[i][0] = r * CLD + (((2 * lh) ^ g) << 2);
            fao[i][1] = r * CLD + (((2 * lh + 1) ^ g) << 2);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int r = wn * WN + j * 32 + lr, g = (r >> 2) & 3;
            fbo[j][0] = r * CLD + (((2 * lh) ^ g) << 2);
            fbo[j][1] = r * CLD + (((2 * lh + 1) ^ g) << 2);
        }
        auto step = [&](auto buf_tag) {
            constexpr int BUF = decltype(buf_tag)::value;
            dma_f(std::integral_constant<int, (BUF + 2) % NBUF>());
            float4 qa[TM][2], qb[TN][2];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                qa[i][0] = *reinterpret_cast<const float4*>(As + BUF * BM * CLD + fao[i][0]);
                qa[i][1] = *reinterpret_cast<const float4*>(As + BUF * BM * CLD + fao[i][1]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                qb[j][0] = *reinterpret_cast<const float4*>(Bs + BUF * BN * CLD + fbo[j][0]);
                qb[j][1] = *reinterpret_cast<const float4*>(Bs + BUF * BN * CLD + fbo[j][1]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float a = kk == 0 ? qa[i][h].x : kk == 1 ? qa[i][h].y : kk == 2 ? qa[i][h].z : qa[i][h].w;
                            const float b = kk == 0 ? qb[j][h].x : kk == 1 ? qb[j][h].y : kk == 2 ? qb[j][h].z : qb[j][h].w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
                        }
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NL) : "memory");
        };
        dma_f(std::integral_constant<int, 0>());
        dma_f(std::integral_constant<int, 1>());
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NL) : "memory");   // chunk 0 landed (chunk 1 still in flight)
        int kc = 0;
        for (; kc + NBUF <= nk; kc += NBUF) {
            step(std::integral_constant<int, 0>());
            step(std::integral_constant<int, 1>());
            step(std::integral_constant<int, 2>());
        }
        if (nk - kc >= 1) step(std::integral_constant<int, 0>());
        if (nk - kc == 2) step(std::integral_constant<int, 1>());
    } else if constexpr (KMODE == 9 || KMODE == 10) {
        static_assert(!ABL && !X3, "the unrolled loops have no measurement / bf16x3 form");
        constexpr int NK = KMODE == 9 ? 11 : 14;             // 7 x 24 = 168 -> 176 floats; 9 x 24 = 216 -> 224 (conv_forward checks Kpad)
        unsigned vat[KMODE == 9 ? 2 : NK][AIT], vb0[BIT];
#pragma unroll
        for (int it = 0; it < BIT; ++it) vb0[it] = vb[it];   // (tap_advance also walks the weight offsets: the loop uses the scalar offset instead)
        if constexpr (KMODE == 9) {
            // The stem's tap walk in closed form.  Quad g = 4 c + kq of the 6-quad kernel rows: row g / 6, quad g % 6.  4 c mod 6 is 0, 4, 2 for
            // c = 0, 1, 2 (mod 3): only in chunks c = 1 (mod 3) do the lanes with kq >= 2 sit one kernel row below the others.  So a lane needs
            // TWO offsets per piece -- vat[0] (its pixel + kq quads) and vat[1] (the same + one image row - 6 quads for kq >= 2) -- and the chunk
            // part (row0(c) image rows + the first quad of the chunk) is wave-uniform and rides in the buffer instruction's scalar offset: no
            // per-chunk vector instruction, 4 instead of 22 offset registers.  (Rows past M read row m_base's pixels -- their results are never
            // stored --, quads past K = 168 read the next window row and meet zero weights; the first version of mode 9 walked the taps like
            // mode 5 does: ~220 vector instructions per tile for an 11-chunk main loop.)
            const int rowq = p.W * 3;                        // floats per image row
#pragma unroll
            for (int it = 0; it < AIT; ++it) {
                vat[0][it] = (unsigned)(a_pix[it] + kq * 4) * 4u;
                vat[1][it] = (unsigned)(a_pix[it] + kq * 4 + (kq >= 2 ? rowq - 24 : 0)) * 4u;
                asm volatile("" : "+v"(vat[0][it]), "+v"(vat[1][it]));
            }
        } else {
#pragma unroll
        for (int c = 0; c < NK; ++c) {                       // the base mode's own tap walk, once: same offsets, out-of-range taps included
            tap_offsets();
#pragma unroll
            for (int it = 0; it < AIT; ++it) {
                vat[c][it] = va[it];
                asm volatile("" : "+v"(vat[c][it]));         // opaque: hipcc would otherwise sink the arithmetic back into the loop
            }
            tap_advance();
        }
        }
        const unsigned wa0 = __builtin_amdgcn_readfirstlane(lds_a + (unsigned)(wave * 16 * CLD * 4));
        const unsigned wb0 = __builtin_amdgcn_readfirstlane(lds_b + (unsigned)(wave * 16 * CLD * 4));
        auto piece9 = [&](const __amdgpu_buffer_rsrc_t& rsrc, unsigned voff, unsigned wbase, auto lds_tag, unsigned soff) {
            unsigned keep;
            // (the k offset of the weight rows rides in the SCALAR offset: the instruction's offset field would also move the LDS address)
            asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %5 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(wbase), "n"(decltype(lds_tag)::value), "s"(soff) : "memory", "scc");
        };
        auto dma9 = [&](auto c_tag) {
            constexpr int C_ = decltype(c_tag)::value, SLOT = C_ % NBUF;
            // mode 9: the chunk's wave-uniform part -- kernel row (4 C) / 6 and first quad (4 C) % 6 of the chunk -- as a scalar byte offset
            const unsigned so_a = KMODE == 9 ? (unsigned)(((4 * C_) / 6) * (p.W * 3) + ((4 * C_) % 6) * 4) * 4u : 0u;
            static_for<AIT>([&](auto it_tag) {
                constexpr int IT = decltype(it_tag)::value;
                piece9(rsrc_a, vat[KMODE == 9 ? (C_ % 3 == 1 ? 1 : 0) : C_][IT], wa0, std::integral_constant<int, (SLOT * BM + IT * RPR) * CLD * 4>(), so_a);
            });
            static_for<BIT>([&](auto it_tag) {
                constexpr int IT = decltype(it_tag)::value;
                piece9(rsrc_b, vb0[IT], wb0, std::integral_constant<int, (SLOT * BN + IT * RPR) * CLD * 4>(), (unsigned)(C_ * CBK * 4));
            });
        };
        int fao[TM][2], fbo[TN][2];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = wm * WM + i * 32 + lr, g = (r >> 2) & 3;
            fao[i][0] = r * CLD + (((2 * lh) ^ g) << 2);
            fao[i][1] = r * CLD + (((2 * lh + 1) ^ g) << 2);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int r = wn * WN + j * 32 + lr, g = (r >> 2) & 3;
            fbo[j][0] = r * CLD + (((2 * lh) ^ g) << 2);
            fbo[j][1] = r * CLD + (((2 * lh + 1) ^ g) << 2);
        }
        dma9(std::integral_constant<int, 0>());
        dma9(std::integral_constant<int, 1>());
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NL) : "memory");   // chunk 0 landed (chunk 1 still in flight)
        static_for<NK>([&](auto kc_tag) {
            constexpr int KC = decltype(kc_tag)::value, BUF = KC % NBUF;
            if constexpr (KC + 2 < NK) dma9(std::integral_constant<int, KC + 2>());
            float4 qa[TM][2], qb[TN][2];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                qa[i][0] = *reinterpret_cast<const float4*>(As + BUF * BM * CLD + fao[i][0]);
                qa[i][1] = *reinterpret_cast<const float4*>(As + BUF * BM * CLD + fao[i][1]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                qb[j][0] = *reinterpret_cast<const float4*>(Bs + BUF * BN * CLD + fbo[j][0]);
                qb[j][1] = *reinterpret_cast<const float4*>(Bs + BUF * BN * CLD + fbo[j][1]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float a = kk == 0 ? qa[i][h].x : kk == 1 ? qa[i][h].y : kk == 2 ? qa[i][h].z : qa[i][h].w;
                            const float b = kk == 0 ? qb[j][h].x : kk == 1 ? qb[j][h].y : kk == 2 ? qb[j][h].z : qb[j][h].w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
                        }
            // chunk KC + 1 landed (chunk KC + 2, when there is one, may still be in flight); the last chunk is followed by the drain below
            if constexpr (KC + 1 < NK)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(KC + 2 < NK ? NL : 0) : "memory");
        });
    } else if constexpr (KMODE == 11) {
        static_assert(!ABL && !X3, "the unrolled loops have no measurement / bf16x3 form");
        // nine taps of slice 0 from mode 1's tap walk (tc = this lane's quad inside the slice; border taps and rows past M: out of range)
        unsigned vt[9][AIT], vb0[BIT];
#pragma unroll
        for (int it = 0; it < BIT; ++it) vb0[it] = vb[it];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            tap_offsets();
#pragma unroll
            for (int it = 0; it < AIT; ++it) {
                vt[t][it] = va[it];
                asm volatile("" : "+v"(vt[t][it]));
            }
            tap_advance();
        }
        const int nsl = nk / 9;                              // 16-channel slices
        const unsigned wa0 = __builtin_amdgcn_readfirstlane(lds_a + (unsigned)(wave * 16 * CLD * 4));
        const unsigned wb0 = __builtin_amdgcn_readfirstlane(lds_b + (unsigned)(wave * 16 * CLD * 4));
        auto piece11 = [&](const __amdgpu_buffer_rsrc_t& rsrc, unsigned voff, unsigned soff, unsigned wbase, auto lds_tag) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %4, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(wbase), "n"(decltype(lds_tag)::value) : "memory", "scc");
        };
        // DMA of tap T of slice sl (chunk 9 sl + T) into ring slot T % 3 (9 is a multiple of the ring depth)
        auto dma11 = [&](auto t_tag, int sl) {
            constexpr int T = decltype(t_tag)::value, SLOT = T % NBUF;
            // past the last slice the scalar offset would leave the tensors (the range check does not see it): zero records instead
            const unsigned live = sl < nsl ? 0xFFFFFFFFu : 0u;
            const unsigned so_a = __builtin_amdgcn_readfirstlane((unsigned)sl * (CBK * 4u));
            const unsigned so_b = __builtin_amdgcn_readfirstlane((unsigned)(sl * 9 + T) * (CBK * 4u));
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_b) + img0 * img_elems, 0, a_bytes & live, 0x00020000);
            const __amdgpu_buffer_rsrc_t rb =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w_b), 0, (unsigned)((int64_t)p.Cout * p.Kpad * 4) & live, 0x00020000);
            static_for<AIT>([&](auto it_tag) {
                constexpr int IT = decltype(it_tag)::value;
                piece11(ra, vt[T][IT], so_a, wa0, std::integral_constant<int, (SLOT * BM + IT * RPR) * CLD * 4>());
            });
            static_for<BIT>([&](auto it_tag) {
                constexpr int IT = decltype(it_tag)::value;
                piece11(rb, vb0[IT], so_b, wb0, std::integral_constant<int, (SLOT * BN + IT * RPR) * CLD * 4>());
            });
        };
        int fao[TM][2], fbo[TN][2];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = wm * WM + i * 32 + lr, g = (r >> 2) & 3;
            fao[i][0] = r * CLD + (((2 * lh) ^ g) << 2);
            fao[i][1] = r * CLD + (((2 * lh + 1) ^ g) << 2);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int r = wn * WN + j * 32 + lr, g = (r >> 2) & 3;
            fbo[j][0] = r * CLD + (((2 * lh) ^ g) << 2);
            fbo[j][1] = r * CLD + (((2 * lh + 1) ^ g) << 2);
        }
        dma11(std::integral_constant<int, 0>(), 0);
        dma11(std::integral_constant<int, 1>(), 0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NL) : "memory");   // chunk 0 landed (chunk 1 still in flight)
        for (int sl = 0; sl < nsl; ++sl) {
            static_for<9>([&](auto t_tag) {
                constexpr int T = decltype(t_tag)::value, BUF = T % NBUF;
                dma11(std::integral_constant<int, (T + 2) % 9>(), T + 2 >= 9 ? sl + 1 : sl);
                float4 qa[TM][2], qb[TN][2];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    qa[i][0] = *reinterpret_cast<const float4*>(As + BUF * BM * CLD + fao[i][0]);
                    qa[i][1] = *reinterpret_cast<const float4*>(As + BUF * BM * CLD + fao[i][1]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    qb[j][0] = *reinterpret_cast<const float4*>(Bs + BUF * BN * CLD + fbo[j][0]);
                    qb[j][1] = *reinterpret_cast<const float4*>(Bs + BUF * BN * CLD + fbo[j][1]);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) {
                                const float a = kk == 0 ? qa[i][h].x : kk == 1 ? qa[i][h].y : kk == 2 ? qa[i][h].z : qa[i][h].w;
                                const float b = kk == 0 ? qb[j][h].x : kk == 1 ? qb[j][h].y : kk == 2 ? qb[j][h].z : qb[j][h].w;
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
                            }
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NL) : "memory");
            });
        }
    } else {
    tap_offsets();
    dma(0);
    tap_advance();
    tap_offsets();
    dma(1);
    tap_advance();
    tap_offsets();
    // fragment read offsets (floats): row r, k-quads 2*lh and 2*lh+1 live in slots (2*lh)^g and (2*lh+1)^g, g = (r>>2)&3
    int fa_off[TM][2], fb_off[TN][2];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * WM + i * 32 + lr, g = (r >> 2) & 3;
        fa_off[i][0] = r * CLD + (((2 * lh) ^ g) << 2);
        fa_off[i][1] = r * CLD + (((2 * lh + 1) ^ g) << 2);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = wn * WN + j * 32 + lr, g = (r >> 2) & 3;
        fb_off[j][0] = r * CLD + (((2 * lh) ^ g) << 2);
        fb_off[j][1] = r * CLD + (((2 * lh + 1) ^ g) << 2);
    }
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NL) : "memory");   // chunk 0 landed (chunk 1 still in flight)
    float4 fa[TM][2], fb[TN][2];
    if (ABL) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i][0] = fa[i][1] = float4{1.f, 2.f, 3.f, 4.f};
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j][0] = fb[j][1] = float4{1.f, 2.f, 3.f, 4.f};
    }
    // One barrier per 16-deep chunk.  Iteration kc: the direct-to-LDS loads of chunk kc+2 are issued first
    // (offsets computed one iteration earlier, no VALU in front of them), then the fragments of chunk kc are
    // read, offsets of chunk kc+3 are computed in the shadow of the 32 MFMAs, and the iteration ends with
    // `s_waitcnt vmcnt(NL) lgkmcnt(0)` (chunk kc+1 landed, chunk kc+2 may still be in flight; THIS wave's fragment
    // reads of chunk kc have completed) + s_barrier.  The lgkmcnt(0) is load-bearing: the asm's "memory" clobber does
    // not order register-only instructions, and hipcc sinks MFMAs -- together with the lgkmcnt wait of the reads that
    // feed them -- below the barrier; without it another wave's next DMA could overwrite a ring slot whose ds_reads
    // were still pending (seen as rare corrupted 32x32 tiles on the short-K 64x64 configuration).  Loads past the
    // last chunk are harmless (range check -> zeros into a ring slot nobody reads).
    int buf = 0, buf2 = 2;
    for (int kc = 0; kc < nk; ++kc) {
        if (!ABL || !(p.ablate & (1 | 128))) dma(buf2);
        if (!ABL || !(p.ablate & 8)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                fa[i][0] = *reinterpret_cast<const float4*>(As + buf * BM * CLD + fa_off[i][0]);
                fa[i][1] = *reinterpret_cast<const float4*>(As + buf * BM * CLD + fa_off[i][1]);
            }
            if constexpr (X3 != 2) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                fb[j][0] = *reinterpret_cast<const float4*>(Bs + buf * BN * CLD + fb_off[j][0]);
                fb[j][1] = *reinterpret_cast<const float4*>(Bs + buf * BN * CLD + fb_off[j][1]);
            }
            }
        }
        if (!ABL || !(p.ablate & (1 | 256))) {
            tap_advance();
            tap_offsets();
        }
        if (ABL && (p.ablate & 64)) __builtin_amdgcn_s_setprio(1);
        if constexpr (X3 == 2) {
            // weights pre-split (three bf16 planes in LDS, one 16-byte read per plane and column tile: lane (lr, lh) takes k = 8 lh .. 8 lh + 7
            // of row wn * WN + j * 32 + lr); the activations are split here as in the X3 == 1 form -- 2 instead of 2 + TN split blocks per chunk.
            // The planes were made with split_bf16x3 itself (bf16x3_split_weights below): same values, same six products in the same order as the
            // in-loop form => bit-identical to it.
            u32x4 Ah[TM], Am[TM], Al[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) split_bf16x3(fa[i][0], fa[i][1], Ah[i], Am[i], Al[i]);
            auto mm16 = [](const u32x4& a, const u32x4& b, f32x16 c) {
                return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
            };
            const char* bs3 = reinterpret_cast<const char*>(Bs) + (buf * 3 * BN) * 32 + lh * 16;
#pragma unroll
            for (int j0 = 0; j0 < TN; j0 += 2) {
                constexpr int JN = TN < 2 ? TN : 2;
                u32x4 Bh[JN], Bm[JN], Bl[JN];
#pragma unroll
                for (int jj = 0; jj < JN; ++jj) {
                    const int r = wn * WN + (j0 + jj) * 32 + lr;
                    Bh[jj] = *reinterpret_cast<const u32x4*>(bs3 + (0 * BN + r) * 32);
                    Bm[jj] = *reinterpret_cast<const u32x4*>(bs3 + (1 * BN + r) * 32);
                    Bl[jj] = *reinterpret_cast<const u32x4*>(bs3 + (2 * BN + r) * 32);
                }
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int jj = 0; jj < JN; ++jj) {
                            const u32x4& a = t == 0 ? Al[i] : t == 1 ? Ah[i] : t == 2 ? Am[i] : t == 3 ? Am[i] : Ah[i];
                            const u32x4& b = t == 0 ? Bh[jj] : t == 1 ? Bl[jj] : t == 2 ? Bm[jj] : t == 3 ? Bh[jj] : t == 4 ? Bm[jj] : Bh[jj];
                            acc[i][j0 + jj] = mm16(a, b, acc[i][j0 + jj]);
                        }
            }
        } else if constexpr (X3 == 1) {
            // a lane's two float4 are k = 8 lh .. 8 lh + 7 of its row: exactly the 8 bf16 the 32x32x16 MFMA takes from it
            u32x4 Ah[TM], Am[TM], Al[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) split_bf16x3(fa[i][0], fa[i][1], Ah[i], Am[i], Al[i]);
            auto mm16 = [](const u32x4& a, const u32x4& b, f32x16 c) {
                return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
            };
            // the A fragments stay split; the B fragments are split one column tile at a time (a 64 x 128 wave tile would not hold all of
            // them).  Smallest terms first into the running fp32 accumulators; the TM sub-tiles of a column are independent chains.
            // (two column tiles at a time: one long split block, then one long MFMA block; a block per column tile measured slower, and so
            //  did a software-pipelined loop that places the split of chunk kc between the MFMAs of chunk kc - 1 -- 160-176 against
            //  182-204 TFLOP/s-equivalent, profiles/r05_bf16x3_variants.txt.  tools/probes/valu_mfma_overlap.hip: on a gfx950 SIMD a block
            //  of these split instructions and a block of bf16 MFMAs take the SUM of their times whichever waves they belong to, so the
            //  mode is bound by VALU issue + MFMA time, ~1.4 x the fp32-MFMA rate instead of the 2.7 x six bf16 products could give)
#pragma unroll
            for (int j0 = 0; j0 < TN; j0 += 2) {
                constexpr int JN = TN < 2 ? TN : 2;
                u32x4 Bh[JN], Bm[JN], Bl[JN];
#pragma unroll
                for (int jj = 0; jj < JN; ++jj) split_bf16x3(fb[j0 + jj][0], fb[j0 + jj][1], Bh[jj], Bm[jj], Bl[jj]);
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int jj = 0; jj < JN; ++jj) {
                            const u32x4& a = t == 0 ? Al[i] : t == 1 ? Ah[i] : t == 2 ? Am[i] : t == 3 ? Am[i] : Ah[i];
                            const u32x4& b = t == 0 ? Bh[jj] : t == 1 ? Bl[jj] : t == 2 ? Bm[jj] : t == 3 ? Bh[jj] : t == 4 ? Bm[jj] : Bh[jj];
                            acc[i][j0 + jj] = mm16(a, b, acc[i][j0 + jj]);
                        }
            }
        } else
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float a = kk == 0 ? fa[i][h].x : kk == 1 ? fa[i][h].y : kk == 2 ? fa[i][h].z : fa[i][h].w;
                        const float b = kk == 0 ? fb[j][h].x : kk == 1 ? fb[j][h].y : kk == 2 ? fb[j][h].z : fb[j][h].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
                    }
        if (ABL && (p.ablate & 64)) __builtin_amdgcn_s_setprio(0);
        if (!ABL || !(p.ablate & 4))
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NL) : "memory");
        buf = buf == NBUF - 1 ? 0 : buf + 1;
        buf2 = buf2 == NBUF - 1 ? 0 : buf2 + 1;
    }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // drain the over-issued loads before LDS is re-used

    // ---- hpool epilogue (round 5): the stem's output is only ever read by MaxPool2d(3, 2, pad 0) -- 6.6 GB written and 10 GB read back
    // per 2 048 frames.  The pool's HORIZONTAL half runs here: the whole 128 x 64 tile goes through LDS as relu(acc + bias), then a thread
    // takes the maximum over pixels (m, m + 1, m + 2) of one channel quad for the even m of the tile (m + 2 only while it is in the same
    // image row: the ceil-mode window at the right border has two columns) and writes pooled pixel m / 2 -- [B, Ho, Wo / 2, 64], half the
    // bytes, and the kernel that finishes the pool (pool_reduce.hip, hp = 1) reads three rows per pixel instead of nine pixels.
    // max is exact and order-free: the pooled tensor is bit-identical to pooling the full stem output.
    if constexpr (TMODE == 5 && BM == 128 && BN == 64) {
        if (hpool) {
            static_assert(TN == 1 && WN == 32, "one 32-channel accumulator column per wave");
            constexpr int TLD = BN + 4;                          // row stride of the staged tile (floats): 16-byte aligned rows, odd in 16-byte units
            static_assert(BM * TLD <= OPER_FLOATS, "the staged tile fits the dead operand ring");
            float* T = lds;
            const float bias = p.bias ? p.bias[wn * WN + lr] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = acc[i][0][e] + bias;
                    T[(wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * TLD + wn * WN + lr] = p.relu ? fmaxf(v, 0.f) : v;
                }
            __syncthreads();
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int idx = tid; idx < (BM / 2 - 1) * (BN / 4); idx += NW * 64) {
                const int u = idx / (BN / 4), q = idx - u * (BN / 4);
                const int m = m_base + 2 * u;
                if (m >= p.M) continue;                          // (M is even: m + 1 < M as well)
                const float4 a = *reinterpret_cast<const float4*>(T + (2 * u) * TLD + 4 * q);
                const float4 b = *reinterpret_cast<const float4*>(T + (2 * u + 1) * TLD + 4 * q);
                const float4 c = *reinterpret_cast<const float4*>(T + (2 * u + 2) * TLD + 4 * q);
                f32x4_t o = {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)};
                if (m - fastdiv(m, p.div_wo_mul, p.div_wo_sh) * p.Wo != p.Wo - 2) o = f32x4_t{fmaxf(o[0], c.x), fmaxf(o[1], c.y), fmaxf(o[2], c.z), fmaxf(o[3], c.w)};
                __builtin_nontemporal_store(o, reinterpret_cast<f32x4_t*>(out_b + (int64_t)(m >> 1) * BN + 4 * q));
            }
            return;
        }
    }

    // ---- fused epilogue.  MFMA C layout: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), i.e. a
    // lane owns a 16-row strip of ONE channel.  Per-lane stores of that layout are 4-byte, row-strided and
    // issue-bound, so each wave transposes its tile through LDS (the operand buffers are dead) and every lane
    // then owns 4 consecutive channels of a row: bias/residual/output all move as 16-byte, 256-byte-per-row
    // coalesced accesses.
    const bool wide = ((p.Cout | p.out_cstride | p.out_coff | p.res_cstride | p.res_coff) & 3) == 0;
    if (wide) {
        constexpr int SLD = WN + 4;                 // staging row stride (floats)
        constexpr int QN = WN / 4;                  // float4 per staged row
        constexpr int RPI = 64 / QN;                // rows covered by one wave-wide float4 access
        float* st = lds + wave * (32 * SLD);        // 32 x WN per wave, re-using the (dead) operand buffers
        const int qc = lane % QN, qr = lane / QN;
        const int n0 = n_base + wn * WN + qc * 4;
        const bool nok = n0 < p.Cout;
        float4 bias4 = {0.f, 0.f, 0.f, 0.f}, ps4 = {1.f, 1.f, 1.f, 1.f}, pt4 = {0.f, 0.f, 0.f, 0.f};
        if (nok && p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + n0);
        if (nok && p.post_scale) {
            ps4 = *reinterpret_cast<const float4*>(p.post_scale + n0);
            pt4 = *reinterpret_cast<const float4*>(p.post_shift + n0);
        }
        // Every wave stages through its OWN 32 x WN region, so no workgroup barrier is needed in here: LDS operations
        // of one wave execute in issue order (its ds_reads see its earlier ds_writes, whichever lane wrote them).  The
        // residual rows of a pass are requested before the pass is staged so their HBM latency overlaps the transpose.
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m0 = m_base + wm * WM + i * 32;
            float4 v[32 / RPI], rsd[32 / RPI];
            if (p.res) {
#pragma unroll
                for (int t = 0; t < 32 / RPI; ++t) {
                    const int m = m0 + t * RPI + qr;
                    const int64_t mo = (nok && m < p.M) ? (int64_t)m * p.res_cstride + p.res_coff + n0 : 0;
                    rsd[t] = *reinterpret_cast<const float4*>(p.res + mo);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    st[((e & 3) + 8 * (e >> 2) + 4 * lh) * SLD + j * 32 + lr] = acc[i][j][e];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int t = 0; t < 32 / RPI; ++t) v[t] = *reinterpret_cast<const float4*>(st + (t * RPI + qr) * SLD + qc * 4);
#pragma unroll
            for (int t = 0; t < 32 / RPI; ++t) {
                const int m = m0 + t * RPI + qr;
                float4 o = v[t];
                o.x += bias4.x; o.y += bias4.y; o.z += bias4.z; o.w += bias4.w;
                if (p.res) { o.x += rsd[t].x; o.y += rsd[t].y; o.z += rsd[t].z; o.w += rsd[t].w; }
                if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                if (p.post_scale) {
                    o.x = o.x * ps4.x + pt4.x; o.y = o.y * ps4.y + pt4.y; o.z = o.z * ps4.z + pt4.z; o.w = o.w * ps4.w + pt4.w;
                }
                if (nok && m < p.M) {
                    typedef float f32x4_t __attribute__((ext_vector_type(4)));
                    const f32x4_t ov = {o.x, o.y, o.z, o.w};
                    __builtin_nontemporal_store(ov, reinterpret_cast<f32x4_t*>(out_b + (int64_t)m * p.out_cstride + p.out_coff + n0));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // pass i+1 re-writes the region these reads came from
        }
        return;
    }
    // narrow fallback (channel counts / strides not multiples of 4, e.g. the 2-wide classifier)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n_base + wn * WN + j * 32 + lr;
        const bool nok = n < p.Cout;
        const float bias = (nok && p.bias) ? p.bias[n] : 0.f;
        const float ps = (nok && p.post_scale) ? p.post_scale[n] : 1.f;
        const float pt = (nok && p.post_shift) ? p.post_shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m_base + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (nok && m < p.M) {
                    float v = acc[i][j][e] + bias;
                    if (p.res) v += p.res[(int64_t)m * p.res_cstride + p.res_coff + n];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.post_scale) v = v * ps + pt;
                    out_b[(int64_t)m * p.out_cstride + p.out_coff + n] = v;
                }
            }
        }
    }
}

// (round 6) weights of a bf16x3 layer split ONCE, with the loop's own split function: out = three bf16 planes [3][n] (h, m, l), n even
__global__ void __launch_bounds__(256) bf16x3_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ o, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;       // (n is a multiple of 16: whole groups of eight)
    const float4 q0 = *reinterpret_cast<const float4*>(w + i), q1 = *reinterpret_cast<const float4*>(w + i + 4);
    u32x4 H, M, L;
    split_bf16x3(q0, q1, H, M, L);
    *reinterpret_cast<u32x4*>(o + i) = H;
    *reinterpret_cast<u32x4*>(o + n + i) = M;
    *reinterpret_cast<u32x4*>(o + 2 * n + i) = L;
}

int bf16x3_split_weights(const float* w, unsigned short* out, int64_t n, hipStream_t stream) {
    if (n <= 0 || n % 16) return MM_ERR_INVALID_ARG;
    hipLaunchKernelGGL(bf16x3_split_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, stream, w, out, n);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

// Per-device caches (relaxed atomics: a race only repeats the query, every thread stores the same value).
constexpr int kMaxDev = 64;
static int cur_dev_slot() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev >= 0 && dev < kMaxDev ? dev : -1;
}

// workgroups of an instantiation that fit one CU (registers, LDS), as the runtime computes it for the current device
template <int BM, int BN, int WGM, int WGN, int KMODE>
static int occ_km() {
    static std::atomic<int> occ[kMaxDev];
    const int d = cur_dev_slot();
    int v = d >= 0 ? occ[d].load(std::memory_order_relaxed) : 0;
    if (!v) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_mfma_kernel<BM, BN, WGM, WGN, KMODE, false>, WGM * WGN * 64, 0) != hipSuccess || n < 1)
            n = 1;
        v = n;
        if (d >= 0) occ[d].store(v, std::memory_order_relaxed);
    }
    return v;
}

static int num_cus() {
    static std::atomic<int> cus[kMaxDev];
    const int d = cur_dev_slot();
    int v = d >= 0 ? cus[d].load(std::memory_order_relaxed) : 0;
    if (!v) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
        if (d >= 0) cus[d].store(v, std::memory_order_relaxed);
    }
    return v;
}

template <int BM, int BN, int WGM, int WGN, int KMODE, bool ABL = false, int X3 = 0>
static int launch_km(ConvParams p, hipStream_t stream) {
    fastdiv_make(p.Ho * p.Wo, p.div_hw_mul, p.div_hw_sh);
    fastdiv_make(p.Wo, p.div_wo_mul, p.div_wo_sh);
    p.tiles_m = (p.M - p.m_off + BM - 1) / BM;
    if (p.hpool) p.tiles_m = (p.M / 2 + (BM / 2 - 1) - 1) / (BM / 2 - 1);     // BM / 2 - 1 pooled pixels per tile (tiles overlap by two rows)
    p.tiles_n = (p.Cout + BN - 1) / BN;
    if (p.batch < 1) p.batch = 1;
    const int64_t blocks = (int64_t)p.tiles_m * p.tiles_n * p.batch;
    if (blocks <= 0 || blocks > 0x7fffffff) return MM_ERR_INVALID_ARG;
    if (prof_enabled()) {
        char tag[64];
        snprintf(tag, sizeof(tag), "M=%d K=%d N=%d k%d s%d t%dx%d b%d%s", p.M - p.m_off, p.K, p.Cout, p.kh, p.stride, BM, BN, p.batch, X3 == 2 ? " x3p x3" : X3 ? " x3" : "");
        prof_before(0, 2.0 * (double)(p.M - p.m_off) * (double)(p.kh * p.kw * p.Cin_real) * (double)p.Cout * (double)p.batch, stream, tag);
    }
    hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WGM, WGN, KMODE, ABL, X3>), dim3((unsigned)blocks), dim3(WGM * WGN * 64), 0, stream, p);
    prof_after(0, stream);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

template <int BM, int BN, int WGM, int WGN>
static int launch_cfg(const ConvParams& p, hipStream_t stream) {
    if (p.korder == 2) {
        // the 7x7 stem (eleven chunks) takes the fully unrolled loop unless the caller asks for the twin (no_sched)
        if constexpr (BN == 64) {
            if (!p.no_sched && p.kh == 7 && p.kw == 7 && p.Kpad == 176 && p.force_tile < 16) return launch_km<BM, BN, WGM, WGN, 9>(p, stream);
        }
        return launch_km<BM, BN, WGM, WGN, 5>(p, stream);
    }
    if constexpr (BN >= 64 && BM != 256) {      // the bf16x3 instantiations exist for the 1x1 forms on the 64x64 / 128x128 tiles (and 128x256 below)
        if (p.x3 && p.in2) return launch_km<BM, BN, WGM, WGN, 6, false, 1>(p, stream);
        if (p.x3 && p.kh == 1 && p.kw == 1 && p.pad == 0) return launch_km<BM, BN, WGM, WGN, 3, false, 1>(p, stream);
    }
    if (p.sched1x1 && p.in2) return launch_km<BM, BN, WGM, WGN, 8>(p, stream);
    if (p.sched1x1) return launch_km<BM, BN, WGM, WGN, 7>(p, stream);
    if (p.in2) return launch_km<BM, BN, WGM, WGN, 6>(p, stream);
    if (p.kh == 1 && p.kw == 1 && p.pad == 0) return launch_km<BM, BN, WGM, WGN, 3>(p, stream);
    if (p.korder == 1) {
        // 3x3 kernels on slice-major k: the loop unrolled by the nine taps (mode 11) unless the caller asks for the twin
        if constexpr (BM == 128 && (BN == 64 || BN == 128) && WGM * WGN == 4) {
            if (!p.no_sched && p.kh == 3 && p.kw == 3 && p.K == p.Kpad && p.batch <= 1 && p.force_tile < 16) return launch_km<BM, BN, WGM, WGN, 11>(p, stream);
        }
        return launch_km<BM, BN, WGM, WGN, 1>(p, stream);
    }
    if (p.Cin >= CBK) {
        // fourteen chunks exactly (3x3 on 24 channels: PhaseNet's first conv): fully unrolled with precomputed tap offsets (mode 10)
        if constexpr (BM == 128 && BN == 64) {
            if (!p.no_sched && p.Kpad == 224 && p.batch <= 1 && p.force_tile < 16) return launch_km<BM, BN, WGM, WGN, 10>(p, stream);
        }
        return launch_km<BM, BN, WGM, WGN, 2>(p, stream);
    }
    if (p.Cin == 4 && p.kw >= 4) return launch_km<BM, BN, WGM, WGN, 4>(p, stream);
    return launch_km<BM, BN, WGM, WGN, 0>(p, stream);
}

int conv_forward(const ConvParams& p0, hipStream_t stream) {
    ConvParams p = p0;
    if (p.korder == 2) {   // packed 3-channel input, zero border in memory (KMODE 5)
        if (p.Cin != 3 || p.in_cstride != 3 || p.in_coff != 0 || p.pad != 0 || p.K != p.kh * ((p.kw * 3 + 3) / 4) * 4 || p.Kpad % CBK ||
            p.K > p.Kpad)
            return MM_ERR_INVALID_ARG;
    } else {
        if (p.Cin % 4 || p.in_cstride % 4 || p.in_coff % 4 || p.Kpad % CBK || p.K > p.Kpad) return MM_ERR_INVALID_ARG;
        if (p.korder != 0 && (p.korder != 1 || p.Cin % CBK)) return MM_ERR_INVALID_ARG;
    }
    // 32-bit byte offsets inside the kernel: the weight matrix, and the few images one block spans (the A descriptor is rebased to
    // the block's first image; the largest block tile is 256 rows -- force_tile 4)
    if ((uint64_t)p.Cout * p.Kpad * 4 >= 0xFFFFF000ull) return MM_ERR_INVALID_ARG;
    {
        const int64_t hw_o = (int64_t)p.Ho * p.Wo;
        const int64_t span = 256 / (hw_o > 0 ? hw_o : 1) + 2;
        if (span * p.H * p.W * p.in_cstride * 4 >= 0x7FFFF000ll) return MM_ERR_INVALID_ARG;
    }
    if (p.in2) {
        if (p.kh != 1 || p.kw != 1 || p.pad != 0 || p.stride != 1 || p.korder != 0 || p.batch > 1 || p.Cin % CBK || p.C2 <= 0 ||
            p.C2 % CBK || p.K != p.Cin + p.C2 || p.Kpad != p.K || p.in2_cstride % 4 || p.in2_coff % 4 || p.stride2 <= 0 ||
            (p.Ho - 1) * p.stride2 >= p.H2 || (p.Wo - 1) * p.stride2 >= p.W2)
            return MM_ERR_INVALID_ARG;
        const int64_t hw_o = (int64_t)p.Ho * p.Wo;
        const int64_t span = 256 / (hw_o > 0 ? hw_o : 1) + 2;
        if (span * p.H2 * p.W2 * p.in2_cstride * 4 >= 0x7FFFF000ll) return MM_ERR_INVALID_ARG;
    }
    if (p.x3 && !(p.kh == 1 && p.kw == 1 && p.pad == 0 && p.korder == 0)) p.x3 = 0;   // bf16x3 exists for the 1x1 forms only
    if (p.hpool) {   // horizontally pooled output: the stem form on the 128x64 tile only (the tile is the whole channel range)
        if (p.korder != 2 || p.Cout != 64 || p.out_cstride != 64 || p.out_coff != 0 || p.Wo % 2 || p.Wo < 4 || p.res || p.post_scale || p.batch > 1 ||
            p.m_off != 0 || p.m_end != 0 || (p.force_tile != 0 && p.force_tile != 2))
            return MM_ERR_INVALID_ARG;
        p.force_tile = 2;
    }
    // 1x1 layers whose K is a whole number of chunks take the scheduled loop (KMODE 7 / 8) unless the caller asks for modes 3 / 6 (no_sched:
    // MM_CONV_SCHED=0 at mm_resnet50_create, the parity twin)
#ifndef MM_SCHED_RES_MINK
#define MM_SCHED_RES_MINK 512   // layers with a residual epilogue and a shorter K keep mode 3 (256 -> 1024 + residual measured 3 % SLOWER on the scheduled loop)
#endif
    p.sched1x1 = !p.no_sched && !p.x3 && !(p.res && p.K < MM_SCHED_RES_MINK) && p.kh == 1 && p.kw == 1 && p.pad == 0 && p.korder == 0 && p.K == p.Kpad && p.force_tile < 16;
    p.M = p.B * p.Ho * p.Wo;
    if (p.m_end > 0 && p.m_end < p.M) p.M = p.m_end;            // (the bulk launch of a tail split ends early)
    if (p.M <= p.m_off) return MM_OK;
    if (p.Cin_real <= 0) p.Cin_real = p.Cin;
    // (round 6) K = 256-class increase layers: the activation panel resident in LDS (conv_panel.hip) -- OPT-IN (use_panel: MM_CONV_PANEL=1),
    // built, bit-identical, measured SLOWER than the engine (profiles/r06_ab_conv_panel.txt).  One workgroup per CU, so a launch is whole
    // rounds of num_cus() panels; the rows of a thin last round go to the engine below instead (same sums in the same order either way).
    if (p.use_panel && p.force_tile == 0 && p.m_off == 0 && p.m_end == 0 && conv_panel_supported(p)) {
        const int pbm = conv_panel_rows(p);
        const int64_t slots = (int64_t)num_cus() * conv_panel_per_cu(p);
        const int64_t tm = (p.M + pbm - 1) / pbm;
        const int64_t full = tm / slots, rest = tm - full * slots;
        if (full >= 1) {
            if (rest == 0 || rest * 2 >= slots) return conv_panel_forward(p, stream);
            ConvParams pb = p, pr = p;
            pb.M = (int)(full * slots * pbm);
            pr.m_off = pb.M;
            pr.use_panel = 0;
            const int64_t t128 = ((p.M - pb.M + 127) / 128) * ((p.Cout + 127) / 128);
            pr.force_tile = t128 * 2 >= (int64_t)num_cus() * 3 ? 1 : 3;
            const int rc = conv_panel_forward(pb, stream);
            return rc != MM_OK ? rc : conv_forward(pr, stream);
        }
    }
    // Tile choice.  Wave tile 64x64 (2x2 MFMA sub-tiles, 4 accumulators) is the efficient shape: 128x128
    // blocks (2x2 waves) when Cout > 64, 256x64 blocks (4x1 waves) for the 64-channel layers; smaller tiles
    // only when the grid would not give every CU (256) a couple of workgroups.
    const int64_t m128 = (p.M + 127) / 128, m256 = (p.M + 255) / 256;
    const int64_t n128 = (p.Cout + 127) / 128, n64 = (p.Cout + 63) / 64;
    int cfg = p.force_tile;
#ifdef MM_MEASURE
    if (cfg >= 16) {  // measurement-only: 128x128 with experiment bits (cfg - 16)
        p.ablate = cfg - 16;
        return launch_km<128, 128, 2, 2, 1, true>(p, stream);  // slice-major 3x3 shapes only
    }
#else
    if (cfg > 5) return MM_ERR_INVALID_ARG;
#endif
    if (cfg == 0) {
        // 1x1 / GEMM shapes whose N is a multiple of 256 (ResNet increase / projection layers, the Winograd GEMMs of
        // conv4_x and conv5_x): one 128x256 workgroup of eight waves covers the full N per 256 columns, so every
        // activation row is fetched once instead of twice and the per-tile prologue is amortised over twice the MFMA
        // work.  Same-box A/B on the whole path: +1.5 % (114.9 vs 116.6 ms per step).
        const int64_t nb5 = p.batch > 1 ? p.batch : 1;
        if (p.kh == 1 && p.kw == 1 && p.pad == 0 && p.Cout % 256 == 0 && m128 * (p.Cout / 256) * nb5 >= 512) cfg = 5;
    }
    if (cfg == 0) {
        const int64_t nb = p.batch > 1 ? p.batch : 1;  // a batched launch (Winograd planes) fills the grid nb times over
        if (p.Cout > 64 && m128 * n128 * nb >= 512) cfg = 1;
        else if (m128 * n64 * nb >= 512) cfg = 2;
        else cfg = 3;
    }
    // Tail split (round 3).  A GEMM whose tile count is not a multiple of the resident workgroup slots ends with a partial round:
    // 3 136 tiles of 128x256 on 512 slots = 6.125 rounds, the seventh running 64 workgroups on a chip that holds 512 (12.5 % of the
    // 1024 -> 256 layers' time; 23 % of the 2048 -> 512 layers').  The rows of the full rounds stay on the big tile; the rows of the
    // partial round go to a second launch on a finer tile that spreads them over the whole chip.  Same kernel body, same k order:
    // every output element is the same sum in the same order (bit-identical to the unsplit launch).  Only when the partial round is
    // less than 30 % full: measured per layer (2 048 frames, one stream) 1024 -> 256 x5 8.11 -> 7.80 ms, 2048 -> 512 x2 3.37 -> 3.01,
    // 1536 -> 2048 4.63 -> 4.56; a half-full last round (256 -> 1024, 768 -> 1024, 512 -> 128) is cheaper left on the big tile.  And only
    // for launches of three or more full rounds: the lanes of the default pipeline run a third of the batch each (two rounds of the
    // 1024 -> 256 layer), and there another lane's kernels already fill a tail -- splitting those cost 0.6 ms per step.
    static const int split_on = getenv("MM_TAIL_SPLIT") ? atoi(getenv("MM_TAIL_SPLIT")) : 1;   // measurement knob
    if (split_on && !p.x3 && p.force_tile == 0 && (cfg == 5 || cfg == 1) && p.kh == 1 && p.kw == 1 && p.pad == 0 && p.batch <= 1 && p.m_off == 0) {
        const int bm = 128, bn = cfg == 5 ? 256 : 128;
        const int occ = cfg == 5 ? (p.in2 ? occ_km<128, 256, 2, 4, 6>() : occ_km<128, 256, 2, 4, 3>())
                                 : (p.in2 ? occ_km<128, 128, 2, 2, 6>() : occ_km<128, 128, 2, 2, 3>());
        const int64_t slots = (int64_t)occ * num_cus();
        const int64_t tn = (p.Cout + bn - 1) / bn, tm = (p.M + bm - 1) / bm;
        const int64_t full = tm * tn / slots;                      // whole rounds
        const int64_t rest = tm * tn - full * slots;               // workgroups of the partial round
        const int64_t tm_bulk = full * slots / tn;                 // m-tiles the whole rounds cover
        if (p.m_end == 0 && full >= 3 && rest > 0 && rest * 100 <= slots * 30 && tm_bulk < tm && tm_bulk * bm < p.M) {
            ConvParams pb = p, pr = p;
            pb.m_end = (int)(tm_bulk * bm);
            pb.force_tile = cfg;
            pr.m_off = pb.m_end;
            const int64_t R = p.M - pb.m_end;
            // remainder tile: 128x128 when that still gives the chip enough workgroups, else 64x64
            const int64_t t128 = ((R + 127) / 128) * ((p.Cout + 127) / 128);
            pr.force_tile = t128 * 2 >= (int64_t)num_cus() * 3 ? 1 : 3;
            const int rc = conv_forward(pb, stream);
            return rc != MM_OK ? rc : conv_forward(pr, stream);
        }
    }
    switch (cfg) {
        case 1: return launch_cfg<128, 128, 2, 2>(p, stream);
        case 2: return launch_cfg<128, 64, 2, 2>(p, stream);
        case 3: return launch_cfg<64, 64, 2, 2>(p, stream);
        case 4: return launch_cfg<256, 64, 4, 1>(p, stream);
        case 5:   // 128x256, eight waves: the whole N of a 256-channel 1x1 layer in one workgroup (A read once)
            if (!(p.kh == 1 && p.kw == 1 && p.pad == 0)) return MM_ERR_INVALID_ARG;
#ifndef MM_X3_WIDE
#define MM_X3_WIDE 1   // 1: the bf16x3 form of the 128x256 tile runs on FOUR waves with 64x128 wave tiles (fewer split instructions per MFMA); 0: eight waves, 64x64
#endif
            // (round 6, opt-in MM_X3_PRESPLIT=1) weights pre-split into three bf16 planes (ConvParams::w3): eight waves, the B fragments need no split
            // in the loop.  Measured SLOWER than the four-wave form below that splits both operands in the loop (154-181 vs 183-206 TFLOP/s per layer,
            // profiles/r06_ab_x3_presplit.txt): 96 B per B row and ring slot make the ring 96 KB -- one workgroup per CU -- and the 64x64 wave tiles
            // amortise the remaining A split over two column tiles instead of four
            if (p.x3 && p.w3) return p.in2 ? launch_km<128, 256, 2, 4, 6, false, 2>(p, stream) : launch_km<128, 256, 2, 4, 3, false, 2>(p, stream);
#if MM_X3_WIDE
            if (p.x3) return p.in2 ? launch_km<128, 256, 2, 2, 6, false, 1>(p, stream) : launch_km<128, 256, 2, 2, 3, false, 1>(p, stream);
#else
            if (p.x3) return p.in2 ? launch_km<128, 256, 2, 4, 6, false, 1>(p, stream) : launch_km<128, 256, 2, 4, 3, false, 1>(p, stream);
#endif
            if (p.sched1x1) return p.in2 ? launch_km<128, 256, 2, 4, 8>(p, stream) : launch_km<128, 256, 2, 4, 7>(p, stream);
            if (p.in2) return launch_km<128, 256, 2, 4, 6>(p, stream);
            return launch_km<128, 256, 2, 4, 3>(p, stream);
        default: return MM_ERR_INVALID_ARG;
    }
}

}  // namespace mm
