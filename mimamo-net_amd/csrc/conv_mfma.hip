// fp32 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One engine serves every dense contraction of the hot path:
//   * the 53 conv+BN(+ReLU)(+residual) layers of the ResNet50 trunk behind
//     Resnet50_Extractor.get_vec (api/resnet50_extractor.py:74-83),
//   * PhaseNet's six 3x3 convs and every nn.Linear of Two_Stream_RNN (api/mimamo_net.py:14-26,
//     41-95,115-122), the GRU input/recurrent products (api/mimamo_net.py:119) -- a Linear is a
//     1x1 conv on a 1x1 image.
// fp32 in / fp32 accumulate: the MFMA is bit-for-bit an fmaf chain, so results differ from the
// reference's fp32 convs only by summation order (the 1e-4 output tolerance of north_star rules
// out bf16/fp16 operands).
//
// GEMM view:  out[m][n] = sum_k A[m][k] * Wt[n][k],  m = (b, ho, wo),  n = cout,  k = (r, s, c)
//   activations NHWC (channel stride/offset allow channel-sliced reads and concat-writes),
//   weights packed [Cout][Kpad] with k contiguous (Kpad = K rounded up to 16, zero filled).
// Workgroup = 256 threads = 2x2 waves; block tile BM x BN x 16; each wave owns (BM/2)x(BN/2) as
// 32x32 MFMA sub-tiles.  Both operand tiles are staged [row][16+4] in LDS: the +4 pad makes the
// ds_read_b128 fragment reads and the global->LDS float4 writes bank-conflict free.  Inside a
// 16-deep chunk lanes 0-31 take k = 0..7 and lanes 32-63 take k = 8..15 (any k order is a valid
// contraction order), so one b128 read feeds four MFMAs.  Global loads for chunk i+1 are issued
// before the MFMAs of chunk i (register prefetch, double-buffered LDS, one barrier per chunk).
// Epilogue (fused): + bias (BN folded on the host) [+ residual] [ReLU] [* post_scale + post_shift]
// (BN placed after ReLU, mimamo_net.py:54-62,115-117), stored as 128-byte channel rows.
#include "mm_common.h"
#include <cstdio>
#include "conv.h"

namespace mm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int CBK = 16;   // k-chunk
constexpr int CLD = 20;   // LDS row stride (floats)

// ABL: measurement-only instantiation whose loop stages can be switched off at run time (p.ablate bits:
// 1 no global loads, 2 no LDS stores, 4 no barrier, 8 no fragment reads) to attribute time; results are wrong.
template <int BM, int BN, int WGM, int WGN, bool ABL = false>
__global__ void __launch_bounds__(256)
conv_mfma_kernel(const ConvParams p) {
    static_assert(WGM * WGN == 4, "four waves per workgroup");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int AIT = BM / 64, BIT = BN / 64;
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * CLD];
    float* As = lds;                      // [2][BM][CLD]
    float* Bs = lds + 2 * BM * CLD;       // [2][BN][CLD]

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
    const int tile_m = logical / p.tiles_n, tile_n = logical - tile_m * p.tiles_n;
    const int m_base = tile_m * BM, n_base = tile_n * BN;

    // ---- operand fetch through buffer descriptors: the hardware range check returns 0 for any offset
    //      >= num_records, which gives zero padding (image border taps, K tail, row/channel tails) without
    //      branches or selects on the data.  The A descriptor is rebased to the first image this block
    //      touches so 32-bit byte offsets always suffice (a block spans a handful of images).
    const int hw_out = p.Ho * p.Wo;
    const int img0 = m_base / hw_out;                                  // wave-uniform
    const int64_t img_elems = (int64_t)p.H * p.W * p.in_cstride;
    const int64_t rem_elems = ((int64_t)p.B - img0) * img_elems;
    const unsigned a_bytes = rem_elems * 4 > 0xFFFFF000ll ? 0xFFFFF000u : (unsigned)(rem_elems * 4);
    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in) + img0 * img_elems, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (unsigned)((int64_t)p.Cout * p.Kpad * 4), 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFFFu;

    // per-thread rows (tid>>2) + 64*it and k-quad (tid&3)
    const int kq = tid & 3, lrow = tid >> 2;
    int a_hi0[AIT], a_wi0[AIT], a_pix[AIT];
    bool a_ok[AIT];
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
        const int m = m_base + lrow + it * 64;
        a_ok[it] = m < p.M;
        const int mm_ = a_ok[it] ? m : m_base;
        const int b = mm_ / hw_out, rem = mm_ - b * hw_out;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        a_hi0[it] = ho * p.stride - p.pad;
        a_wi0[it] = wo * p.stride - p.pad;
        // element offset of tap (0,0), channel 0, relative to the descriptor base (may be negative: padding)
        a_pix[it] = (b - img0) * (int)img_elems + (a_hi0[it] * p.W + a_wi0[it]) * p.in_cstride + p.in_coff;
    }
    unsigned vb[BIT];
#pragma unroll
    for (int it = 0; it < BIT; ++it) {
        const int n = n_base + lrow + it * 64;
        vb[it] = n < p.Cout ? (unsigned)(n * p.Kpad + kq * 4) * 4u : OOB;
    }
    // tap state of this lane's k-quad, advanced by one chunk (16) per iteration.  Two K orders:
    //   korder 0:  k = (r*kw + s)*Cin + c              (any Cin % 4 == 0)
    //   korder 1:  k = ((c/16 * kh + r)*kw + s)*16 + c%16   (Cin % 16 == 0): all taps of a 16-channel slice are
    //              consecutive chunks, so the ~kh*kw-fold re-use of every input pixel happens within a few
    //              chunks and is served by L1/L2 instead of the fabric (measured: conv4_x 3x3 fetched 8.5x its
    //              input with korder 0).
    int tk = kq * 4, tr, ts, tc;
    if (p.korder == 0) {
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
        const int tapoff = (tr * p.W + ts) * p.in_cstride + tc;
        const bool kok = tk < p.K;
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            const int hi = a_hi0[it] + tr, wi = a_wi0[it] + ts;
            const bool ok = kok && a_ok[it] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            va[it] = ok ? (unsigned)(a_pix[it] + tapoff) * 4u : OOB;
        }
    };
    auto tap_advance = [&]() {
        tk += CBK;
        if (p.korder == 0) {
            tc += CBK;
            while (tc >= p.Cin) {
                tc -= p.Cin;
                if (++ts == p.kw) { ts = 0; ++tr; }
            }
        } else if (++ts == p.kw) {
            ts = 0;
            if (++tr == p.kh) { tr = 0; tc += CBK; }
        }
#pragma unroll
        for (int it = 0; it < BIT; ++it) vb[it] = vb[it] == OOB ? OOB : vb[it] + CBK * 4u;
    };

    u32x4 ga[AIT], gb[BIT];
    auto gload = [&]() {
#pragma unroll
        for (int it = 0; it < AIT; ++it) ga[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, va[it], 0, 0);
#pragma unroll
        for (int it = 0; it < BIT; ++it) gb[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, vb[it], 0, 0);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int it = 0; it < AIT; ++it)
            *reinterpret_cast<u32x4*>(As + (buf * BM + lrow + it * 64) * CLD + kq * 4) = ga[it];
#pragma unroll
        for (int it = 0; it < BIT; ++it)
            *reinterpret_cast<u32x4*>(Bs + (buf * BN + lrow + it * 64) * CLD + kq * 4) = gb[it];
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
    // One barrier per 16-deep chunk.  Iteration kc: fragments of chunk kc LDS->registers, loads of chunk kc+1
    // issued with offsets computed one iteration earlier (no VALU in front of them), offsets of chunk kc+2
    // computed in the shadow of the 32 MFMAs, then chunk kc+1 registers->LDS (other buffer) and the barrier.
    // Loads/stores past the last chunk are harmless (range check -> zeros into a buffer nobody reads).
    tap_offsets();
    gload();
    tap_advance();
    tap_offsets();
    lstore(0);
    __syncthreads();
    float4 fa[TM][2], fb[TN][2];
    if (ABL) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i][0] = fa[i][1] = float4{1.f, 2.f, 3.f, 4.f};
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j][0] = fb[j][1] = float4{1.f, 2.f, 3.f, 4.f};
    }
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (!ABL || !(p.ablate & 8)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float4* q = reinterpret_cast<const float4*>(As + (buf * BM + wm * WM + i * 32 + lr) * CLD + lh * 8);
                fa[i][0] = q[0];
                fa[i][1] = q[1];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float4* q = reinterpret_cast<const float4*>(Bs + (buf * BN + wn * WN + j * 32 + lr) * CLD + lh * 8);
                fb[j][0] = q[0];
                fb[j][1] = q[1];
            }
        }
        if (!ABL || !(p.ablate & 1)) {
            gload();
            tap_advance();
            tap_offsets();
        }
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
        if (!ABL || !(p.ablate & 2)) lstore(buf ^ 1);
        if (!ABL || !(p.ablate & 4)) __syncthreads();
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
        float* st = lds + wave * (32 * SLD);        // 32 x WN per wave; 4 * 32 * SLD <= 2 * (BM + BN) * CLD
        static_assert(4 * 32 * SLD <= 2 * (BM + BN) * CLD, "staging fits in the operand buffers");
        const int qc = lane % QN, qr = lane / QN;
        const int n0 = n_base + wn * WN + qc * 4;
        const bool nok = n0 < p.Cout;
        float4 bias4 = {0.f, 0.f, 0.f, 0.f}, ps4 = {1.f, 1.f, 1.f, 1.f}, pt4 = {0.f, 0.f, 0.f, 0.f};
        if (nok && p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + n0);
        if (nok && p.post_scale) {
            ps4 = *reinterpret_cast<const float4*>(p.post_scale + n0);
            pt4 = *reinterpret_cast<const float4*>(p.post_shift + n0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    st[((e & 3) + 8 * (e >> 2) + 4 * lh) * SLD + j * 32 + lr] = acc[i][j][e];
            __syncthreads();
            const int m0 = m_base + wm * WM + i * 32;
            float4 v[32 / RPI], rsd[32 / RPI];
#pragma unroll
            for (int t = 0; t < 32 / RPI; ++t) {
                const int row = t * RPI + qr;
                v[t] = *reinterpret_cast<const float4*>(st + row * SLD + qc * 4);
                const int m = m0 + row;
                if (p.res) {
                    const int64_t mo = (nok && m < p.M) ? (int64_t)m * p.res_cstride + p.res_coff + n0 : 0;
                    rsd[t] = *reinterpret_cast<const float4*>(p.res + mo);
                }
            }
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
                if (nok && m < p.M) *reinterpret_cast<float4*>(p.out + (int64_t)m * p.out_cstride + p.out_coff + n0) = o;
            }
            if (i + 1 < TM) __syncthreads();
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
                    p.out[(int64_t)m * p.out_cstride + p.out_coff + n] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WGM, int WGN, bool ABL = false>
static int launch_cfg(ConvParams p, hipStream_t stream) {
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.Cout + BN - 1) / BN;
    const int64_t blocks = (int64_t)p.tiles_m * p.tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffff) return MM_ERR_INVALID_ARG;
    if (prof_enabled()) {
        char tag[64];
        snprintf(tag, sizeof(tag), "M=%d K=%d N=%d k%d s%d t%dx%d", p.M, p.K, p.Cout, p.kh, p.stride, BM, BN);
        prof_before(0, 2.0 * (double)p.M * (double)(p.kh * p.kw * p.Cin_real) * (double)p.Cout, stream, tag);
    }
    hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WGM, WGN, ABL>), dim3((unsigned)blocks), dim3(256), 0, stream, p);
    prof_after(0, stream);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

int conv_forward(const ConvParams& p0, hipStream_t stream) {
    ConvParams p = p0;
    if (p.Cin % 4 || p.in_cstride % 4 || p.in_coff % 4 || p.Kpad % CBK || p.K > p.Kpad) return MM_ERR_INVALID_ARG;
    if (p.korder != 0 && (p.korder != 1 || p.Cin % CBK)) return MM_ERR_INVALID_ARG;
    // 32-bit byte offsets inside the kernel: the weight matrix, and the few images one 128-row block spans
    if ((uint64_t)p.Cout * p.Kpad * 4 >= 0xFFFFF000ull) return MM_ERR_INVALID_ARG;
    {
        const int64_t hw_o = (int64_t)p.Ho * p.Wo;
        const int64_t span = 128 / (hw_o > 0 ? hw_o : 1) + 2;
        if (span * p.H * p.W * p.in_cstride * 4 >= 0x7FFFF000ll) return MM_ERR_INVALID_ARG;
    }
    p.M = p.B * p.Ho * p.Wo;
    if (p.M <= 0) return MM_OK;
    if (p.Cin_real <= 0) p.Cin_real = p.Cin;
    // Tile choice.  Wave tile 64x64 (2x2 MFMA sub-tiles, 4 accumulators) is the efficient shape: 128x128
    // blocks (2x2 waves) when Cout > 64, 256x64 blocks (4x1 waves) for the 64-channel layers; smaller tiles
    // only when the grid would not give every CU (256) a couple of workgroups.
    const int64_t m128 = (p.M + 127) / 128, m256 = (p.M + 255) / 256;
    const int64_t n128 = (p.Cout + 127) / 128, n64 = (p.Cout + 63) / 64;
    int cfg = p.force_tile;
    if (cfg >= 16) {  // measurement-only: 128x128 with ablation bits (cfg - 16)
        p.ablate = cfg - 16;
        return launch_cfg<128, 128, 2, 2, true>(p, stream);
    }
    if (cfg == 0) {
        if (p.Cout > 64 && m128 * n128 >= 512) cfg = 1;
        else if (p.Cout <= 64 && m256 * n64 >= 512) cfg = 4;
        else if (m128 * n64 >= 512) cfg = 2;
        else cfg = 3;
    }
    switch (cfg) {
        case 1: return launch_cfg<128, 128, 2, 2>(p, stream);
        case 2: return launch_cfg<128, 64, 2, 2>(p, stream);
        case 3: return launch_cfg<64, 64, 2, 2>(p, stream);
        case 4: return launch_cfg<256, 64, 4, 1>(p, stream);
        default: return MM_ERR_INVALID_ARG;
    }
}

}  // namespace mm
