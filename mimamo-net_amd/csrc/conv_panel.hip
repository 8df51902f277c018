// 1x1 layers with a short K and a wide N -- ResNet50's 256 -> 1024 (+ residual) increase convs of conv4_x
// (api/resnet50_extractor.py:74-83: five of the trunk's launches, 8.9 ms of a 2 048-frame step at 0.76 of the MFMA roof in the engine) --
// with the ACTIVATION PANEL RESIDENT IN LDS (round 6, round-5 verdict item 2).
//
// The engine (conv_mfma.hip) gives a workgroup one 128 x 256 output tile: sixteen 16-deep chunks, prologue + epilogue as long as a quarter of
// the MFMA burst, every activation row fetched N / 256 = 4 times, one workgroup barrier per chunk.  Here a workgroup owns 128 rows for ALL N:
//   * the 128 x K panel (K <= 256: 128 KB) is loaded ONCE by LDS-DMA into the engine's swizzled 16-float rows, one barrier;
//   * the weights never touch LDS: a wave's B fragments -- lane (l & 31, l >> 5) takes W[n0 + (l & 31)][16 c + 8 (l >> 5) .. + 7], exactly the
//     eight floats the engine's ds_read_b128 pair hands it -- come straight from global memory (1 MB, L2-resident) through a four-deep
//     register ring, so the main loop has NO workgroup barrier and no LDS write: the eight waves run free and one wave's epilogue
//     (residual read, bias, ReLU, stores) hides under the other waves' MFMAs;
//   * the product is taken TRANSPOSED (a = weight fragment, b = activation fragment): the accumulators then hold four consecutive output
//     channels of one pixel per lane and register quad, so bias / residual / output move as 16-byte accesses straight from the
//     accumulators -- no staging through LDS (which the panel does not leave room for), no transpose.
// Same products in the same order per output element as the engine's loop (chunk by chunk, k-quad halves h = 0, 1, k = 0..3 inside):
// bit-identical to it (GPU tests).
//
// MEASURED SLOWER, so OPT-IN (MM_CONV_PANEL=1; the default keeps the engine) -- profiles/r06_ab_conv_panel.txt, same box, five launches of a
// 2 048-frame step: engine 8.67-8.82 ms (119-121 TFLOP/s); this kernel 13.1-13.5 ms with non-temporal stores, 10.8 ms with plain ones (shipped
// form), 8.4 ms with the epilogue's memory operations REMOVED (122 TFLOP/s: the barrier-free loop alone only matches the engine's whole launch),
// 12.7 ms without the weight loads, priorities null; PMC: matrix pipes 49.5 % busy, 37 % of the wave time at waitcnt.  Why: (1) one workgroup
// per CU runs in chip-wide lock-step rounds -- every round starts with 256 workgroups requesting 32 MB of panels at once and nothing to
// compute under it (the engine's two small-tile workgroups per CU are at random phases); (2) the two waves of a SIMD are symmetric (same
// n range, the two m halves), start together and reach their epilogues together, so the 3.3 GB of residual reads + stores per launch are
// exposed instead of hidden (0.5 ms per launch); (3) 32 contiguous bytes per pixel and store instruction: non-temporal stores of that
// granularity cost another 0.5 ms per launch.  What would fix (1) and (2) -- a persistent kernel that refills each 8 KB chunk slot with the NEXT
// panel's chunk as the last N-step releases it, and a deliberate half-step skew between the waves of a SIMD -- is a second kernel on top of
// this one for at most the 8.4 ms the loop itself reaches: not built.
#include "mm_common.h"
#include <cstdio>
#include <cstdlib>
#include "conv.h"

namespace mm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int PCH = 16;           // floats per chunk (the engine's CBK / CLD)
// PBM rows per panel, PNW = PBM / 16 waves: 128 rows -> 2 (m) x 4 (n) waves, one 128 KB workgroup per CU; 64 rows -> 1 x 4 waves, 64 KB: TWO
// workgroups per CU at independent phases (the second form, use_panel == 2: built after the first one's ablations pointed at its lock-step
// rounds and exposed epilogues).  Wave tile 64 x 64 of a PBM x 256 N-step either way.
constexpr int PRING = 4;          // chunks of B fragments in flight per wave
#ifndef MM_PANEL_ABL
#define MM_PANEL_ABL 0            // variant builds only (tools/_ab): 1 = no residual loads / one store per tile (results wrong), 2 = no weight loads in the
#endif                            // loop (results wrong), 4 = the wm = 0 waves at a higher issue priority, 8 = non-temporal instead of plain stores

template <int NKC, int PBM>       // chunks of the panel: K = 16 NKC (NKC <= 16); rows of the panel
__global__ void __launch_bounds__(PBM / 16 * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv_panel_kernel(const ConvParams p) {
    constexpr int PNW = PBM / 16;
    extern __shared__ __attribute__((aligned(16))) float As[];       // [NKC][PBM][16], slot s of row m holds k-quad s ^ ((m >> 2) & 3)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int lr = lane & 31, lh = lane >> 5;
    const int m_base = p.m_off + blockIdx.x * PBM;

    // ---- the panel: one LDS-DMA piece per wave and chunk (16 rows x 64 B), rows past M come back as zeros from the range check
    {
        const int rows = p.M - m_base < PBM ? p.M - m_base : PBM;
        const float* base = p.in + (int64_t)m_base * p.in_cstride + p.in_coff;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (unsigned)(rows * p.in_cstride * 4), 0x00020000);
        const int lrow = wave * 16 + (lane >> 2);
        const int kq = (lane & 3) ^ ((lrow >> 2) & 3);
        const unsigned voff = (unsigned)(lrow * p.in_cstride + kq * 4) * 4u;
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)As + (unsigned)(wave * 16 * PCH * 4));
#pragma unroll
        for (int c = 0; c < NKC; ++c) {
            unsigned keep;      // M0 is compiler-reserved: save it, point it at this wave's 1 KB piece of chunk c, restore it, all in one statement
            // (the chunk's k offset rides in the SCALAR offset: the range check looks at the vector offset only, which is what marks rows past M)
            asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %5 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds0), "s"((unsigned)(c * PBM * PCH * 4)), "s"((unsigned)(c * PCH * 4)) : "memory", "scc");
        }
    }

    // ---- B fragments straight from global memory: rows n_step + wn * 64 + j * 32 + lr, floats [16 c + 8 lh, + 8)
    const float* wrow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) wrow[j] = p.w + (int64_t)(wn * 64 + j * 32 + lr) * p.Kpad + 8 * lh;
    const int n_steps = p.Cout / 256;
    float4 bq[PRING][2][2];
    // chunk c (compile-time) of the N-step whose rows start at wp[j]: the k offset is an instruction immediate
    auto load_b = [&](int slot, const float* const (&wp)[2], int c) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bq[slot][j][0] = *reinterpret_cast<const float4*>(wp[j] + c * PCH);
            bq[slot][j][1] = *reinterpret_cast<const float4*>(wp[j] + c * PCH + 4);
        }
    };
    static_assert(NKC % PRING == 0, "the chunk loop is unrolled by the ring depth");
#pragma unroll
    for (int s = 0; s < PRING - 1; ++s) load_b(s, wrow, s);

    // A fragment offsets (floats) inside a chunk: row r, k-quads 2 lh and 2 lh + 1 in slots (2 lh) ^ g and (2 lh + 1) ^ g, g = (r >> 2) & 3
    int fao[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wm * 64 + i * 32 + lr, g = (r >> 2) & 3;
        fao[i][0] = r * PCH + (((2 * lh) ^ g) << 2);
        fao[i][1] = r * PCH + (((2 * lh + 1) ^ g) << 2);
    }

    // chunks 8..15 lie beyond the 16-bit offset field of ds_read_b128: their own base registers (opaque, or hipcc re-adds the constant per read)
    int fao_hi[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            fao_hi[i][h] = fao[i][h] + 8 * PBM * PCH;
            asm volatile("" : "+v"(fao_hi[i][h]));
        }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((PRING - 1) * 4) : "memory");    // the panel landed (the B loads issued after it may still fly)

    const bool wide_res = p.res != nullptr;
    // A fragments one chunk ahead of the MFMAs that use them (two register sets), B fragments PRING - 1 chunks ahead; a scheduling barrier per
    // chunk keeps hipcc from hoisting a whole N-step's loads to the top of the unrolled loop (it renames the ring away and spills 176 registers)
    float4 qa[2][2][2];
    auto read_a = [&](int set, int c) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // (assume_aligned: behind the opaque base hipcc no longer sees the 16-byte alignment and splits the read into two ds_read2_b32)
            qa[set][i][0] = *reinterpret_cast<const float4*>(__builtin_assume_aligned(As + (c & 7) * PBM * PCH + (c < 8 ? fao[i][0] : fao_hi[i][0]), 16));
            qa[set][i][1] = *reinterpret_cast<const float4*>(__builtin_assume_aligned(As + (c & 7) * PBM * PCH + (c < 8 ? fao[i][1] : fao_hi[i][1]), 16));
        }
    };
    read_a(0, 0);
    if ((MM_PANEL_ABL & 4) && wm == 0) __builtin_amdgcn_s_setprio(2);
    for (int ns = 0; ns < n_steps; ++ns) {
        const int nn = ns + 1 < n_steps ? ns + 1 : 0;
        const float* const wcur[2] = {wrow[0] + (int64_t)ns * 256 * p.Kpad, wrow[1] + (int64_t)ns * 256 * p.Kpad};
        const float* const wnext[2] = {wrow[0] + (int64_t)nn * 256 * p.Kpad, wrow[1] + (int64_t)nn * 256 * p.Kpad};
#pragma unroll
        for (int c = 0; c < NKC; ++c) {
            // (unconditional -- past the last N-step the stream wraps to the first: behind a branch hipcc's counted vmcnt waits must assume the
            //  path WITHOUT the new loads and so wait for the new loads as well, which exposes the full L2 latency in every chunk)
            if (!(MM_PANEL_ABL & 2)) {
            if (c + PRING - 1 < NKC) load_b((c + PRING - 1) % PRING, wcur, c + PRING - 1);
            else load_b((c + PRING - 1) % PRING, wnext, c + PRING - 1 - NKC);
            }
            read_a((c + 1) & 1, (c + 1) % NKC);          // (the last chunk of an N-step reads chunk 0 again: the next N-step's first)
            __builtin_amdgcn_sched_barrier(0);           // loads and fragment reads first, then the chunk's MFMA burst
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float4& A4 = qa[c & 1][i][h];
                            const float4& B4 = bq[c % PRING][j][h];
                            const float a = kk == 0 ? A4.x : kk == 1 ? A4.y : kk == 2 ? A4.z : A4.w;
                            const float b = kk == 0 ? B4.x : kk == 1 ? B4.y : kk == 2 ? B4.z : B4.w;
                            // transposed: rows of the accumulator = output channels, columns = pixels
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[i][j], 0, 0, 0);
                        }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue of this N-step, straight from the accumulators: lane (lr, lh), register quad g of acc[i][j] = channels
        //      n0 + 8 g + 4 lh .. + 3 of pixel m_base + wm * 64 + i * 32 + lr
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m_base + wm * 64 + i * 32 + lr;
            const bool mok = m < p.M;
            const int64_t mo = mok ? (int64_t)m : (int64_t)m_base;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n0 = ns * 256 + wn * 64 + j * 32 + 4 * lh;
                float4 rs[4];
                if (wide_res && !(MM_PANEL_ABL & 1)) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) rs[g] = *reinterpret_cast<const float4*>(p.res + mo * p.res_cstride + p.res_coff + n0 + 8 * g);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + 8 * g;
                    float4 o = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    if (p.bias) {
                        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
                        o.x += b4.x; o.y += b4.y; o.z += b4.z; o.w += b4.w;
                    }
                    if (wide_res && !(MM_PANEL_ABL & 1)) { o.x += rs[g].x; o.y += rs[g].y; o.z += rs[g].z; o.w += rs[g].w; }
                    if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                    if (mok && (!(MM_PANEL_ABL & 1) || o.x == 12345.678f)) {
                        const f32x4_t ov = {o.x, o.y, o.z, o.w};
                        // (plain stores: 32 contiguous bytes per pixel and instruction -- the non-temporal hint the engine's 256-byte rows use costs
                        //  this granularity 0.5 ms per launch, MM_PANEL_ABL & 8)
                        if (MM_PANEL_ABL & 8) __builtin_nontemporal_store(ov, reinterpret_cast<f32x4_t*>(p.out + mo * p.out_cstride + p.out_coff + n));
                        else *reinterpret_cast<f32x4_t*>(p.out + mo * p.out_cstride + p.out_coff + n) = ov;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = 0.f;
                }
            }
        }
    }
}

bool conv_panel_supported(const ConvParams& p) {
    return p.kh == 1 && p.kw == 1 && p.pad == 0 && p.stride == 1 && p.korder == 0 && !p.in2 && !p.x3 && !p.post_scale && !p.hpool && p.batch <= 1 &&
           p.H == p.Ho && p.W == p.Wo && p.K == p.Kpad && (p.K == 256 || p.K == 128) && p.Cin == p.K && p.in_cstride >= p.K && p.Cout % 256 == 0 &&
           p.Cout >= 512 && ((p.in_cstride | p.in_coff | p.out_cstride | p.out_coff | p.res_cstride | p.res_coff) & 3) == 0 &&
           (int64_t)128 * p.in_cstride * 4 < 0x7FFFF000ll;
}

template <int NKC, int PBM>
static int launch_panel(const ConvParams& p, int tiles, hipStream_t stream) {
    const int lds = NKC * PBM * PCH * 4;
    MM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_panel_kernel<NKC, PBM>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((conv_panel_kernel<NKC, PBM>), dim3((unsigned)tiles), dim3(PBM / 16 * 64), lds, stream, p);
    return MM_OK;
}

// rows of one panel / panels resident per CU for the form ConvParams::use_panel selects
int conv_panel_rows(const ConvParams& p) { return p.use_panel == 2 ? 64 : 128; }
int conv_panel_per_cu(const ConvParams& p) { return p.use_panel == 2 ? 2 : 1; }

// rows [p.m_off, p.M) -- conv_forward has set M (and m_off / m_end for a tail split)
int conv_panel_forward(const ConvParams& p, hipStream_t stream) {
    const int pbm = conv_panel_rows(p);
    const int tiles = (p.M - p.m_off + pbm - 1) / pbm;
    if (tiles <= 0) return MM_OK;
    const int nkc = p.K / PCH;
    if (prof_enabled()) {
        char tag[64];
        snprintf(tag, sizeof(tag), "M=%d K=%d N=%d k1 s1 t%dxNp b1", p.M - p.m_off, p.K, p.Cout, pbm);
        prof_before(0, 2.0 * (double)(p.M - p.m_off) * (double)p.K * (double)p.Cout, stream, tag);
    }
    int rc;
    if (pbm == 128) rc = nkc == 16 ? launch_panel<16, 128>(p, tiles, stream) : launch_panel<8, 128>(p, tiles, stream);
    else rc = nkc == 16 ? launch_panel<16, 64>(p, tiles, stream) : launch_panel<8, 64>(p, tiles, stream);
    prof_after(0, stream);
    if (rc != MM_OK) return rc;
    MM_LAUNCH_CHECK();
    return MM_OK;
}

}  // namespace mm
