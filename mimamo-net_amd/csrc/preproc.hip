// Frame preprocessing on the GPU, bit-exact with the PIL pipeline the reference runs in DataLoader workers
// (SURVEY.md 8f-1):
//   gray : Image.open(bmp).convert('L') -> GroupScale(48, Image.LANCZOS) -> /255
//          (api/sampler/snippet_sampler.py:163,177-185; api/utils/data_utils.py:80)
//   rgb  : Resize(256) [PIL bilinear] -> CenterCrop(224) -> ToTensor -> x*255 -> Normalize(mean, 1)
//          (api/utils/model_utils.py:29-39; api/sampler/image_sampler.py)
// PIL resamples 8-bit images in fixed point (Pillow src/libImaging/Resample.c): per axis a table of
// double-precision filter weights normalised to sum 1, converted to integers with 22 fractional bits
// (round half away from zero), a horizontal pass then a vertical pass, each accumulating
// 2^21 + sum(u8 * coeff) in int32 and storing clip8(acc >> 22) -- i.e. the intermediate image is uint8.
// The tables are rebuilt here on the host with the same formulas; the kernels are pure integer work
// followed by the reference's fp32 epilogue evaluated with explicitly un-fused IEEE operations.
#include <cmath>
#include <new>
#include <vector>
#include <cstdlib>
#include "mm_common.h"

namespace mm {

constexpr int PRECISION_BITS = 32 - 8 - 2;

struct ResampleTable {
    int in_size = 0, out_size = 0, ksize = 0;
    std::vector<int> bounds;  // [out][2]: xmin, count
    std::vector<int> kk;      // [out][ksize]
};

static double sinc_filter(double x) {
    if (x == 0.0) return 1.0;
    x = x * 3.14159265358979323846;
    return std::sin(x) / x;
}
static double lanczos_filter(double x) { return (-3.0 <= x && x < 3.0) ? sinc_filter(x) * sinc_filter(x / 3) : 0.0; }
static double bilinear_filter(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

// Pillow precompute_coeffs + normalize_coeffs_8bpc for a full-axis resize (box = whole image).
// filter: 0 = bilinear (support 1), 1 = lanczos (support 3)
int build_resample_table(int in_size, int out_size, int filter, ResampleTable& t) {
    if (in_size <= 0 || out_size <= 0 || filter < 0 || filter > 1) return MM_ERR_INVALID_ARG;
    double (*f)(double) = filter == 0 ? bilinear_filter : lanczos_filter;
    const double fsupport = filter == 0 ? 1.0 : 3.0;
    double filterscale, scale;
    filterscale = scale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = fsupport * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    t.in_size = in_size;
    t.out_size = out_size;
    t.ksize = ksize;
    t.bounds.assign((size_t)out_size * 2, 0);
    t.kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int x = 0;
        for (; x < xmax; ++x) {
            const double w = f((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; ++x) k[x] = 0;
        t.bounds[xx * 2] = xmin;
        t.bounds[xx * 2 + 1] = xmax;
        for (x = 0; x < ksize; ++x) {
            const double v = k[x] * (double)(1 << PRECISION_BITS);
            t.kk[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
        }
    }
    return MM_OK;
}

__device__ __forceinline__ int clip8(int acc) {
    const int v = acc >> PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// ---- gray: one workgroup per frame ------------------------------------------------------------------
// LDS: L plane [S][S] u8, horizontal-pass image [S][G] u8.
__global__ void __launch_bounds__(256)
preproc_gray_kernel(const uint8_t* __restrict__ frames, int S, int G, int ksize, const int* __restrict__ bounds,
                    const int* __restrict__ kk, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* L = smem;            // [S*S]
    unsigned char* T = smem + S * S;    // [S*G]
    const int64_t n = blockIdx.x;
    const uint8_t* src = frames + n * (int64_t)S * S * 3;
    const int tid = threadIdx.x;
    // convert('L'): (19595 R + 38470 G + 7471 B + 0x8000) >> 16
    for (int i = tid; i < S * S; i += 256) {
        const unsigned r = src[i * 3], g = src[i * 3 + 1], b = src[i * 3 + 2];
        L[i] = (unsigned char)((19595u * r + 38470u * g + 7471u * b + 0x8000u) >> 16);
    }
    __syncthreads();
    for (int i = tid; i < S * G; i += 256) {  // horizontal pass
        const int y = i / G, xx = i - y * G;
        const int xmin = bounds[xx * 2], cnt = bounds[xx * 2 + 1];
        const int* k = kk + xx * ksize;
        int acc = 1 << (PRECISION_BITS - 1);
        for (int x = 0; x < cnt; ++x) acc += (int)L[y * S + xmin + x] * k[x];
        T[i] = (unsigned char)clip8(acc);
    }
    __syncthreads();
    for (int i = tid; i < G * G; i += 256) {  // vertical pass + /255
        const int yy = i / G, xx = i - yy * G;
        const int ymin = bounds[yy * 2], cnt = bounds[yy * 2 + 1];
        const int* k = kk + yy * ksize;
        int acc = 1 << (PRECISION_BITS - 1);
        for (int y = 0; y < cnt; ++y) acc += (int)T[(ymin + y) * G + xx] * k[y];
        out[n * (int64_t)G * G + i] = __fdiv_rn((float)clip8(acc), 255.0f);
    }
}

// ---- gray, fast form (round 6).  The kernel above spends its time in the LDS pipe: byte-wide LDS accesses cost ~26 LDS cycles per wave
// instruction on this chip (PMC: SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS; a ds_read_b32 costs 2), and with the 15 taps of the Lanczos filter
// (112 -> 48) it issues thirty accesses per output: 0.173 ms per 2 048 frames, LDS 73 % busy.  Here every LDS access is a word or wider: the
// source arrives as 32-bit words (four pixels = three words), the L plane stays bytes but is read as five words + four funnel shifts per
// output, the horizontal-pass image is kept as ints, and the tables (bounds, weights padded to sixteen per position: the builder zero-fills
// beyond a position's taps, so all sixteen are applied unconditionally) sit in LDS.  Same integer arithmetic on the same values: bit-exact
// with PIL like the kernel above (tests/test_preproc.py).  Needs ksize <= 16, S * S % 4 == 0 and 4-byte aligned frames; the launcher falls
// back to the kernel above otherwise.
constexpr int GRAY_KW = 16;
__device__ __forceinline__ int dot4(unsigned bytes, const int4 k) {
    return (int)(bytes & 255u) * k.x + (int)((bytes >> 8) & 255u) * k.y + (int)((bytes >> 16) & 255u) * k.z + (int)(bytes >> 24) * k.w;
}
__global__ void __launch_bounds__(256)
preproc_gray_words_kernel(const uint8_t* __restrict__ frames, int S, int G, int ksize, const int* __restrict__ bounds,
                          const int* __restrict__ kk, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* L4 = reinterpret_cast<unsigned*>(smem);                       // [S*S / 4 + 8] bytes of L, as words (+ slack for the last taps)
    int* T = reinterpret_cast<int*>(smem) + ((S * S / 4 + 8 + 3) & ~3);     // [S*G] ints
    int* s_kk = T + ((S * G + 3) & ~3);                                     // [G][16]
    int* s_bounds = s_kk + GRAY_KW * G;                                     // [G][2]
    const int64_t n = blockIdx.x;
    const unsigned* w = reinterpret_cast<const unsigned*>(frames + n * (int64_t)S * S * 3);
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * G; i += 256) s_bounds[i] = bounds[i];
    for (int i = tid; i < GRAY_KW * G; i += 256) s_kk[i] = (i & (GRAY_KW - 1)) < ksize ? kk[(i / GRAY_KW) * ksize + (i & (GRAY_KW - 1))] : 0;
    if (tid < 8) L4[S * S / 4 + tid] = 0u;
    // convert('L'): (19595 R + 38470 G + 7471 B + 0x8000) >> 16, four pixels per step
    for (int q = tid; q < S * S / 4; q += 256) {
        const unsigned w0 = w[3 * q], w1 = w[3 * q + 1], w2 = w[3 * q + 2];
        const unsigned l0 = (19595u * (w0 & 255u) + 38470u * ((w0 >> 8) & 255u) + 7471u * ((w0 >> 16) & 255u) + 0x8000u) >> 16;
        const unsigned l1 = (19595u * (w0 >> 24) + 38470u * (w1 & 255u) + 7471u * ((w1 >> 8) & 255u) + 0x8000u) >> 16;
        const unsigned l2 = (19595u * ((w1 >> 16) & 255u) + 38470u * (w1 >> 24) + 7471u * (w2 & 255u) + 0x8000u) >> 16;
        const unsigned l3 = (19595u * ((w2 >> 8) & 255u) + 38470u * ((w2 >> 16) & 255u) + 7471u * (w2 >> 24) + 0x8000u) >> 16;
        L4[q] = (l0 & 255u) | ((l1 & 255u) << 8) | ((l2 & 255u) << 16) | (l3 << 24);
    }
    __syncthreads();
    for (int i = tid; i < S * G; i += 256) {  // horizontal pass: sixteen bytes from byte offset o of L
        const int y = i / G, xx = i - y * G;
        const int o = y * S + s_bounds[xx * 2];
        const unsigned* p = L4 + (o >> 2);
        const unsigned a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4];
        const unsigned sh = (unsigned)(o & 3) * 8u;
        const int4* k4 = reinterpret_cast<const int4*>(s_kk + xx * GRAY_KW);
        int acc = 1 << (PRECISION_BITS - 1);
        acc += dot4(__builtin_amdgcn_alignbit(a1, a0, sh), k4[0]);
        acc += dot4(__builtin_amdgcn_alignbit(a2, a1, sh), k4[1]);
        acc += dot4(__builtin_amdgcn_alignbit(a3, a2, sh), k4[2]);
        acc += dot4(__builtin_amdgcn_alignbit(a4, a3, sh), k4[3]);
        T[i] = clip8(acc);
    }
    __syncthreads();
    for (int i = tid; i < G * G; i += 256) {  // vertical pass + /255
        const int yy = i / G, xx = i - yy * G;
        const int ymin = s_bounds[yy * 2], cnt = s_bounds[yy * 2 + 1];
        const int* k = s_kk + yy * GRAY_KW;
        int acc = 1 << (PRECISION_BITS - 1);
        for (int y = 0; y < cnt; ++y) acc += T[(ymin + y) * G + xx] * k[y];
        out[n * (int64_t)G * G + i] = __fdiv_rn((float)clip8(acc), 255.0f);
    }
}

// ---- rgb, zero-bordered packed NHWC3 output (the stem's fastest input): one workgroup per (frame, 16 output rows).
// The general kernel below evaluates the horizontal pass once per (output row, vertical tap): 16 rows x 2 taps = 32 evaluations per
// output column for the 8-10 input rows a block really touches.  Here the block's input rows go through the horizontal pass ONCE
// into LDS (uint8, what PIL stores between its passes), the vertical pass reads bytes from LDS, and every lane writes one float of
// the packed row (row starts are only 4-byte aligned in the bordered layout): 256 contiguous bytes per wave store.
constexpr int RGB_ROWS = 16, RGB_MAX_IN = 24;
__global__ void __launch_bounds__(256)
preproc_rgb3_kernel(const uint8_t* __restrict__ frames, int S, int R, int C, int ksize, const int* __restrict__ bounds,
                    const int* __restrict__ kk, float mean0, float mean1, float mean2, float* __restrict__ out, int max_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hrow[];      // [max_in][C][3] uint8, then the staged source rows [max_in][S][3]
    const int64_t n = blockIdx.x;
    const int row0 = blockIdx.y * RGB_ROWS;
    const int off = (int)rintf((R - C) / 2.0f);  // CenterCrop: int(round((256-224)/2.)) = 16
    const uint8_t* src = frames + n * (int64_t)S * S * 3;
    const int CP = C + 6;
    float* img = out + n * (int64_t)CP * CP * 3;
    const int rows = min(RGB_ROWS, C - row0);
    // the 3-pixel zero border of this block's rows (left / right), and the three full rows above / below by the first / last block
    for (int i = threadIdx.x; i < rows * 18; i += 256) {
        const int ry = i / 18, e = i - ry * 18;
        img[((int64_t)(row0 + ry + 3) * CP + (e < 9 ? 0 : C + 3)) * 3 + (e < 9 ? e : e - 9)] = 0.f;
    }
    if (blockIdx.y == 0)
        for (int i = threadIdx.x; i < 3 * CP * 3; i += 256) img[i] = 0.f;
    if (blockIdx.y == gridDim.y - 1)
        for (int i = threadIdx.x; i < 3 * CP * 3; i += 256) img[(int64_t)(C + 3) * CP * 3 + i] = 0.f;
    // input rows this block's vertical taps touch
    const int yy_first = row0 + off, yy_last = row0 + rows - 1 + off;
    const int in_lo = bounds[yy_first * 2];
    const int n_in = bounds[yy_last * 2] + bounds[yy_last * 2 + 1] - in_lo;
    // (round 5) the block's source rows -- one contiguous range of the frame -- come into LDS as coalesced words first, and a thread keeps its
    // output column's bounds and weights in registers over the rows: the first version chased bounds -> weights -> source bytes through
    // global memory for every (row, column) item, nine dependent rounds of L2 latency per thread
    unsigned char* srow = hrow + ((max_in * C * 3 + 15) & ~15);               // [n_in][S][3] uint8
    {
        const uint8_t* s0 = src + (int64_t)in_lo * S * 3;
        const int nb = n_in * S * 3;
        if (((nb | (S * 3)) & 3) == 0 && (reinterpret_cast<uintptr_t>(s0) & 3) == 0) {
            for (int i = threadIdx.x; i < nb / 4; i += 256) reinterpret_cast<unsigned*>(srow)[i] = reinterpret_cast<const unsigned*>(s0)[i];
        } else {
            for (int i = threadIdx.x; i < nb; i += 256) srow[i] = s0[i];
        }
    }
    __syncthreads();
    // horizontal pass, once per (input row, cropped output column)
    for (int cx = threadIdx.x; cx < C; cx += 256) {
        const int xx = cx + off;
        const int xmin = bounds[xx * 2], xcnt = bounds[xx * 2 + 1];
        int kx[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) kx[x] = x < ksize ? kk[xx * ksize + x] : 0;      // (ksize <= 4: checked by the launcher)
        for (int r = 0; r < n_in; ++r) {
            const unsigned char* rowp = srow + (r * S + xmin) * 3;
            int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
#pragma unroll
            for (int x = 0; x < 4; ++x)
                if (x < xcnt) {
                    a0 += (int)rowp[x * 3] * kx[x];
                    a1 += (int)rowp[x * 3 + 1] * kx[x];
                    a2 += (int)rowp[x * 3 + 2] * kx[x];
                }
            unsigned char* h = hrow + (r * C + cx) * 3;
            h[0] = (unsigned char)clip8(a0); h[1] = (unsigned char)clip8(a1); h[2] = (unsigned char)clip8(a2);
        }
    }
    // per-row vertical taps and the epilogue as a table: the reference's ToTensor (/255), *255.0, -mean are three separately rounded
    // fp32 operations of a uint8 value -- 256 possible results per channel, evaluated once per block with contraction off
    __shared__ int s_rel[RGB_ROWS], s_cnt[RGB_ROWS], s_k[RGB_ROWS][4];
    __shared__ float s_lut[3][256];
    if (threadIdx.x < rows) {
        const int yy = row0 + threadIdx.x + off;
        s_rel[threadIdx.x] = bounds[yy * 2] - in_lo;
        s_cnt[threadIdx.x] = bounds[yy * 2 + 1];
        for (int y = 0; y < 4; ++y) s_k[threadIdx.x][y] = y < ksize ? kk[yy * ksize + y] : 0;
    }
    for (int i = threadIdx.x; i < 768; i += 256) {
#pragma clang fp contract(off)
        const int c = i >> 8;
        const float q = (float)(i & 255) / 255.0f;
        const float m255 = q * 255.0f;
        s_lut[c][i & 255] = m255 - (c == 0 ? mean0 : c == 1 ? mean1 : mean2);
    }
    __syncthreads();
    // vertical pass: a lane owns up to three floats (columns fl = cx * 3 + c) of the packed row and walks the block's rows, so a
    // wave store covers 256 contiguous bytes
    const int RF = C * 3;
    for (int fl = threadIdx.x; fl < RF; fl += 256) {
        const float* lut = s_lut[fl % 3];
        for (int ry = 0; ry < rows; ++ry) {
            const unsigned char* hp = hrow + s_rel[ry] * RF + fl;
            int acc = 1 << (PRECISION_BITS - 1);
            const int cnt = s_cnt[ry];
            for (int y = 0; y < cnt; ++y) acc += (int)hp[y * RF] * s_k[ry][y];
            img[((int64_t)(row0 + ry + 3) * CP + 3) * 3 + fl] = lut[clip8(acc)];
        }
    }
}

// ---- rgb3, word-wide LDS form (round 6).  preproc_rgb3_kernel above is LDS-bound on byte accesses (PMC: 3.8e7 LDS instructions at 5.3 LDS
// cycles each, the LDS pipe 68 % busy over its 0.49 ms; a byte-wide access costs ~26 cycles, a ds_read_b32 two): twelve byte reads + three
// byte writes per (input row, column) in the horizontal pass, a byte read per tap and float in the vertical one.  Here the horizontal pass
// reads the twelve source bytes of its four taps as four words + three funnel shifts and writes its three channel bytes as ONE packed word;
// the vertical pass is a lane per pixel (one word per tap for the three channels) and stores the pixel's three floats together.  Same
// integer arithmetic on the same values, same look-up table: bit-exact with PIL like the kernel above.  Needs C <= 256, S * 3 % 4 == 0 and
// 4-byte aligned frames (the launcher checks, and falls back).
__global__ void __launch_bounds__(256)
preproc_rgb3_words_kernel(const uint8_t* __restrict__ frames, int S, int R, int C, int ksize, const int* __restrict__ bounds,
                          const int* __restrict__ kk, float mean0, float mean1, float mean2, float* __restrict__ out, int max_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hrow[];
    unsigned* hw = reinterpret_cast<unsigned*>(hrow);                         // [max_in][C] packed (c0 | c1 << 8 | c2 << 16)
    unsigned* srow = hw + (((max_in * C) + 3) & ~3);                          // [n_in][S * 3 / 4] source rows as words (+ 4 words of slack)
    const int64_t n = blockIdx.x;
    const int row0 = blockIdx.y * RGB_ROWS;
    const int off = (int)rintf((R - C) / 2.0f);  // CenterCrop: int(round((256-224)/2.)) = 16
    const uint8_t* src = frames + n * (int64_t)S * S * 3;
    const int CP = C + 6;
    float* img = out + n * (int64_t)CP * CP * 3;
    const int rows = min(RGB_ROWS, C - row0);
    for (int i = threadIdx.x; i < rows * 18; i += 256) {      // the 3-pixel zero border of this block's rows
        const int ry = i / 18, e = i - ry * 18;
        img[((int64_t)(row0 + ry + 3) * CP + (e < 9 ? 0 : C + 3)) * 3 + (e < 9 ? e : e - 9)] = 0.f;
    }
    if (blockIdx.y == 0)
        for (int i = threadIdx.x; i < 3 * CP * 3; i += 256) img[i] = 0.f;
    if (blockIdx.y == gridDim.y - 1)
        for (int i = threadIdx.x; i < 3 * CP * 3; i += 256) img[(int64_t)(C + 3) * CP * 3 + i] = 0.f;
    const int yy_first = row0 + off, yy_last = row0 + rows - 1 + off;
    const int in_lo = bounds[yy_first * 2];
    const int n_in = bounds[yy_last * 2] + bounds[yy_last * 2 + 1] - in_lo;
    const int SW = S * 3 / 4;                                 // words per source row
    {
        const unsigned* s0 = reinterpret_cast<const unsigned*>(src + (int64_t)in_lo * S * 3);
        for (int i = threadIdx.x; i < n_in * SW; i += 256) srow[i] = s0[i];
        if (threadIdx.x < 4) srow[n_in * SW + threadIdx.x] = 0u;
    }
    __shared__ int s_rel[RGB_ROWS], s_cnt[RGB_ROWS], s_k[RGB_ROWS][4];
    __shared__ float s_lut[3][256];
    if (threadIdx.x < rows) {
        const int yy = row0 + threadIdx.x + off;
        s_rel[threadIdx.x] = bounds[yy * 2] - in_lo;
        s_cnt[threadIdx.x] = bounds[yy * 2 + 1];
        for (int y = 0; y < 4; ++y) s_k[threadIdx.x][y] = y < ksize ? kk[yy * ksize + y] : 0;
    }
    for (int i = threadIdx.x; i < 768; i += 256) {
#pragma clang fp contract(off)
        const int c = i >> 8;
        const float q = (float)(i & 255) / 255.0f;
        const float m255 = q * 255.0f;
        s_lut[c][i & 255] = m255 - (c == 0 ? mean0 : c == 1 ? mean1 : mean2);
    }
    __syncthreads();
    const int cx = threadIdx.x;
    if (cx < C) {     // horizontal pass, once per (input row, cropped output column): taps beyond a position's count carry zero weights
        const int xx = cx + off;
        const int xmin = bounds[xx * 2];
        int kx[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) kx[x] = x < ksize ? kk[xx * ksize + x] : 0;
        for (int r = 0; r < n_in; ++r) {
            const int bo = (r * S + xmin) * 3;
            const unsigned* p = srow + (bo >> 2);
            const unsigned a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
            const unsigned sh = (unsigned)(bo & 3) * 8u;
            const unsigned b0 = __builtin_amdgcn_alignbit(a1, a0, sh), b1 = __builtin_amdgcn_alignbit(a2, a1, sh),
                           b2 = __builtin_amdgcn_alignbit(a3, a2, sh);
            const int rnd = 1 << (PRECISION_BITS - 1);
            const int c0 = rnd + (int)(b0 & 255u) * kx[0] + (int)(b0 >> 24) * kx[1] + (int)((b1 >> 16) & 255u) * kx[2] + (int)((b2 >> 8) & 255u) * kx[3];
            const int c1 = rnd + (int)((b0 >> 8) & 255u) * kx[0] + (int)(b1 & 255u) * kx[1] + (int)(b1 >> 24) * kx[2] + (int)((b2 >> 16) & 255u) * kx[3];
            const int c2 = rnd + (int)((b0 >> 16) & 255u) * kx[0] + (int)((b1 >> 8) & 255u) * kx[1] + (int)(b2 & 255u) * kx[2] + (int)(b2 >> 24) * kx[3];
            hw[r * C + cx] = (unsigned)clip8(c0) | ((unsigned)clip8(c1) << 8) | ((unsigned)clip8(c2) << 16);
        }
    }
    __syncthreads();
    if (cx < C) {     // vertical pass: a lane owns a pixel column, three floats per store
        for (int ry = 0; ry < rows; ++ry) {
            const unsigned* hp = hw + s_rel[ry] * C + cx;
            const int cnt = s_cnt[ry];
            int c0 = 1 << (PRECISION_BITS - 1), c1 = c0, c2 = c0;
            for (int y = 0; y < cnt; ++y) {
                const unsigned v = hp[y * C];
                const int k = s_k[ry][y];
                c0 += (int)(v & 255u) * k;
                c1 += (int)((v >> 8) & 255u) * k;
                c2 += (int)((v >> 16) & 255u) * k;
            }
            float* dst = img + ((int64_t)(row0 + ry + 3) * CP + 3 + cx) * 3;
            // streaming stores: 1.33 GB that the stem reads once, from HBM -- keeping it out of L2 is worth 15 % of this write-bound kernel
            // (0.362 -> 0.306 ms, profiles/r06_ab_preproc_words.txt; the same hint on the window kernels' outputs measured slower)
            __builtin_nontemporal_store(s_lut[0][clip8(c0)], dst);
            __builtin_nontemporal_store(s_lut[1][clip8(c1)], dst + 1);
            __builtin_nontemporal_store(s_lut[2][clip8(c2)], dst + 2);
        }
    }
}

// ---- rgb: bilinear S -> R, centre crop C, normalise.  grid (frames, row tiles of 16 output rows) -------
__global__ void __launch_bounds__(256)
preproc_rgb_kernel(const uint8_t* __restrict__ frames, int S, int R, int C, int ksize, const int* __restrict__ bounds,
                   const int* __restrict__ kk, float mean0, float mean1, float mean2, float* __restrict__ out, int nchw) {
    const int64_t n = blockIdx.x;
    const int row0 = blockIdx.y * 16;
    const int off = (int)rintf((R - C) / 2.0f);  // CenterCrop: int(round((256-224)/2.)) = 16
    const uint8_t* src = frames + n * (int64_t)S * S * 3;
    const float mean[3] = {mean0, mean1, mean2};
    if (nchw == 2) {
        // the 3-pixel zero border of this block's 16 rows (left / right), and the three full rows above / below by the first / last block
        const int CP = C + 6;
        float* img = out + n * (int64_t)CP * CP * 3;
        for (int i = threadIdx.x; i < 16 * 18; i += 256) {
            const int ry = i / 18, e = i - ry * 18;
            const int oy = row0 + ry;
            if (oy < C) img[((int64_t)(oy + 3) * CP + (e < 9 ? 0 : C + 3)) * 3 + (e < 9 ? e : e - 9)] = 0.f;
        }
        if (blockIdx.y == 0)
            for (int i = threadIdx.x; i < 3 * CP * 3; i += 256) img[i] = 0.f;
        if (blockIdx.y == gridDim.y - 1)
            for (int i = threadIdx.x; i < 3 * CP * 3; i += 256) img[(int64_t)(C + 3) * CP * 3 + i] = 0.f;
    }
    for (int i = threadIdx.x; i < 16 * C; i += 256) {
        const int ry = i / C, cx = i - ry * C;
        const int oy = row0 + ry;
        if (oy >= C) break;
        const int yy = oy + off, xx = cx + off;                 // position in the R x R resized image
        const int ymin = bounds[yy * 2], ycnt = bounds[yy * 2 + 1];
        const int xmin = bounds[xx * 2], xcnt = bounds[xx * 2 + 1];
        const int* ky = kk + yy * ksize;
        const int* kx = kk + xx * ksize;
        int accv[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
        for (int y = 0; y < ycnt; ++y) {
            // horizontal pass value of input row (ymin + y) at column xx: uint8 after clip8, as PIL stores it
            int acch[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
            const uint8_t* rowp = src + ((int64_t)(ymin + y) * S + xmin) * 3;
            for (int x = 0; x < xcnt; ++x) {
                const int w = kx[x];
                acch[0] += (int)rowp[x * 3] * w;
                acch[1] += (int)rowp[x * 3 + 1] * w;
                acch[2] += (int)rowp[x * 3 + 2] * w;
            }
            const int w = ky[y];
            accv[0] += clip8(acch[0]) * w;
            accv[1] += clip8(acch[1]) * w;
            accv[2] += clip8(acch[2]) * w;
        }
        float v[3];
        {
            // ToTensor (/255), *255.0, -mean are three separately rounded fp32 operations in the reference; HIP's
            // default -ffp-contract=fast would fuse the last two into one FMA (1-ulp differences), so contraction
            // is switched off for this block.
#pragma clang fp contract(off)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float q = (float)clip8(accv[c]) / 255.0f;
                const float m255 = q * 255.0f;
                v[c] = m255 - mean[c];
            }
        }
        if (nchw == 2) {
            // zero-bordered packed NHWC3 [n][C + 6][C + 6][3]: the stem's padding lives in memory (conv_mfma.hip KMODE 5)
            const int CP = C + 6;
            float* o = out + ((n * CP + oy + 3) * (int64_t)CP + cx + 3) * 3;
            o[0] = v[0];
            o[1] = v[1];
            o[2] = v[2];
        } else if (nchw) {
            float* o = out + n * 3 * (int64_t)C * C + (int64_t)oy * C + cx;
            o[0] = v[0];
            o[(int64_t)C * C] = v[1];
            o[2 * (int64_t)C * C] = v[2];
        } else {
            reinterpret_cast<float4*>(out)[(n * C + oy) * (int64_t)C + cx] = float4{v[0], v[1], v[2], 0.f};
        }
    }
}

}  // namespace mm

struct mm_preproc {
    int in_size, gray_size, resize, crop;
    float mean[3];
    mm::ResampleTable lan, bil;
    int *d_lan_bounds = nullptr, *d_lan_kk = nullptr, *d_bil_bounds = nullptr, *d_bil_kk = nullptr;
    int rgb3_max_in = 0;   // most input rows a 16-row block of preproc_rgb3_kernel touches, from the bounds table itself
    int device = 0;
};

extern "C" {

int mm_preproc_host_coeffs(int in_size, int out_size, int filter, int* ksize, int* bounds, int* kk, int kk_capacity) {
    if (!ksize) return MM_ERR_INVALID_ARG;
    mm::ResampleTable t;
    int rc = mm::build_resample_table(in_size, out_size, filter, t);
    if (rc != MM_OK) return rc;
    *ksize = t.ksize;
    if (bounds && kk) {
        if (kk_capacity < (int)t.kk.size()) return MM_ERR_WORKSPACE;
        for (size_t i = 0; i < t.bounds.size(); ++i) bounds[i] = t.bounds[i];
        for (size_t i = 0; i < t.kk.size(); ++i) kk[i] = t.kk[i];
    }
    return MM_OK;
}

int mm_preproc_create(mm_preproc_t** out, int in_size, int gray_size, int resize, int crop, const float* mean3) {
    if (!out) return MM_ERR_INVALID_ARG;
    *out = nullptr;
    if (in_size <= 0 || gray_size <= 0 || resize < crop || crop <= 0 || !mean3 || in_size * in_size + in_size * gray_size > 160 * 1024)
        return MM_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return MM_ERR_NO_DEVICE;
    mm_preproc* h = new (std::nothrow) mm_preproc();
    if (!h) return MM_ERR_INVALID_ARG;
    h->in_size = in_size; h->gray_size = gray_size; h->resize = resize; h->crop = crop;
    h->device = mm::current_device_or(0);
    for (int c = 0; c < 3; ++c) h->mean[c] = mean3[c];
    int rc = mm::build_resample_table(in_size, gray_size, 1, h->lan);
    if (rc == MM_OK) rc = mm::build_resample_table(in_size, resize, 0, h->bil);
    auto up = [&](const std::vector<int>& v, int** d) -> int {
        MM_HIP(hipMalloc((void**)d, v.size() * sizeof(int)));
        MM_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
        return MM_OK;
    };
    if (rc == MM_OK) rc = up(h->lan.bounds, &h->d_lan_bounds);
    if (rc == MM_OK) rc = up(h->lan.kk, &h->d_lan_kk);
    if (rc == MM_OK) rc = up(h->bil.bounds, &h->d_bil_bounds);
    if (rc == MM_OK) rc = up(h->bil.kk, &h->d_bil_kk);
    if (rc == MM_OK) {
        // exact LDS need of preproc_rgb3_kernel: the kernel stages n_in = bounds[last].min + bounds[last].count - bounds[first].min
        // input rows per block of RGB_ROWS cropped output rows; take the maximum over the blocks the launch will have
        const int off = (int)rintf((resize - crop) / 2.0f);
        for (int row0 = 0; row0 < crop; row0 += mm::RGB_ROWS) {
            const int rows = crop - row0 < mm::RGB_ROWS ? crop - row0 : mm::RGB_ROWS;
            const int yf = row0 + off, yl = row0 + rows - 1 + off;
            const int n_in = h->bil.bounds[yl * 2] + h->bil.bounds[yl * 2 + 1] - h->bil.bounds[yf * 2];
            if (n_in > h->rgb3_max_in) h->rgb3_max_in = n_in;
        }
    }
    if (rc != MM_OK) {
        mm_preproc_destroy(h);
        return rc;
    }
    *out = h;
    return MM_OK;
}

int mm_preproc_destroy(mm_preproc_t* h) {
    if (!h) return MM_OK;
    for (int* p : {h->d_lan_bounds, h->d_lan_kk, h->d_bil_bounds, h->d_bil_kk})
        if (p) (void)hipFree(p);
    delete h;
    return MM_OK;
}

int mm_preproc_forward(mm_preproc_t* h, const uint8_t* frames, int64_t n, float* gray_out, float* rgb_out, int rgb_nchw,
                       void* stream_) {
    if (!h || n < 0 || (n > 0 && !frames) || (!gray_out && !rgb_out)) return MM_ERR_INVALID_ARG;
    if (n == 0) return MM_OK;
    MM_CHECK_DEVICE(h);
    hipStream_t s = (hipStream_t)stream_;
    if (gray_out) {
        const int S = h->in_size, G = h->gray_size;
        // word-wide LDS form when it applies (MM_PREPROC_WORDS=0: the byte form, for the A/B; read once)
        static const bool words_on = !(getenv("MM_PREPROC_WORDS") && atoi(getenv("MM_PREPROC_WORDS")) == 0);
        const int lds_w = (((S * S / 4 + 8 + 3) & ~3) + ((S * G + 3) & ~3) + (mm::GRAY_KW + 2) * G) * 4;
        const bool words = words_on && h->lan.ksize <= mm::GRAY_KW && (S * S) % 4 == 0 && (reinterpret_cast<uintptr_t>(frames) & 3) == 0 && lds_w <= 64 * 1024;
        const int lds = words ? lds_w : S * S + S * G;
        if (lds > 64 * 1024)
            MM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mm::preproc_gray_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        mm::prof_before(4, (double)n * (3.0 * h->in_size * h->in_size + 4.0 * h->gray_size * h->gray_size), s, "preproc_gray");
        if (words)
            hipLaunchKernelGGL(mm::preproc_gray_words_kernel, dim3((unsigned)n), dim3(256), lds, s, frames, S, G, h->lan.ksize, h->d_lan_bounds,
                               h->d_lan_kk, gray_out);
        else
            hipLaunchKernelGGL(mm::preproc_gray_kernel, dim3((unsigned)n), dim3(256), lds, s, frames, h->in_size, h->gray_size,
                               h->lan.ksize, h->d_lan_bounds, h->d_lan_kk, gray_out);
        mm::prof_after(4, s);
        MM_LAUNCH_CHECK();
    }
    if (rgb_out && rgb_nchw == 2) {
        // rows of the block after the horizontal pass: rgb3_max_in (exact, from the bounds table at create) input rows x crop
        // columns x 3 bytes of LDS; the kernel's tap arrays hold four vertical taps
        const int max_in = h->rgb3_max_in;
        const int64_t rgb3_lds = (((int64_t)max_in * h->crop * 3 + 15) & ~15) + (int64_t)max_in * h->in_size * 3 + 16;
        if (max_in <= 0 || max_in > mm::RGB_MAX_IN || h->bil.ksize > 4 || rgb3_lds > 60 * 1024)
            return MM_ERR_UNSUPPORTED;   // (a down-scaling resize: not the reference's 112 -> 256)
        dim3 grid((unsigned)n, (unsigned)((h->crop + mm::RGB_ROWS - 1) / mm::RGB_ROWS));
        mm::prof_before(4, (double)n * (3.0 * h->in_size * h->in_size + 12.0 * (h->crop + 6) * (h->crop + 6)), s, "preproc_rgb3");
        // word-wide LDS form when it applies (MM_PREPROC_WORDS=0: the byte form, for the A/B; read once)
        static const bool rgb_words_on = !(getenv("MM_PREPROC_WORDS") && atoi(getenv("MM_PREPROC_WORDS")) == 0);
        const int64_t lds_w = ((((int64_t)max_in * h->crop + 3) & ~3) + (int64_t)max_in * (h->in_size * 3 / 4) + 4) * 4;
        if (rgb_words_on && h->crop <= 256 && (h->in_size * 3) % 4 == 0 && (reinterpret_cast<uintptr_t>(frames) & 3) == 0 && lds_w <= 56 * 1024)
            hipLaunchKernelGGL(mm::preproc_rgb3_words_kernel, grid, dim3(256), (int)lds_w, s, frames, h->in_size, h->resize, h->crop,
                               h->bil.ksize, h->d_bil_bounds, h->d_bil_kk, h->mean[0], h->mean[1], h->mean[2], rgb_out, max_in);
        else
            hipLaunchKernelGGL(mm::preproc_rgb3_kernel, grid, dim3(256), (int)rgb3_lds, s, frames, h->in_size, h->resize, h->crop,
                               h->bil.ksize, h->d_bil_bounds, h->d_bil_kk, h->mean[0], h->mean[1], h->mean[2], rgb_out, max_in);
        mm::prof_after(4, s);
        MM_LAUNCH_CHECK();
    } else if (rgb_out) {
        dim3 grid((unsigned)n, (unsigned)((h->crop + 15) / 16));
        mm::prof_before(4, (double)n * (3.0 * h->in_size * h->in_size + (rgb_nchw ? 12.0 : 16.0) * h->crop * h->crop), s, "preproc_rgb");
        hipLaunchKernelGGL(mm::preproc_rgb_kernel, grid, dim3(256), 0, s, frames, h->in_size, h->resize, h->crop, h->bil.ksize,
                           h->d_bil_bounds, h->d_bil_kk, h->mean[0], h->mean[1], h->mean[2], rgb_out, rgb_nchw);
        mm::prof_after(4, s);
        MM_LAUNCH_CHECK();
    }
    return MM_OK;
}

}  // extern "C"
