// Generic Phase_Difference_Extractor.extract (api/phase_difference_extractor.py:93-134) for any plane size up to 4096
// pixels, any window length and any number of bands: the same arithmetic as phase_window.hip (atan2 / magnitude + 1e-10,
// fmod unwrap over time, amplitude-weighted separable 11-tap blur with zero padding, temporal difference, spatial-mean
// removal, clamp to +-5 pi), written for generality instead of speed.  The inference pipeline never comes here: it is
// what makes the drop-in class usable with other constructor arguments than api/tester.py's (height / nbands / levels /
// frame size / window length), together with the general pyramid of scfpyr.hip.
#include "mm_common.h"
#include "phase_math.h"

namespace mm {
namespace {

constexpr int GTAP = 11, GR = 5, GNT = 256, GMAXPP = 16;   // 16 pixels per thread x 256 threads = 4096 pixels

__device__ constexpr float g_gauss[GTAP] = {0.043936934322118759f, 0.1353352814912796f, 0.32465246319770813f,
                                            0.60653066635131836f,  0.88249689340591431f, 1.0f,
                                            0.88249689340591431f,  0.60653066635131836f, 0.32465246319770813f,
                                            0.1353352814912796f,   0.043936934322118759f};

// coeff [planes][P][R][C][2] -> out [planes][P-1][R][C]; one workgroup per plane set (b, band)
__global__ void __launch_bounds__(GNT)
phase_extract_generic_kernel(const float* __restrict__ coeff, float* __restrict__ out, float* __restrict__ den_out, int P, int R,
                             int C) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int npix = R * C;
    float* in_num = sm;
    float* in_den = sm + npix;
    float* tmp_num = sm + 2 * npix;
    float* tmp_den = sm + 3 * npix;
    __shared__ float red[GNT / 64], red2[GNT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* cp = coeff + (size_t)blockIdx.x * P * npix * 2;
    float* op = out + (size_t)blockIdx.x * (P - 1) * npix;
    const float LIM = 5.f * 3.14159265358979323846f;
    float prev_ph[GMAXPP], cum[GMAXPP], prev_blur[GMAXPP], dcur[GMAXPP];
    float* dp = den_out ? den_out + (size_t)blockIdx.x * P * npix : nullptr;
    for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int k = 0; k < GMAXPP; ++k) {
            const int i = tid + k * GNT;
            if (i < npix) {
                const float2 c = reinterpret_cast<const float2*>(cp + (size_t)p * npix * 2)[i];
                float ph, mag;
                to_polar(c.x, c.y, ph, mag);
                float up = ph;
                if (p == 0) cum[k] = 0.f;
                else up = unwrap_step(ph, prev_ph[k], cum[k]);
                prev_ph[k] = ph;
                in_num[i] = mag * up;
                in_den[i] = mag;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < GMAXPP; ++k) {   // blur along the last dimension
            const int i = tid + k * GNT;
            if (i < npix) {
                const int y = i / C, x = i - y * C;
                float sn = 0.f, sd = 0.f;
#pragma unroll
                for (int t = 0; t < GTAP; ++t) {
                    const int xx = x + t - GR;
                    const bool ok = xx >= 0 && xx < C;
                    sn = fmaf(g_gauss[t], ok ? in_num[y * C + xx] : 0.f, sn);
                    sd = fmaf(g_gauss[t], ok ? in_den[y * C + xx] : 0.f, sd);
                }
                tmp_num[i] = sn;
                tmp_den[i] = sd;
            }
        }
        __syncthreads();
        float part = 0.f, bsum = 0.f;
#pragma unroll
        for (int k = 0; k < GMAXPP; ++k) {   // blur along the first dimension, ratio, temporal difference
            const int i = tid + k * GNT;
            if (i < npix) {
                const int y = i / C, x = i - y * C;
                float sn = 0.f, sd = 0.f;
#pragma unroll
                for (int t = 0; t < GTAP; ++t) {
                    const int yy = y + t - GR;
                    const bool ok = yy >= 0 && yy < R;
                    sn = fmaf(g_gauss[t], ok ? tmp_num[yy * C + x] : 0.f, sn);
                    sd = fmaf(g_gauss[t], ok ? tmp_den[yy * C + x] : 0.f, sd);
                }
                const float blur = sn / sd;
                bsum += blur;
                if (p > 0) {
                    dcur[k] = blur - prev_blur[k];
                    part += dcur[k];
                }
                prev_blur[k] = blur;
            }
        }
        if (dp) {   // training-side option (Aff-wild-exps/utils.py:410-412,417): the mean-centred denoised phase itself
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) bsum += __shfl_down(bsum, off, 64);
            if (lane == 0) red2[wave] = bsum;
            __syncthreads();
            float s2 = 0.f;
#pragma unroll
            for (int w = 0; w < GNT / 64; ++w) s2 += red2[w];
            const float bmean = s2 / (float)npix;
#pragma unroll
            for (int k = 0; k < GMAXPP; ++k) {
                const int i = tid + k * GNT;
                if (i < npix) dp[(size_t)p * npix + i] = prev_blur[k] - bmean;   // prev_blur holds this frame's value now
            }
        }
        if (p > 0) {   // spatial mean of this difference plane, then mean removal + clamp (:130-133)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
            if (lane == 0) red[wave] = part;
            __syncthreads();
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < GNT / 64; ++w) s += red[w];
            const float mean = s / (float)npix;
#pragma unroll
            for (int k = 0; k < GMAXPP; ++k) {
                const int i = tid + k * GNT;
                if (i < npix) op[(size_t)(p - 1) * npix + i] = fminf(fmaxf(dcur[k] - mean, -LIM), LIM);
            }
        }
        __syncthreads();   // red[] and the LDS planes are rewritten by the next frame
    }
}

// The same arithmetic for planes of ANY size (round 5): the four blur planes and the per-pixel scan state (previous phase, running
// unwrap correction, previous blurred value, current difference) live in a caller-provided global workspace (8 floats per pixel and
// plane set) instead of LDS / registers; one workgroup per plane set walks its pixels with a stride loop.  Same functions, same tap
// order, same reduction tree as the LDS kernel: bit-identical results on plane sizes both kernels take (GPU test).  Written for
// completeness of the drop-in `extract` (api/phase_difference_extractor.py:93-134 takes any W x H), not for speed.
__global__ void __launch_bounds__(GNT)
phase_extract_generic_big_kernel(const float* __restrict__ coeff, float* __restrict__ out, float* __restrict__ den_out, float* __restrict__ ws,
                                 int P, int R, int C) {
    const int npix = R * C;
    float* base = ws + (size_t)blockIdx.x * 8 * npix;
    float* in_num = base;
    float* in_den = base + npix;
    float* tmp_num = base + 2 * (size_t)npix;
    float* tmp_den = base + 3 * (size_t)npix;
    float* prev_ph = base + 4 * (size_t)npix;
    float* cum = base + 5 * (size_t)npix;
    float* prev_blur = base + 6 * (size_t)npix;
    float* dcur = base + 7 * (size_t)npix;
    __shared__ float red[GNT / 64], red2[GNT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* cp = coeff + (size_t)blockIdx.x * P * npix * 2;
    float* op = out + (size_t)blockIdx.x * (P - 1) * npix;
    float* dp = den_out ? den_out + (size_t)blockIdx.x * P * npix : nullptr;
    const float LIM = 5.f * 3.14159265358979323846f;
    for (int p = 0; p < P; ++p) {
        for (int i = tid; i < npix; i += GNT) {
            const float2 c = reinterpret_cast<const float2*>(cp + (size_t)p * npix * 2)[i];
            float ph, mag;
            to_polar(c.x, c.y, ph, mag);
            float up = ph;
            float cm = 0.f;
            if (p > 0) {
                cm = cum[i];
                up = unwrap_step(ph, prev_ph[i], cm);
            }
            cum[i] = cm;
            prev_ph[i] = ph;
            in_num[i] = mag * up;
            in_den[i] = mag;
        }
        __syncthreads();          // workgroup-scope fence + barrier: the planes are global memory written and read by this workgroup only
        for (int i = tid; i < npix; i += GNT) {
            const int y = i / C, x = i - y * C;
            float sn = 0.f, sd = 0.f;
#pragma unroll
            for (int t = 0; t < GTAP; ++t) {
                const int xx = x + t - GR;
                const bool ok = xx >= 0 && xx < C;
                sn = fmaf(g_gauss[t], ok ? in_num[y * C + xx] : 0.f, sn);
                sd = fmaf(g_gauss[t], ok ? in_den[y * C + xx] : 0.f, sd);
            }
            tmp_num[i] = sn;
            tmp_den[i] = sd;
        }
        __syncthreads();
        float part = 0.f, bsum = 0.f;
        for (int i = tid; i < npix; i += GNT) {
            const int y = i / C, x = i - y * C;
            float sn = 0.f, sd = 0.f;
#pragma unroll
            for (int t = 0; t < GTAP; ++t) {
                const int yy = y + t - GR;
                const bool ok = yy >= 0 && yy < R;
                sn = fmaf(g_gauss[t], ok ? tmp_num[yy * C + x] : 0.f, sn);
                sd = fmaf(g_gauss[t], ok ? tmp_den[yy * C + x] : 0.f, sd);
            }
            const float blur = sn / sd;
            bsum += blur;
            if (p > 0) {
                const float d = blur - prev_blur[i];
                dcur[i] = d;
                part += d;
            }
            prev_blur[i] = blur;
        }
        if (dp) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) bsum += __shfl_down(bsum, off, 64);
            if (lane == 0) red2[wave] = bsum;
            __syncthreads();
            float s2 = 0.f;
#pragma unroll
            for (int w = 0; w < GNT / 64; ++w) s2 += red2[w];
            const float bmean = s2 / (float)npix;
            for (int i = tid; i < npix; i += GNT) dp[(size_t)p * npix + i] = prev_blur[i] - bmean;   // own writes: same thread, same index
        }
        if (p > 0) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
            if (lane == 0) red[wave] = part;
            __syncthreads();
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < GNT / 64; ++w) s += red[w];
            const float mean = s / (float)npix;
            for (int i = tid; i < npix; i += GNT) op[(size_t)(p - 1) * npix + i] = fminf(fmaxf(dcur[i] - mean, -LIM), LIM);
        }
        __syncthreads();
    }
}

}  // namespace
}  // namespace mm

extern "C" int64_t mm_phase_extract_generic_workspace_bytes(int64_t planes, int P, int R, int C) {
    if (planes < 0 || P < 2 || R <= 0 || C <= 0) return MM_ERR_INVALID_ARG;
    if ((int64_t)R * C > 0x7ffffff) return MM_ERR_UNSUPPORTED;
    return (int64_t)R * C <= (int64_t)mm::GNT * mm::GMAXPP ? 0 : planes * (int64_t)R * C * 8 * (int64_t)sizeof(float);
}

extern "C" int mm_phase_extract_generic_ws(const float* coeff, int64_t planes, int P, int R, int C, float* out, float* denoised,
                                           void* workspace, int64_t workspace_bytes, void* stream) {
    if (planes < 0 || P < 2 || R <= 0 || C <= 0 || (planes > 0 && (!coeff || !out))) return MM_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes <= 0) return mm_phase_extract_generic(coeff, planes, P, R, C, out, denoised, stream);
    if ((int64_t)R * C > 0x7ffffff || planes > 0x7fffffff) return MM_ERR_UNSUPPORTED;
    if (workspace_bytes < planes * (int64_t)R * C * 8 * (int64_t)sizeof(float)) return MM_ERR_WORKSPACE;
    if (planes == 0) return MM_OK;
    hipLaunchKernelGGL(mm::phase_extract_generic_big_kernel, dim3((unsigned)planes), dim3(mm::GNT), 0, (hipStream_t)stream, coeff, out, denoised,
                       (float*)workspace, P, R, C);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

extern "C" int mm_phase_extract_generic(const float* coeff, int64_t planes, int P, int R, int C, float* out, float* denoised,
                                        void* stream) {
    if (planes < 0 || P < 2 || R <= 0 || C <= 0 || (planes > 0 && (!coeff || !out))) return MM_ERR_INVALID_ARG;
    if ((int64_t)R * C > (int64_t)mm::GNT * mm::GMAXPP || planes > 0x7fffffff) return MM_ERR_UNSUPPORTED;
    if (planes == 0) return MM_OK;
    const size_t lds = (size_t)4 * R * C * sizeof(float);
    MM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mm::phase_extract_generic_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(mm::phase_extract_generic_kernel, dim3((unsigned)planes), dim3(mm::GNT), lds, (hipStream_t)stream, coeff, out,
                       denoised, P, R, C);
    MM_LAUNCH_CHECK();
    return MM_OK;
}
