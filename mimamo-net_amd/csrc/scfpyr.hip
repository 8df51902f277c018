// General complex steerable pyramid: the whole return list of SCFpyr_PyTorch.build
// (api/steerable/SCFpyr_PyTorch.py:70-208) -- hi-pass residual, every oriented band of every level, low-pass
// residual -- for arbitrary (non-mirrored) square images up to 1024x1024, even or odd (the intermediate of each 2-D transform lives
// in LDS up to 96x96, in a global scratch above), any height / number of bands, fp32 or fp64 I/O.
//
// This is the API-completeness path, not the hot path: the inference pipeline uses pyramid.hip, which exploits the
// mirror symmetry of its input and keeps only the coefficients the phase stage consumes.  Here every transform is a
// plain DFT-by-summation with float64 accumulation and float64 twiddles (gfx950 issues fp64 FMAs at the fp32 rate),
// so precision=32 results are exact to the final rounding and precision=64 matches a float64 FFT to ~1e-15.
//
//   forward : F[kr][kc] = sum_{r,c} x[r][c] e^{-2 pi i (kr r + kc c)/n0}        one workgroup per image, rows then
//                                                                                columns through LDS
//   inverse : out_o[y][x] = sum_{a,b} F[a' ][b'] T_o[a][b] e^{+2 pi i (a y + b x)/m}   one workgroup per
//             (image, output o); T_o = the reference's mask product for that output on its (cropped) grid, in FFT
//             order, with 1/m^2 and the (-i)^(nbands-1) band factor folded in (mm_masks.cpp); a' = a's signed
//             frequency taken modulo n0 (the reference's centre crops of the shifted spectrum keep signed frequency).
#include <algorithm>
#include <cmath>
#include <new>
#include "mm_common.h"

struct mm_scfpyr {
    int size, height, nbands, scale_factor;
    int device;
    int n_out;
    std::vector<int> side, is_complex;
    std::vector<double2*> d_table;  // per output, device
    std::vector<double2*> d_tw;     // per output: e^{+2 pi i k / side}, [side] (outputs of one side share the table)
    std::vector<double2*> d_tw_own; // the distinct twiddle allocations
    double2* d_twiddle;             // [size]  e^{+2 pi i k / size}
};

namespace mm {
namespace {

constexpr int kScfThreads = 256;
constexpr int kScfLdsSide = 96;   // LDS-resident intermediate: side^2 complex float64 = 147 456 B at 96
constexpr int kScfMaxSide = 1024; // above kScfLdsSide the intermediate goes through a global scratch; O(side^3) per image: completeness, not speed

template <typename TIn>
__global__ __launch_bounds__(kScfThreads) void scf_forward_kernel(const TIn* __restrict__ im, double2* __restrict__ F,
                                                                  const double2* __restrict__ tw, int n0, double2* scratch) {
    extern __shared__ __attribute__((aligned(16))) double2 sm[];
    double2* w = sm;                                                                     // [n0]
    double2* X1 = scratch ? scratch + (size_t)blockIdx.x * n0 * n0 : sm + n0;           // [n0][n0] row transforms
    for (int k = threadIdx.x; k < n0; k += kScfThreads) w[k] = tw[k];
    __syncthreads();
    const TIn* x = im + (size_t)blockIdx.x * n0 * n0;
    for (int idx = threadIdx.x; idx < n0 * n0; idx += kScfThreads) {
        const int r = idx / n0, k = idx - r * n0;
        double re = 0.0, imv = 0.0;
        int j = 0;
        for (int c = 0; c < n0; ++c) {
            const double v = (double)x[r * n0 + c];
            const double2 t = w[j];
            re = fma(v, t.x, re);
            imv = fma(-v, t.y, imv);  // e^{-i...}
            j += k;
            if (j >= n0) j -= n0;
        }
        X1[idx] = make_double2(re, imv);
    }
    __syncthreads();
    double2* Fo = F + (size_t)blockIdx.x * n0 * n0;
    for (int idx = threadIdx.x; idx < n0 * n0; idx += kScfThreads) {
        const int kr = idx / n0, kc = idx - kr * n0;
        double re = 0.0, imv = 0.0;
        int j = 0;
        for (int r = 0; r < n0; ++r) {
            const double2 a = X1[r * n0 + kc];
            const double2 t = w[j];
            // a * conj(t)
            re = fma(a.x, t.x, fma(a.y, t.y, re));
            imv = fma(a.y, t.x, fma(-a.x, t.y, imv));
            j += kr;
            if (j >= n0) j -= n0;
        }
        Fo[idx] = make_double2(re, imv);
    }
}

template <typename TOut>
__global__ __launch_bounds__(kScfThreads) void scf_inverse_kernel(const double2* __restrict__ F, const double2* __restrict__ T,
                                                                  const double2* __restrict__ tw, TOut* __restrict__ out,
                                                                  int n0, int m, int is_complex, double2* scratch) {
    extern __shared__ __attribute__((aligned(16))) double2 sm[];
    double2* w = sm;                                                                 // [m]   e^{+2 pi i k/m}: this level's own table
    double2* Y = scratch ? scratch + (size_t)blockIdx.x * m * m : sm + m;           // [m][m]
    for (int k = threadIdx.x; k < m; k += kScfThreads) w[k] = tw[k];
    __syncthreads();
    const double2* Fi = F + (size_t)blockIdx.x * n0 * n0;
    const int h = (m + 1) / 2;             // FFT indices [0, h) are the non-negative frequencies (odd m: one more than the negative ones)
    for (int idx = threadIdx.x; idx < m * m; idx += kScfThreads) {
        const int a = idx / m, x = idx - a * m;
        const int sa = a < h ? a : a - m + n0;  // signed frequency modulo n0
        const double2* Frow = Fi + (size_t)sa * n0;
        const double2* Trow = T + (size_t)a * m;
        double re = 0.0, imv = 0.0;
        int j = 0;
        for (int b = 0; b < m; ++b) {
            const int sb = b < h ? b : b - m + n0;
            const double2 f = Frow[sb], t = Trow[b];
            const double sr = f.x * t.x - f.y * t.y, si = f.x * t.y + f.y * t.x;
            const double2 e = w[j];
            re = fma(sr, e.x, fma(-si, e.y, re));
            imv = fma(sr, e.y, fma(si, e.x, imv));
            j += x;
            if (j >= m) j -= m;
        }
        Y[idx] = make_double2(re, imv);
    }
    __syncthreads();
    const size_t per = (size_t)m * m * (is_complex ? 2 : 1);
    TOut* o = out + (size_t)blockIdx.x * per;
    for (int idx = threadIdx.x; idx < m * m; idx += kScfThreads) {
        const int y = idx / m, x = idx - y * m;
        double re = 0.0, imv = 0.0;
        int j = 0;
        for (int a = 0; a < m; ++a) {
            const double2 v = Y[a * m + x];
            const double2 e = w[j];
            re = fma(v.x, e.x, fma(-v.y, e.y, re));
            imv = fma(v.x, e.y, fma(v.y, e.x, imv));
            j += y;
            if (j >= m) j -= m;
        }
        if (is_complex) {
            o[(size_t)idx * 2] = (TOut)re;
            o[(size_t)idx * 2 + 1] = (TOut)imv;
        } else {
            o[idx] = (TOut)re;
        }
    }
}

template <typename K>
int raise_lds(K kernel, size_t bytes) {
    MM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return MM_OK;
}

int scf_check(int size, int height, int nbands, int scale_factor) {
    if (size <= 0 || height < 1 || nbands < 1 || scale_factor < 1) return MM_ERR_INVALID_ARG;
    // SCFpyr_PyTorch.py:90-91
    if (height > (int)std::floor(std::log2((double)size)) - 2) return MM_ERR_TOO_SMALL;
    // nbands == 1 recurses forever in the reference (math_utils.py:79-84, quirk Q7); height < 2 has no residual pair
    if (nbands < 2 || nbands > 16 || height < 2 || size > kScfMaxSide) return MM_ERR_UNSUPPORTED;
    return MM_OK;
}

}  // namespace
}  // namespace mm

extern "C" {

int mm_scfpyr_host_table(int size, int height, int nbands, int scale_factor, int index, double* out, int* side,
                         int* is_complex) {
    if (!side || !is_complex) return MM_ERR_INVALID_ARG;
    int rc = mm::scf_check(size, height, nbands, scale_factor);
    if (rc != MM_OK) return rc;
    std::vector<mm::ScfOutput> outs;
    rc = mm::build_scf_full_tables(size, height, nbands, scale_factor, outs);
    if (rc != MM_OK) return rc;
    if (index < 0 || index >= (int)outs.size()) return MM_ERR_INVALID_ARG;
    *side = outs[index].side;
    *is_complex = outs[index].is_complex;
    if (out)
        for (size_t i = 0; i < outs[index].table.size(); ++i) out[i] = outs[index].table[i];
    return MM_OK;
}

int mm_scfpyr_create(mm_scfpyr_t** out, int size, int height, int nbands, int scale_factor) {
    if (!out) return MM_ERR_INVALID_ARG;
    *out = nullptr;
    int rc = mm::scf_check(size, height, nbands, scale_factor);
    if (rc != MM_OK) return rc;
    std::vector<mm::ScfOutput> outs;
    rc = mm::build_scf_full_tables(size, height, nbands, scale_factor, outs);
    if (rc != MM_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return MM_ERR_NO_DEVICE;
    mm_scfpyr* h = new (std::nothrow) mm_scfpyr();
    if (!h) return MM_ERR_INVALID_ARG;
    h->size = size; h->height = height; h->nbands = nbands; h->scale_factor = scale_factor;
    h->n_out = (int)outs.size();
    h->d_twiddle = nullptr;
    hipError_t e = hipGetDevice(&h->device);
    for (size_t i = 0; i < outs.size() && e == hipSuccess; ++i) {
        double2* d = nullptr;
        e = hipMalloc((void**)&d, outs[i].table.size() * sizeof(double));
        if (e == hipSuccess) {
            h->d_table.push_back(d);
            h->side.push_back(outs[i].side);
            h->is_complex.push_back(outs[i].is_complex);
            e = hipMemcpy(d, outs[i].table.data(), outs[i].table.size() * sizeof(double), hipMemcpyHostToDevice);
        }
    }
    auto upload_twiddle = [&](int n, double2** dst) {
        std::vector<double> tw((size_t)n * 2);
        const double pi = 3.14159265358979323846;
        for (int k = 0; k < n; ++k) {
            // exact symmetric reduction keeps e^{i pi/2 multiples} exact
            tw[2 * k] = std::cos(2.0 * pi * k / n);
            tw[2 * k + 1] = std::sin(2.0 * pi * k / n);
            if (4 * k == n) { tw[2 * k] = 0.0; tw[2 * k + 1] = 1.0; }
            if (2 * k == n) { tw[2 * k] = -1.0; tw[2 * k + 1] = 0.0; }
            if (4 * k == 3 * n) { tw[2 * k] = 0.0; tw[2 * k + 1] = -1.0; }
        }
        hipError_t er = hipMalloc((void**)dst, tw.size() * sizeof(double));
        if (er == hipSuccess) {
            er = hipMemcpy(*dst, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice);
            if (er != hipSuccess) {            // the caller only records a table it got hipSuccess for: free it here or it leaks
                (void)hipFree(*dst);
                *dst = nullptr;
            }
        }
        return er;
    };
    if (e == hipSuccess) e = upload_twiddle(size, &h->d_twiddle);
    // one twiddle table per distinct output side (a cropped level's side need not divide the image side: 100 -> 50 -> 25 -> 13)
    for (size_t i = 0; i < h->side.size() && e == hipSuccess; ++i) {
        double2* t = nullptr;
        for (size_t j = 0; j < i; ++j)
            if (h->side[j] == h->side[i]) { t = h->d_tw[j]; break; }
        if (!t) {
            if (h->side[i] == size) t = h->d_twiddle;
            else {
                e = upload_twiddle(h->side[i], &t);
                if (e == hipSuccess) h->d_tw_own.push_back(t);
            }
        }
        h->d_tw.push_back(t);
    }
    if (e != hipSuccess) {
        mm_scfpyr_destroy(h);
        return mm::hip_fail(e);
    }
    *out = h;
    return MM_OK;
}

int mm_scfpyr_destroy(mm_scfpyr_t* h) {
    if (!h) return MM_OK;
    for (double2* d : h->d_table)
        if (d) (void)hipFree(d);
    for (double2* d : h->d_tw_own)
        if (d) (void)hipFree(d);
    if (h->d_twiddle) (void)hipFree(h->d_twiddle);
    delete h;
    return MM_OK;
}

int mm_scfpyr_num_outputs(const mm_scfpyr_t* h) { return h ? h->n_out : MM_ERR_INVALID_ARG; }

int mm_scfpyr_output_info(const mm_scfpyr_t* h, int index, int* side, int* is_complex) {
    if (!h || index < 0 || index >= h->n_out || !side || !is_complex) return MM_ERR_INVALID_ARG;
    *side = h->side[index];
    *is_complex = h->is_complex[index];
    return MM_OK;
}

int64_t mm_scfpyr_workspace_bytes(const mm_scfpyr_t* h, int64_t n) {
    if (!h || n < 0) return MM_ERR_INVALID_ARG;
    // spectrum, plus (sides above the LDS-resident limit) one intermediate plane per image
    return n * (int64_t)h->size * h->size * (int64_t)sizeof(double2) * (h->size > mm::kScfLdsSide ? 2 : 1);
}

int mm_scfpyr_build(const mm_scfpyr_t* h, const void* images, int precision, int64_t n, void* const* outputs,
                    void* workspace, int64_t workspace_bytes, void* stream) {
    if (!h || (precision != 32 && precision != 64) || n < 0) return MM_ERR_INVALID_ARG;
    if (n == 0) return MM_OK;
    if (!images || !outputs || !workspace) return MM_ERR_INVALID_ARG;
    if (workspace_bytes < mm_scfpyr_workspace_bytes(h, n)) return MM_ERR_WORKSPACE;
    for (int i = 0; i < h->n_out; ++i)
        if (!outputs[i]) return MM_ERR_INVALID_ARG;
    MM_CHECK_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    double2* F = (double2*)workspace;
    const int n0 = h->size;
    double2* scratch = n0 > mm::kScfLdsSide ? F + (size_t)n * n0 * n0 : nullptr;   // [n][n0][n0], reused by every launch
    auto lds_bytes = [&](int m) { return ((scratch ? 0 : (size_t)m * m) + m) * sizeof(double2); };
    const size_t lds_f = lds_bytes(n0);
    // the largest request of any launch: an LDS-resident level (side <= 96: intermediate + twiddles) or the twiddles of the largest side
    const size_t lds_max = std::max(((size_t)mm::kScfLdsSide * mm::kScfLdsSide + mm::kScfLdsSide) * sizeof(double2),
                                    (size_t)mm::kScfMaxSide * sizeof(double2));
    const dim3 grid((unsigned)n), block(mm::kScfThreads);
    int rc;
    if (precision == 32) {
        if ((rc = mm::raise_lds(mm::scf_forward_kernel<float>, lds_f)) != MM_OK) return rc;
        hipLaunchKernelGGL(mm::scf_forward_kernel<float>, grid, block, lds_f, s, (const float*)images, F, h->d_twiddle, n0, scratch);
    } else {
        if ((rc = mm::raise_lds(mm::scf_forward_kernel<double>, lds_f)) != MM_OK) return rc;
        hipLaunchKernelGGL(mm::scf_forward_kernel<double>, grid, block, lds_f, s, (const double*)images, F, h->d_twiddle, n0, scratch);
    }
    MM_LAUNCH_CHECK();
    for (int i = 0; i < h->n_out; ++i) {
        const int m = h->side[i];
        // a level whose grid fits keeps its intermediate in LDS even when the full-size levels do not
        double2* sc = m > mm::kScfLdsSide ? scratch : nullptr;
        const size_t lds_i = ((sc ? 0 : (size_t)m * m) + m) * sizeof(double2);
        if (precision == 32) {
            if ((rc = mm::raise_lds(mm::scf_inverse_kernel<float>, lds_max)) != MM_OK) return rc;
            hipLaunchKernelGGL(mm::scf_inverse_kernel<float>, grid, block, lds_i, s, F, h->d_table[i], h->d_tw[i],
                               (float*)outputs[i], n0, m, h->is_complex[i], sc);
        } else {
            if ((rc = mm::raise_lds(mm::scf_inverse_kernel<double>, lds_max)) != MM_OK) return rc;
            hipLaunchKernelGGL(mm::scf_inverse_kernel<double>, grid, block, lds_i, s, F, h->d_table[i], h->d_tw[i],
                               (double*)outputs[i], n0, m, h->is_complex[i], sc);
        }
        MM_LAUNCH_CHECK();
    }
    return MM_OK;
}

}  // extern "C"
