// Shared declarations for libmimamo_hip.so (gfx950 only; no CUDA/compat paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "../../include/mimamo_hip.h"

namespace mm {

extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) {
    g_last_hip_error = (int)e;
    return MM_ERR_HIP;
}

#define MM_HIP(call)                                  \
    do {                                              \
        hipError_t _e = (call);                       \
        if (_e != hipSuccess) return mm::hip_fail(_e); \
    } while (0)

#define MM_LAUNCH_CHECK()                             \
    do {                                              \
        hipError_t _e = hipGetLastError();            \
        if (_e != hipSuccess) return mm::hip_fail(_e); \
    } while (0)

// A handle's tables / weights live on the device that was current at mm_*_create; using it with another current
// device would hand the kernels pointers of a different GPU.
inline int current_device_or(int fallback) {
    int d = fallback;
    return hipGetDevice(&d) == hipSuccess ? d : fallback;
}
#define MM_CHECK_DEVICE(h)                                                   \
    do {                                                                     \
        if (mm::current_device_or(-1) != (h)->device) return MM_ERR_INVALID_ARG; \
    } while (0)

// XCD-aware work order: workgroup b runs on XCD b % 8 (each XCD has its own L2).  Returns the logical work item of block b so that
// every XCD walks a CONTIGUOUS run of items: neighbours that share operands (n-tiles of one m-tile, consecutive 13-frame windows)
// then share an L2 instead of being fetched by all eight.
__device__ __forceinline__ int xcd_contiguous(int b, int nblk) {
    const int xcd = b & 7, idx = b >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- optional launch timing (profile.hip)
bool prof_enabled();
void prof_before(int cat, double work, hipStream_t s, const char* tag = nullptr);
void prof_after(int cat, hipStream_t s);

// ---- host-side pyramid constants (mm_masks.cpp) -------------------------------------
struct PyramidConfig {
    int size;          // un-mirrored side (48)
    int height, nbands, scale_factor;
};

// Real mask product on the level grid (fftshift-ed), float64; level in {1,2}.
// side = 2*size >> (level-1).  crop = bounds applied going into that level.
int host_level_mask(const PyramidConfig& c, int level, int band, std::vector<double>& out, int& side, int crop[2]);

// Complex band tables in the kernels' half-plane layout (see pyramid.hip), fp32 (re,im interleaved).
struct PyramidTables {
    std::vector<float> dct;      // [48][48]  D[f][m] = 2 cos(pi f (2m+1)/96)
    std::vector<float> ec, es;   // [48][48]  cos / sin (2 pi f q / 96)
    std::vector<float> m1[2];    // level-1 band masks, complex: band0 [96][48][2], band1 [48][96][2]
    std::vector<float> m2[2];    // level-2 band masks, complex: band0 [48][24][2], band1 [24][48][2]
};
int build_pyramid_tables(const PyramidConfig& c, PyramidTables& t);

// Full pyramid (SCFpyr_PyTorch.build's whole return list) for a general square n0 x n0 image: one complex float64
// multiplier table per output, in FFT (un-shifted) index order of that output's grid, with the ifft 1/side^2 and the
// (-i)^(nbands-1) band factor folded in.  Order: hi-pass residual, bands level-major, low-pass residual.
struct ScfOutput {
    int side;
    int is_complex;             // bands: 1; residuals (real part kept): 0
    std::vector<double> table;  // [side][side][2]
};
int build_scf_full_tables(int n0, int height, int nbands, int scale_factor, std::vector<ScfOutput>& outs);

}  // namespace mm

struct mm_pyramid {
    mm::PyramidConfig cfg;
    int device;
    float* d_tables;   // one HBM allocation, sub-tables at fixed offsets (pyramid.hip)
    int64_t table_floats;
};
