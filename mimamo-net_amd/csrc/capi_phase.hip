// extern "C" entry points: steerable pyramid + phase difference (see include/mimamo_hip.h).
#include <cmath>
#include <new>
#include "mm_common.h"

namespace mm {
thread_local int g_last_hip_error = 0;

int launch_pyramid(const mm_pyramid* h, const float* frames, int64_t n, int64_t group, float* c1, int64_t gs1,
                   int64_t is1, int64_t bs1, float* c2, int64_t gs2, int64_t is2, int64_t bs2, int polar, hipStream_t stream);
int pyramid_table_floats();
int pack_pyramid_tables(const PyramidTables& t, std::vector<float>& packed);
int launch_phase_window(const float* coeff, const int32_t* ids, int64_t img_stride, int64_t band_stride, int64_t J,
                        int W, float* out, int out_nhwc, int out_cstride, int out_coffset, int polar, hipStream_t stream);

int64_t phase_frames_floats(int W, int64_t n);
int launch_pyramid_frames(const mm_pyramid* h, const float* frames, int64_t n, float* f1, float* f2, hipStream_t stream);
bool phase_window2_pair_applies(int out0_nhwc, int out1_nhwc);
int launch_phase_window2_pair(const float* fr1, const float* fr2, const int32_t* ids, int64_t n, int64_t J, float* out0, int out0_nhwc,
                              int out0_cstride, int out0_coffset, float* out1, int out1_nhwc, int out1_cstride, int out1_coffset,
                              hipStream_t s);
int launch_phase_window2(const float* fr, const int32_t* ids, int64_t n, int64_t J, int W, float* out, int out_nhwc, int out_cstride,
                         int out_coffset, hipStream_t s);

static int check_config(int size, int height, int nbands, int scale_factor) {
    if (size <= 0 || height < 1 || nbands < 1 || scale_factor < 1) return MM_ERR_INVALID_ARG;
    // SCFpyr_PyTorch.py:90-91, evaluated on the mirrored side (phase_difference_extractor.py:44-47)
    if (height > (int)std::floor(std::log2((double)(2 * size))) - 2) return MM_ERR_TOO_SMALL;
    // nbands == 1 recurses forever in the reference (math_utils.py:79-84, quirk Q7)
    if (size != 48 || height != 4 || nbands != 2 || scale_factor != 2) return MM_ERR_UNSUPPORTED;
    return MM_OK;
}
}  // namespace mm

extern "C" {

int mm_version(void) { return MM_VERSION; }

const char* mm_status_string(int s) {
    switch (s) {
        case MM_OK: return "ok";
        case MM_ERR_INVALID_ARG: return "invalid argument";
        case MM_ERR_TOO_SMALL: return "Cannot build the requested number of levels, image too small.";
        case MM_ERR_UNSUPPORTED: return "configuration not supported by this build (supported: size=48, height=4, nbands=2, scale_factor=2)";
        case MM_ERR_HIP: return "HIP runtime error";
        case MM_ERR_NO_DEVICE: return "no gfx950 device";
        case MM_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown status";
    }
}

int mm_last_hip_error(void) { return mm::g_last_hip_error; }

int mm_pyramid_host_mask(int size, int height, int nbands, int level, int band, double* out, int* crop) {
    if (!out || !crop) return MM_ERR_INVALID_ARG;
    if (size <= 0 || height < 3 || nbands < 2) return MM_ERR_INVALID_ARG;
    if (height > (int)std::floor(std::log2((double)(2 * size))) - 2) return MM_ERR_TOO_SMALL;
    mm::PyramidConfig c{size, height, nbands, 2};
    std::vector<double> m;
    int side = 0;
    int rc = mm::host_level_mask(c, level, band, m, side, crop);
    if (rc != MM_OK) return rc;
    for (size_t i = 0; i < m.size(); ++i) out[i] = m[i];
    return MM_OK;
}

int64_t mm_pyramid_host_tables(int size, int height, int nbands, int scale_factor, float* out, int64_t capacity) {
    int rc = mm::check_config(size, height, nbands, scale_factor);
    if (rc != MM_OK) return rc;
    mm::PyramidTables t;
    rc = mm::build_pyramid_tables(mm::PyramidConfig{size, height, nbands, scale_factor}, t);
    std::vector<float> packed;
    if (rc == MM_OK) rc = mm::pack_pyramid_tables(t, packed);
    if (rc != MM_OK) return rc;
    if (out) {
        if (capacity < (int64_t)packed.size()) return MM_ERR_WORKSPACE;
        std::copy(packed.begin(), packed.end(), out);
    }
    return (int64_t)packed.size();
}

int mm_pyramid_create(mm_pyramid_t** out, int size, int height, int nbands, int scale_factor) {
    if (!out) return MM_ERR_INVALID_ARG;
    *out = nullptr;
    int rc = mm::check_config(size, height, nbands, scale_factor);
    if (rc != MM_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return MM_ERR_NO_DEVICE;
    mm_pyramid* h = new (std::nothrow) mm_pyramid();
    if (!h) return MM_ERR_INVALID_ARG;
    h->cfg = mm::PyramidConfig{size, height, nbands, scale_factor};
    h->d_tables = nullptr;
    MM_HIP(hipGetDevice(&h->device));
    mm::PyramidTables t;
    rc = mm::build_pyramid_tables(h->cfg, t);
    std::vector<float> packed;
    if (rc == MM_OK) rc = mm::pack_pyramid_tables(t, packed);
    if (rc != MM_OK) {
        delete h;
        return rc;
    }
    h->table_floats = (int64_t)packed.size();
    hipError_t e = hipMalloc((void**)&h->d_tables, packed.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(h->d_tables, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (h->d_tables) (void)hipFree(h->d_tables);
        delete h;
        return mm::hip_fail(e);
    }
    *out = h;
    return MM_OK;
}

int mm_pyramid_destroy(mm_pyramid_t* h) {
    if (!h) return MM_OK;
    if (h->d_tables) (void)hipFree(h->d_tables);
    delete h;
    return MM_OK;
}

int mm_pyramid_build(mm_pyramid_t* h, const float* frames, int64_t n, float* c1, int64_t is1, int64_t bs1, float* c2,
                     int64_t is2, int64_t bs2, void* stream) {
    if (!h || n < 0 || (n > 0 && (!frames || !c1 || !c2))) return MM_ERR_INVALID_ARG;
    MM_CHECK_DEVICE(h);
    return mm::launch_pyramid(h, frames, n, n, c1, 0, is1, bs1, c2, 0, is2, bs2, 0, (hipStream_t)stream);
}

// build_pyramid layout (phase_difference_extractor.py:82-85): image (b,p), band k -> plane [b][k][p],
// i.e. groups of P images with group stride nb*P*plane, image stride plane, band stride P*plane.
int mm_pyramid_build_batch(mm_pyramid_t* h, const float* im_batch, int64_t B, int64_t P, float* c1, float* c2,
                           void* stream) {
    if (!h || B < 0 || P <= 0 || (B > 0 && (!im_batch || !c1 || !c2))) return MM_ERR_INVALID_ARG;
    MM_CHECK_DEVICE(h);
    const int64_t S = h->cfg.size, nb = h->cfg.nbands;
    const int64_t plane1 = S * S * 2, plane2 = (S / 2) * (S / 2) * 2;
    return mm::launch_pyramid(h, im_batch, B * P, P, c1, nb * P * plane1, plane1, P * plane1, c2, nb * P * plane2,
                              plane2, P * plane2, 0, (hipStream_t)stream);
}

int mm_phase_extract(mm_pyramid_t* h, const float* coeff, const int32_t* ids, int64_t img_stride, int64_t band_stride,
                     int64_t J, int P, int W, float* out, int out_nhwc, int out_cstride, int out_coffset, void* stream) {
    if (!h || J < 0 || (J > 0 && (!coeff || !ids || !out))) return MM_ERR_INVALID_ARG;
    MM_CHECK_DEVICE(h);
    if (P != 13) return MM_ERR_UNSUPPORTED;  // num_phase = 12 (api/tester.py:28)
    if (W != h->cfg.size && W != h->cfg.size / 2) return MM_ERR_UNSUPPORTED;
    if (out_nhwc && (out_cstride < out_coffset + 2 * (P - 1) || out_coffset < 0 || (out_cstride | out_coffset) & 3))
        return MM_ERR_INVALID_ARG;  // 16-byte channel groups
    return mm::launch_phase_window(coeff, ids, img_stride, band_stride, J, W, out, out_nhwc, out_cstride, out_coffset,
                                   0, (hipStream_t)stream);
}

int64_t mm_phase_workspace_bytes(mm_pyramid_t* h, int64_t n) {
    if (!h || n < 0) return MM_ERR_INVALID_ARG;
    const int64_t S = h->cfg.size;
    // the per-frame planes of both levels (mag, B, R, phase per band: pyramid_frames.hip) -- 92 KB per frame
    return (mm::phase_frames_floats((int)S, n) + mm::phase_frames_floats((int)S / 2, n)) * (int64_t)sizeof(float);
}

int mm_phase_diff_frames(mm_pyramid_t* h, const float* frames, int64_t n, const int32_t* ids, int64_t J, float* out0,
                         int out0_nhwc, int out0_cstride, int out0_coffset, float* out1, int out1_nhwc,
                         int out1_cstride, int out1_coffset, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!h || n <= 0 || J < 0 || !frames || !ids || !out0 || !out1 || !workspace) return MM_ERR_INVALID_ARG;
    if (workspace_bytes < mm_phase_workspace_bytes(h, n)) return MM_ERR_WORKSPACE;
    MM_CHECK_DEVICE(h);
    const int64_t S = h->cfg.size;
    float* f1 = (float*)workspace;           // frame planes, level 1: [n][band]{mag, B, R, phase}[S][S]
    float* f2 = f1 + mm::phase_frames_floats((int)S, n);
    hipStream_t s = (hipStream_t)stream;
    if (out0_nhwc && (out0_cstride < out0_coffset + 24 || out0_coffset < 0 || (out0_cstride | out0_coffset) & 3)) return MM_ERR_INVALID_ARG;
    if (out1_nhwc && (out1_cstride < out1_coffset + 24 || out1_coffset < 0 || (out1_cstride | out1_coffset) & 3)) return MM_ERR_INVALID_ARG;
    // once per unique frame: pyramid, atan2 / magnitude, the frame-only blurs (B, R) -- one kernel; then one blur per (window, frame)
    int rc = mm::launch_pyramid_frames(h, frames, n, f1, f2, s);
    if (rc != MM_OK) return rc;
    // both levels of a (window, band) in one workgroup (round 6) when the two outputs share a layout; else one launch per level
    if (mm::phase_window2_pair_applies(out0_nhwc, out1_nhwc)) {
        mm::prof_before(2, (double)J * 2 * 12 * (S * S + (S / 2) * (S / 2)) * 4, s, "phase_window2<48+24>");
        rc = mm::launch_phase_window2_pair(f1, f2, ids, n, J, out0, out0_nhwc, out0_cstride, out0_coffset, out1, out1_nhwc, out1_cstride, out1_coffset, s);
        mm::prof_after(2, s);
        return rc;
    }
    mm::prof_before(2, (double)J * 2 * 12 * (S * S) * 4, s, "phase_window2<48>");               // algorithmic write: 24 difference planes
    rc = mm::launch_phase_window2(f1, ids, n, J, (int)S, out0, out0_nhwc, out0_cstride, out0_coffset, s);
    mm::prof_after(2, s);
    if (rc != MM_OK) return rc;
    mm::prof_before(2, (double)J * 2 * 12 * ((S / 2) * (S / 2)) * 4, s, "phase_window2<24>");
    rc = mm::launch_phase_window2(f2, ids, n, J, (int)S / 2, out1, out1_nhwc, out1_cstride, out1_coffset, s);
    mm::prof_after(2, s);
    return rc;
}

int mm_phase_diff_planes(mm_pyramid_t* h, const float* planes, int64_t n, const int32_t* ids, int64_t J, int W, float* out,
                         int out_nhwc, int out_cstride, int out_coffset, void* stream) {
    if (!h || n <= 0 || J < 0 || !planes || (J > 0 && (!ids || !out))) return MM_ERR_INVALID_ARG;
    MM_CHECK_DEVICE(h);
    if (W != h->cfg.size && W != h->cfg.size / 2) return MM_ERR_UNSUPPORTED;
    if (out_nhwc && (out_cstride < out_coffset + 24 || out_coffset < 0 || (out_cstride | out_coffset) & 3)) return MM_ERR_INVALID_ARG;
    return mm::launch_phase_window2(planes, ids, n, J, W, out, out_nhwc, out_cstride, out_coffset, (hipStream_t)stream);
}

}  // extern "C"
