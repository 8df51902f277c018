// Shared device math of the phase stage, so the pyramid epilogue (polar output) and the window kernel
// (rectangular input) produce bit-identical phase / magnitude values.
#pragma once
#include <hip/hip_runtime.h>

namespace mm {

// phase = atan2(im, re); mag = sqrt(im^2 + re^2) + 1e-10  (api/phase_difference_extractor.py:100-104).
// torch evaluates pow(im,2) + pow(re,2) with separately rounded operations: contraction is switched off.
__device__ __forceinline__ void to_polar(float re, float im, float& phase, float& mag) {
#pragma clang fp contract(off)
    phase = atan2f(im, re);
    const float ii = im * im;
    const float rr = re * re;
    mag = sqrtf(ii + rr) + 1e-10f;
}

// One step of torch_unwrap along time (api/utils/phase_utils.py:5-20) for a single pixel: given this frame's wrapped
// phase, the previous frame's wrapped phase and the running correction, returns the unwrapped phase and updates the
// correction.  ddmod = fmod(dd + pi, 2 pi) - pi with C fmod semantics (only positive jumps get corrected, quirk Q2);
// for dd + pi in [-pi, 3 pi] -- differences of two atan2 values -- fmod is the identity below 2 pi and an exact
// subtraction above it, so the two-case form is bit-identical to fmodf.
__device__ __forceinline__ float unwrap_step(float ph, float prev_ph, float& cum) {
    const float PI_F = 3.14159265358979323846f, TWO_PI_F = 6.28318530717958647692f;
    const float dd = ph - prev_ph;
    const float xs = dd + PI_F;
    float ddmod = (xs >= TWO_PI_F ? xs - TWO_PI_F : xs) - PI_F;
    if (ddmod == -PI_F && dd > 0.f) ddmod = PI_F;
    float corr = ddmod - dd;
    if (fabsf(dd) < PI_F) corr = 0.f;
    cum += corr;
    return ph + cum;
}

}  // namespace mm
