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

}  // namespace mm
