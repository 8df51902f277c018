// Layout of the packed pyramid tables in HBM (one allocation per mm_pyramid handle), shared by pyramid.hip (drop-in
// build_pyramid: complex coefficients out) and pyramid_frames.hip (fused hot path: per-frame phase-stage planes out).
#pragma once

namespace mm {
namespace pyr {

constexpr int S = 48;        // frame side

// global table offsets (floats)
constexpr int OFF_DCT = 0;                          // [48][48]  D[f][m] = 2 cos(pi f (2m+1)/96)
constexpr int OFF_EC = OFF_DCT + S * S;             // [48][48]  cos(2 pi f q / 96)
constexpr int OFF_ES = OFF_EC + S * S;              // [48][48]  sin(2 pi f q / 96)
constexpr int OFF_M1B0 = OFF_ES + S * S;            // [96][48][2]
constexpr int OFF_M1B1 = OFF_M1B0 + 96 * 48 * 2;    // [48][96][2]
constexpr int OFF_M2B0 = OFF_M1B1 + 96 * 48 * 2;    // [48][24][2]
constexpr int OFF_M2B1 = OFF_M2B0 + 48 * 24 * 2;    // [24][48][2]
constexpr int TABLE_FLOATS = OFF_M2B1 + 48 * 24 * 2;

}  // namespace pyr
}  // namespace mm
