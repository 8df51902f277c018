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
// The same four band masks once more in MFMA-fragment order for pyramid_wave.hip (one wave per frame): the value lane (li, lk) of a wave
// multiplies into its A fragment of k-step ks of 16-row tile tr sits at [(tr * KS + ks) * 64 + lane], so that a wave's load is one
// coalesced 512-byte row.  Row of the tile: 16 tr + frag_row(li); column: 4 ks + lk (see frag_row below and pack_pyramid_tables).
constexpr int OFF_F1B0 = OFF_M2B1 + 48 * 24 * 2;    // [6][12][64][2]   rows r = fu + 48 of band 0, columns fv
constexpr int OFF_F1B1 = OFF_F1B0 + 96 * 48 * 2;    // [6][12][64][2]   rows c = fv + 48 of band 1 (transposed), columns fu
constexpr int OFF_F2B0 = OFF_F1B1 + 96 * 48 * 2;    // [3][6][64][2]
constexpr int OFF_F2B1 = OFF_F2B0 + 48 * 24 * 2;    // [3][6][64][2]
constexpr int TABLE_FLOATS = OFF_F2B1 + 48 * 24 * 2;

// Level-1 band spectra vanish beyond radius 48: in tile rows 0 and 5 of the fragment-ordered masks the last EDGE_ZERO_KSTEPS k-steps (columns
// 36..47) are exactly zero -- pack_pyramid_tables refuses tables for which that does not hold.
constexpr int EDGE_ZERO_KSTEPS = 3;

// Row a lane supplies inside a 16-row tile when the product's accumulator is used directly as an operand of the next MFMA: accumulator
// register e of lane (li, lk) holds row 4 lk + e, and k-slot lk of the next product's step e must be row 4 e + lk (ascending k, the
// order of the round-3 kernels), so lane li = 4 a + b loads row 4 b + a.
constexpr int frag_row(int li) { return 4 * (li & 3) + (li >> 2); }

}  // namespace pyr
}  // namespace mm
