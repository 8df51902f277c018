// Internal interfaces of the conv / GEMM engine and the small elementwise kernels.
#pragma once
#include "mm_common.h"

namespace mm {

struct ConvParams {
    const float* in;          // NHWC activations
    const float* w;           // [Cout][Kpad], k = (r, s, c), zero padded
    const float* bias;        // [Cout] or null
    const float* res;         // residual (NHWC, same M) or null
    const float* post_scale;  // affine applied AFTER ReLU (BN placed after ReLU) or null
    const float* post_shift;
    float* out;
    int B, H, W, Cin, in_cstride, in_coff;
    int Ho, Wo, Cout, out_cstride, out_coff;
    int res_cstride, res_coff;
    int kh, kw, stride, pad;
    int K, Kpad;
    int korder;      // 0: k=(r,s,c)   1: k=(c/16,r,s,c%16), needs Cin % 16 == 0   2: packed 3-channel rows (see conv_mfma.hip)
                     //    FINITE INPUTS ASSUMED for korder 2 on the unrolled stem loop (KMODE 9): k-quads past K = 168 and rows past M read real
                     //    memory (the window's 8th row, row m_base's pixels) and rely on zero weights / never-stored rows, so an Inf / NaN in
                     //    that extra row gives 0 * Inf = NaN in output pixels whose 7x7 window does not contain it -- base mode 5 (no_sched)
                     //    and the reference's conv do not.  pool5 averages every pixel anyway, so a non-finite frame poisons its features in
                     //    both forms; preprocessing (uint8 in) cannot produce one.
    int relu;
    int force_tile;  // 0 auto, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 256x64, 5 = 128x256 (8 waves, 1x1 only), 16+bits = ablation build
    int ablate;
    int Cin_real;    // un-padded input channels (FLOP accounting only; 0 = Cin)
    int batch;       // >1: `batch` independent problems of this shape in one launch (0/1 = single)
    int64_t in_bstride, w_bstride, out_bstride;   // element strides between the problems of a batch
    int M, tiles_m, tiles_n;  // filled by conv_forward (M = exclusive end row of the launch: B * Ho * Wo, or m_end when that is smaller)
    int m_end;       // caller-side cap on the rows of this launch: rows [m_off, m_end) are computed (0 = up to B * Ho * Wo)
    int m_off;       // first output row of this launch (conv_forward's tail split: bulk rows on the big tile, the last partial
                     // round of workgroups as a second launch on a finer tile); 0 from callers
    // second K source of a 1x1 layer (null = none): out = W[:, :Cin] * in(pixel) + W[:, Cin:] * in2(pixel * stride2), i.e. a residual
    // block's increase conv and its projection shortcut as ONE contraction over K = Cin + C2 (the shortcut tensor is never written
    // or re-read).  in2 is NHWC [B, H2, W2, in2_cstride]; needs kh = kw = 1, pad = 0, stride = 1, Cin % 16 == 0, C2 % 16 == 0.
    const float* in2;
    int H2, W2, C2, in2_cstride, in2_coff, stride2;
    int sched1x1;    // set by conv_forward: 1x1 layer with K % 16 == 0 on the scheduled loop (conv_mfma.hip KMODE 7 / 8)
    int no_sched;    // 1: keep modes 3 / 6 for such a layer (the parity twin of the scheduled loop)
    int use_panel;   // 1 / 2: a layer conv_panel.hip applies to runs there on 128-row / 64-row panels (MM_CONV_PANEL=1 | 2: opt-in, measured slower than
                     // the engine -- the tested twins)
    int x3;          // 1: 1x1 layer on the bf16 matrix pipes through a three-way bf16 split of both fp32 operands (conv_mfma.hip X3; `extra` only)
    const unsigned short* w3;   // (round 6, x3 only) the same weights pre-split into three bf16 planes [3][w3_plane] (bf16x3_split_weights): the 128x256
    int64_t w3_plane;           // tile then reads the B fragments ready-made (no split of the weights in the loop); null = split in the loop.  Elements
                                // per plane = batch * Cout * Kpad (a batched launch indexes a plane with w_bstride like `w`)
    // exact division of a row index m < 2^31 by Ho * Wo and by Wo as multiply-high + shift (filled by conv_forward; the emulated 32-bit divisions of
    // the tile prologue were a third of its vector instructions, and vector instructions are matrix time on this chip)
    unsigned div_hw_mul, div_hw_sh, div_wo_mul, div_wo_sh;
    int hpool;       // 1 (korder 2 -- the stem -- with Cout == 64, Wo even, out_cstride == 64 only): the epilogue writes the HORIZONTAL half of
                     // MaxPool2d(3, 2, pad 0): out [B, Ho, Wo / 2, 64], out(b, y, j) = max over x in {2j, 2j+1, 2j+2 (if < Wo)} of relu(conv + bias)
                     // -- half the bytes; maxpool_reduce64(..., hp = 1) finishes the pool vertically (round 5, conv_mfma.hip "hpool")
};

int conv_forward(const ConvParams& p, hipStream_t stream);
// fp32 weights [n] (n % 16 == 0) -> three bf16 planes [3][n]: x = h + m + l, each the round-to-nearest bf16 of what is left (the split the
// bf16x3 loop applies to its fragments, applied once)
int bf16x3_split_weights(const float* w, unsigned short* out, int64_t n, hipStream_t stream);
// (round 6) 1x1 layers with K = 256 (or 128) and N a multiple of 256, N >= 512, stride 1: the 128-row activation panel resident in LDS, weights
// streamed through registers, no barrier in the main loop, epilogue straight from the accumulators (conv_panel.hip).  conv_forward routes there
// only with ConvParams::use_panel (measured slower: profiles/r06_ab_conv_panel.txt); bit-identical to the engine.
bool conv_panel_supported(const ConvParams& p);
int conv_panel_forward(const ConvParams& p, hipStream_t stream);
int conv_panel_rows(const ConvParams& p);      // 128 (use_panel == 1) or 64 (use_panel == 2: two workgroups per CU)
int conv_panel_per_cu(const ConvParams& p);

// NCHW [N,C,HW] -> NHWC [N,HW,cstride] at channel offset coff; channels [C, cpad) are zero-filled
int nchw_to_nhwc(const float* in, float* out, int64_t N, int C, int HW, int cstride, int coff, int cpad, hipStream_t s);
// NCHW [N,3,S,S] -> zero-bordered packed NHWC3 [N,S+2pad,S+2pad,3]
int nchw3_to_bordered_nhwc3(const float* in, float* out, int64_t N, int S, int pad, hipStream_t s);
// MaxPool2d(k=3, s=2, pad=0, ceil_mode) on NHWC
int maxpool3x3s2(const float* in, float* out, int64_t N, int H, int W, int C, int Ho, int Wo, hipStream_t s);
// the same pool AND the 1x1 conv that follows it at conv2_1 (64 -> 64, w [64][64] BN-folded, + bias, ReLU) in one kernel (pool_reduce.hip;
// the pooled values reach the MFMA through a wave-private LDS stage):
// x_out = the pooled tensor (bit-identical to maxpool3x3s2), y_out = relu?(w x + bias); NHWC, 64 channels
// hp = 1: `in` is the horizontally pooled stem output [N, H, ceil(W / 2), 64] (ConvParams::hpool): three vertical taps per pixel instead of nine
int maxpool_reduce64(const float* in, const float* w, const float* bias, float* x_out, float* y_out, int64_t N, int H, int W, int Ho, int Wo,
                     int relu, hipStream_t s, int hp = 0);
// global average pool over HW (AvgPool2d(k=HW side)); optional ReLU afterwards
int avgpool_hw(const float* in, float* out, int64_t N, int HW, int C, int out_cstride, int out_coff, int relu, hipStream_t s);
// Winograd F(m x m, 3x3) transforms around a batched GEMM (winograd.hip), m = 2 or 4, a = m + 2
//   input : x NHWC [B,H,W,C] (pad 1)             -> V [a*a][B*TH*TW][C],  TH = ceil(H/m), TW = ceil(W/m)
//   output: M [a*a][B*TH*TW][Cout] + bias, ReLU  -> y NHWC [B,H,W,Cout]
//   x_channels (m = 4 only; 0 = C): x has x_channels <= C channels per pixel, channels [x_channels, C) of V are zero (K padded to the
//   GEMM kernels' granularity; the matching weight columns are zero too)
int wino_input_transform(const float* x, float* V, int B, int H, int W, int C, int m, hipStream_t s, int x_channels = 0);
int wino_output_transform(const float* M, const float* bias, float* y, int B, int H, int W, int Cout, int relu, int m, hipStream_t s);

// F(4x4,3x3) position GEMMs + output transform in one kernel (wino_fused.hip): V [36][tiles][Cin] (wino_input_transform, m = 4),
// U [36][Cout][Cin] -> y NHWC + bias, ReLU.  shape: workgroup variant (0 default, see wino_fused.hip; measurement knob).
// Needs Cin % 64 == 0 and Cout % 32 == 0 (MM_ERR_UNSUPPORTED otherwise).
// generic_loop (all three entry points): 1 = the runtime-scheduled main loop whatever K is; 0 (default) = the compile-time-scheduled one for
// K = 64 / 128 / 256 (wino_fused.hip KSL: no address arithmetic left in a slab) -- same operations in the same order, bit-identical results
int wino_gemm_output_fused(const float* V, const float* U, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                           int relu, int shape, hipStream_t s, int generic_loop = 0);
// The same kernel with the residual block's 1x1 increase conv inside its epilogue -- (Cout, C2) == (64, 256): conv2_x blocks 2, 3, or
// (128, 512): conv3_x blocks 2-4 (eight-wave workgroup):
// out [B,H,W,C2] = relu( W2 relu(conv3x3(x) + bias) + bias2 + res ); the Cout-channel tensor in between never reaches HBM.
// next_w / next_bias / next_out (round 6, (64, 256) only; null = off): the NEXT block's 1x1 reduce conv, [64][C2] BN-folded + ReLU, applied to `out`
// before it leaves the CU: next_out [B,H,W,64] = relu(next_w out + next_bias) -- that block's 256 -> 64 launch and its re-read of `out` disappear.
// shape: 0 = default, 8 = eight-wave workgroups without next_* (A/B of the shape), 2 = -DMM_MEASURE cost proxy (wrong results)
int wino_gemm_output_fused_inc(const float* V, const float* U, const float* bias, const float* W2, const float* bias2, const float* res,
                               float* out, int B, int H, int W, int Cin, int Cout, int C2, int relu, hipStream_t s, int generic_loop = 0,
                               const float* next_w = nullptr, const float* next_bias = nullptr, float* next_out = nullptr, int shape = 0);
bool wino_fused_inc_supported(int64_t ntile, int Cin, int Cout, int C2);
// ... and with increase conv + stride-1 projection shortcut as one contraction over [relu(conv3x3) ; x] (conv2_x block 1): W2 [C2][128]
int wino_gemm_output_fused_incproj(const float* V, const float* U, const float* bias, const float* W2, const float* bias2, const float* x,
                                   float* out, int B, int H, int W, int Cin, int Cout, int C2, int relu, hipStream_t s, int generic_loop = 0);
// whether wino_gemm_output_fused takes the shape (channel granularity, 32-bit offsets inside a position plane)
bool wino_fused_supported(int64_t ntile, int Cin, int Cout);

// one GRU time step for a batch of Bt rows (PyTorch gate order r,z,n):
//   gi [Bt, gi_stride] (+gi_off) = W_ih x + b_ih ; gh [Bt, 3H] = W_hh h + b_hh (or null with bhh => h == 0)
//   h_out[b, out_stride*b + out_off + j] = (1-z)*n + z*h_prev
int gru_gates(const float* gi, int gi_stride, int gi_off, const float* gh, const float* bhh, const float* h_prev,
              int hp_stride, int hp_off, float* h_out, int out_stride, int out_off, int64_t Bt, int H, hipStream_t s);

}  // namespace mm
