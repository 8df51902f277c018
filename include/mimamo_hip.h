/*
 * mimamo_hip.h -- C ABI of libmimamo_hip.so: MI355X (gfx950) kernels for MIMAMO-Net's
 * per-video inference hot path.
 *
 * The reference (wtomin/MIMAMO-Net) has no FFI layer: its operator surface is four Python
 * classes that api/tester.py constructs and calls.  Each entry point below is the native
 * replacement for one of those Python call sites (cited as file:line, relative to the
 * reference repo); the Python classes in mimamo-net_amd/ with the reference's names bind
 * them through ctypes, and INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C: device pointers are raw `float*` / `int32_t*` into HBM, sizes are int/int64_t,
 *     `stream` is a hipStream_t passed as void* (NULL = the null stream).  No torch types.
 *   - every function returns 0 (MM_OK) or a negative mm_status code; nothing throws.
 *   - caller owns every buffer (inputs, outputs, workspace).  A handle owns only immutable
 *     per-device constant tables (masks, DFT twiddles, BN-folded weights).
 *   - calls only ENQUEUE work on `stream`; no hidden synchronisation (the reference's
 *     device->host assert at api/phase_difference_extractor.py:105 is deliberately dropped).
 *   - one handle per (process, device); calls on different handles are re-entrant.  A handle is bound to
 *     the device that was current at mm_*_create: enqueueing with another current device returns
 *     MM_ERR_INVALID_ARG instead of handing one GPU's pointers to another.
 *   - all arithmetic is fp32 (the reference pins torch.float32, SCFpyr_PyTorch.py:57-59).
 */
#ifndef MIMAMO_HIP_H
#define MIMAMO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_VERSION 106 /* 0.1.6: round 6 adds mm_pyramid_host_tables (host-only, for tests); 0.1.5 removed the CU-mask stream entry points of 0.1.4 */

typedef enum mm_status {
    MM_OK = 0,
    MM_ERR_INVALID_ARG = -1,   /* bad pointer / shape / unsupported configuration          */
    MM_ERR_TOO_SMALL = -2,     /* "Cannot build N levels, image too small" (SCFpyr_PyTorch.py:90-91) */
    MM_ERR_UNSUPPORTED = -3,   /* configuration the reference accepts but this build does not */
    MM_ERR_HIP = -4,           /* a HIP runtime call failed; see mm_last_hip_error()        */
    MM_ERR_NO_DEVICE = -5,     /* no gfx950 device visible                                  */
    MM_ERR_WORKSPACE = -6      /* caller-provided workspace too small                       */
} mm_status;

int mm_version(void);
const char* mm_status_string(int status);
/* hipError_t of the last failing HIP call on this thread (0 if none). */
int mm_last_hip_error(void);

/* ---------------------------------------------------------------------------------------
 * Steerable pyramid + phase difference
 * replaces: Phase_Difference_Extractor.{__init__,build_pyramid,extract}
 *           (api/phase_difference_extractor.py:7-37, 38-87, 93-134),
 *           SCFpyr_PyTorch.build/_build_levels (api/steerable/SCFpyr_PyTorch.py:70-208),
 *           symmetric_extension_batch/torch_unwrap/torch_diff/amplitude_based_gaussian_blur
 *           (api/utils/phase_utils.py:5-40,78-129), Tester.phase_diff_output (api/tester.py:122-139)
 * ------------------------------------------------------------------------------------- */
typedef struct mm_pyramid mm_pyramid_t;

/* size = side of the (un-mirrored) square input; the kernels mirror it to 2*size
 * (symmetric_extension_batch).  Supported: size=48, height=4, nbands=2, scale_factor=2 --
 * the one configuration api/tester.py:28-32 uses; anything else the reference would accept
 * returns MM_ERR_UNSUPPORTED, height > floor(log2(2*size))-2 returns MM_ERR_TOO_SMALL. */
int mm_pyramid_create(mm_pyramid_t** out, int size, int height, int nbands, int scale_factor);
int mm_pyramid_destroy(mm_pyramid_t* h);

/* Host-side constant builder exposed for testing (no GPU needed): writes the real-valued
 * product of the reference's float64 masks for pyramid list item `level` (1 or 2), band
 * `band`, on that level's (2*size / 2^(level-1))^2 grid, fftshift-ed layout exactly as
 * SCFpyr_PyTorch builds them:  lo0 * [lomask_0 ...] * himask_level * anglemask_level_band.
 * `out` must hold side*side doubles.  Also returns the crop bounds used going INTO that level
 * (level 1: {0, 2*size}) in crop[0..1]. */
int mm_pyramid_host_mask(int size, int height, int nbands, int level, int band, double* out, int* crop);

/* Host-side, for testing (no GPU needed): the packed fp32 constant tables a handle of this configuration uploads -- the DCT and twiddle
 * tables, the four complex band masks in the kernels' half-plane layout and the same masks once more in MFMA-fragment order for
 * pyramid_wave_kernel (csrc/pyramid_tables.h gives the offsets).  Returns the number of floats; writes them when out != NULL and
 * capacity is large enough (MM_ERR_WORKSPACE otherwise).  Negative: an MM_ERR_* status (unsupported configuration, or a table that
 * violates an assumption a kernel relies on, e.g. the spectrum blocks pyramid_wave_kernel skips as exactly zero). */
int64_t mm_pyramid_host_tables(int size, int height, int nbands, int scale_factor, float* out, int64_t capacity);

/* frames: device f32 [n, size, size].  For every image and band writes the kept quadrant of the
 * level-1 and level-2 complex band coefficients (interleaved re,im):
 *   c1 + img*img_stride1 + band*band_stride1  ->  [size,   size,   2]
 *   c2 + img*img_stride2 + band*band_stride2  ->  [size/2, size/2, 2]
 * (strides in floats).  build_pyramid's return layout [B, nbands, P, W, H, 2] for im_batch
 * [B,P,W,H] is obtained with n=B*P, image index (b*P+p): see mm_pyramid_build_batch. */
int mm_pyramid_build(mm_pyramid_t* h, const float* frames, int64_t n,
                     float* c1, int64_t img_stride1, int64_t band_stride1,
                     float* c2, int64_t img_stride2, int64_t band_stride2, void* stream);

/* Drop-in layout helper: im_batch [B,P,size,size] -> c1 [B,nbands,P,size,size,2],
 * c2 [B,nbands,P,size/2,size/2,2] (phase_difference_extractor.py:76-86). */
int mm_pyramid_build_batch(mm_pyramid_t* h, const float* im_batch, int64_t B, int64_t P,
                           float* c1, float* c2, void* stream);

/* Phase difference of J windows of P=13 coefficient planes (extract(), :93-134).
 *   plane(j, i, band) = coeff + ids[j*P + i]*img_stride + band*band_stride   -> [W, W, 2]
 * ids: device int32 [J*P].  W in {size, size/2}.
 * out element (j, c = band*(P-1)+k, y, x):
 *   out_nhwc == 0:  out[((j*C + c)*W + y)*W + x]                 (C = nbands*(P-1); tester.py:133-138)
 *   out_nhwc == 1:  out[((j*W + y)*W + x)*out_cstride + out_coffset + c]   (feeds the head's NHWC convs)
 */
int mm_phase_extract(mm_pyramid_t* h, const float* coeff, const int32_t* ids,
                     int64_t img_stride, int64_t band_stride, int64_t J, int P, int W,
                     float* out, int out_nhwc, int out_cstride, int out_coffset, void* stream);

/* extract() for ANY configuration (other band counts, plane sizes, window lengths than api/tester.py's):
 * coeff device f32 [planes, P, R, C, 2] (planes = batch*nbands, the reference's view at :97-98) ->
 * out [planes, P-1, R, C].  Same arithmetic as mm_phase_extract, generic and slower; R*C <= 4096 (larger planes: the _ws
 * variant below), P >= 2, otherwise MM_ERR_UNSUPPORTED.  Needs no handle (the Gaussian taps are compile-time constants).
 * denoised (optional, may be NULL): [planes, P, R, C], the amplitude-blurred unwrapped phase with its spatial
 * mean removed -- what the training-side `Steerable_Pyramid_Phase.extract_phase(return_phase=True)` returns
 * (Aff-wild-exps/utils.py:367-418). */
int mm_phase_extract_generic(const float* coeff, int64_t planes, int P, int R, int C, float* out, float* denoised,
                             void* stream);
/* The same for planes of ANY size (the reference's extract takes any W x H, api/phase_difference_extractor.py:93): the blur planes and
 * the per-pixel scan state live in a caller-owned workspace of mm_phase_extract_generic_workspace_bytes() bytes (0 = the plane fits the
 * LDS kernel above and no workspace is needed; 8 floats per pixel and plane set otherwise).  A non-NULL workspace of sufficient size
 * selects the workspace kernel whatever the plane size (bit-identical results where both apply); NULL / 0 bytes = mm_phase_extract_generic. */
int64_t mm_phase_extract_generic_workspace_bytes(int64_t planes, int P, int R, int C);
int mm_phase_extract_generic_ws(const float* coeff, int64_t planes, int P, int R, int C, float* out, float* denoised,
                                void* workspace, int64_t workspace_bytes, void* stream);

/* Fused, de-duplicated driver for one batch of frames (the build's fast path): ONE kernel per unique frame (pyramid,
 * atan2 / magnitude, the frame-only blurs; csrc/pyramid_frames.hip), then J windows gathered through window ids
 * (snippet_sampler.py:144-152; csrc/phase_frames.hip).
 * frames [n,size,size]; ids int32 [J*13] in [0,n): ANY frame pattern -- the unwrap decisions are taken between the
 * consecutive frames of each window from the frames' phase planes, as the reference takes them (an id outside [0,n) is
 * clamped into range, never dereferenced).  Outputs as in mm_phase_extract for W=size (out0) and W=size/2 (out1).
 * workspace: mm_phase_workspace_bytes(n) bytes of HBM: the per-frame planes, level 1 then level 2, each
 * [n][nbands]{magnitude, B = blur(mag phase)/blur(mag), R = 1/blur(mag), phase}[W][W] f32. */
int64_t mm_phase_workspace_bytes(mm_pyramid_t* h, int64_t n);
int mm_phase_diff_frames(mm_pyramid_t* h, const float* frames, int64_t n, const int32_t* ids, int64_t J,
                         float* out0, int out0_nhwc, int out0_cstride, int out0_coffset,
                         float* out1, int out1_nhwc, int out1_cstride, int out1_coffset,
                         void* workspace, int64_t workspace_bytes, void* stream);

/* The window half of mm_phase_diff_frames on its own: J windows from per-frame planes of ONE level that the caller keeps
 * (the layout mm_phase_diff_frames leaves in its workspace: [n][nbands]{mag, B, R, phase}[W][W], W = size or size/2), e.g.
 * to re-window cached frames with other ids.  out as in mm_phase_extract. */
int mm_phase_diff_planes(mm_pyramid_t* h, const float* planes, int64_t n, const int32_t* ids, int64_t J, int W,
                         float* out, int out_nhwc, int out_cstride, int out_coffset, void* stream);

/* ---------------------------------------------------------------------------------------
 * General complex steerable pyramid -- replaces `SCFpyr_PyTorch(height, nbands, scale_factor,
 * device, precision).build(im_batch)` (api/steerable/SCFpyr_PyTorch.py:51-125 and
 * _build_levels :127-208) with its FULL return list: hi-pass residual, every oriented band of every
 * level, low-pass residual.  Arbitrary (non-mirrored) square images of side <= 1024, even or odd, odd level grids included (each
 * 2-D transform's intermediate lives in LDS up to side 96, in the workspace above; the transforms are DFTs by summation, O(side^3) per
 * image: completeness, not speed), height >= 2, 2 <= nbands <= 16; otherwise MM_ERR_UNSUPPORTED;
 * `height > floor(log2(size)) - 2` returns MM_ERR_TOO_SMALL (the reference's RuntimeError, :90-91).
 * The inference hot path does not go through here (mm_pyramid_* exploits the mirrored input).
 * ------------------------------------------------------------------------------------- */
typedef struct mm_scfpyr mm_scfpyr_t;
int mm_scfpyr_create(mm_scfpyr_t** out, int size, int height, int nbands, int scale_factor);
int mm_scfpyr_destroy(mm_scfpyr_t* h);
/* Host-side constant builder exposed for testing (no GPU needed): the complex float64 multiplier
 * of output `index` ([side][side][2], FFT index order of that output's grid; 1/side^2 and the
 * (-i)^(nbands-1) band factor folded in).  out may be NULL to query side / is_complex only. */
int mm_scfpyr_host_table(int size, int height, int nbands, int scale_factor, int index, double* out, int* side,
                         int* is_complex);
/* outputs in the order of the reference's list, flattened: 0 = hi-pass residual, then
 * level-major bands (level 1 band 0 .. nbands-1, level 2 ...), last = low-pass residual */
int mm_scfpyr_num_outputs(const mm_scfpyr_t* h);
int mm_scfpyr_output_info(const mm_scfpyr_t* h, int index, int* side, int* is_complex);
int64_t mm_scfpyr_workspace_bytes(const mm_scfpyr_t* h, int64_t n);
/* images: device [n, size, size] float (precision 32) or double (precision 64) -- the reference's
 * [N,1,H,W] batch.  outputs: HOST array of mm_scfpyr_num_outputs device pointers; output i is
 * [n, side_i, side_i] (residuals, real) or [n, side_i, side_i, 2] (bands, re/im) in the same
 * precision.  Internally float64 throughout. */
int mm_scfpyr_build(const mm_scfpyr_t* h, const void* images, int precision, int64_t n, void* const* outputs,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Convolutional networks (fp32 MFMA implicit-GEMM engine)
 * replaces: Resnet50_Extractor.get_vec (api/resnet50_extractor.py:74-83) and
 *           Two_Stream_RNN.forward (api/mimamo_net.py:129-143)
 * ------------------------------------------------------------------------------------- */
typedef struct mm_resnet50 mm_resnet50_t;
typedef struct mm_head mm_head_t;

/* The conv engine itself (one launch): NHWC fp32 convolution / linear layer with fused epilogue
 *   out = [post_scale *] relu?( conv(in, w) + bias [+ residual] ) [+ post_shift]
 * in  [B,H,W,in_cstride]  channels [in_coff, in_coff+Cin) are read   (Cin, in_cstride, in_coff multiples of 4)
 * w   [Cout][Kpad] packed weights, Kpad = K rounded up to 16, zero filled;
 *     korder 0: k = (r*kw + s)*Cin + c;  korder 1 (Cin % 16 == 0): k = ((c/16*kh + r)*kw + s)*16 + c%16
 *     (slice-major: the taps of a 16-channel slice are adjacent, which keeps the kh*kw-fold input re-use in L2)
 * out [B,Ho,Wo,out_cstride] channels [out_coff, out_coff+Cout) are written; residual [B,Ho,Wo,res_cstride].
 * bias/residual/post_scale/post_shift may be NULL.  tile: 0 auto, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 256x64, 5 = 128x256 (1x1 kernels only).
 * This is what torch.nn.functional.conv2d / linear + BatchNorm(eval) + ReLU lower to on this path
 * (api/mimamo_net.py:14-26,68-78; the third-party ResNet50's conv/bn/relu triples). */
int mm_conv2d_nhwc(const float* in, const float* w, const float* bias, const float* residual,
                   const float* post_scale, const float* post_shift, float* out,
                   int B, int H, int W, int Cin, int in_cstride, int in_coff,
                   int Cout, int out_cstride, int out_coff, int res_cstride,
                   int kh, int kw, int stride, int pad, int relu, int tile, int korder, void* stream);

/* Weight blob: host f32 array, tensors concatenated in the order of mm_resnet50_blob_floats /
 * documented in mimamo-net_amd/weights.py (per conv: weight OIHW, then BN gamma, beta,
 * running_mean, running_var).  BN (eps) is folded into the conv on the host at create time.
 * stride_on_first_1x1: 1 = Caffe-style (the third-party model file), 0 = torchvision-style. */
int64_t mm_resnet50_blob_floats(void);
int mm_resnet50_create(mm_resnet50_t** out, const float* host_blob, int64_t n_floats,
                       int stride_on_first_1x1, int maxpool_ceil_mode, float bn_eps);
int mm_resnet50_destroy(mm_resnet50_t* h);
/* The stride-1 3x3 layers of conv2_x..conv5_x (16 layers) run by default as Winograd F(4x4,3x3) (4x fewer multiply-adds,
 * same result up to fp32 rounding).  mode: 0 = every layer in the direct implicit-GEMM form, 2 = F(2x2,3x3) as three kernels
 * (input transform, batched position GEMMs, output transform), 4 = F(4x4,3x3) as three kernels, 5 = F(4x4,3x3) with the output
 * transform fused into the position GEMMs (csrc/wino_fused.hip: the M planes never reach HBM), 1 = the default: variant 5 for
 * the layers with <= 256 input channels (conv2_x..conv4_x), variant 4 for conv5_x. */
int mm_resnet50_set_winograd(mm_resnet50_t* h, int enable);
int64_t mm_resnet50_workspace_bytes(mm_resnet50_t* h, int64_t batch);
/* images: device f32, [batch,3,224,224] (nchw=1, the reference's layout), channels-last padded to four
 * channels [batch,224,224,4] (nchw=0; the 4th channel is ignored) or zero-bordered packed rows [batch,230,230,3]
 * (nchw=2, what mm_preproc_forward writes with rgb_nchw=2: the 7x7/2 stem then runs with K = 168 instead of 196);
 * already normalised (255*x - mean, api/utils/model_utils.py:36-39).  out: [batch, 2048]
 * = relu(pool5_7x7_s1) on the device (the reference copies it to a CPU tensor and squeeze()s
 * it -- quirk Q8, not reproduced). */
/* Arithmetic of the 1x1 layers with K >= 512 (conv3_x's 512 -> 128 reduce convs, all of conv4_x / conv5_x's 1x1 layers and projection
 * contractions, conv5_x's Winograd position GEMMs).  0 (default, the headline and every parity claim): fp32 operands on the fp32 matrix
 * pipes (v_mfma_f32_32x32x2_f32), bit-for-bit an fmaf chain.  1: both fp32 operands split three ways into bf16 in registers
 * (x = h + m + l, 3 x 8 mantissa bits) and six of the nine partial products accumulated in fp32 on v_mfma_f32_32x32x16_bf16 (16 x the
 * fp32-MFMA rate; the dropped terms are below 2^-24 relative).  Inputs, outputs, weights and every other layer stay fp32.  Reported by
 * bench.py as extra.bf16x3 against the bf16 roof -- never as the headline (the reference computes in fp32). */
int mm_resnet50_set_precision(mm_resnet50_t* h, int mode);
int mm_resnet50_forward(mm_resnet50_t* h, const float* images, int nchw, int64_t batch, float* out,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* Two-stream head.  Blob: the Two_Stream_RNN state_dict float tensors in state_dict order
 * (num_batches_tracked skipped); see weights.py `two_stream_blob`. */
int64_t mm_head_blob_floats(void);
int mm_head_create(mm_head_t** out, const float* host_blob, int64_t n_floats);
/* The same with another MLP on the ResNet feature stream: units = Two_Stream_RNN's mlp_hidden_units (api/mimamo_net.py:6-26,
 * 97-98), e.g. {2048, 512, 256}: n_units >= 2, units[last] == 256 (the reference asserts it), every entry a multiple of 4;
 * the rgb input of mm_head_forward is then [bs, T, units[0]].  mm_head_create == units {2048, 256, 256} (api/tester.py:45). */
int64_t mm_head_blob_floats_mlp(int n_units, const int* units);
int mm_head_create_mlp(mm_head_t** out, const float* host_blob, int64_t n_floats, int n_units, const int* units);
/* Two_Stream_RNN(mlp_hidden_units, num_phase=...) (api/mimamo_net.py:97-112): PhaseNet takes 2 * num_phase channels per level
 * (phase_0 [bs,T,2*num_phase,48,48], phase_1 [bs,T,2*num_phase,24,24]); the state_dict's conv_net.0.0 / conv_net.1.0 weights have
 * 2*num_phase / 64 + 2*num_phase input channels.  Any num_phase in [1, 128] (MM_ERR_UNSUPPORTED beyond, from both functions).  An odd
 * num_phase (2*num_phase not a multiple of the 16-byte channel group) runs on internally zero-padded channels and takes NCHW phase
 * inputs only (mm_head_forward's phase_nhwc == 0; the channels-last entry points answer MM_ERR_UNSUPPORTED).
 * mm_head_create_mlp == num_phase 12 (api/tester.py:28). */
int64_t mm_head_blob_floats_cfg(int n_units, const int* units, int num_phase);
int mm_head_create_cfg(mm_head_t** out, const float* host_blob, int64_t n_floats, int n_units, const int* units, int num_phase);
int mm_head_destroy(mm_head_t* h);
int64_t mm_head_workspace_bytes(mm_head_t* h, int64_t bs, int64_t T);
/* phase_0 [bs,T,24,48,48] / phase_1 [bs,T,24,24,24] (phase_nhwc=0, reference layout) or
 * NHWC [bs*T,48,48,24] / [bs*T,24,24,24] (phase_nhwc=1), or NHWC with phase_1 already placed at channels
 * 64..87 of a [bs*T,24,24,88] buffer that the call then completes in place (phase_nhwc=2, the fused
 * pipeline: PhaseNet concatenates conv features and level-1 phase, mimamo_net.py:85);
 * rgb [bs,T,2048] ([bs,T,units[0]] for mm_head_create_mlp); out [bs,T,2]
 * (col 0 valence, col 1 arousal, tester.py:52).  The GRU runs over dim 0 (bs) with T as its
 * batch, exactly like nn.GRU without batch_first (mimamo_net.py:119,139; quirk Q1). */
int mm_head_forward(mm_head_t* h, const float* phase_0, const float* phase_1, int phase_nhwc,
                    const float* rgb, int64_t bs, int64_t T, float* out,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Frame preprocessing, bit-exact with the PIL pipeline of the reference's DataLoader workers
 * replaces: Image.open(bmp).convert('L') + GroupScale(48, LANCZOS) + /255
 *             (api/sampler/snippet_sampler.py:163,177-185; api/utils/data_utils.py:80)
 *           Resize(256, bilinear) + CenterCrop(224) + ToTensor + x*255 + Normalize(mean, std=1)
 *             (api/utils/model_utils.py:29-39; api/sampler/image_sampler.py)
 * ------------------------------------------------------------------------------------- */
typedef struct mm_preproc mm_preproc_t;
/* Host-side table builder exposed for testing (no GPU): Pillow's fixed-point resampling coefficients for a full-axis
 * resize in_size -> out_size; filter 0 = bilinear, 1 = lanczos.  Call with bounds = kk = NULL to query *ksize;
 * bounds [out_size][2] = {first input index, count}, kk [out_size][ksize] (22 fractional bits). */
int mm_preproc_host_coeffs(int in_size, int out_size, int filter, int* ksize, int* bounds, int* kk, int kk_capacity);
/* in_size: side of the aligned-face frames (112); gray_size 48; resize 256; crop 224; mean3: meta['mean']. */
int mm_preproc_create(mm_preproc_t** out, int in_size, int gray_size, int resize, int crop, const float* mean3);
int mm_preproc_destroy(mm_preproc_t* h);
/* frames: device uint8 [n, in_size, in_size, 3] (RGB, the BMP pixel order).  gray_out: f32 [n,gray,gray] in [0,1];
 * rgb_out: f32 [n,3,crop,crop] (rgb_nchw=1), channels-last padded to four channels [n,crop,crop,4] (rgb_nchw=0) or packed
 * three-channel rows with the stem's 3-pixel zero border in memory [n,crop+6,crop+6,3] (rgb_nchw=2: input mode 2 of
 * mm_resnet50_forward), 255*x - mean.  Either output may be NULL. */
int mm_preproc_forward(mm_preproc_t* h, const uint8_t* frames, int64_t n, float* gray_out, float* rgb_out,
                       int rgb_nchw, void* stream);

/* ---------------------------------------------------------------------------------------
 * Measurement hook (bench.py roofline leg; not part of the reference surface).  Between begin and
 * end every kernel launch is bracketed by hipEvents on its own stream.  Categories: 0 conv/GEMM
 * engine (work = FLOPs executed on the matrix cores, 2*M*K*Cout per problem), 1 pyramid kernel, 2 phase-window kernel
 * (work = algorithmic HBM bytes: frame in / phase planes out), 3 Winograd input/output transforms (work = bytes moved), 4 the other
 * data-movement kernels (preprocessing, max-pool, average pools, layout conversions, GRU gates; work = algorithmic bytes).  Arrays hold
 * MM_PROF_CATEGORIES entries.
 * mm_profile_end synchronises the device.
 * ------------------------------------------------------------------------------------- */
#define MM_PROF_CATEGORIES 5
int mm_profile_begin(void);
int mm_profile_end(double* ms, double* work, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* MIMAMO_HIP_H */
