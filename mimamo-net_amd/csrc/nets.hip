// Network executors on top of the conv engine: ResNet50 trunk to pool5 and the two-stream head.
//
//   mm_resnet50_*  <- Resnet50_Extractor.get_vec (api/resnet50_extractor.py:74-83).  The layer graph is the
//                     third-party `resnet50_ferplus_dag` model the reference loads by name
//                     (api/utils/model_utils.py:65-79; Caffe-style ResNet-50: stride on the first 1x1 of a
//                     stage, ceil-mode 3x3/2 max pool, 7x7 average pool `pool5_7x7_s1`).
//   mm_head_*      <- Two_Stream_RNN.forward (api/mimamo_net.py:129-143): MLP (:14-26), PhaseNet (:41-95),
//                     transform (:115-118), 2-layer bidirectional GRU over dim 0 (:119,139), classifier (:120-122).
// Eval-mode BatchNorm is folded into the preceding conv/linear on the host (float64) when it directly follows it;
// BatchNorm placed after a ReLU (PhaseNet.fc, transform) runs as the conv engine's post-ReLU affine.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include "conv.h"

namespace mm {

struct Layer {
    float *w = nullptr, *bias = nullptr, *ps = nullptr, *pt = nullptr;
    float* wino_u = nullptr;   // [16][cout][cin] F(2x2,3x3)-domain weights (stride-1 3x3 layers with cin >= 128), or null
    float* wino_u4 = nullptr;  // [36][cout][cin] F(4x4,3x3)-domain weights
    int cin = 0, cin_p = 0, cout = 0, k = 1, stride = 1, pad = 0, K = 0, Kpad = 0, relu = 0, korder = 0;
    int wino_cin = 0;          // channels of the Winograd-domain weights: cin, or cin padded with zero columns (wino_pad)
    int tile = 0;              // conv engine tile for this layer (0 = automatic; measurement knob MM_STEM_TILE for the stem)
    // (round 6) bf16x3 mode: `w` / `wino_u4` pre-split into three bf16 planes (made by mm_resnet50_set_precision(h, 1) for the layers the
    // mode touches whose tile is 128x256); null = the loop splits the weight fragments itself
    unsigned short *w3 = nullptr, *wino_u4_3 = nullptr;
};

struct DeviceArena {
    std::vector<void*> ptrs;
    int upload(const std::vector<float>& h, float** out) {
        *out = nullptr;
        if (h.empty()) return MM_OK;
        void* d = nullptr;
        MM_HIP(hipMalloc(&d, h.size() * sizeof(float)));
        ptrs.push_back(d);
        MM_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        *out = (float*)d;
        return MM_OK;
    }
    void release() {
        for (void* p : ptrs) (void)hipFree(p);
        ptrs.clear();
    }
};

struct BN {
    const float *gamma, *beta, *mean, *var;
};

// Build one conv/linear layer from host tensors.  w: [cout][cin][k][k] (OIHW), bias may be null.
// fold: BN directly after the conv (scale into the weights).  post: BN after the ReLU.
static int make_layer(DeviceArena& A, Layer& L, const float* w, const float* bias, int cout, int cin, int k, int stride,
                      int pad, int relu, const BN* fold, const BN* post, float eps, bool wino = false, bool packed3 = false,
                      int wino_pad = 0) {
    if (packed3) {
        // 3-channel input rows packed NHWC3 with the zero border in memory (conv_mfma.hip KMODE 5): k = r * RG + s * 3 + c, RG = the
        // k * 3 floats of a kernel row rounded up to a multiple of 4 (zero weights in the slack)
        if (cin != 3) return MM_ERR_INVALID_ARG;
        const int RG = (k * 3 + 3) / 4 * 4;
        L.cin = 3; L.cin_p = 3; L.cout = cout; L.k = k; L.stride = stride; L.pad = 0; L.relu = relu;
        L.K = k * RG; L.Kpad = (L.K + 15) / 16 * 16; L.korder = 2;
        std::vector<float> hw((size_t)cout * L.Kpad, 0.f), hb(cout, 0.f);
        for (int o = 0; o < cout; ++o) {
            double sc = 1.0, sh = 0.0;
            if (fold) {
                sc = (double)fold->gamma[o] / std::sqrt((double)fold->var[o] + (double)eps);
                sh = (double)fold->beta[o] - (double)fold->mean[o] * sc;
            }
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < k; ++r)
                    for (int q = 0; q < k; ++q)
                        hw[(size_t)o * L.Kpad + r * RG + q * 3 + c] = (float)((double)w[(((size_t)o * 3 + c) * k + r) * k + q] * sc);
            hb[o] = (float)((bias ? (double)bias[o] : 0.0) * sc + sh);
        }
        int rc = A.upload(hw, &L.w);
        if (rc == MM_OK) rc = A.upload(hb, &L.bias);
        return rc;
    }
    L.cin = cin;
    L.cin_p = (cin + 3) / 4 * 4;
    L.cout = cout;
    L.k = k;
    L.stride = stride;
    L.pad = pad;
    L.relu = relu;
    L.K = k * k * L.cin_p;
    L.Kpad = (L.K + 15) / 16 * 16;
    L.korder = (k > 1 && L.cin_p % 16 == 0) ? 1 : 0;  // slice-major K: taps of a 16-channel slice adjacent
    std::vector<float> hw((size_t)cout * L.Kpad, 0.f), hb(cout, 0.f);
    for (int o = 0; o < cout; ++o) {
        double sc = 1.0, sh = 0.0;
        if (fold) {
            sc = (double)fold->gamma[o] / std::sqrt((double)fold->var[o] + (double)eps);
            sh = (double)fold->beta[o] - (double)fold->mean[o] * sc;
        }
        for (int c = 0; c < cin; ++c)
            for (int r = 0; r < k; ++r)
                for (int s = 0; s < k; ++s)
                {
                    const size_t kidx = L.korder == 0 ? (size_t)(r * k + s) * L.cin_p + c
                                                      : ((size_t)((c / 16) * k + r) * k + s) * 16 + c % 16;
                    hw[(size_t)o * L.Kpad + kidx] = (float)((double)w[(((size_t)o * cin + c) * k + r) * k + s] * sc);
                }
        hb[o] = (float)((bias ? (double)bias[o] : 0.0) * sc + sh);
    }
    int rc = A.upload(hw, &L.w);
    if (rc == MM_OK) rc = A.upload(hb, &L.bias);
    // wino_pad: Winograd-domain K rounded up to a multiple of it with zero columns (a 88-channel input on the 64-deep fused kernel)
    const int wcin = wino_pad > 0 ? (cin + wino_pad - 1) / wino_pad * wino_pad : cin;
    if (rc == MM_OK && wino && k == 3 && stride == 1 && pad == 1 && wcin % 16 == 0 && cin >= 64 && cin % 4 == 0 && cout % 4 == 0) {
        L.wino_cin = wcin;
        // U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], on the BN-folded filter, float64 -> fp32
        static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        std::vector<float> hu((size_t)16 * cout * wcin, 0.f);
        for (int o = 0; o < cout; ++o) {
            const double sc = fold ? (double)fold->gamma[o] / std::sqrt((double)fold->var[o] + (double)eps) : 1.0;
            for (int c = 0; c < cin; ++c) {
                double g[3][3], t[4][3];
                for (int r = 0; r < 3; ++r)
                    for (int q = 0; q < 3; ++q) g[r][q] = (double)w[(((size_t)o * cin + c) * 3 + r) * 3 + q] * sc;
                for (int i = 0; i < 4; ++i)
                    for (int q = 0; q < 3; ++q) t[i][q] = G[i][0] * g[0][q] + G[i][1] * g[1][q] + G[i][2] * g[2][q];
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j)
                        hu[((size_t)(i * 4 + j) * cout + o) * wcin + c] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
            }
        }
        rc = A.upload(hu, &L.wino_u);
        // F(4x4,3x3): G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
        static const double G4[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                        {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
        std::vector<float> hu4((size_t)36 * cout * wcin, 0.f);
        for (int o = 0; o < cout && rc == MM_OK; ++o) {
            const double sc = fold ? (double)fold->gamma[o] / std::sqrt((double)fold->var[o] + (double)eps) : 1.0;
            for (int c = 0; c < cin; ++c) {
                double g[3][3], t[6][3];
                for (int r = 0; r < 3; ++r)
                    for (int q = 0; q < 3; ++q) g[r][q] = (double)w[(((size_t)o * cin + c) * 3 + r) * 3 + q] * sc;
                for (int i = 0; i < 6; ++i)
                    for (int q = 0; q < 3; ++q) t[i][q] = G4[i][0] * g[0][q] + G4[i][1] * g[1][q] + G4[i][2] * g[2][q];
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j)
                        hu4[((size_t)(i * 6 + j) * cout + o) * wcin + c] = (float)(t[i][0] * G4[j][0] + t[i][1] * G4[j][1] + t[i][2] * G4[j][2]);
            }
        }
        if (rc == MM_OK) rc = A.upload(hu4, &L.wino_u4);
    }
    if (rc == MM_OK && post) {
        std::vector<float> ps(cout), pt(cout);
        for (int o = 0; o < cout; ++o) {
            const double sc = (double)post->gamma[o] / std::sqrt((double)post->var[o] + (double)eps);
            ps[o] = (float)sc;
            pt[o] = (float)((double)post->beta[o] - (double)post->mean[o] * sc);
        }
        rc = A.upload(ps, &L.ps);
        if (rc == MM_OK) rc = A.upload(pt, &L.pt);
    }
    return rc;
}

// A residual block's increase conv (w1 [cout][c1], BN bn1) and its projection shortcut (w2 [cout][c2], BN bn2) as one 1x1 layer over
// K = c1 + c2: relu(BN1(w1 a) + BN2(w2 x)) = relu([s1 w1 | s2 w2] [a; x] + (t1 + t2))  (conv_mfma.hip KMODE 6).
static int make_layer_dual(DeviceArena& A, Layer& L, const float* w1, const BN& bn1, int c1, const float* w2, const BN& bn2, int c2,
                           int cout, float eps) {
    if (c1 % 16 || c2 % 16) return MM_ERR_UNSUPPORTED;
    L.cin = c1; L.cin_p = c1; L.cout = cout; L.k = 1; L.stride = 1; L.pad = 0; L.relu = 1; L.korder = 0;
    L.K = L.Kpad = c1 + c2;
    std::vector<float> hw((size_t)cout * L.K), hb(cout);
    for (int o = 0; o < cout; ++o) {
        const double s1 = (double)bn1.gamma[o] / std::sqrt((double)bn1.var[o] + (double)eps);
        const double s2 = (double)bn2.gamma[o] / std::sqrt((double)bn2.var[o] + (double)eps);
        for (int c = 0; c < c1; ++c) hw[(size_t)o * L.K + c] = (float)((double)w1[(size_t)o * c1 + c] * s1);
        for (int c = 0; c < c2; ++c) hw[(size_t)o * L.K + c1 + c] = (float)((double)w2[(size_t)o * c2 + c] * s2);
        hb[o] = (float)(((double)bn1.beta[o] - (double)bn1.mean[o] * s1) + ((double)bn2.beta[o] - (double)bn2.mean[o] * s2));
    }
    int rc = A.upload(hw, &L.w);
    if (rc == MM_OK) rc = A.upload(hb, &L.bias);
    return rc;
}

constexpr int kX3MinK = 512;   // mm_resnet50_set_precision(h, 1): 1x1 layers with K >= this run on the bf16 pipes (three-way split)

static int run_layer(const Layer& L, const float* in, int B, int H, int W, int in_cstride, int in_coff, float* out,
                     int out_cstride, int out_coff, const float* res, int res_cstride, hipStream_t s, int* Ho_ = nullptr,
                     int* Wo_ = nullptr, int x3 = 0, int no_sched = 0, int hpool = 0, int use_panel = 0) {
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.in = in; p.w = L.w; p.bias = L.bias; p.res = res; p.post_scale = L.ps; p.post_shift = L.pt; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = L.cin_p; p.in_cstride = in_cstride; p.in_coff = in_coff;
    p.Ho = (H + 2 * L.pad - L.k) / L.stride + 1;
    p.Wo = (W + 2 * L.pad - L.k) / L.stride + 1;
    p.Cout = L.cout; p.out_cstride = out_cstride; p.out_coff = out_coff;
    p.res_cstride = res_cstride; p.res_coff = 0;
    p.kh = L.k; p.kw = L.k; p.stride = L.stride; p.pad = L.pad;
    p.K = L.K; p.Kpad = L.Kpad; p.relu = L.relu; p.Cin_real = L.cin; p.korder = L.korder; p.force_tile = L.tile;
    p.x3 = x3 && L.k == 1 && L.Kpad >= kX3MinK;
    p.w3 = p.x3 ? L.w3 : nullptr; p.w3_plane = (int64_t)L.cout * L.Kpad;
    p.no_sched = no_sched;
    p.hpool = hpool;
    p.use_panel = use_panel;
    if (Ho_) *Ho_ = p.Ho;
    if (Wo_) *Wo_ = p.Wo;
    return conv_forward(p, s);
}

// Stride-1 3x3 layer through Winograd F(2x2,3x3): input transform, ONE batched GEMM launch (16 problems), output
// transform with the fused bias/ReLU.  V and M are caller-provided scratch of 16*B*ceil(H/2)*ceil(W/2)*C floats.
// out = relu(L over [in | in2 sampled at stride2]); in [B,H,W,L.cin], in2 [B,H2,W2,C2]
static int run_layer_dual(const Layer& L, const float* in, int B, int H, int W, const float* in2, int H2, int W2, int C2, int stride2,
                          float* out, hipStream_t s, int x3 = 0, int no_sched = 0) {
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.in = in; p.w = L.w; p.bias = L.bias; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = L.cin; p.in_cstride = L.cin; p.Ho = H; p.Wo = W;
    p.Cout = L.cout; p.out_cstride = L.cout; p.kh = 1; p.kw = 1; p.stride = 1;
    p.K = L.K; p.Kpad = L.Kpad; p.relu = 1; p.Cin_real = L.K;
    p.in2 = in2; p.H2 = H2; p.W2 = W2; p.C2 = C2; p.in2_cstride = C2; p.stride2 = stride2;
    p.x3 = x3 && L.Kpad >= kX3MinK;
    p.w3 = p.x3 ? L.w3 : nullptr; p.w3_plane = (int64_t)L.cout * L.Kpad;
    p.no_sched = no_sched;
    return conv_forward(p, s);
}

static int g_wino_fused_max_cin = 256;   // fused kernel for Cin <= this (conv5_x stays on the three-kernel form); -DMM_MEASURE builds: MM_WF_MAX_CIN
static int g_wino_fused_shape = 0;       // workgroup shape of the fused kernel, 0 = auto; -DMM_MEASURE builds: MM_WINO_FUSED_SHAPE

// inc / inc_res / inc_out (optional, fused form only): the residual block's increase layer applied inside the fused kernel --
// inc_out [B,H,W,inc->cout] = relu(inc(relu(L(in))) + inc_res); `out` is then not written.  Returns MM_ERR_UNSUPPORTED (before
// anything is launched) when the shape is not the one wino_fused.hip's INC instantiation takes.
// inc_two_src: `inc` is a make_layer_dual layer (K = L.cout + L.cout: increase | stride-1 projection) and inc_res the block input x
// [B,H,W,L.cout] -- inc_out = relu(inc([relu(L(in)); x])), no residual.
// next / next_out (optional, with inc on the (64, 256) shape only): the NEXT block's stride-1 1x1 reduce layer (inc->cout -> 64) run inside the
// same kernel on the block output -- next_out [B,H,W,64] = relu(next(inc_out)); inc_shape: workgroup variant of that kernel (measurement knob)
static int run_layer_wino(const Layer& L, const float* in, int B, int H, int W, float* out, float* V, float* M, int m, hipStream_t s,
                          const Layer* inc = nullptr, const float* inc_res = nullptr, float* inc_out = nullptr, bool inc_two_src = false,
                          int x3 = 0, int generic_loop = 0, int no_sched = 0, const Layer* next = nullptr, float* next_out = nullptr,
                          int inc_shape = 0) {
    const int mt = m == 5 ? 4 : m;
    const int TH = (H + mt - 1) / mt, TW = (W + mt - 1) / mt, npos = (mt + 2) * (mt + 2);
    const int64_t ntile = (int64_t)B * TH * TW;
    if (ntile > 0x7fffffff) return MM_ERR_INVALID_ARG;
    const bool fused = m == 5;          // F(4x4,3x3) with the output transform inside the GEMM kernel (wino_fused.hip)
    if (fused) m = 4;
    const int wc = L.wino_cin;          // K of the position GEMMs (= cin unless the layer was built with zero-padded columns)
    if (wc != L.cin_p && m != 4) return MM_ERR_UNSUPPORTED;
    // no plane set for the three-kernel form and the fused kernel declines the shape: say so before transforming anything
    if (!M && !(fused && wino_fused_supported(ntile, wc, L.cout))) return MM_ERR_UNSUPPORTED;
    // (the two-source form exists for Cout == 64 only: decline here, before the input transform is launched)
    if (inc && !(fused && inc->k == 1 && inc->stride == 1 && inc->Kpad == (inc_two_src ? 2 : 1) * L.cout && inc->korder == 0 && inc->relu &&
                 !inc->ps && (!inc_two_src || L.cout == 64) && wino_fused_inc_supported(ntile, wc, L.cout, inc->cout)))
        return MM_ERR_UNSUPPORTED;
    if (next && !(inc && !inc_two_src && L.cout == 64 && inc->cout == 256 && next_out && next->k == 1 && next->stride == 1 && next->pad == 0 &&
                  next->cin_p == 256 && next->Kpad == 256 && next->cout == 64 && next->korder == 0 && next->relu && !next->ps && next->bias))
        return MM_ERR_UNSUPPORTED;
    int rc = wino_input_transform(in, V, B, H, W, wc, m, s, L.cin_p);
    if (rc != MM_OK) return rc;
    if (inc && inc_two_src)
        return wino_gemm_output_fused_incproj(V, L.wino_u4, L.bias, inc->w, inc->bias, inc_res, inc_out, B, H, W, wc, L.cout, inc->cout,
                                              L.relu, s, generic_loop);
    if (inc)
        return wino_gemm_output_fused_inc(V, L.wino_u4, L.bias, inc->w, inc->bias, inc_res, inc_out, B, H, W, wc, L.cout, inc->cout,
                                          L.relu, s, generic_loop, next ? next->w : nullptr, next ? next->bias : nullptr, next_out, inc_shape);
    if (fused) {
        rc = wino_gemm_output_fused(V, L.wino_u4, L.bias, out, B, H, W, wc, L.cout, L.relu, g_wino_fused_shape, s, generic_loop);
        if (rc != MM_ERR_UNSUPPORTED) return rc;
    }
    if (!M) return MM_ERR_UNSUPPORTED;  // caller provided no plane set for the three-kernel form
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.in = V; p.w = m == 4 ? L.wino_u4 : L.wino_u; p.out = M;
    p.B = (int)ntile; p.H = 1; p.W = 1; p.Cin = wc; p.in_cstride = wc; p.Ho = 1; p.Wo = 1;
    p.Cout = L.cout; p.out_cstride = L.cout; p.kh = 1; p.kw = 1; p.stride = 1; p.K = wc; p.Kpad = wc; p.Cin_real = wc;
    p.batch = npos; p.in_bstride = ntile * wc; p.w_bstride = (int64_t)L.cout * wc; p.out_bstride = ntile * L.cout;
    p.x3 = x3 && wc >= kX3MinK;      // the position GEMMs of the three-kernel form (conv5_x: K = 512)
    p.w3 = (p.x3 && m == 4) ? L.wino_u4_3 : nullptr; p.w3_plane = (int64_t)npos * L.cout * wc;
    p.no_sched = no_sched;
    rc = conv_forward(p, s);
    if (rc != MM_OK) return rc;
    return wino_output_transform(M, L.bias, out, B, H, W, L.cout, L.relu, m, s);
}

struct Bump {
    char* base;
    int64_t off = 0, cap;
    Bump(void* b, int64_t c) : base((char*)b), cap(c) {}
    float* take(int64_t floats) {
        const int64_t bytes = (floats * 4 + 255) / 256 * 256;
        float* p = (float*)(base + off);
        off += bytes;
        return p;
    }
    static int64_t size_of(int64_t floats) { return (floats * 4 + 255) / 256 * 256; }
};

// ======================================= ResNet50 =====================================================
static const int kStages[4][4] = {{3, 64, 256, 1}, {4, 128, 512, 2}, {6, 256, 1024, 2}, {3, 512, 2048, 2}};

struct Bottleneck {
    Layer proj, reduce, conv3, increase;
    Layer inc_proj;   // increase + projection shortcut as one contraction (make_layer_dual); w == null when not built
    int proj_stride;
    bool has_proj;
};

}  // namespace mm

struct mm_resnet50 {
    mm::DeviceArena arena;
    mm::Layer stem;    // 7x7/2 on NHWC4 input (K = 196 -> 208)
    mm::Layer stem3;   // the same layer on zero-bordered NHWC3 input (K = 168 -> 176): input mode 2 of mm_resnet50_forward
    std::vector<mm::Bottleneck> blocks;
    int ceil_mode;
    int winograd;  // 0 direct, 2 = F(2x2,3x3), 4 = F(4x4,3x3) for the layers that have Winograd-domain weights
    int fuse_proj; // 1 (default): the first block of a stage runs increase + projection as one launch
    int fuse_pool; // pool1 and conv2_1's 1x1 reduce conv run as one kernel (pool_reduce.hip): 0 = two launches (the parity twin), 1 = one kernel over
                   // the full stem output, 2 (default) = the packed-NHWC3 stem also pools horizontally in its epilogue (conv_mfma.hip hpool) and the
                   // pool kernel finishes vertically -- the 112 x 112 x 64 stem output never exists
    int no_sched;  // MM_CONV_SCHED=0 at create time: 1x1 layers on the engine's modes 3 / 6 instead of the scheduled loop (modes 7 / 8): the parity twin
    int use_panel; // MM_CONV_PANEL=1 at create time: the 256 -> 1024 (+ residual) layers on the LDS-resident-panel kernel (conv_panel.hip, round 6:
                   // built, bit-identical, measured slower -- default off, the tested twin)
    int wf_generic;// MM_WF_KSL=0 at create time: the fused Winograd kernels' runtime-scheduled main loop (the parity twin of round 5's compile-time one)
    int precision; // 0 (default): every contraction on the fp32 matrix pipes; 1: 1x1 layers with K >= 512 through the three-way bf16 split
                   // (mm_resnet50_set_precision; bench.py's extra.bf16x3 -- never the headline)
    int fuse_next; // round 6: the fused conv2_x kernel of a block also runs the NEXT block's 256 -> 64 reduce conv on its output (wino_fused.hip NEXT): 0
                   // (default) = off, 1 (MM_FUSE_NEXT=1) = conv2_x block 2 -> block 3.  Built, parity-tested, measured SLOWER (+0.3 ms per step,
                   // profiles/r06_ab_next_reduce.txt: the kernel's time is MFMA time + memory time, added matrix work hides under nothing)
    int inc_shape; // workgroup variant of the conv2_x fused kernel (measurement knob MM_INC1_SHAPE: 8 = eight waves without NEXT)
    int fuse_inc;  // 3x3 + increase conv (+ residual | + projection) in ONE kernel (wino_fused.hip INC): 0 = never (the parity twin), 1 = conv2_x
                   // blocks 2-3, 2 = also conv2_x block 1 (increase | projection over two K sources), 3 (default) = also conv3_x blocks 2-4
    int device;
};

struct mm_head {
    mm::DeviceArena arena;
    std::vector<mm::Layer> mlp;   // MLP(hidden_units): Linear -> BN -> ReLU per layer (api/mimamo_net.py:6-26); published: 2048 -> 256 -> 256
    int feat_dim, mlp_max;        // hidden_units[0] (width of the rgb features), widest hidden layer
    mm::Layer conv[6], fc1, fc2, transform, gru_ih[2], gru_hh[2][2], classifier;
    float* bhh[2][2];
    int winograd;   // 1 (default): PhaseNet's 128 -> 256 3x3 layer through the fused F(4x4,3x3) kernel; MM_HEAD_WINOGRAD=0: direct form
    int no_sched;   // MM_CONV_SCHED=0 at create time: the conv engine's base loops instead of the scheduled ones (modes 7, 10, 11): the parity twin
    int pc;         // PhaseNet input channels = 2 * num_phase (api/mimamo_net.py:112): 24 for the published model
    int pcp;        // pc rounded up to a multiple of 4: channel stride of the NHWC phase buffers (== pc unless num_phase is odd)
    int device;
};

namespace mm {

static int64_t resnet_blob_floats() {
    int64_t n = 64 * 3 * 49 + 4 * 64;
    int cin = 64;
    for (auto& st : kStages)
        for (int b = 0; b < st[0]; ++b) {
            const int mid = st[1], cout = st[2];
            if (b == 0) n += (int64_t)cout * cin + 4 * cout;
            n += (int64_t)mid * cin + 4 * mid;
            n += (int64_t)mid * mid * 9 + 4 * mid;
            n += (int64_t)cout * mid + 4 * cout;
            cin = cout;
        }
    return n;
}

// per-frame workspace floats (see mm_resnet50_forward)
static const int64_t kRsIn4 = 224 * 224 * 4, kRsBig = 112 * 112 * 64, kRsMid = 56 * 56 * 128;
static const int64_t kRsWino = 36 * 14 * 14 * 64;   // largest Winograd plane set: conv2_x under F(4x4,3x3) (> kRsMid)

static const int kMlpPublished[3] = {2048, 256, 256};   // api/tester.py:45

// hidden_units as api/mimamo_net.py:7-12 takes them: >= 2 entries, last == 256; here also multiples of 4 (16-byte channel groups)
static bool mlp_units_ok(int n_units, const int* units) {
    if (n_units < 2 || n_units > 16 || !units || units[n_units - 1] != 256) return false;
    for (int i = 0; i < n_units; ++i)
        if (units[i] <= 0 || units[i] % 4 || units[i] > 65536) return false;
    return true;
}

// PhaseNet input channels pc = 2 * num_phase (api/mimamo_net.py:112): any num_phase up to 128.  Channel groups are 16 bytes, so an odd
// num_phase (pc % 4 == 2) runs on buffers padded to pcp = pc rounded up to 4 with zero channels meeting zero weights; the concat layer takes
// the fused Winograd kernel while 64 + pcp <= 128 and pc % 4 == 0, the direct form otherwise.
static bool phase_channels_ok(int pc) { return pc >= 2 && pc <= 256 && pc % 2 == 0; }
static int pad4(int c) { return (c + 3) / 4 * 4; }

static int64_t head_blob_floats(int n_units, const int* units, int pc = 24) {
    int64_t n = 0;
    auto lin = [&](int o, int i) { n += (int64_t)o * i + o; };
    auto bn = [&](int c) { n += 4 * c; };
    for (int i = 1; i < n_units; ++i) { lin(units[i], units[i - 1]); bn(units[i]); }
    const int ch[3][2] = {{pc, 64}, {64 + pc, 128}, {128, 256}};
    for (auto& c : ch) { n += (int64_t)c[1] * c[0] * 9 + c[1]; bn(c[1]); n += (int64_t)c[1] * c[1] * 9 + c[1]; bn(c[1]); }
    lin(256, 256); bn(256); lin(256, 256); bn(256); lin(1, 256); bn(1);
    lin(256, 512); bn(256);
    n += 4 * (384 * 256 + 384 * 128 + 384 + 384);
    lin(2, 256); bn(2);
    return n;
}

}  // namespace mm

extern "C" {

int mm_conv2d_nhwc(const float* in, const float* w, const float* bias, const float* residual, const float* post_scale,
                   const float* post_shift, float* out, int B, int H, int W, int Cin, int in_cstride, int in_coff,
                   int Cout, int out_cstride, int out_coff, int res_cstride, int kh, int kw, int stride, int pad,
                   int relu, int tile, int korder, void* stream) {
    using namespace mm;
    if (!in || !w || !out || B < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 ||
        pad < 0 || tile < 0)
        return MM_ERR_INVALID_ARG;
#ifdef MM_MEASURE
    if (tile > 2047) return MM_ERR_INVALID_ARG;     // tile >= 16: the ablation instantiation of conv_mfma.hip (measurement builds only)
#else
    if (tile > 5) return MM_ERR_INVALID_ARG;        // the documented range; the default library has no measurement modes
#endif
    if ((post_scale == nullptr) != (post_shift == nullptr)) return MM_ERR_INVALID_ARG;
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.in = in; p.w = w; p.bias = bias; p.res = residual; p.post_scale = post_scale; p.post_shift = post_shift; p.out = out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.in_cstride = in_cstride; p.in_coff = in_coff;
    p.Ho = (H + 2 * pad - kh) / stride + 1;
    p.Wo = (W + 2 * pad - kw) / stride + 1;
    if (p.Ho <= 0 || p.Wo <= 0 || (int64_t)B * p.Ho * p.Wo > 0x7fffffff) return MM_ERR_INVALID_ARG;
    p.Cout = Cout; p.out_cstride = out_cstride; p.out_coff = out_coff; p.res_cstride = res_cstride;
    p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad;
    p.K = kh * kw * Cin; p.Kpad = (p.K + 15) / 16 * 16; p.relu = relu; p.force_tile = tile; p.korder = korder;
    {
        const char* cp = getenv("MM_CONV_PANEL");      // read per call (a test switches it): opt-in panel kernel for the shapes it takes
        p.use_panel = cp ? (atoi(cp) == 1 ? 1 : atoi(cp) == 2 ? 2 : 0) : 0;
    }
    return conv_forward(p, (hipStream_t)stream);
}

int64_t mm_resnet50_blob_floats(void) { return mm::resnet_blob_floats(); }

int mm_resnet50_create(mm_resnet50_t** out, const float* blob, int64_t n_floats, int stride_on_first_1x1,
                       int maxpool_ceil_mode, float bn_eps) {
    using namespace mm;
    if (!out) return MM_ERR_INVALID_ARG;
    *out = nullptr;
    if (!blob || n_floats != resnet_blob_floats()) return MM_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return MM_ERR_NO_DEVICE;
    mm_resnet50* h = new (std::nothrow) mm_resnet50();
    if (!h) return MM_ERR_INVALID_ARG;
    h->ceil_mode = maxpool_ceil_mode;
    h->device = current_device_or(0);
    const float* p = blob;
    int rc = MM_OK;
    h->winograd = 1;
    h->precision = 0;
    {
        const char* wk = getenv("MM_WF_KSL");      // measurement knob / parity twin: 0 = the generic main loop in the fused Winograd kernels
        h->wf_generic = wk ? atoi(wk) == 0 : 0;
        const char* cs = getenv("MM_CONV_SCHED");  // measurement knob / parity twin: 0 = no scheduled 1x1 loop in the conv engine
        h->no_sched = cs ? atoi(cs) == 0 : 0;
        const char* cp = getenv("MM_CONV_PANEL");  // measurement knob / parity twin: 1 = the LDS-resident-panel kernel for the K = 256 increase layers
        h->use_panel = cp ? (atoi(cp) == 1 ? 1 : atoi(cp) == 2 ? 2 : 0) : 0;
        const char* fp = getenv("MM_FUSE_PROJ");   // measurement knob: 0 = projection shortcut as its own launch + residual read
        h->fuse_proj = fp ? atoi(fp) : 1;
        const char* fpl = getenv("MM_FUSE_POOL");  // measurement knob: 0 = max-pool and conv2_1's reduce conv as two launches (the parity twin),
        h->fuse_pool = fpl ? atoi(fpl) : 2;        // 1 = one kernel on the full stem output, 2 = horizontal half of the pool in the stem's epilogue
        const char* fi = getenv("MM_FUSE_INC");    // measurement knob: 0 = 3x3 and increase conv as separate launches (the parity twin)
        // 1 = conv2_x blocks 2, 3 only; 2 = also block 1 (increase | projection over two K sources); 3 = also conv3_x blocks 2-4
        h->fuse_inc = fi ? atoi(fi) : 3;
        const char* fn = getenv("MM_FUSE_NEXT");   // measurement knob: 1 = the next block's reduce conv inside the fused conv2_x kernel (measured slower)
        h->fuse_next = fn ? atoi(fn) : 0;
        const char* is = getenv("MM_INC1_SHAPE");  // measurement knob: 8 = the conv2_x fused kernel as eight-wave workgroups (without NEXT)
        h->inc_shape = is ? atoi(is) : 0;
    }
    auto conv_bn = [&](Layer& L, int cout, int cin, int k, int stride, int pad, int relu) {
        const float* w = p;
        p += (int64_t)cout * cin * k * k;
        BN bn{p, p + cout, p + 2 * cout, p + 3 * cout};
        p += 4 * cout;
        if (rc == MM_OK) rc = make_layer(h->arena, L, w, nullptr, cout, cin, k, stride, pad, relu, &bn, nullptr, bn_eps, true);
    };
    {   // both stem forms share the blob entry
        const float* w = p;
        BN bn{p + 64 * 3 * 49, p + 64 * 3 * 49 + 64, p + 64 * 3 * 49 + 128, p + 64 * 3 * 49 + 192};
        rc = make_layer(h->arena, h->stem3, w, nullptr, 64, 3, 7, 2, 0, 1, &bn, nullptr, bn_eps, false, true);
    }
    conv_bn(h->stem, 64, 3, 7, 2, 3, 1);
#ifdef MM_MEASURE
    if (const char* st = getenv("MM_STEM_TILE")) {   // measurement builds only: 1 = 128x128, 2 = 128x64 (automatic choice), 3 = 64x64, 4 = 256x64
        const int t = atoi(st);
        if (t >= 0 && t <= 4) h->stem.tile = h->stem3.tile = t;
    }
#endif
    int cin = 64;
    for (auto& st : kStages)
        for (int b = 0; b < st[0]; ++b) {
            const int mid = st[1], cout = st[2], s = b == 0 ? st[3] : 1;
            const int s1 = stride_on_first_1x1 ? s : 1, s3 = stride_on_first_1x1 ? 1 : s;
            h->blocks.emplace_back();
            Bottleneck& B = h->blocks.back();
            B.has_proj = b == 0;
            B.proj_stride = s;
            const float* w_proj = p;
            if (b == 0) conv_bn(B.proj, cout, cin, 1, s, 0, 0);
            conv_bn(B.reduce, mid, cin, 1, s1, 0, 1);
            conv_bn(B.conv3, mid, mid, 3, s3, 1, 1);
            const float* w_inc = p;
            conv_bn(B.increase, cout, mid, 1, 1, 0, 1);  // ReLU applies after the residual add (fused epilogue)
            if (b == 0 && rc == MM_OK) {
                const float* q1 = w_inc + (int64_t)cout * mid;
                const float* q2 = w_proj + (int64_t)cout * cin;
                const BN bn1{q1, q1 + cout, q1 + 2 * cout, q1 + 3 * cout}, bn2{q2, q2 + cout, q2 + 2 * cout, q2 + 3 * cout};
                const int rd = make_layer_dual(h->arena, B.inc_proj, w_inc, bn1, mid, w_proj, bn2, cin, cout, bn_eps);
                if (rd != MM_OK && rd != MM_ERR_UNSUPPORTED) rc = rd;
            }
            cin = cout;
        }
    if (rc != MM_OK) {
        h->arena.release();
        delete h;
        return rc;
    }
    *out = h;
    return MM_OK;
}

int mm_resnet50_destroy(mm_resnet50_t* h) {
    if (!h) return MM_OK;
    h->arena.release();
    delete h;
    return MM_OK;
}

int64_t mm_resnet50_workspace_bytes(mm_resnet50_t* h, int64_t batch) {
    using namespace mm;
    if (!h || batch < 0) return MM_ERR_INVALID_ARG;
    return Bump::size_of(batch * kRsIn4) + 3 * Bump::size_of(batch * kRsBig) + 2 * Bump::size_of(batch * kRsMid) +
           2 * Bump::size_of(batch * kRsWino);
}

int mm_resnet50_set_winograd(mm_resnet50_t* h, int enable) {
    if (!h) return MM_ERR_INVALID_ARG;
    if (enable != 0 && enable != 1 && enable != 2 && enable != 4 && enable != 5) return MM_ERR_INVALID_ARG;
    h->winograd = enable;   // 1 = default: F(4x4,3x3), output transform fused into the GEMMs where that is faster (Cin <= 256)
#ifdef MM_MEASURE
    const char* mc = getenv("MM_WF_MAX_CIN");         // measurement builds only (tuning sweeps of rounds 2-4)
    if (mc) mm::g_wino_fused_max_cin = atoi(mc);
    const char* sh = getenv("MM_WINO_FUSED_SHAPE");
    mm::g_wino_fused_shape = sh ? atoi(sh) : 0;
#endif
    return MM_OK;
}

// bf16x3 with MM_X3_PRESPLIT=1 (round 6, verdict item 7 -- OPT-IN: built, bit-identical, measured SLOWER, profiles/r06_ab_x3_presplit.txt): the
// weights of the layers the mode touches, split once into three bf16 planes; the 128x256 tile then needs no split of its B fragments in the
// loop.  Default: every split in the loop (round-5 form).
static int ensure_x3_weights(mm_resnet50_t* h) {
    using namespace mm;
    const char* e = getenv("MM_X3_PRESPLIT");
    if (!e || atoi(e) != 1) return MM_OK;
    auto split = [&](const float* w, int64_t n, unsigned short** out) -> int {
        if (*out || !w || n <= 0 || n % 16) return MM_OK;
        void* d = nullptr;
        MM_HIP(hipMalloc(&d, (size_t)n * 3 * sizeof(unsigned short)));
        h->arena.ptrs.push_back(d);
        const int rc = bf16x3_split_weights(w, (unsigned short*)d, n, nullptr);
        if (rc != MM_OK) return rc;
        *out = (unsigned short*)d;
        return MM_OK;
    };
    for (auto& Bk : h->blocks) {
        for (Layer* L : {&Bk.proj, &Bk.reduce, &Bk.increase, &Bk.inc_proj})
            if (L->w && L->k == 1 && L->Kpad >= kX3MinK && L->cout % 256 == 0) {
                const int rc = split(L->w, (int64_t)L->cout * L->Kpad, &L->w3);
                if (rc != MM_OK) return rc;
            }
        Layer& C = Bk.conv3;
        if (C.wino_u4 && C.wino_cin >= kX3MinK && C.cout % 256 == 0) {
            const int rc = split(C.wino_u4, (int64_t)36 * C.cout * C.wino_cin, &C.wino_u4_3);
            if (rc != MM_OK) return rc;
        }
    }
    MM_HIP(hipDeviceSynchronize());
    return MM_OK;
}

int mm_resnet50_set_precision(mm_resnet50_t* h, int mode) {
    if (!h || (mode != 0 && mode != 1)) return MM_ERR_INVALID_ARG;
    if (mode == 1) {
        MM_CHECK_DEVICE(h);
        const int rc = ensure_x3_weights(h);
        if (rc != MM_OK) return rc;
    }
    h->precision = mode;
    return MM_OK;
}

int mm_resnet50_forward(mm_resnet50_t* h, const float* images, int nchw, int64_t batch, float* out, void* workspace,
                        int64_t workspace_bytes, void* stream_) {
    using namespace mm;
    if (!h || batch < 0 || (batch > 0 && (!images || !out || !workspace))) return MM_ERR_INVALID_ARG;
    if (batch == 0) return MM_OK;
    if (batch > 40000) return MM_ERR_INVALID_ARG;  // M = batch*112*112 must fit int32
    MM_CHECK_DEVICE(h);
    if (workspace_bytes < mm_resnet50_workspace_bytes(h, batch)) return MM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream_;
    const int B = (int)batch;
    Bump ws(workspace, workspace_bytes);
    float* in4 = ws.take(batch * kRsIn4);
    float* big[3] = {ws.take(batch * kRsBig), ws.take(batch * kRsBig), ws.take(batch * kRsBig)};
    float* y1 = ws.take(batch * kRsMid);
    float* y2 = ws.take(batch * kRsMid);
    float* wv = ws.take(batch * kRsWino);  // Winograd V: (m+2)^2 * tiles * C floats per frame
    float* wm = ws.take(batch * kRsWino);  // Winograd M
    int rc;
    const float* x0 = images;
    int H = 224, W = 224, Ho, Wo;
    // pool1 + conv2_1_1x1_reduce in one kernel when that block starts with a stride-1 64 -> 64 1x1 conv (the published graph); with the
    // packed three-channel stem (fuse_pool 2) the stem's epilogue does the horizontal half of the pool
    const mm::Layer& R0 = h->blocks.front().reduce;
    const bool pooled_reduce = h->fuse_pool && R0.k == 1 && R0.stride == 1 && R0.pad == 0 && R0.cin_p == 64 && R0.cout == 64 && R0.Kpad == 64 &&
                               R0.korder == 0 && !R0.ps;
    const bool hpool = pooled_reduce && h->fuse_pool >= 2 && nchw;
    if (nchw) {
        // zero-bordered packed NHWC3 [batch, 230, 230, 3]: handed over (2, mm_preproc_forward layout 2) or converted from the
        // reference's NCHW (1) into the workspace (230 * 230 * 3 < 224 * 224 * 4 floats); stem with K = 168
        if (nchw == 1) {
            rc = nchw3_to_bordered_nhwc3(images, in4, batch, 224, 3, s);
            if (rc != MM_OK) return rc;
            x0 = in4;
        }
        rc = run_layer(h->stem3, x0, B, 230, 230, 3, 0, big[0], 64, 0, nullptr, 0, s, &Ho, &Wo, 0, h->no_sched, hpool);
    } else {
        rc = run_layer(h->stem, x0, B, H, W, 4, 0, big[0], 64, 0, nullptr, 0, s, &Ho, &Wo, 0, h->no_sched);   // NHWC4: K = 196
    }
    if (rc != MM_OK) return rc;
    if (Ho != 112 || Wo != 112) return MM_ERR_UNSUPPORTED;
    H = Ho; W = Wo;
    // MaxPool2d(3, 2, pad 0, ceil_mode)
    if (h->ceil_mode) {
        Ho = (H - 3 + 1) / 2 + 1; Wo = (W - 3 + 1) / 2 + 1;
        if ((Ho - 1) * 2 >= H) --Ho;
        if ((Wo - 1) * 2 >= W) --Wo;
    } else {
        Ho = (H - 3) / 2 + 1; Wo = (W - 3) / 2 + 1;
    }
    if (pooled_reduce)
        rc = maxpool_reduce64(big[0], R0.w, R0.bias, big[1], y1, batch, H, W, Ho, Wo, R0.relu, s, hpool);
    else
        rc = maxpool3x3s2(big[0], big[1], batch, H, W, 64, Ho, Wo, s);
    if (rc != MM_OK) return rc;
    H = Ho; W = Wo;
    int xi = 1;  // index of the buffer holding the block input
    int C = 64;
    const int x3 = h->precision == 1, ns = h->no_sched;
    bool reduce_done = false;   // the previous block's fused kernel has already written this block's reduce output into y1 (fuse_next)
    for (size_t bi = 0; bi < h->blocks.size(); ++bi) {
        const Bottleneck& Bk = h->blocks[bi];
        const Bottleneck* Nx = bi + 1 < h->blocks.size() ? &h->blocks[bi + 1] : nullptr;
        float* x = big[xi];
        float* sc = big[(xi + 1) % 3];
        float* o = big[(xi + 2) % 3];
        int H1, W1, H2, W2, H3, W3;
        const float* resid = x;
        const bool dual = Bk.has_proj && h->fuse_proj && Bk.inc_proj.w;
        if (Bk.has_proj && !dual) {
            rc = run_layer(Bk.proj, x, B, H, W, C, 0, sc, Bk.proj.cout, 0, nullptr, 0, s, nullptr, nullptr, x3, ns);
            if (rc != MM_OK) return rc;
            resid = sc;
        }
        if ((pooled_reduce && bi == 0) || reduce_done) {
            H1 = H; W1 = W;                        // y1 was written by the pool kernel / by the previous block's fused kernel
            reduce_done = false;
        } else {
            rc = run_layer(Bk.reduce, x, B, H, W, C, 0, y1, Bk.reduce.cout, 0, nullptr, 0, s, &H1, &W1, x3, ns);
            if (rc != MM_OK) return rc;
        }
        // default (1): conv2_x..conv4_x (Cin <= 256) take the fused kernel, conv5_x the three-kernel form (Cin = 512: its
        // position GEMMs are matrix-core bound, 135 vs 109 TFLOP/s, and its M planes are small); per layer in DESIGN.md
        const int wm_ = h->winograd == 1 ? (Bk.conv3.cin <= g_wino_fused_max_cin ? 5 : 4) : h->winograd;
        const int wt_ = wm_ == 5 ? 4 : wm_;   // tile side of the variant
        bool inc_done = false;
        if (wm_ && Bk.conv3.wino_u &&
            (int64_t)(wt_ + 2) * (wt_ + 2) * ((H1 + wt_ - 1) / wt_) * ((W1 + wt_ - 1) / wt_) * Bk.conv3.cin <= kRsWino) {
            rc = MM_ERR_UNSUPPORTED;
            if (wm_ == 5 && h->fuse_inc && !Bk.has_proj && (Bk.conv3.cout == 64 || h->fuse_inc >= 3)) {
                // conv2_x blocks 2, 3 (Cin = Cout = 64 -> 256) and, with MM_FUSE_INC >= 3, conv3_x blocks 2-4 (128 -> 512): 3x3 +
                // increase + residual + ReLU in one kernel
                // ... and (fuse_next) the NEXT block's stride-1 256 -> 64 reduce conv on the block output, written to y1 -- dead once the input transform
                // has read it (same stream): that block then starts at its 3x3 layer
                const bool nx = h->fuse_next && Bk.conv3.cout == 64 && Nx && !Nx->has_proj && Nx->reduce.stride == 1;
                if (nx) {
                    rc = run_layer_wino(Bk.conv3, y1, B, H1, W1, nullptr, wv, nullptr, 5, s, &Bk.increase, x, o, false, 0, h->wf_generic, 0, &Nx->reduce, y1,
                                        h->inc_shape);
                    reduce_done = rc == MM_OK;
                }
                if (!nx || rc == MM_ERR_UNSUPPORTED)
                    rc = run_layer_wino(Bk.conv3, y1, B, H1, W1, nullptr, wv, nullptr, 5, s, &Bk.increase, x, o, false, 0, h->wf_generic, 0, nullptr, nullptr,
                                        h->inc_shape);
                inc_done = rc == MM_OK;
            } else if (wm_ == 5 && h->fuse_inc >= 2 && dual && Bk.proj_stride == 1 && C == Bk.conv3.cout && H1 == H && W1 == W) {
                // conv2_x block 1: 3x3 + (increase | projection of the block input) + ReLU in one kernel
                rc = run_layer_wino(Bk.conv3, y1, B, H1, W1, nullptr, wv, nullptr, 5, s, &Bk.inc_proj, x, o, true, 0, h->wf_generic);
                inc_done = rc == MM_OK;
            }
            if (rc == MM_ERR_UNSUPPORTED) rc = run_layer_wino(Bk.conv3, y1, B, H1, W1, y2, wv, wm, wm_, s, nullptr, nullptr, nullptr, false, x3, h->wf_generic, ns);
            H2 = H1; W2 = W1;
        } else {
            rc = run_layer(Bk.conv3, y1, B, H1, W1, Bk.reduce.cout, 0, y2, Bk.conv3.cout, 0, nullptr, 0, s, &H2, &W2, 0, ns);   // ns: the twin runs base mode 1, not the unrolled mode 11
        }
        if (rc != MM_OK) return rc;
        if (inc_done) {
            H3 = H2; W3 = W2;
        } else if (dual) {
            // relu(BN(increase(y2)) + BN(proj(x))) in one accumulation: the shortcut tensor is never written or re-read
            rc = run_layer_dual(Bk.inc_proj, y2, B, H2, W2, x, H, W, C, Bk.proj_stride, o, s, x3, ns);
            H3 = H2; W3 = W2;
        } else {
            rc = run_layer(Bk.increase, y2, B, H2, W2, Bk.conv3.cout, 0, o, Bk.increase.cout, 0, resid, Bk.increase.cout, s, &H3, &W3, x3, ns, 0, h->use_panel);
        }
        if (rc != MM_OK) return rc;
        H = H3; W = W3; C = Bk.increase.cout;
        xi = (xi + 2) % 3;
    }
    // pool5_7x7_s1 + relu(squeeze)  (resnet50_extractor.py:83)
    if (H != 7 || W != 7) return MM_ERR_UNSUPPORTED;
    return avgpool_hw(big[xi], out, batch, H * W, C, C, 0, 1, s);
}

// ============================================ head =====================================================
int64_t mm_head_blob_floats(void) { return mm::head_blob_floats(3, mm::kMlpPublished); }

int64_t mm_head_blob_floats_mlp(int n_units, const int* units) {
    return mm::mlp_units_ok(n_units, units) ? mm::head_blob_floats(n_units, units) : (int64_t)MM_ERR_INVALID_ARG;
}

int mm_head_create(mm_head_t** out, const float* blob, int64_t n_floats) {
    return mm_head_create_mlp(out, blob, n_floats, 3, mm::kMlpPublished);
}

int mm_head_create_mlp(mm_head_t** out, const float* blob, int64_t n_floats, int n_units, const int* units) {
    return mm_head_create_cfg(out, blob, n_floats, n_units, units, 12);
}

int64_t mm_head_blob_floats_cfg(int n_units, const int* units, int num_phase) {
    if (!mm::mlp_units_ok(n_units, units) || num_phase < 1) return (int64_t)MM_ERR_INVALID_ARG;
    if (!mm::phase_channels_ok(2 * num_phase)) return (int64_t)MM_ERR_UNSUPPORTED;   // as mm_head_create_cfg: more than 128 differences
    return mm::head_blob_floats(n_units, units, 2 * num_phase);
}

int mm_head_create_cfg(mm_head_t** out, const float* blob, int64_t n_floats, int n_units, const int* units, int num_phase) {
    using namespace mm;
    if (!out) return MM_ERR_INVALID_ARG;
    *out = nullptr;
    const int pc = 2 * num_phase;
    if (!mlp_units_ok(n_units, units)) return MM_ERR_INVALID_ARG;
    if (num_phase < 1) return MM_ERR_INVALID_ARG;
    if (!phase_channels_ok(pc)) return MM_ERR_UNSUPPORTED;      // more than 128 differences
    if (!blob || n_floats != head_blob_floats(n_units, units, pc)) return MM_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return MM_ERR_NO_DEVICE;
    mm_head* h = new (std::nothrow) mm_head();
    if (!h) return MM_ERR_INVALID_ARG;
    h->device = current_device_or(0);
    h->pc = pc;
    h->pcp = pad4(pc);
    const float eps = 1e-5f;
    const float* p = blob;
    int rc = MM_OK;
    auto take = [&](int64_t n) { const float* q = p; p += n; return q; };
    auto take_bn = [&](int c) { BN b{p, p + c, p + 2 * c, p + 3 * c}; p += 4 * c; return b; };
    // Linear -> BN -> ReLU  (MLP, mimamo_net.py:14-20)
    auto lin_bn_relu = [&](Layer& L, int o, int i) {
        const float* w = take((int64_t)o * i); const float* b = take(o); BN bn = take_bn(o);
        if (rc == MM_OK) rc = make_layer(h->arena, L, w, b, o, i, 1, 1, 0, 1, &bn, nullptr, eps);
    };
    // Conv3x3(+bias) -> BN -> ReLU  (PhaseNet._make_conv_layer, :68-78)
    auto conv_bn_relu = [&](Layer& L, int o, int i, int stride, bool wino) {
        const float* w = take((int64_t)o * i * 9); const float* b = take(o); BN bn = take_bn(o);
        // Winograd-domain weights for the fused F(4x4,3x3) kernel (stride 1, K a multiple of 64) only for the two layers
        // mm_head_forward runs through it: (64 + pc) -> 128 at 24x24 (the concat layer gets zero columns up to 128; only while
        // 64 + pc <= 128 and pc % 4 == 0 -- make_layer leaves wino_u4 null otherwise and the layer runs in the direct form) and
        // 128 -> 256 at 12x12
        if (rc == MM_OK) rc = make_layer(h->arena, L, w, b, o, i, 3, stride, 1, 1, &bn, nullptr, eps, wino && i <= 128, false, 64);
    };
    // Linear -> ReLU -> BN  (PhaseNet.fc :54-62, transform :115-117)
    auto lin_relu_bn = [&](Layer& L, int o, int i) {
        const float* w = take((int64_t)o * i); const float* b = take(o); BN bn = take_bn(o);
        if (rc == MM_OK) rc = make_layer(h->arena, L, w, b, o, i, 1, 1, 0, 1, nullptr, &bn, eps);
    };
    {
        const char* e = getenv("MM_HEAD_WINOGRAD");   // measurement knob
        h->winograd = e ? atoi(e) : 1;
        const char* cs = getenv("MM_CONV_SCHED");     // measurement knob / parity twin, as in mm_resnet50_create
        h->no_sched = cs ? atoi(cs) == 0 : 0;
    }
    h->feat_dim = units[0];
    h->mlp_max = 256;
    h->mlp.resize(n_units - 1);
    for (int i = 1; i < n_units; ++i) {
        lin_bn_relu(h->mlp[i - 1], units[i], units[i - 1]);
        if (units[i] > h->mlp_max) h->mlp_max = units[i];
    }
    const int ch[3][2] = {{pc, 64}, {64 + pc, 128}, {128, 256}};
    for (int i = 0; i < 3; ++i) {
        conv_bn_relu(h->conv[2 * i], ch[i][1], ch[i][0], 1, i >= 1);      // conv[2], conv[4]
        conv_bn_relu(h->conv[2 * i + 1], ch[i][1], ch[i][1], 2, false);
    }
    lin_relu_bn(h->fc1, 256, 256);
    lin_relu_bn(h->fc2, 256, 256);
    take(256 + 1); take_bn(1);  // phasenet.classifier: unused with feature=True (:91-92)
    lin_relu_bn(h->transform, 256, 512);
    // GRU: state_dict order per (layer, direction): weight_ih, weight_hh, bias_ih, bias_hh
    for (int l = 0; l < 2 && rc == MM_OK; ++l) {
        std::vector<float> wih(768 * 256), bih(768);
        for (int d = 0; d < 2; ++d) {
            const float* w_ih = take(384 * 256); const float* w_hh = take(384 * 128);
            const float* b_ih = take(384); const float* b_hh = take(384);
            std::memcpy(wih.data() + (size_t)d * 384 * 256, w_ih, sizeof(float) * 384 * 256);
            std::memcpy(bih.data() + (size_t)d * 384, b_ih, sizeof(float) * 384);
            if (rc == MM_OK) rc = make_layer(h->arena, h->gru_hh[l][d], w_hh, b_hh, 384, 128, 1, 1, 0, 0, nullptr, nullptr, eps);
            std::vector<float> bh(b_hh, b_hh + 384);
            if (rc == MM_OK) rc = h->arena.upload(bh, &h->bhh[l][d]);
        }
        if (rc == MM_OK) rc = make_layer(h->arena, h->gru_ih[l], wih.data(), bih.data(), 768, 256, 1, 1, 0, 0, nullptr, nullptr, eps);
    }
    {   // classifier: Dropout, Linear(256,2), BatchNorm1d(2)  (:120-122)
        const float* w = take(2 * 256); const float* b = take(2); BN bn = take_bn(2);
        if (rc == MM_OK) rc = make_layer(h->arena, h->classifier, w, b, 2, 256, 1, 1, 0, 0, &bn, nullptr, eps);
    }
    if (rc == MM_OK && p - blob != n_floats) rc = MM_ERR_INVALID_ARG;
    if (rc != MM_OK) {
        h->arena.release();
        delete h;
        return rc;
    }
    *out = h;
    return MM_OK;
}

int mm_head_destroy(mm_head_t* h) {
    if (!h) return MM_OK;
    h->arena.release();
    delete h;
    return MM_OK;
}

namespace {
struct HeadWs {
    int64_t p0n, a0, cat, a1, a2, a3, a4, pool, fc1, m1, feat, f, gi, gh, l0, l1, wv;
};
HeadWs head_sizes(int64_t N, int64_t T, int64_t mlp_max, bool wino, int pc) {   // pc: the PADDED channel count (mm_head::pcp)
    HeadWs s;
    s.p0n = N * 48 * 48 * pc; s.a0 = N * 48 * 48 * 64; s.cat = N * 24 * 24 * (64 + pc); s.a1 = N * 24 * 24 * 128;
    s.a2 = N * 12 * 12 * 128; s.a3 = N * 12 * 12 * 256; s.a4 = N * 6 * 6 * 256; s.pool = N * 256; s.fc1 = N * 256;
    s.m1 = 2 * N * mlp_max; s.feat = N * 512; s.f = N * 256; s.gi = N * 768; s.gh = T * 384; s.l0 = N * 256; s.l1 = N * 256;
    // Winograd planes of the 24x24 layer (36 positions x 36 tiles x 128 channels); the 12x12 layer's fit too
    s.wv = wino ? N * 36 * 36 * 128 : 0;
    return s;
}
}  // namespace

int64_t mm_head_workspace_bytes(mm_head_t* h, int64_t bs, int64_t T) {
    using mm::Bump;
    if (!h || bs < 0 || T < 0) return MM_ERR_INVALID_ARG;
    const HeadWs s = head_sizes(bs * T, T, h->mlp_max, h->winograd != 0, h->pcp);
    const int64_t all[] = {s.p0n, s.a0, s.cat, s.a1, s.a2, s.a3, s.a4, s.pool, s.fc1, s.m1, s.feat, s.f, s.gi, s.gh, s.l0, s.l1, s.wv};
    int64_t tot = 0;
    for (int64_t v : all) tot += Bump::size_of(v);
    return tot;
}

int mm_head_forward(mm_head_t* h, const float* phase_0, const float* phase_1, int phase_nhwc, const float* rgb,
                    int64_t bs, int64_t T, float* out, void* workspace, int64_t workspace_bytes, void* stream_) {
    using namespace mm;
    if (!h || bs < 0 || T < 0) return MM_ERR_INVALID_ARG;
    const int64_t N64 = bs * T;
    if (N64 == 0) return MM_OK;
    if (!phase_0 || !phase_1 || !rgb || !out || !workspace || N64 > 400000) return MM_ERR_INVALID_ARG;
    if (workspace_bytes < mm_head_workspace_bytes(h, bs, T)) return MM_ERR_WORKSPACE;
    MM_CHECK_DEVICE(h);
    hipStream_t s = (hipStream_t)stream_;
    const int N = (int)N64;
    const HeadWs z = head_sizes(N64, T, h->mlp_max, h->winograd != 0, h->pcp);
    Bump ws(workspace, workspace_bytes);
    float* p0n = ws.take(z.p0n); float* a0 = ws.take(z.a0); float* cat = ws.take(z.cat); float* a1 = ws.take(z.a1);
    float* a2 = ws.take(z.a2); float* a3 = ws.take(z.a3); float* a4 = ws.take(z.a4); float* pool = ws.take(z.pool);
    float* fc1 = ws.take(z.fc1); float* m1 = ws.take(z.m1); float* feat = ws.take(z.feat); float* f = ws.take(z.f);
    float* gi = ws.take(z.gi); float* gh = ws.take(z.gh); float* l0 = ws.take(z.l0); float* l1 = ws.take(z.l1);
    float* wv = ws.take(z.wv);
    int rc;
#define MM_TRY(x) do { rc = (x); if (rc != MM_OK) return rc; } while (0)
    // ---- temporal stream: PhaseNet (mimamo_net.py:79-92)
    const float* x0 = phase_0;
    const int pc = h->pc, pcp = h->pcp, catc = 64 + pcp;      // 24 / 24 / 88 for the published num_phase = 12
    if (phase_nhwc && pcp != pc) return MM_ERR_UNSUPPORTED;   // odd num_phase: channels-last inputs would need 16-byte channel groups; NCHW only
    if (!phase_nhwc) {
        // (channels [pc, pcp) of both buffers are zero-filled; their weights are zero columns)
        MM_TRY(nchw_to_nhwc(phase_0, p0n, N64, pc, 48 * 48, pcp, 0, pcp, s));
        MM_TRY(nchw_to_nhwc(phase_1, cat, N64, pc, 24 * 24, catc, 64, pcp, s));  // torch.cat([conv1, level1], dim=1) (:85)
        x0 = p0n;
    } else if (phase_nhwc == 2) {
        // phase_1 IS the concat buffer [N,24,24,64+pc] with the level-1 channels already at 64.. (written
        // there by mm_phase_diff_frames); conv[1] fills channels 0..63 in place.
        cat = const_cast<float*>(phase_1);
    } else {
        // phase_1 given as [N,24,24,pc] NHWC: place it behind the 64 conv channels
        MM_HIP(hipMemcpy2DAsync(cat + 64, catc * sizeof(float), phase_1, pc * sizeof(float), pc * sizeof(float),
                                (size_t)N64 * 24 * 24, hipMemcpyDeviceToDevice, s));
    }
    MM_TRY(run_layer(h->conv[0], x0, N, 48, 48, pcp, 0, a0, 64, 0, nullptr, 0, s, nullptr, nullptr, 0, h->no_sched));
    MM_TRY(run_layer(h->conv[1], a0, N, 48, 48, 64, 0, cat, catc, 0, nullptr, 0, s, nullptr, nullptr, 0, h->no_sched));
    // the fused Winograd kernel declines (MM_ERR_UNSUPPORTED) what its 32-bit plane offsets cannot address: direct form then
    rc = MM_ERR_UNSUPPORTED;
    if (h->winograd && h->conv[2].wino_u4) rc = run_layer_wino(h->conv[2], cat, N, 24, 24, a1, wv, nullptr, 5, s);   // K = 88 padded to 128
    if (rc == MM_ERR_UNSUPPORTED) rc = run_layer(h->conv[2], cat, N, 24, 24, catc, 0, a1, 128, 0, nullptr, 0, s, nullptr, nullptr, 0, h->no_sched);
    if (rc != MM_OK) return rc;
    MM_TRY(run_layer(h->conv[3], a1, N, 24, 24, 128, 0, a2, 128, 0, nullptr, 0, s, nullptr, nullptr, 0, h->no_sched));
    rc = MM_ERR_UNSUPPORTED;
    if (h->winograd && h->conv[4].wino_u4) rc = run_layer_wino(h->conv[4], a2, N, 12, 12, a3, wv, nullptr, 5, s);   // 3x3 tiles per map
    if (rc == MM_ERR_UNSUPPORTED) rc = run_layer(h->conv[4], a2, N, 12, 12, 128, 0, a3, 256, 0, nullptr, 0, s, nullptr, nullptr, 0, h->no_sched);
    if (rc != MM_OK) return rc;
    MM_TRY(run_layer(h->conv[5], a3, N, 12, 12, 256, 0, a4, 256, 0, nullptr, 0, s, nullptr, nullptr, 0, h->no_sched));
    MM_TRY(avgpool_hw(a4, pool, N64, 36, 256, 256, 0, 0, s));
    MM_TRY(run_layer(h->fc1, pool, N, 1, 1, 256, 0, fc1, 256, 0, nullptr, 0, s, nullptr, nullptr, 0, h->no_sched));
    MM_TRY(run_layer(h->fc2, fc1, N, 1, 1, 256, 0, feat, 512, 256, nullptr, 0, s, nullptr, nullptr, 0, h->no_sched));  // cat([spatial, temporal]) (:136)
    // ---- spatial stream: MLP (:22-26)
    {
        const float* xin = rgb;
        int cin = h->feat_dim;
        for (size_t i = 0; i < h->mlp.size(); ++i) {
            const bool last = i + 1 == h->mlp.size();
            float* dst = last ? feat : m1 + (i & 1) * N64 * h->mlp_max;     // the last layer writes the spatial half of `feat`
            MM_TRY(run_layer(h->mlp[i], xin, N, 1, 1, cin, 0, dst, last ? 512 : h->mlp[i].cout, 0, nullptr, 0, s, nullptr, nullptr, 0, h->no_sched));
            xin = dst;
            cin = h->mlp[i].cout;
        }
    }
    MM_TRY(run_layer(h->transform, feat, N, 1, 1, 512, 0, f, 256, 0, nullptr, 0, s, nullptr, nullptr, 0, h->no_sched));
    // ---- nn.GRU(256,128,bidirectional,num_layers=2) WITHOUT batch_first: the [bs,T,256] tensor is read as
    //      (seq_len = bs, batch = T)  (:119,139 -- quirk Q1, trained in, reproduced on purpose)
    const int Ti = (int)T, S = (int)bs;
    const float* x = f;
    float* lay[2] = {l0, l1};
    for (int l = 0; l < 2; ++l) {
        MM_TRY(run_layer(h->gru_ih[l], x, N, 1, 1, 256, 0, gi, 768, 0, nullptr, 0, s, nullptr, nullptr, 0, h->no_sched));
        for (int d = 0; d < 2; ++d) {
            for (int step = 0; step < S; ++step) {
                const int t = d == 0 ? step : S - 1 - step;
                float* hout = lay[l] + (int64_t)t * Ti * 256;
                const float* git = gi + (int64_t)t * Ti * 768;
                if (step == 0) {
                    MM_TRY(gru_gates(git, 768, d * 384, nullptr, h->bhh[l][d], nullptr, 0, 0, hout, 256, d * 128, Ti, 128, s));
                } else {
                    const int tp = d == 0 ? t - 1 : t + 1;
                    const float* hprev = lay[l] + (int64_t)tp * Ti * 256;
                    MM_TRY(run_layer(h->gru_hh[l][d], hprev, Ti, 1, 1, 256, d * 128, gh, 384, 0, nullptr, 0, s, nullptr, nullptr, 0, h->no_sched));
                    MM_TRY(gru_gates(git, 768, d * 384, gh, nullptr, hprev, 256, d * 128, hout, 256, d * 128, Ti, 128, s));
                }
            }
        }
        x = lay[l];
    }
    MM_TRY(run_layer(h->classifier, l1, N, 1, 1, 256, 0, out, 2, 0, nullptr, 0, s, nullptr, nullptr, 0, h->no_sched));
#undef MM_TRY
    return MM_OK;
}

}  // extern "C"
