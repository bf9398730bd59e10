// C ABI of libsrmi.so (see include/srmi.h): validation + kernel dispatch.  No allocation, no sync.
#include <stdarg.h>
#include <stdlib.h>

#include "common.h"

namespace srmi {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// SRMI_DBG (ablation build only, -DSRMI_ABLATE): profiling-only ablation switches (results are WRONG with bits 1..16 or 512 set; never set in production).
//   1/2/4/8/16  fused ResBlock: skip staging / conv 1 / epilogue / conv 2 / T write (phase timing; 1/2/4 also in the
//               Cin=64 kernel: staging / MFMA loop / epilogue)
//   32          disable the XCD-aware block -> tile remap        128  4-wave fused ResBlock
//   256         4-wave Cin=64 / K-chunked kernels                512  fused ResBlock without the residual read
//   bits 23-24  thin-Cout kernel: cache policy of its row loads (0 = nt, the default; 1 = none; 2 = nt+sc1...), bit 25: nt stores
//   f32 path (results stay correct): 256 sub-pixel dgrad on the exact kernel, 512 no halo-tile split kernel, 4096 no full-map
//               forms, 16384 gathering stride-2 dgrad, bit 24 / 25 the no-LDS forward / old wgrad; bits 26-30 ablate the tiled
//               exact-f32 kernel (results wrong).  SRMI_F32_EXACT=1 (separate variable): no split-bf16 kernels at all.
#ifdef SRMI_ABLATE
int dbg_flags() {
    static const int v = [] { const char* e = getenv("SRMI_DBG"); return e ? atoi(e) : 0; }();
    return v;
}

int exp_flags() {
    static const int v = [] { const char* e = getenv("SRMI_EXP"); return e ? atoi(e) : 0; }();
    return v;
}

int dbg_thin_rows() {
    static const int v = [] { const char* e = getenv("SRMI_THIN_RB"); return e ? atoi(e) : 0; }();
    return v;
}
#endif

int hip_check(hipError_t e, const char* what) {
    if (e == hipSuccess) return SRMI_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return SRMI_EHIP;
}

int layout_convert(const void* src, void* dst, int N, int C, int H, int W, int sl, int sd, int dl, int dd,
                   hipStream_t s);

int pair_loss(int op, const float* a, const float* b, size_t n, float* out, void* ws, hipStream_t s);
int tv_loss(const float* x, int N, int C, int H, int W, float weight, float* out, void* ws, hipStream_t s);
int y_metrics(const float* sr, const float* gt, int nimg, int H, int W, int gt_H, int gt_W, int border, int clip,
              int win, double* out, void* ws, hipStream_t s);

size_t train_workspace_bytes(int C);
int train_max_slots(int C);
int bn_finalize_launch(const void* ws, int nslots, long long M, int C, float eps, float momentum, float* mean, float* rstd,
                       float* rmean, float* rvar, long long* nbt, int updates, hipStream_t s);
int bn_train_stats(const float* z, long long M, int C, float eps, float momentum, float* mean, float* rstd,
                   float* rmean, float* rvar, long long* nbt, int updates, void* ws, hipStream_t s);
int epilogue_fwd(const srmi_epi_desc& d, const float* z, const float* mean, const float* rstd, const float* gamma,
                 const float* beta, const float* alpha, const float* residual, float* y, hipStream_t s);
int epilogue_bwd(const srmi_epi_desc& d, const float* z, const float* dy, const float* mean, const float* rstd,
                 const float* gamma, const float* beta, const float* alpha, float* dz, float* dsum1, float* dsum2,
                 float* dalpha, float* dzero, void* ws, hipStream_t s);
int maxpool_fwd(const float* x, float* y, int N, int H, int W, int C, hipStream_t s);
int maxpool_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, hipStream_t s);
int mse_bwd(const float* a, const float* b, const float* gout, size_t n, float* da, float* db, hipStream_t s);
int bce_bwd(const float* p, const float* t, const float* gout, size_t n, float* dp, hipStream_t s);
int adam_step(float* param, const float* grad, float* m, float* v, size_t n, double lr, double beta1, double beta2,
              double eps, int step, hipStream_t s);
int tv_bwd(const float* x, const float* gout, int N, int C, int H, int W, float weight, float* dx, hipStream_t s);
size_t generic_dgrad_packed_bytes(const srmi_conv_desc& d);
int generic_dgrad_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s);
size_t generic_dgrad_workspace_bytes(const srmi_conv_desc& d, const ConvGeom& g);
size_t wgrad_workspace_bytes(const srmi_conv_desc& d, const ConvGeom& g);
int generic_dgrad_launch(const srmi_conv_desc& d, const ConvGeom& g, const float* dz, const float* w_dgrad, float* dx,
                         hipStream_t s, void* ws, size_t ws_bytes);
int wgrad_launch(const srmi_conv_desc& d, const ConvGeom& g, const float* x, const float* dz, float* dw,
                 hipStream_t s, void* ws, size_t ws_bytes);

int u8_to_f32(const unsigned char* src, float* dst, int N, int H, int W, int C, hipStream_t s);
int f32_to_u8(const float* src, unsigned char* dst, int N, int H, int W, int C, int clip, hipStream_t s);
int pool_up_launch(int op, const void* x, void* y, int N, int H, int W, int C, int cs, int co, int dtype, hipStream_t s);
int crop_augment(const unsigned char* lr_pack, const unsigned char* gt_pack, const srmi_crop_item* items, int B, int patch,
                 int scale, int C, int f32_arith, float* lr_out, float* gt_out, hipStream_t s);

static bool is_dtype(int v) { return v == SRMI_F32 || v == SRMI_BF16; }
static bool is_layout(int v) { return v == SRMI_NHWC || v == SRMI_NCHW; }

int conv_geom(const srmi_conv_desc& d, ConvGeom* g) {
    if (d.N < 0 || d.H < 0 || d.W < 0 || d.Cin <= 0 || d.Cout <= 0 || d.kh <= 0 || d.kw <= 0 || d.stride <= 0 ||
        d.pad < 0) {
        set_error("conv desc: bad dims N=%d H=%d W=%d Cin=%d Cout=%d k=%dx%d stride=%d pad=%d", d.N, d.H, d.W, d.Cin,
                  d.Cout, d.kh, d.kw, d.stride, d.pad);
        return SRMI_EINVAL;
    }
    if (!is_dtype(d.in_dtype) || !is_dtype(d.out_dtype) || !is_dtype(d.res_dtype) || !is_layout(d.in_layout) ||
        !is_layout(d.out_layout)) {
        set_error("conv desc: bad dtype/layout code");
        return SRMI_EINVAL;
    }
    if (d.compute != SRMI_COMPUTE_F32 && d.compute != SRMI_COMPUTE_BF16) {
        set_error("conv desc: bad compute code %d", d.compute);
        return SRMI_EINVAL;
    }
    if (d.act < SRMI_ACT_NONE || d.act > SRMI_ACT_ADD_RELU) {
        set_error("conv desc: bad activation code %d", d.act);
        return SRMI_EINVAL;
    }
    if (d.shuffle < 1 || d.Cout % (d.shuffle * d.shuffle) != 0) {
        set_error("conv desc: Cout=%d not divisible by shuffle^2 (r=%d)", d.Cout, d.shuffle);
        return SRMI_EINVAL;
    }
    if (d.shuffle_order != SRMI_SHUFFLE_CRD && d.shuffle_order != SRMI_SHUFFLE_DCR) {
        set_error("conv desc: bad shuffle order %d", d.shuffle_order);
        return SRMI_EINVAL;
    }
    const int He = d.H + 2 * d.pad - d.kh, We = d.W + 2 * d.pad - d.kw;
    g->Ho = (d.N == 0 || d.H == 0 || d.W == 0 || He < 0) ? 0 : He / d.stride + 1;
    g->Wo = (d.N == 0 || d.H == 0 || d.W == 0 || We < 0) ? 0 : We / d.stride + 1;
    g->Cf = d.Cout / (d.shuffle * d.shuffle);
    g->Hf = g->Ho * d.shuffle;
    g->Wf = g->Wo * d.shuffle;
    g->in_cs = d.in_cstride ? d.in_cstride : d.Cin;
    g->out_cs = d.out_cstride ? d.out_cstride : g->Cf;
    g->res_cs = d.res_cstride ? d.res_cstride : g->Cf;
    if (g->in_cs < d.Cin || d.out_coffset < 0 || g->out_cs < g->Cf + d.out_coffset || g->res_cs < g->Cf) {
        set_error("conv desc: channel slice out of range (in_cs=%d Cin=%d out_cs=%d Cf=%d off=%d)", g->in_cs, d.Cin,
                  g->out_cs, g->Cf, d.out_coffset);
        return SRMI_EINVAL;
    }
    if ((d.in_layout == SRMI_NCHW && d.in_cstride && d.in_cstride != d.Cin) ||
        (d.out_layout == SRMI_NCHW && (d.out_coffset || (d.out_cstride && d.out_cstride != g->Cf) ||
                                       (d.res_cstride && d.res_cstride != g->Cf)))) {
        set_error("conv desc: channel slices need NHWC");
        return SRMI_EINVAL;
    }
    return SRMI_OK;
}

int select_kernel(const srmi_conv_desc& d) {
    if (c64_supported(d)) return K_C64_BF16;
    if (cin_supported(d)) return K_CIN_BF16;
    if (thin_supported(d)) return K_THIN_BF16;
    if (k9c1_supported(d)) return K_K9C1_BF16;
    if (k9c3_supported(d)) return K_K9C3_BF16;
    if (pw_supported(d)) return K_PW_BF16;
    if (c1_supported(d)) return K_C1OUT_BF16;
    return K_GENERIC_F32;
}

const char* kernel_label(int k) {
    switch (k) {
        case K_C64_BF16: return "c64_bf16";
        case K_THIN_BF16: return "c64_thin_bf16";
        case K_K9C3_BF16: return "k9c3_bf16";
        case K_CIN_BF16: return "cin_bf16";
        case K_PW_BF16: return "pw_bf16";
        case K_K9C1_BF16: return "k9c1_bf16";
        case K_C1OUT_BF16: return "c1out_bf16";
        default: return "generic_f32";
    }
}

static int conv_fwd_one(const srmi_conv_args* a, hipStream_t s, void* ws = nullptr, size_t ws_bytes = 0) {
    if (!a) { set_error("conv: NULL args"); return SRMI_EINVAL; }
    ConvGeom g;
    int rc = conv_geom(a->d, &g);
    if (rc) return rc;
    if ((long long)a->d.N * g.Ho * g.Wo == 0) return SRMI_OK;  // empty batch / image: nothing to do
    if (!a->x || !a->w || !a->y) {
        set_error("conv: NULL x/w/y");
        return SRMI_EINVAL;
    }
    if ((!a->scale || !a->shift) && select_kernel(a->d) != K_GENERIC_F32) {
        set_error("conv: NULL scale/shift (= 1 / 0) is only accepted by the generic f32 kernel");
        return SRMI_EINVAL;
    }
    if (a->d.act == SRMI_ACT_PRELU && !a->alpha) {
        set_error("conv: PReLU needs alpha");
        return SRMI_EINVAL;
    }
    if (a->x_copy && select_kernel(a->d) != K_C64_BF16) {
        set_error("conv: x_copy is served by the c64_bf16 kernel only (this descriptor runs %s)", kernel_label(select_kernel(a->d)));
        return SRMI_EUNSUPPORTED;
    }
    switch (select_kernel(a->d)) {
        case K_C64_BF16: return c64_launch(*a, g, s);
        case K_THIN_BF16: return thin_launch(*a, g, s);   // rejects a residual (SRMI_EUNSUPPORTED)
        case K_K9C3_BF16: return k9c3_launch(*a, g, s);
        case K_CIN_BF16: return cin_launch(*a, g, s);
        case K_PW_BF16: return pw_launch(*a, g, s);
        case K_K9C1_BF16: return k9c1_launch(*a, g, s);
        case K_C1OUT_BF16: return c1_launch(*a, g, s);
        default: return generic_launch(*a, g, s, ws, ws_bytes);
    }
}

}  // namespace srmi

using namespace srmi;

extern "C" {

int srmi_version(void) { return SRMI_VERSION; }

const char* srmi_last_error(void) { return g_err; }

int srmi_conv2d_kernel_name(const srmi_conv_desc* d, char* buf, size_t buf_len) {
    if (!d || !buf || buf_len == 0) { set_error("kernel_name: NULL argument"); return SRMI_EINVAL; }
    ConvGeom g;
    int rc = conv_geom(*d, &g);
    if (rc) return rc;
    const int k = select_kernel(*d);
    snprintf(buf, buf_len, "%s", k == K_GENERIC_F32 ? generic_kernel_label(*d, g) : kernel_label(k));
    return SRMI_OK;
}

int srmi_conv2d_packed_weight_bytes(const srmi_conv_desc* d, size_t* bytes) {
    if (!d || !bytes) { set_error("packed_weight_bytes: NULL argument"); return SRMI_EINVAL; }
    ConvGeom g;
    int rc = conv_geom(*d, &g);
    if (rc) return rc;
    switch (select_kernel(*d)) {
        case K_C64_BF16: *bytes = c64_packed_bytes(*d); break;
        case K_THIN_BF16: *bytes = thin_packed_bytes(*d); break;
        case K_K9C3_BF16: *bytes = k9c3_packed_bytes(*d); break;
        case K_CIN_BF16: *bytes = cin_packed_bytes(*d); break;
        case K_PW_BF16: *bytes = pw_packed_bytes(*d); break;
        case K_K9C1_BF16: *bytes = k9c1_packed_bytes(*d); break;
        case K_C1OUT_BF16: *bytes = c1_packed_bytes(*d); break;
        default: *bytes = generic_packed_bytes(*d);
    }
    return SRMI_OK;
}

int srmi_conv2d_pack_weights(const srmi_conv_desc* d, const float* w_oihw, void* w_packed, void* stream) {
    if (!d || !w_oihw || !w_packed) { set_error("pack_weights: NULL argument"); return SRMI_EINVAL; }
    ConvGeom g;
    int rc = conv_geom(*d, &g);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    switch (select_kernel(*d)) {
        case K_C64_BF16: return c64_pack(*d, w_oihw, w_packed, s);
        case K_THIN_BF16: return thin_pack(*d, w_oihw, w_packed, s);
        case K_K9C3_BF16: return k9c3_pack(*d, w_oihw, w_packed, s);
        case K_CIN_BF16: return cin_pack(*d, w_oihw, w_packed, s);
        case K_PW_BF16: return pw_pack(*d, w_oihw, w_packed, s);
        case K_K9C1_BF16: return k9c1_pack(*d, w_oihw, w_packed, s);
        case K_C1OUT_BF16: return c1_pack(*d, w_oihw, w_packed, s);
        default: return generic_pack(*d, w_oihw, w_packed, s);
    }
}

int srmi_conv2d_fwd(const srmi_conv_args* a, void* stream) { return conv_fwd_one(a, (hipStream_t)stream); }

static int srcnn_layer(const srmi_conv_args* a, void* stream, int kernel, const char* what) {
    if (!a) { set_error("%s: NULL args", what); return SRMI_EINVAL; }
    ConvGeom g;
    int rc = conv_geom(a->d, &g);
    if (rc) return rc;
    if (select_kernel(a->d) != kernel) {
        set_error("%s: not this kernel's shape (the dispatcher picks %s)", what, kernel_label(select_kernel(a->d)));
        return SRMI_EUNSUPPORTED;
    }
    return conv_fwd_one(a, (hipStream_t)stream);
}

int srmi_srcnn_layer_c1in(const srmi_conv_args* a, void* stream) { return srcnn_layer(a, stream, K_K9C1_BF16, "srcnn_layer_c1in"); }
int srmi_srcnn_layer_c1out(const srmi_conv_args* a, void* stream) { return srcnn_layer(a, stream, K_C1OUT_BF16, "srcnn_layer_c1out"); }

int srmi_conv2d_workspace_bytes(const srmi_conv_desc* d, size_t* bytes) {
    if (!d || !bytes) { set_error("workspace_bytes: NULL argument"); return SRMI_EINVAL; }
    ConvGeom g;
    int rc = conv_geom(*d, &g);
    if (rc) return rc;
    *bytes = select_kernel(*d) == K_GENERIC_F32 ? generic_workspace_bytes(*d, g) : 0;
    return SRMI_OK;
}

int srmi_conv2d_fwd_ws(const srmi_conv_args* a, void* ws, size_t ws_bytes, void* stream) {
    if (ws && ((uintptr_t)ws & 15)) { set_error("conv2d_fwd_ws: workspace must be 16-byte aligned"); return SRMI_EINVAL; }
    return conv_fwd_one(a, (hipStream_t)stream, ws, ws_bytes);
}

// 0: not fusable; 1: the ResBlock pair (conv_resblock_bf16.hip); 2: up conv + PixelShuffle + last conv (conv_upfuse_bf16.hip);
// 3: 9x9 conv of a one-channel image + 1x1 conv (conv_k9c1_bf16.hip)
static int fuse_kind(const srmi_conv_args* a1, const srmi_conv_args* a2) {
    ConvGeom g;
    if (conv_geom(a1->d, &g) || conv_geom(a2->d, &g)) return 0;
    if (a1->x_copy || a2->x_copy) return 0;          // the fused kernels do not write input copies: two launches
    const bool rb = resblock_supported(a1->d, a2->d);
    const bool uf = !rb && upfuse_supported(a1->d, a2->d);
    const bool kp = !rb && !uf && k9pw_supported(a1->d, a2->d);
    if (!rb && !uf && !kp) return 0;
    if (a1->residual || !a1->x || !a1->w || !a2->w || !a2->y || !a1->scale || !a1->shift || !a2->scale || !a2->shift)
        return 0;
    if ((a1->d.act == SRMI_ACT_PRELU && !a1->alpha) || (a2->d.act == SRMI_ACT_PRELU && !a2->alpha)) return 0;
    if (a2->x && a2->x != a1->y) return 0;
    if (a2->y == a1->x) return 0;
    if ((uf || kp) && a2->residual) return 0;
    return rb ? 1 : uf ? 2 : 3;
}

static bool can_fuse_pair(const srmi_conv_args* a1, const srmi_conv_args* a2) { return fuse_kind(a1, a2) == 1; }

int srmi_conv2d_can_fuse(const srmi_conv_args* conv1, const srmi_conv_args* conv2) {
    return (conv1 && conv2) ? fuse_kind(conv1, conv2) : 0;
}

int srmi_conv2d_fwd_pair(const srmi_conv_args* conv1, const srmi_conv_args* conv2, void* stream) {
    if (!conv1 || !conv2) { set_error("fwd_pair: NULL args"); return SRMI_EINVAL; }
    ConvGeom g;
    int rc = conv_geom(conv1->d, &g);
    if (rc) return rc;
    rc = conv_geom(conv2->d, &g);
    if (rc) return rc;
    switch (fuse_kind(conv1, conv2)) {
    case 1: return resblock_launch(*conv1, *conv2, (hipStream_t)stream);
    case 2: return upfuse_launch(*conv1, *conv2, (hipStream_t)stream);
    case 3: return k9pw_launch(*conv1, *conv2, (hipStream_t)stream);
    }
    set_error("fwd_pair: not a fusable pair (see srmi_conv2d_can_fuse)");
    return SRMI_EUNSUPPORTED;
}

int srmi_resblock_fwd(const srmi_conv_args* conv1, const srmi_conv_args* conv2, void* stream) {
    if (!conv1 || !conv2) { set_error("resblock: NULL args"); return SRMI_EINVAL; }
    ConvGeom g;
    int rc = conv_geom(conv1->d, &g);
    if (rc) return rc;
    rc = conv_geom(conv2->d, &g);
    if (rc) return rc;
    if (!can_fuse_pair(conv1, conv2)) {
        set_error("resblock: not a fusable pair (need 3x3 64->64->64 NHWC bf16, SRMI_COMPUTE_BF16, y != x)");
        return SRMI_EUNSUPPORTED;
    }
    return resblock_launch(*conv1, *conv2, (hipStream_t)stream);
}

int srmi_chain_status(void) { return chain_status(); }

int srmi_conv2d_chain_length(const srmi_conv_args* a, int32_t n) {
    if (!a || n < 4) return 0;
    if (!(a[0].flags & SRMI_CONV_FUSE_CHAIN)) return 0;
    return chain_run_length(a, n, can_fuse_pair);
}

int srmi_conv2d_fwd_seq(const srmi_conv_args* a, int32_t n, void* stream) {
    if (n < 0 || (n > 0 && !a)) { set_error("fwd_seq: bad arguments"); return SRMI_EINVAL; }
    for (int32_t i = 0; i < n; ++i) {
        if (a[i].flags & SRMI_CONV_FUSE_CHAIN) {         // the first pair of a chain, or the head link in front of one
            int npairs = 0, kinds[2] = {0, 0}, head = 0;
            const int nc = chain_run_length(a + i, n - i, can_fuse_pair, &npairs, kinds, &head);
            if (nc >= 4) {
                int rc = chain_launch(a + i, head, npairs, kinds, (hipStream_t)stream);
                if (rc) return rc;
                i += nc - 1;
                continue;
            }
        }
        if ((a[i].flags & SRMI_CONV_FUSE_NEXT) && i + 1 < n && can_fuse_pair(a + i, a + i + 1)) {
            int rc = resblock_launch(a[i], a[i + 1], (hipStream_t)stream);
            if (rc) return rc;
            ++i;
            continue;
        }
        if ((a[i].flags & SRMI_CONV_FUSE_NEXT) && i + 1 < n) {
            const int kind = fuse_kind(a + i, a + i + 1);
            if (kind == 2 || kind == 3) {
                int rc = kind == 2 ? upfuse_launch(a[i], a[i + 1], (hipStream_t)stream) : k9pw_launch(a[i], a[i + 1], (hipStream_t)stream);
                if (rc) return rc;
                ++i;
                continue;
            }
        }
        int rc = conv_fwd_one(a + i, (hipStream_t)stream);
        if (rc) return rc;
    }
    return SRMI_OK;
}

int srmi_maxpool2x2_nhwc(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t in_cstride,
                         int32_t dtype, void* stream) {
    if (N < 0 || H < 0 || W < 0 || C <= 0 || in_cstride < 0 || (in_cstride && in_cstride < C) || !is_dtype(dtype)) {
        set_error("maxpool2x2_nhwc: bad arguments");
        return SRMI_EINVAL;
    }
    if ((size_t)N * (H / 2) * (W / 2) == 0) return SRMI_OK;
    if (!x || !y) { set_error("maxpool2x2_nhwc: NULL buffer"); return SRMI_EINVAL; }
    return pool_up_launch(0, x, y, N, H, W, C, in_cstride ? in_cstride : C, 0, dtype, (hipStream_t)stream);
}

int srmi_upsample2x_nhwc(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t out_cstride,
                         int32_t out_coffset, int32_t dtype, void* stream) {
    const int cs = out_cstride ? out_cstride : C;
    if (N < 0 || H < 0 || W < 0 || C <= 0 || out_cstride < 0 || out_coffset < 0 || out_coffset + C > cs || !is_dtype(dtype)) {
        set_error("upsample2x_nhwc: bad arguments");
        return SRMI_EINVAL;
    }
    if ((size_t)N * H * W == 0) return SRMI_OK;
    if (!x || !y) { set_error("upsample2x_nhwc: NULL buffer"); return SRMI_EINVAL; }
    return pool_up_launch(1, x, y, N, H, W, C, cs, out_coffset, dtype, (hipStream_t)stream);
}

int srmi_crop_augment_u8(const uint8_t* lr_pack, const uint8_t* gt_pack, const srmi_crop_item* items, int32_t B,
                         int32_t patch, int32_t scale, int32_t C, int32_t f32_arith, float* lr_out, float* gt_out,
                         void* stream) {
    if (B < 0 || patch < 0 || scale < 1 || C < 1) { set_error("crop_augment: bad arguments"); return SRMI_EINVAL; }
    if ((size_t)B * patch == 0) return SRMI_OK;
    if (!lr_pack || !gt_pack || !items || !lr_out || !gt_out) { set_error("crop_augment: NULL buffer"); return SRMI_EINVAL; }
    if ((long long)patch * scale > 0x7fff) { set_error("crop_augment: patch too large"); return SRMI_EINVAL; }
    return crop_augment(lr_pack, gt_pack, items, B, patch, scale, C, f32_arith, lr_out, gt_out, (hipStream_t)stream);
}

int srmi_layout_convert(const void* src, void* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t src_layout,
                        int32_t src_dtype, int32_t dst_layout, int32_t dst_dtype, void* stream) {
    if (N < 0 || C < 0 || H < 0 || W < 0 || !is_layout(src_layout) || !is_layout(dst_layout) || !is_dtype(src_dtype) ||
        !is_dtype(dst_dtype)) {
        set_error("layout_convert: bad arguments");
        return SRMI_EINVAL;
    }
    if ((size_t)N * C * H * W == 0) return SRMI_OK;
    if (!src || !dst) { set_error("layout_convert: NULL buffer"); return SRMI_EINVAL; }
    return layout_convert(src, dst, N, C, H, W, src_layout, src_dtype, dst_layout, dst_dtype, (hipStream_t)stream);
}

int srmi_u8hwc_to_f32nchw(const uint8_t* src, float* dst, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    if (N < 0 || H < 0 || W < 0 || C <= 0) { set_error("u8hwc_to_f32nchw: bad dims"); return SRMI_EINVAL; }
    if ((size_t)N * H * W == 0) return SRMI_OK;
    if (!src || !dst) { set_error("u8hwc_to_f32nchw: NULL buffer"); return SRMI_EINVAL; }
    return u8_to_f32(src, dst, N, H, W, C, (hipStream_t)stream);
}

int srmi_f32nchw_to_u8hwc(const float* src, uint8_t* dst, int32_t N, int32_t H, int32_t W, int32_t C, int32_t clip,
                          void* stream) {
    if (N < 0 || H < 0 || W < 0 || C <= 0) { set_error("f32nchw_to_u8hwc: bad dims"); return SRMI_EINVAL; }
    if ((size_t)N * H * W == 0) return SRMI_OK;
    if (!src || !dst) { set_error("f32nchw_to_u8hwc: NULL buffer"); return SRMI_EINVAL; }
    return f32_to_u8(src, dst, N, H, W, C, clip, (hipStream_t)stream);
}

static int check_reduce(const char* what, const void* a, const void* b, const void* out, const void* ws) {
    if (!a || !b || !out || !ws) { set_error("%s: NULL argument", what); return SRMI_EINVAL; }
    if ((uintptr_t)ws % 8) { set_error("%s: workspace must be 8-byte aligned", what); return SRMI_EINVAL; }
    return SRMI_OK;
}

int srmi_mse_loss(const float* a, const float* b, size_t n, float* out, void* ws, void* stream) {
    int rc = check_reduce("mse_loss", a, b, out, ws);
    if (rc) return rc;
    return pair_loss(0, a, b, n, out, ws, (hipStream_t)stream);  // n == 0: 0/0 = NaN, as torch
}

int srmi_bce_loss(const float* p, const float* t, size_t n, float* out, void* ws, void* stream) {
    int rc = check_reduce("bce_loss", p, t, out, ws);
    if (rc) return rc;
    return pair_loss(1, p, t, n, out, ws, (hipStream_t)stream);
}

int srmi_tv_loss(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, float weight, float* out, void* ws,
                 void* stream) {
    int rc = check_reduce("tv_loss", x, x, out, ws);
    if (rc) return rc;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) { set_error("tv_loss: bad dims %dx%dx%dx%d", N, C, H, W); return SRMI_EINVAL; }
    return tv_loss(x, N, C, H, W, weight, out, ws, (hipStream_t)stream);
}

static int y_metrics_check(const char* what, const float* sr, const float* gt, int32_t H, int32_t W, int32_t gt_H,
                           int32_t gt_W, int32_t border, int32_t win, double* out, void* ws) {
    int rc = check_reduce(what, sr, gt, out, ws);
    if (rc) return rc;
    if ((uintptr_t)out % 8) { set_error("%s: out must be 8-byte aligned", what); return SRMI_EINVAL; }
    if (H <= 0 || W <= 0 || gt_H < H || gt_W < W || border < 0 || H - 2 * border <= 0 || W - 2 * border <= 0) {
        set_error("%s: bad geometry H=%d W=%d gt=%dx%d border=%d", what, H, W, gt_H, gt_W, border);
        return SRMI_EINVAL;
    }
    const int side = (H < W ? H : W) - 2 * border;
    if (win != 0 && (win < 3 || win > 7 || win % 2 == 0 || win > side)) {
        set_error("%s: window %d must be odd, 3..7 and <= the cropped side %d", what, win, side);
        return SRMI_EINVAL;
    }
    return SRMI_OK;
}

int srmi_y_psnr_ssim(const float* sr, const float* gt, int32_t H, int32_t W, int32_t gt_H, int32_t gt_W,
                     int32_t border, int32_t clip, int32_t win, double* out, void* ws, void* stream) {
    int rc = y_metrics_check("y_psnr_ssim", sr, gt, H, W, gt_H, gt_W, border, win, out, ws);
    if (rc) return rc;
    return y_metrics(sr, gt, 1, H, W, gt_H, gt_W, border, clip, win, out, ws, (hipStream_t)stream);
}

int srmi_y_psnr_ssim_batch(const float* sr, const float* gt, int32_t n_images, int32_t H, int32_t W, int32_t gt_H,
                           int32_t gt_W, int32_t border, int32_t clip, int32_t win, double* out, void* ws,
                           void* stream) {
    if (n_images == 0) return SRMI_OK;
    if (n_images < 0 || n_images > SRMI_METRICS_MAX_BATCH) {
        set_error("y_psnr_ssim_batch: n_images = %d, must be 0..%d (split larger batches)", n_images,
                  SRMI_METRICS_MAX_BATCH);
        return SRMI_EINVAL;
    }
    int rc = y_metrics_check("y_psnr_ssim_batch", sr, gt, H, W, gt_H, gt_W, border, win, out, ws);
    if (rc) return rc;
    return y_metrics(sr, gt, n_images, H, W, gt_H, gt_W, border, clip, win, out, ws, (hipStream_t)stream);
}

static int epi_check(const char* what, const srmi_epi_desc* d) {
    if (!d) { set_error("%s: NULL descriptor", what); return SRMI_EINVAL; }
    if (d->N < 0 || d->H < 0 || d->W < 0 || d->C <= 0 || d->C > (1 << 22) || d->shuffle < 1 || d->C % (d->shuffle * d->shuffle) != 0 ||
        d->act < SRMI_ACT_NONE || d->act > SRMI_ACT_SIGMOID ||
        (d->shuffle_order != SRMI_SHUFFLE_CRD && d->shuffle_order != SRMI_SHUFFLE_DCR)) {
        set_error("%s: bad descriptor N=%d H=%d W=%d C=%d r=%d act=%d", what, d->N, d->H, d->W, d->C, d->shuffle, d->act);
        return SRMI_EINVAL;
    }
    if ((long long)d->N * d->H * d->W > 0x7fffffffLL) { set_error("%s: N*H*W exceeds int32", what); return SRMI_EINVAL; }
    return SRMI_OK;
}

int srmi_train_workspace_bytes(int32_t C, size_t* bytes) {
    if (C <= 0 || !bytes) { set_error("train_workspace_bytes: bad arguments"); return SRMI_EINVAL; }
    *bytes = train_workspace_bytes(C);
    return SRMI_OK;
}

int srmi_bn_train_stats_n(const float* z, int64_t M, int32_t C, float eps, float momentum, float* mean, float* rstd,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, int32_t updates, void* ws,
                          void* stream) {
    if (!z || !mean || !rstd || !ws || M <= 0 || C <= 0 || (uintptr_t)ws % 8 || updates < 0 || updates > 64) {
        set_error("bn_train_stats: bad arguments (M=%lld C=%d updates=%d)", (long long)M, C, updates);
        return SRMI_EINVAL;
    }
    return bn_train_stats(z, M, C, eps, momentum, mean, rstd, running_mean, running_var, (long long*)num_batches_tracked, updates,
                          ws, (hipStream_t)stream);
}

int srmi_conv2d_fwd_bn_stats(const srmi_conv_args* a, void* conv_ws, size_t conv_ws_bytes, float eps, float momentum,
                             float* mean, float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                             int32_t updates, void* stat_ws, void* stream) {
    if (!a || !mean || !rstd || !stat_ws || (uintptr_t)stat_ws % 8 || updates < 0 || updates > 64) {
        set_error("conv2d_fwd_bn_stats: bad arguments");
        return SRMI_EINVAL;
    }
    if (conv_ws && ((uintptr_t)conv_ws & 15)) { set_error("conv2d_fwd_bn_stats: workspace must be 16-byte aligned"); return SRMI_EINVAL; }
    const srmi_conv_desc& d = a->d;
    ConvGeom g;
    int rc = conv_geom(d, &g);
    if (rc) return rc;
    const long long M = (long long)d.N * g.Ho * g.Wo;
    if (d.out_layout != SRMI_NHWC || d.out_dtype != SRMI_F32 || d.shuffle != 1 || d.out_coffset != 0 ||
        (d.out_cstride && d.out_cstride != d.Cout) || d.act != SRMI_ACT_NONE || a->residual || a->scale || M <= 0) {
        set_error("conv2d_fwd_bn_stats: the conv must write its dense NHWC f32 output z = conv + shift (no scale, activation, "
                  "residual or sub-pixel store) over a non-empty batch");
        return SRMI_EUNSUPPORTED;
    }
    // the halo-tile split kernel leaves per-workgroup sums of its own output in the statistics slots (one slot per tile)
    const long long tiles = b3h_tiles(d);
    const bool fused = select_kernel(d) == K_GENERIC_F32 && b3h_fwd_ok(d) && tiles <= train_max_slots(d.Cout) && a->x && a->w && a->y;
    if (fused) {
        rc = generic_launch(*a, g, (hipStream_t)stream, conv_ws, conv_ws_bytes, (double*)stat_ws);
        if (rc) return rc;
        return bn_finalize_launch(stat_ws, (int)tiles, M, d.Cout, eps, momentum, mean, rstd, running_mean, running_var,
                                  (long long*)num_batches_tracked, updates, (hipStream_t)stream);
    }
    rc = conv_fwd_one(a, (hipStream_t)stream, conv_ws, conv_ws_bytes);
    if (rc) return rc;
    return bn_train_stats((const float*)a->y, M, d.Cout, eps, momentum, mean, rstd, running_mean, running_var,
                          (long long*)num_batches_tracked, updates, stat_ws, (hipStream_t)stream);
}

int srmi_bn_train_stats(const float* z, int64_t M, int32_t C, float eps, float momentum, float* mean, float* rstd,
                        float* running_mean, float* running_var, int64_t* num_batches_tracked, void* ws, void* stream) {
    return srmi_bn_train_stats_n(z, M, C, eps, momentum, mean, rstd, running_mean, running_var, num_batches_tracked, 1, ws, stream);
}

int srmi_epilogue_fwd(const srmi_epi_desc* d, const float* z, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, const float* alpha, const float* residual, float* y,
                      void* stream) {
    int rc = epi_check("epilogue_fwd", d);
    if (rc) return rc;
    if (!z || !y || (d->bn && (!mean || !rstd || !gamma || !beta)) || (d->act == SRMI_ACT_PRELU && !alpha)) {
        set_error("epilogue_fwd: NULL argument");
        return SRMI_EINVAL;
    }
    return epilogue_fwd(*d, z, mean, rstd, gamma, beta, alpha, residual, y, (hipStream_t)stream);
}

int srmi_epilogue_bwd(const srmi_epi_desc* d, const float* z, const float* dy, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, const float* alpha, float* dz, float* dsum1,
                      float* dsum2, float* dalpha, float* dzero, void* ws, void* stream) {
    int rc = epi_check("epilogue_bwd", d);
    if (rc) return rc;
    if (!z || !dy || !dz || !ws || (uintptr_t)ws % 8 || (d->bn && (!mean || !rstd || !gamma || !beta)) ||
        (d->act == SRMI_ACT_PRELU && !alpha)) {
        set_error("epilogue_bwd: NULL or misaligned argument");
        return SRMI_EINVAL;
    }
    return epilogue_bwd(*d, z, dy, mean, rstd, gamma, beta, alpha, dz, dsum1, dsum2, dalpha, dzero, ws, (hipStream_t)stream);
}

static int grad_geom(const char* what, const srmi_conv_desc* d, ConvGeom* g) {
    if (!d) { set_error("%s: NULL descriptor", what); return SRMI_EINVAL; }
    int rc = conv_geom(*d, g);
    if (rc) return rc;
    if (d->in_layout != SRMI_NHWC || d->out_layout != SRMI_NHWC || d->in_dtype != SRMI_F32 || d->out_dtype != SRMI_F32 ||
        d->shuffle != 1 || d->out_coffset != 0 || (d->out_cstride && d->out_cstride != d->Cout)) {
        set_error("%s: gradients are NHWC f32, dense dz, no sub-pixel store (run the epilogue backward first)", what);
        return SRMI_EUNSUPPORTED;
    }
    return SRMI_OK;
}

int srmi_conv2d_dgrad_packed_weight_bytes(const srmi_conv_desc* d, size_t* bytes) {
    ConvGeom g;
    int rc = grad_geom("dgrad_packed_weight_bytes", d, &g);
    if (rc) return rc;
    if (!bytes) { set_error("dgrad_packed_weight_bytes: NULL argument"); return SRMI_EINVAL; }
    *bytes = generic_dgrad_packed_bytes(*d);
    return SRMI_OK;
}

int srmi_conv2d_dgrad_pack_weights(const srmi_conv_desc* d, const float* w_oihw, void* w_packed, void* stream) {
    ConvGeom g;
    int rc = grad_geom("dgrad_pack_weights", d, &g);
    if (rc) return rc;
    if (!w_oihw || !w_packed) { set_error("dgrad_pack_weights: NULL argument"); return SRMI_EINVAL; }
    return generic_dgrad_pack(*d, w_oihw, w_packed, (hipStream_t)stream);
}

int srmi_conv2d_dgrad_workspace_bytes(const srmi_conv_desc* d, size_t* bytes) {
    ConvGeom g;
    int rc = grad_geom("dgrad_workspace_bytes", d, &g);
    if (rc) return rc;
    if (!bytes) { set_error("dgrad_workspace_bytes: NULL argument"); return SRMI_EINVAL; }
    *bytes = generic_dgrad_workspace_bytes(*d, g);
    return SRMI_OK;
}

int srmi_conv2d_dgrad(const srmi_conv_desc* d, const float* dz, const void* w_packed, float* dx, void* ws,
                      size_t ws_bytes, void* stream) {
    ConvGeom g;
    int rc = grad_geom("dgrad", d, &g);
    if (rc) return rc;
    if (d->in_cstride && d->in_cstride != d->Cin) { set_error("dgrad: dx must be dense"); return SRMI_EUNSUPPORTED; }
    if (ws && ((uintptr_t)ws & 15)) { set_error("dgrad: workspace must be 16-byte aligned"); return SRMI_EINVAL; }
    if ((long long)d->N * d->H * d->W == 0) return SRMI_OK;
    if (!dz || !w_packed || !dx) { set_error("dgrad: NULL argument"); return SRMI_EINVAL; }
    if ((long long)d->N * g.Ho * g.Wo == 0)
        return hip_check(hipMemsetAsync(dx, 0, (size_t)d->N * d->H * d->W * d->Cin * sizeof(float), (hipStream_t)stream),
                         "dgrad memset");
    return generic_dgrad_launch(*d, g, dz, (const float*)w_packed, dx, (hipStream_t)stream, ws, ws ? ws_bytes : 0);
}

int srmi_conv2d_wgrad_workspace_bytes(const srmi_conv_desc* d, size_t* bytes) {
    ConvGeom g;
    int rc = grad_geom("wgrad_workspace_bytes", d, &g);
    if (rc) return rc;
    if (!bytes) { set_error("wgrad_workspace_bytes: NULL argument"); return SRMI_EINVAL; }
    *bytes = wgrad_workspace_bytes(*d, g);
    return SRMI_OK;
}

int srmi_conv2d_wgrad(const srmi_conv_desc* d, const float* x, const float* dz, float* dw, void* ws, size_t ws_bytes,
                      void* stream) {
    ConvGeom g;
    int rc = grad_geom("wgrad", d, &g);
    if (rc) return rc;
    if (!dw) { set_error("wgrad: NULL dw"); return SRMI_EINVAL; }
    if (ws && ((uintptr_t)ws & 15)) { set_error("wgrad: workspace must be 16-byte aligned"); return SRMI_EINVAL; }
    if ((long long)d->N * g.Ho * g.Wo == 0)     // no pixels: the gradient is zero (every other case overwrites all of dw)
        return hip_check(hipMemsetAsync(dw, 0, (size_t)d->Cout * d->Cin * d->kh * d->kw * sizeof(float),
                                        (hipStream_t)stream), "wgrad memset");
    if (!x || !dz) { set_error("wgrad: NULL argument"); return SRMI_EINVAL; }
    return wgrad_launch(*d, g, x, dz, dw, (hipStream_t)stream, ws, ws ? ws_bytes : 0);
}

int srmi_maxpool2x2_fwd(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    if (N < 0 || H < 0 || W < 0 || C <= 0) { set_error("maxpool_fwd: bad dims"); return SRMI_EINVAL; }
    if ((size_t)N * (H / 2) * (W / 2) == 0) return SRMI_OK;
    if (!x || !y) { set_error("maxpool_fwd: NULL argument"); return SRMI_EINVAL; }
    return maxpool_fwd(x, y, N, H, W, C, (hipStream_t)stream);
}

int srmi_maxpool2x2_bwd(const float* x, const float* dy, float* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                        void* stream) {
    if (N < 0 || H < 0 || W < 0 || C <= 0) { set_error("maxpool_bwd: bad dims"); return SRMI_EINVAL; }
    if ((size_t)N * H * W == 0) return SRMI_OK;
    if (!x || !dx || (!dy && (H / 2) * (W / 2) > 0)) { set_error("maxpool_bwd: NULL argument"); return SRMI_EINVAL; }
    return maxpool_bwd(x, dy, dx, N, H, W, C, (hipStream_t)stream);
}

int srmi_mse_loss_bwd(const float* a, const float* b, const float* gout, size_t n, float* da, float* db,
                      void* stream) {
    if (!a || !b || !gout || (!da && !db)) { set_error("mse_loss_bwd: NULL argument"); return SRMI_EINVAL; }
    return mse_bwd(a, b, gout, n, da, db, (hipStream_t)stream);
}

int srmi_bce_loss_bwd(const float* p, const float* t, const float* gout, size_t n, float* dp, void* stream) {
    if (!p || !t || !gout || !dp) { set_error("bce_loss_bwd: NULL argument"); return SRMI_EINVAL; }
    return bce_bwd(p, t, gout, n, dp, (hipStream_t)stream);
}

int srmi_tv_loss_bwd(const float* x, const float* gout, int32_t N, int32_t C, int32_t H, int32_t W, float weight,
                     float* dx, void* stream) {
    if (!x || !gout || !dx) { set_error("tv_loss_bwd: NULL argument"); return SRMI_EINVAL; }
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) { set_error("tv_loss_bwd: bad dims"); return SRMI_EINVAL; }
    return tv_bwd(x, gout, N, C, H, W, weight, dx, (hipStream_t)stream);
}

int srmi_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                   double beta1, double beta2, double eps, int32_t step, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1) {
        set_error("adam_step: NULL argument or step < 1");
        return SRMI_EINVAL;
    }
    return adam_step(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, (hipStream_t)stream);
}

}  // extern "C"
