/*
 * srmi.h - C ABI of libsrmi.so: hand-written gfx950 (MI355X) kernels for the super-resolution
 * generator hot path of Amir-Rouhbakhsh/super-resolution.
 *
 * The reference has NO native code and NO FFI: its hot path reaches third-party kernels
 * (cuDNN / oneDNN through ATen) from `torch.nn` call sites.  Every entry point below therefore
 * cites the reference `torch.nn` call site(s) it replaces
 * (P/ = "pytorch srgan with lpips,dists/SRGAN_CustomDataset-main/").  The binding a maintainer
 * adds on the reference side is a ctypes stub - see INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no torch types; every buffer is caller-owned DEVICE memory; the library
 *     allocates nothing and frees nothing, keeps no global state except read-only tables;
 *   - all work is enqueued on the caller's stream (a hipStream_t passed as void*; NULL = the
 *     null stream); no host synchronisation, no default-stream use => graph-capturable;
 *   - every function returns 0 on success or a negative SRMI_E* code; the message for the
 *     calling thread's last failure is srmi_last_error(); shapes/dtypes/alignments are checked
 *     on the host BEFORE any launch; nothing throws or aborts.
 */
#ifndef SRMI_H
#define SRMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRMI_VERSION 100 /* 0.1.0 */

/* error codes */
#define SRMI_OK 0
#define SRMI_EINVAL (-1)      /* bad descriptor / shape / dtype / alignment */
#define SRMI_EUNSUPPORTED (-2) /* valid request that no kernel in this build serves */
#define SRMI_EHIP (-3)        /* HIP runtime error at launch (message has hipGetErrorString) */

/* storage dtypes */
#define SRMI_F32 0
#define SRMI_BF16 1
/* layouts */
#define SRMI_NHWC 0
#define SRMI_NCHW 1
/* compute paths */
#define SRMI_COMPUTE_F32 0  /* exact f32 MFMA (v_mfma_f32_32x32x2_f32): the <=1e-4 parity path */
#define SRMI_COMPUTE_BF16 1 /* bf16 MFMA, f32 accumulate: the throughput path */
/* activations (applied after scale/shift, before the residual add) */
#define SRMI_ACT_NONE 0
#define SRMI_ACT_PRELU 1   /* alpha[0] (scalar) or alpha[co] when alpha_per_channel; nn.PReLU, P/srgan_model.py:8 */
#define SRMI_ACT_LRELU 2   /* slope = act_param; nn.LeakyReLU, P/srgan_model.py:43 */
#define SRMI_ACT_RELU 3
#define SRMI_ACT_TANH 4    /* nn.Tanh, P/srgan_model.py:25 */
#define SRMI_ACT_SIGMOID 5 /* nn.Sigmoid, P/srgan_model.py:60 */
/* sub-pixel store orders (shuffle > 1) */
#define SRMI_SHUFFLE_CRD 0 /* nn.PixelShuffle: conv channel = c*r*r + i*r + j   (P/ops.py:75) */
#define SRMI_SHUFFLE_DCR 1 /* tf.nn.depth_to_space: conv channel = (i*r + j)*C + c (ESRGAN/untitled0.py:60) */

/*
 * One fused convolution:  y = store( act( conv(x, w) * scale[co] + shift[co] ) + residual )
 *
 * Replaces the chain  nn.Conv2d -> [nn.BatchNorm2d (eval)] -> [PReLU|LeakyReLU|Tanh|...]
 * -> [nn.PixelShuffle] -> [residual add]  built by `conv`, `ResBlock`, `Upsampler`,
 * `discrim_block` (P/ops.py:18-35, 37-48, 70-84, 86-96) and used by `Generator.forward`
 * (P/srgan_model.py:27-39) and `Discriminator.forward` (P/srgan_model.py:65-73).
 * Bias and eval-mode BatchNorm are folded by the caller into scale/shift:
 *   scale = gamma / sqrt(running_var + eps),  shift = beta + (bias - running_mean) * scale
 * (scale = 1, shift = bias for a conv without BN).
 */
typedef struct srmi_conv_desc {
    int32_t N, H, W;        /* input batch and spatial size */
    int32_t Cin, Cout;      /* conv channels (Cout BEFORE any sub-pixel shuffle) */
    int32_t kh, kw;         /* odd or even, any size */
    int32_t stride, pad;    /* `_conv` always uses pad = k/2 (P/ops.py:8-9); any value accepted */
    int32_t in_dtype;       /* SRMI_F32 | SRMI_BF16 */
    int32_t out_dtype;      /* SRMI_F32 | SRMI_BF16 */
    int32_t res_dtype;      /* dtype of `residual` (ignored when residual == NULL) */
    int32_t in_layout;      /* SRMI_NHWC | SRMI_NCHW */
    int32_t out_layout;     /* SRMI_NHWC | SRMI_NCHW (applies to y and residual) */
    int32_t compute;        /* SRMI_COMPUTE_F32 | SRMI_COMPUTE_BF16 */
    int32_t act;            /* SRMI_ACT_* */
    float act_param;        /* LeakyReLU slope */
    int32_t alpha_per_channel; /* PReLU: 0 = one scalar (reference), 1 = alpha[Cout] */
    int32_t shuffle;        /* sub-pixel factor r folded into the store addressing; 1 = none */
    int32_t shuffle_order;  /* SRMI_SHUFFLE_CRD | SRMI_SHUFFLE_DCR */
    int32_t in_cstride;     /* NHWC only: channels per pixel of the x buffer (0 = Cin); x points at the
                               first channel read, so a channel slice of a wider buffer can be consumed */
    int32_t out_cstride;    /* NHWC only: channels per pixel of the y buffer (0 = Cout / r^2) */
    int32_t out_coffset;    /* NHWC only: first channel written (concat-by-slice) */
    int32_t res_cstride;    /* NHWC only: channels per pixel of the residual buffer (0 = dense = Cout / r^2): lets the
                               residual be a channel slice of a concat buffer (RRDB `x + 0.2*y`, ESRGAN/untitled0.py:31-35) */
} srmi_conv_desc;

typedef struct srmi_conv_args {
    srmi_conv_desc d;
    const void* x;         /* input activations */
    const void* w;         /* weights packed by srmi_conv2d_pack_weights for THIS descriptor */
    const float* scale;    /* [Cout] f32 */
    const float* shift;    /* [Cout] f32 */
    const float* alpha;    /* PReLU slope(s), f32, device memory (read in-kernel: no host sync) */
    const void* residual;  /* NULL or a dense tensor shaped like the final output */
    void* y;               /* output */
    int32_t flags;         /* SRMI_CONV_* bits, 0 = none */
    int32_t reserved;      /* must be 0 */
} srmi_conv_args;

/* srmi_conv_args.flags */
#define SRMI_CONV_FUSE_NEXT 1 /* inside srmi_conv2d_fwd_seq: y is consumed ONLY by the next conv of the
                                 sequence (whose x is this y), so the pair may run as one fused launch
                                 that never writes y (the ResBlock conv-BN-act-conv-BN-add, P/ops.py:45-48).
                                 y must still be a valid scratch buffer: shapes no fused kernel serves
                                 run as two launches. */

/* library version (SRMI_VERSION of the build). */
int srmi_version(void);

/* message for the calling thread's most recent failure ("" if none). */
const char* srmi_last_error(void);

/* name of the kernel the dispatcher selects for `d` ("generic_f32", "c64_bf16", ...). */
int srmi_conv2d_kernel_name(const srmi_conv_desc* d, char* buf, size_t buf_len);

/* bytes of the kernel-native packed weight buffer for `d`. */
int srmi_conv2d_packed_weight_bytes(const srmi_conv_desc* d, size_t* bytes);

/* OIHW f32 (the `nn.Conv2d.weight` layout, P/ops.py:11) -> kernel-native packed layout; runs on
 * the device (w_oihw and w_packed are device pointers).  Linear layers (P/srgan_model.py:56-59)
 * are 1x1 convs over a 1x1 image: pass weight[out,in] as OIHW with kh = kw = 1. */
int srmi_conv2d_pack_weights(const srmi_conv_desc* d, const float* w_oihw, void* w_packed, void* stream);

/* one fused conv (see srmi_conv_desc). */
int srmi_conv2d_fwd(const srmi_conv_args* a, void* stream);

/* n fused convs enqueued back to back on `stream` from native code (one FFI crossing per forward
 * of `Generator`, P/srgan_model.py:27-39: 37 convs). */
int srmi_conv2d_fwd_seq(const srmi_conv_args* a, int32_t n, void* stream);

/* ResBlock in one launch: y = residual + affine2(conv2(act(affine1(conv1(x))))); the intermediate stays
 * in LDS.  conv2->x is ignored; conv2->y must not alias conv1->x.  Returns SRMI_EUNSUPPORTED when the
 * pair is not a fusable shape (3x3, 64->64->64, NHWC bf16, SRMI_COMPUTE_BF16). Replaces `ResBlock.forward`
 * (P/ops.py:45-48). */
int srmi_resblock_fwd(const srmi_conv_args* conv1, const srmi_conv_args* conv2, void* stream);

/* 1 if srmi_conv2d_fwd_seq would run (conv1 flagged SRMI_CONV_FUSE_NEXT, conv2) as one fused launch, else 0. */
int srmi_conv2d_can_fuse(const srmi_conv_args* conv1, const srmi_conv_args* conv2);

/* dense tensor re-layout / cast: src (N,C,H,W logical) in (src_layout, src_dtype) -> dst.
 * Replaces the implicit NCHW contract at the module edge (P/dataset.py:46-47, P/mode.py:396-400). */
int srmi_layout_convert(const void* src, void* dst, int32_t N, int32_t C, int32_t H, int32_t W,
                        int32_t src_layout, int32_t src_dtype, int32_t dst_layout, int32_t dst_dtype,
                        void* stream);

/* ---- losses and metrics (HBM-bound reductions) ------------------------------------------------
 * All results are written to DEVICE memory (no host synchronisation).  `ws` is a caller-owned device
 * workspace of at least SRMI_REDUCE_WS_BYTES bytes, 8-byte aligned; it holds the per-workgroup f64
 * partials of the deterministic two-stage reduction (no atomics: same input => same bits). */
#define SRMI_REDUCE_WS_BYTES 16384

/* out[0] = mean((a-b)^2) over n f32 elements.  Replaces nn.MSELoss (P/mode.py:89, :169; P/losses.py:30). */
int srmi_mse_loss(const float* a, const float* b, size_t n, float* out, void* ws, void* stream);

/* out[0] = mean(-(t*log(p) + (1-t)*log(1-p))), logs clamped at -100.  Replaces nn.BCELoss on the
 * discriminator probabilities (P/mode.py:136, :153-154, :171). */
int srmi_bce_loss(const float* p, const float* t, size_t n, float* out, void* ws, void* stream);

/* out[0] = weight * 2 * (sum(dh^2)/count_h + sum(dw^2)/count_w) / N on an NCHW f32 tensor.
 * Replaces TVLoss.forward (P/losses.py:46-55). */
int srmi_tv_loss(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, float weight, float* out, void* ws,
                 void* stream);

/* Y-channel PSNR and SSIM of one image pair, both 3xHxW planar f32 in [-1,1] (the generator's range):
 *   ((v+1)/2) -> rgb2ycbcr luma (BT.601, 16..235) -> crop `border` pixels on every side ->
 *   out[0] = mse(Y/255), out[1] = PSNR in dB (data_range 1), out[2] = mean SSIM with a win x win uniform
 *   window, sample covariance, data_range 1.0 on the UNSCALED luma exactly as the reference calls it.
 * `sr` is dense 3xHxW; `gt` is the top-left HxW corner of a dense 3 x gt_H x gt_W tensor (P/mode.py:353).
 * clip != 0 clamps sr to [-1,1] first (P/mode.py:360 does, P/mode.py:24-45 does not).  win = 0 skips SSIM
 * (out[2] = NaN); otherwise win is odd, 3 <= win <= 7 and <= the cropped side (P/mode.py:40-43).
 * out is 3 f64 in device memory.  Replaces calculate_metrics' PSNR/SSIM (P/mode.py:24-45) and
 * test()'s PSNR (P/mode.py:369-373). */
int srmi_y_psnr_ssim(const float* sr, const float* gt, int32_t H, int32_t W, int32_t gt_H, int32_t gt_W,
                     int32_t border, int32_t clip, int32_t win, double* out, void* ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SRMI_H */
