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
 *     allocates nothing and frees nothing, keeps no global state except read-only tables
 *     (ONE exception, opt-in per conv: SRMI_CONV_FUSE_CHAIN, which see);
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

#define SRMI_VERSION 119 /* 0.1.9: srmi_bn_train_stats_n, srmi_conv2d_fwd_bn_stats; 0.1.8: SRMI_FUSE_K9PW, srmi_epilogue_bwd takes NULL sum pointers; 0.1.7: SRMI_CONV_FUSE_CHAIN + srmi_chain_status + SRMI_ETIMEOUT; 0.1.6: srmi_conv_args.x_copy; 0.1.5: srmi_adam_step takes double hyper-parameters (0.1.4: bn_train_stats counts num_batches_tracked, epilogue_bwd zero-fills the BN-fed bias gradient;
                            srmi_conv2d_kernel_name reports the f32 kernel that runs) */

/* error codes */
#define SRMI_OK 0
#define SRMI_EINVAL (-1)      /* bad descriptor / shape / dtype / alignment */
#define SRMI_EUNSUPPORTED (-2) /* valid request that no kernel in this build serves */
#define SRMI_EHIP (-3)        /* HIP runtime error at launch (message has hipGetErrorString) */
#define SRMI_ETIMEOUT (-4)    /* a persistent (SRMI_CONV_FUSE_CHAIN) launch on this device gave up waiting: see srmi_chain_status */

/* storage dtypes */
#define SRMI_F32 0
#define SRMI_BF16 1
/* layouts */
#define SRMI_NHWC 0
#define SRMI_NCHW 1
/* compute paths */
#define SRMI_COMPUTE_F32 0  /* f32 results: the <=1e-4 parity path.  Convs with Cin, Cout multiples of 32 run as six bf16
                               MFMA products per k on operands split into three bf16 (an f32 FMA chain's rounding,
                               kernels "b3_f32" / "b3h_f32"); everything else, and everything under SRMI_F32_EXACT=1,
                               on the exact f32 MFMA v_mfma_f32_32x32x2_f32 ("f32t" / "generic_f32") */
#define SRMI_COMPUTE_BF16 1 /* bf16 MFMA, f32 accumulate: the throughput path */
/* activations (applied after scale/shift, before the residual add) */
#define SRMI_ACT_NONE 0
#define SRMI_ACT_PRELU 1   /* alpha[0] (scalar) or alpha[co] when alpha_per_channel; nn.PReLU, P/srgan_model.py:8 */
#define SRMI_ACT_LRELU 2   /* slope = act_param; nn.LeakyReLU, P/srgan_model.py:43 */
#define SRMI_ACT_RELU 3
#define SRMI_ACT_TANH 4    /* nn.Tanh, P/srgan_model.py:25 */
#define SRMI_ACT_SIGMOID 5 /* nn.Sigmoid, P/srgan_model.py:60 */
#define SRMI_ACT_ADD_RELU 6 /* the one activation applied AFTER the residual: y = relu(conv*scale + shift + residual), the
                               `add([shortcut, conv])` -> ReLU of res_conv_block (Res-U-SRGAN/untitled0.py:67-72).  Served by the
                               generic kernel and, for bf16 3x3 convs, the K-chunked bf16 kernel; residual == NULL makes
                               it a plain ReLU */
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
    const float* scale;    /* [Cout] f32; NULL = all ones (SRMI_COMPUTE_F32 generic kernel only) */
    const float* shift;    /* [Cout] f32; NULL = all zeros (same restriction) */
    const float* alpha;    /* PReLU slope(s), f32, device memory (read in-kernel: no host sync) */
    const void* residual;  /* NULL or a dense tensor shaped like the final output */
    void* y;               /* output */
    int32_t flags;         /* SRMI_CONV_* bits, 0 = none */
    int32_t reserved;      /* must be 0 */
    float* x_copy;         /* NULL, or a dense [N, Cin, H, W] f32 NCHW buffer: the launch ALSO writes its input there,
                              re-laid out and widened from the halo tile it stages in LDS anyway (no extra pass over x:
                              `feat` of Generator.forward, P/srgan_model.py:34,39, which the first Upsampler conv reads
                              as NHWC bf16 and the caller gets as NCHW f32).  Served by "c64_bf16" launches with a dense
                              64-channel bf16 input, W % 4 == 0 and a 16-byte aligned buffer; every other kernel refuses
                              a non-NULL x_copy with SRMI_EUNSUPPORTED (srmi_layout_convert is the stand-alone pass) */
} srmi_conv_args;

/* srmi_conv_args.flags */
#define SRMI_CONV_FUSE_NEXT 1 /* inside srmi_conv2d_fwd_seq: y is consumed ONLY by the next conv of the
                                 sequence (whose x is this y), so the pair may run as one fused launch
                                 that never writes y (the ResBlock conv-BN-act-conv-BN-add, P/ops.py:45-48).
                                 y must still be a valid scratch buffer: shapes no fused kernel serves
                                 run as two launches. */
#define SRMI_CONV_FUSE_CHAIN 2 /* with SRMI_CONV_FUSE_NEXT on the first conv of a fusable ResBlock pair whose residual is its
                                 own input: consecutive such pairs of one srmi_conv2d_fwd_seq call, each reading the
                                 previous one's output, may run as ONE persistent launch ("resblock_chain_kernel": every
                                 workgroup keeps its tile in LDS from block to block and exchanges the two-pixel halo with
                                 its neighbours through flags in device memory; the 16-block trunk of Generator.forward,
                                 P/srgan_model.py:32-33).  Results are bit-identical to the launch-per-block form, and every
                                 block's y is still written in full.  Taken only when every workgroup of a launch is
                                 resident at once (W > 64, H >= 12, >= 2 blocks, and T = ceil(H/12) * ceil(W/48) <= the
                                 device's CU count): N * T <= CUs is one launch; a bigger batch runs as several chained
                                 launches of floor(CUs / T) whole images each when those fill the chip well enough to beat
                                 the launch-per-block form (6 * ceil(N T / CUs) >= 5 * launches), e.g. 32 images of 16 tiles,
                                 or the five 124 x 976 strips of a 960 x 540 frame; otherwise the pairs run one launch each.  What the flag opts into: (1) the
                                 library allocates 4 KiB of device memory and a 64-byte host-mapped status word per device
                                 at the first such launch (never freed) - make that first call outside a graph capture;
                                 (2) a chained launch on another stream than the previous one first waits on the host for
                                 that stream (two persistent grids at once would starve each other); (3) residency is a
                                 property of the whole GPU: if another PROCESS holds CUs for long, a workgroup gives up
                                 after ~50 ms of waiting, the launch still ends, its outputs are invalid, and every later
                                 chained launch and srmi_chain_status() on that device return SRMI_ETIMEOUT.
                                 TAIL LINKS: the flag (without SRMI_CONV_FUSE_NEXT) on the single convs that directly follow
                                 the last chained pair lets them run as the chain's last links: first a 3x3 64->64 bf16
                                 conv on the last pair's y with a dense bf16 residual that no block of the chain writes
                                 (conv02 + the long skip add of Generator.forward, P/srgan_model.py:33-34; it honours
                                 SRMI_CONV_Y_TRANSIENT when the next link follows), then - always last - a 3x3 conv
                                 64 -> 64..256 on the previous link's y, any shuffle / store mode of "c64_bf16", no
                                 residual, x_copy allowed (the first Upsampler conv, P/ops.py:74-78).  Either may be absent.
                                 HEAD LINK: the flag (without SRMI_CONV_FUSE_NEXT) on the conv right in front of the first
                                 chained pair lets it run as the chain's first link when it is a "k9c3_bf16" conv (9x9,
                                 <= 3 input channels in any layout that kernel reads, 64 dense bf16 output channels = the
                                 first pair's x, no residual): conv01 of Generator.forward, P/srgan_model.py:29.  Its y is
                                 written in full. */
#define SRMI_CONV_Y_TRANSIENT 4 /* with SRMI_CONV_FUSE_CHAIN on the first conv of a pair: the pair's output y (conv2->y) is read
                                 ONLY by the next pair of the sequence, so when both run inside one chained launch y need
                                 not be written in full: the launch stores just the two outermost pixel rows and columns of
                                 every 12 x 48 tile (what the neighbouring tiles' halos read) - 38 % of the bytes - and the
                                 rest of y keeps whatever it held.  The last block of a chained launch, and every pair that
                                 runs as its own launch, always writes y in full. */

/* library version (SRMI_VERSION of the build). */
int srmi_version(void);

/* message for the calling thread's most recent failure ("" if none). */
const char* srmi_last_error(void);

/* name of the kernel that RUNS the forward conv of `d`: bf16 path "c64_bf16", "cin_bf16" (K-chunked 3x3; launches with at least
 * four 8 x 32 tiles per CU run its persistent, wave-specialised form conv_cinp_kernel: same packed weights, same bits),
 * "c64_thin_bf16", "k9c3_bf16" (9x9 / pad 4 or 3x3 / pad 1 conv of a <= 3-channel image),
 * "k9c1_bf16", "pw_bf16", "c1out_bf16"; f32 path "b3h_f32" (split-bf16, 3x3 / 64 input channels, halo tile in LDS),
 * "b3_f32" (split-bf16, general), "f32t" (exact f32 MFMA, LDS-tiled), "thin_f32" (<= 4 channels on one side, stride 1,
 * <= 9 taps, shift-only epilogue plus - on the <= 4 -> 64 side - a slope activation: fmaf chains on the vector ALU, HBM-bound), "generic_f32" (exact f32 MFMA, no LDS: NCHW or
 * bf16-storage inputs, odd shapes).  The packed-weight FORMAT is a property of this name (see below). */
int srmi_conv2d_kernel_name(const srmi_conv_desc* d, char* buf, size_t buf_len);

/* The HBM-bound SRCNN layers (SRCNN/srcnn.py:205-210) by name: the same descriptors go through srmi_conv2d_fwd, these
 * entry points only REFUSE (SRMI_EUNSUPPORTED) anything the dedicated kernel does not run, so a caller that asks for
 * them never lands on the generic kernel silently.  Weights packed by srmi_conv2d_pack_weights for the same descriptor.
 *   c1in : 9x9, ONE input channel (f32 plane) -> 64 g channels, NHWC bf16 out, none | ReLU | LeakyReLU | scalar PReLU
 *          ("k9c1_bf16": the 81 taps packed on the MFMA K axis; 4 B in + 2 Cout B out per pixel).
 *   c1out: k x k (3 | 5), 32 | 64 channels (NHWC bf16) -> ONE channel, f32 | bf16 out, any pointwise activation
 *          ("c1out_bf16": banded MFMA operand, every input row streamed once, no atomics; 2 Cin B in + 4 B out per pixel).
 * The 1x1 middle layer of the 9-1-5 variant is "pw_bf16" (Cin 32|64|128 -> Cout 16|32|64, streaming bf16 GEMM). */
int srmi_srcnn_layer_c1in(const srmi_conv_args* a, void* stream);
int srmi_srcnn_layer_c1out(const srmi_conv_args* a, void* stream);

/* bytes of the kernel-native packed weight buffer for `d`.  The format (and size) belongs to the kernel the dispatcher
 * selects for `d` (srmi_conv2d_kernel_name).  PACK WITH THE DESCRIPTOR YOU LAUNCH WITH: a packed buffer may be shared by
 * two descriptors only if srmi_conv2d_kernel_name returns the same name for both and their Cin, Cout, kh, kw agree.
 * The name never depends on N / H / W (one packed buffer serves every input size), but it does depend on more than
 * the weight's shape:
 *   "b3_f32"   SRMI_COMPUTE_F32, NHWC f32 input, Cin and Cout multiples of 32, <= 25 taps, kh*kw*Cin >= 256:
 *              [plane][tap][n][k] bf16, three planes (6 B per weight);
 *   "b3h_f32"  the same AND 3x3, stride 1, pad 1, Cin == 64, Cout % 64 == 0, act one of NONE / RELU / LRELU / PRELU (the
 *              slope activations), shuffle == 1, NHWC f32 output and f32 residual: [plane][tap][nt][ks][lane][8] bf16 -
 *              the SAME byte count as "b3_f32" in a different order, so e.g. an act = RELU and an act = TANH descriptor
 *              of one 64 -> 64 3x3 weight need two packed buffers (a mismatch is not detectable from the size);
 *   "f32t" / "thin_f32" / "generic_f32"  [tap][Cin_pad][Cout_pad] f32 (one format: these three may share a buffer).
 * Environment SRMI_F32_EXACT=1 (read once, at the first call) keeps every f32 conv on the exact-f32 MFMA kernels and
 * their format. */
int srmi_conv2d_packed_weight_bytes(const srmi_conv_desc* d, size_t* bytes);

/* OIHW f32 (the `nn.Conv2d.weight` layout, P/ops.py:11) -> kernel-native packed layout; runs on
 * the device (w_oihw and w_packed are device pointers).  Linear layers (P/srgan_model.py:56-59)
 * are 1x1 convs over a 1x1 image: pass weight[out,in] as OIHW with kh = kw = 1. */
int srmi_conv2d_pack_weights(const srmi_conv_desc* d, const float* w_oihw, void* w_packed, void* stream);

/* one fused conv (see srmi_conv_desc). */
int srmi_conv2d_fwd(const srmi_conv_args* a, void* stream);

/* Workspace variant for the linear-epilogue f32 convs of the training path (scale == NULL, act NONE, no
 * residual): spatially small, deep layers are split along K to fill the chip; with a workspace of
 * srmi_conv2d_workspace_bytes(d) bytes the K slices are stored there and added in slice order (same input =>
 * same bits, which keeps every activation on the same side of its ReLU kink from run to run).  Without a
 * workspace (srmi_conv2d_fwd, or ws == NULL / too small) the conv does not split: slower on such layers, same
 * determinism.  The library contains NO floating-point atomics.  bytes = 0: the conv never splits. */
int srmi_conv2d_workspace_bytes(const srmi_conv_desc* d, size_t* bytes);
int srmi_conv2d_fwd_ws(const srmi_conv_args* a, void* ws, size_t ws_bytes, void* stream);

/* n fused convs enqueued back to back on `stream` from native code (one FFI crossing per forward
 * of `Generator`, P/srgan_model.py:27-39: 37 convs). */
int srmi_conv2d_fwd_seq(const srmi_conv_args* a, int32_t n, void* stream);

/* ResBlock in one launch: y = residual + affine2(conv2(act(affine1(conv1(x))))); the intermediate stays
 * in LDS.  conv2->x is ignored; conv2->y must not alias conv1->x.  Returns SRMI_EUNSUPPORTED when the
 * pair is not a fusable shape (3x3, 64->64->64, NHWC bf16, SRMI_COMPUTE_BF16). Replaces `ResBlock.forward`
 * (P/ops.py:45-48). */
int srmi_resblock_fwd(const srmi_conv_args* conv1, const srmi_conv_args* conv2, void* stream);

/* SRMI_OK, or SRMI_ETIMEOUT if a SRMI_CONV_FUSE_CHAIN launch on the calling thread's current device has ever given up
 * waiting for a neighbouring workgroup (sticky; reads a host-mapped word, no synchronisation: call it after the stream
 * has been synchronised to cover the launches before).  No counterpart in the reference (P/ops.py:45-48 is a Python
 * loop over modules). */
int srmi_chain_status(void);

/* Number of convs ([head conv +] 2 x blocks [+ up to 2 tail convs], >= 4) that srmi_conv2d_fwd_seq(a, n, .) would run as ONE
 * SRMI_CONV_FUSE_CHAIN launch starting at a[0]; 0 if a[0] starts no chain (then it goes out as a fused pair or a single
 * conv). */
int srmi_conv2d_chain_length(const srmi_conv_args* a, int32_t n);

/* Non-zero if srmi_conv2d_fwd_seq would run (conv1 flagged SRMI_CONV_FUSE_NEXT, conv2) as one fused launch:
 * SRMI_FUSE_RESBLOCK (the pair above), SRMI_FUSE_UPTAIL (below) or SRMI_FUSE_K9PW; 0 if they go out as two launches. */
#define SRMI_FUSE_RESBLOCK 1
#define SRMI_FUSE_UPTAIL 2
#define SRMI_FUSE_K9PW 3     /* "k9c1pw_bf16": 9x9 conv of a one-channel f32 image to 64 channels + activation, then a 1x1 conv
                                64 -> 32 + activation (SRCNN 9-1-5, SRCNN/srcnn.py:205-208 with the 64/32 widths): the 64-channel
                                tensor never reaches memory (4 B in, 64 B out per pixel instead of 4 + 128 + 128 + 64); conv1 packed
                                for "k9c1_bf16", conv2 for "pw_bf16", dense bf16 NHWC in between, no residuals.  Bit-identical to
                                the two launches. */
int srmi_conv2d_can_fuse(const srmi_conv_args* conv1, const srmi_conv_args* conv2);

/* Any fusable pair in one launch (the kind srmi_conv2d_can_fuse reports); SRMI_EUNSUPPORTED otherwise.
 * SRMI_FUSE_UPTAIL is the generator's tail: conv1 = conv3x3 64->256 + PixelShuffle(2) (CRD order) + shared-slope
 * activation, NHWC bf16 in; conv2 = conv3x3 64->3 + any pointwise activation, NCHW f32 out, on conv1's output
 * (conv2->d.H/W = 2x conv1's).  conv1's output tensor is never written: conv1->y is ignored (may be any non-aliasing
 * pointer), conv2->x is ignored.  The last conv's nine tap sums are accumulated in int32 fixed point (2^-18 units)
 * in LDS, so the result is the same for every run and every tiling; it differs from the two-launch path by that
 * quantisation (<= 2e-5 on the pre-activation) and by f32 summation order only.  RANGE: each of the nine tap
 * contributions to an output value is clamped to +-900 before the fixed-point conversion (9 x 900 x 2^18 < 2^31); a
 * network whose last conv has tap sums beyond that (pre-activations in the thousands: tanh / sigmoid saturated long
 * before) saturates silently here and would differ from the two-launch path - run such a pair as two launches (do
 * not set SRMI_CONV_FUSE_NEXT on conv1; Python: `srmi.engine.set_fuse_tail(False)`).  Replaces the second `Upsampler`
 * + `last_conv` + tanh of `Generator.forward` (P/ops.py:74-78, P/srgan_model.py:36-37). */
int srmi_conv2d_fwd_pair(const srmi_conv_args* conv1, const srmi_conv_args* conv2, void* stream);

/* dense tensor re-layout / cast: src (N,C,H,W logical) in (src_layout, src_dtype) -> dst.
 * Replaces the implicit NCHW contract at the module edge (P/dataset.py:46-47, P/mode.py:396-400). */
int srmi_layout_convert(const void* src, void* dst, int32_t N, int32_t C, int32_t H, int32_t W,
                        int32_t src_layout, int32_t src_dtype, int32_t dst_layout, int32_t dst_dtype,
                        void* stream);

/* Image I/O at the edge of the hot path.  uint8 HWC (what PIL/numpy hand over, P/dataset.py:21-22)
 * -> f32 NCHW in [-1,1], `v / 127.5 - 1` rounded once from f64 exactly as P/dataset.py:40-47 computes it. */
int srmi_u8hwc_to_f32nchw(const uint8_t* src, float* dst, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);

/* f32 NCHW generator output -> uint8 HWC: [clip to [-1,1] when clip != 0, P/mode.py:360] -> (v+1)/2 -> *255 ->
 * truncate like numpy's astype(np.uint8) (P/mode.py:397-400); out-of-range values saturate. */
int srmi_f32nchw_to_u8hwc(const float* src, uint8_t* dst, int32_t N, int32_t H, int32_t W, int32_t C, int32_t clip,
                          void* stream);

/* Inference-path addressing ops of the U-Net / autoencoder generator bodies (Keras MaxPooling2D((2,2)),
 * UpSampling2D((2,2)) nearest, concatenate(axis=3): U-SRGAN/srganunet.py:44-77).  NHWC, dtype = SRMI_F32 | SRMI_BF16.
 * Concatenation is never a copy: producers write channel slices of one buffer.
 *   srmi_maxpool2x2_nhwc:   x has in_cstride channels per pixel (0 = C; x points at the slice's first channel), y is a
 *                           dense [N, H/2, W/2, C] tensor; H, W are the INPUT size (floor, Keras padding='valid').
 *   srmi_upsample2x_nhwc:   x dense [N, H, W, C]; y[n, 2h+i, 2w+j, out_coffset + c] = x[n, h, w, c] in a buffer with
 *                           out_cstride channels per pixel (0 = C). */
int srmi_maxpool2x2_nhwc(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t in_cstride,
                         int32_t dtype, void* stream);
int srmi_upsample2x_nhwc(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t out_cstride,
                         int32_t out_coffset, int32_t dtype, void* stream);

/* Training data pipeline on the device: `crop` + `augmentation` + normalisation + HWC->CHW of
 * `mydata.__getitem__` (P/dataset.py:27-49, :81-132) for a whole batch in one launch.  The dataset lives in device
 * memory as two packed uint8 HWC buffers (LR images, GT images; GT = scale x LR in both dimensions); `items` is a
 * DEVICE array of B descriptors that the host fills from the same random draws the reference makes
 * (P/dataset.py:92-93: ix, iy; :111-113: hor_flip, ver_flip, rot, applied in that order: fliplr, flipud, transpose).
 * Outputs: lr_out [B, C, patch, patch], gt_out [B, C, scale*patch, scale*patch], f32 NCHW, `v / 127.5 - 1`
 * evaluated in f32 when f32_arith != 0 (the in-memory path, P/dataset.py:33-34 + :40-41) or in f64 and rounded once
 * (the from-disk path, :37-41).  Bit-exact with the reference's numpy expressions. */
typedef struct srmi_crop_item {
    int64_t lr_offset;   /* byte offset of the LR image in lr_pack */
    int64_t gt_offset;   /* byte offset of the GT image in gt_pack */
    int32_t lr_h, lr_w;  /* LR image size; the GT image is scale*lr_h x scale*lr_w */
    int32_t iy, ix;      /* top-left corner of the LR patch */
    int32_t flags;       /* SRMI_AUG_* */
    int32_t reserved;    /* must be 0 */
} srmi_crop_item;
#define SRMI_AUG_HFLIP 1
#define SRMI_AUG_VFLIP 2
#define SRMI_AUG_TRANSPOSE 4
int srmi_crop_augment_u8(const uint8_t* lr_pack, const uint8_t* gt_pack, const srmi_crop_item* items, int32_t B,
                         int32_t patch, int32_t scale, int32_t C, int32_t f32_arith, float* lr_out, float* gt_out,
                         void* stream);

/* ---- losses and metrics (HBM-bound reductions) ------------------------------------------------
 * All results are written to DEVICE memory (no host synchronisation).  `ws` is a caller-owned device
 * workspace of at least SRMI_REDUCE_WS_BYTES bytes, 8-byte aligned; it holds the per-workgroup f64
 * partials of the deterministic two-stage reduction (no atomics: same input => same bits). */
#define SRMI_REDUCE_WS_BYTES 16384

/* out[0] = mean((a-b)^2) over n f32 elements.  Replaces nn.MSELoss (P/mode.py:89, :169; P/losses.py:30). */
int srmi_mse_loss(const float* a, const float* b, size_t n, float* out, void* ws, void* stream);

/* srmi_mse_loss for callers that can neither allocate nor hand over a workspace (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  NO counterpart in the reference.  The f64 partials live in a
 * SRMI_REDUCE_WS_BYTES (16 KiB) file-scope device array of the library (one copy per GPU, held until the process
 * ends; its address is resolved once per device on the first call there).  The call enqueues two kernels on `stream`
 * and calls no hipMalloc / hipFree / allocating routine.  n > SRMI_NOALLOC_MAX_ELEMS is rejected with SRMI_EINVAL.
 * The FIRST call on each device resolves the array's address with hipGetSymbolAddress, which is not a capturable
 * operation: make one call per device outside graph capture first (any n >= 0; the result can be discarded).
 * `stream` must belong to the CURRENT device (hipGetDevice): the scratch copy is chosen by the current device and a
 * stream of another GPU is rejected with SRMI_EINVAL rather than run against the wrong GPU's array.
 * Every call on a GPU shares that GPU's array: calls that overlap in time on one GPU (two streams, two threads) RACE
 * on it - serialise them on one stream, or use srmi_mse_loss with a workspace per stream.  The array needs no clearing
 * between calls (every slot read was written by the same call).  Result bit-identical to srmi_mse_loss. */
#define SRMI_NOALLOC_MAX_ELEMS ((size_t)1 << 32)
int srmi_mse_loss_noalloc(const float* a, const float* b, size_t n, float* out, void* stream);

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

/* The same for every image pair of a batch in THREE launches (every training variant scores a validation
 * set each epoch, P/mode.py:116-119; the reference moves one image at a time to the host): sr is a dense
 * [n_images, 3, H, W] tensor, gt a dense [n_images, 3, gt_H, gt_W] tensor, out holds n_images x 3 f64
 * ([mse, psnr, ssim] per image).  The workspace's partial slots are divided between the images, so
 * n_images <= SRMI_METRICS_MAX_BATCH; results are bit-identical to n_images calls of srmi_y_psnr_ssim only when
 * the per-image grids coincide (they do for n_images = 1); otherwise they agree to f64 rounding. */
#define SRMI_METRICS_MAX_BATCH 256
int srmi_y_psnr_ssim_batch(const float* sr, const float* gt, int32_t n_images, int32_t H, int32_t W, int32_t gt_H,
                           int32_t gt_W, int32_t border, int32_t clip, int32_t win, double* out, void* ws,
                           void* stream);

/* ---- training step (fp32): train-mode BatchNorm, epilogue fwd/bwd, dgrad, wgrad, pool, loss grads, Adam ----
 * The training path keeps the conv and its epilogue as separate launches because train-mode BatchNorm needs
 * the batch statistics of the whole conv output before it can normalise (P/ops.py:24-26 with the module in
 * .train(), P/mode.py:101, :149, :164):   z = conv(x) + bias  (srmi_conv2d_fwd, act NONE)  ->  batch stats  ->
 * y = store(act(bn(z))) + residual.  All tensors dense NHWC f32.  `ws` must hold srmi_train_workspace_bytes(C)
 * bytes, 8-byte aligned: the per-channel reductions leave one f64 partial per workgroup and channel there and a
 * finalize launch adds them in a fixed order (no atomics: same input => same bits); it needs no clearing. */
typedef struct srmi_epi_desc {
    int32_t N, H, W, C;        /* geometry of z, the conv output BEFORE any sub-pixel shuffle */
    int32_t act;               /* SRMI_ACT_* */
    float act_param;           /* LeakyReLU slope */
    int32_t alpha_per_channel; /* PReLU: alpha[C / r^2] instead of a scalar */
    int32_t shuffle;           /* sub-pixel factor of the store (y is [N, H*r, W*r, C/r^2]); 1 = none */
    int32_t shuffle_order;     /* SRMI_SHUFFLE_* */
    int32_t bn;                /* 1: y = act((z - mean) * rstd * gamma + beta) */
} srmi_epi_desc;

int srmi_train_workspace_bytes(int32_t C, size_t* bytes);

/* Batch statistics of z [M pixels, C channels]: mean[c], rstd[c] = 1/sqrt(biased_var + eps); when
 * running_mean / running_var are non-NULL they are updated in place as nn.BatchNorm2d does
 * (r = (1-momentum) r + momentum * stat, unbiased variance).  P/ops.py:24-26. */
int srmi_bn_train_stats(const float* z, int64_t M, int32_t C, float eps, float momentum, float* mean, float* rstd,
                        float* running_mean, float* running_var, int64_t* num_batches_tracked, void* ws, void* stream);
/* (num_batches_tracked, may be NULL: the module's int64 counter, incremented by 1 on the device by the same launch that
 *  updates the running statistics - nn.BatchNorm2d's `self.num_batches_tracked.add_(1)`.) */
/* The same with the running statistics (and the counter) moved `updates` times, each time rounded to f32 as a forward of
 * its own would leave them: `updates` = 2 stands for TWO forwards of the module over the same batch with unchanged weights -
 * the SRGAN iteration runs the generator twice per step (P/mode.py:149 and :164) and the second forward's outputs are the
 * first's, bit for bit; only BatchNorm's buffers tell the difference.  updates = 0: statistics only. */
int srmi_bn_train_stats_n(const float* z, int64_t M, int32_t C, float eps, float momentum, float* mean, float* rstd,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, int32_t updates, void* ws,
                          void* stream);

/* srmi_conv2d_fwd_ws(a, conv_ws, conv_ws_bytes) followed by srmi_bn_train_stats_n on its output, as ONE call - and, where
 * the conv runs on the halo-tile split kernel ("b3h_f32": the generator's 64 -> 64 body convs) with at most as many tiles as the
 * statistics workspace has slots, as one pass: every workgroup leaves the per-channel sum and sum of squares of the 64
 * output pixels it has just computed in its own slot (f64, fixed order), and only the small finalize launch follows - no
 * second read of the conv output, one launch fewer per conv + BatchNorm (P/ops.py:22-26 under P/mode.py:101).  `a` must
 * describe z = conv + shift: dense NHWC f32 output, no scale / activation / residual / sub-pixel store.  stat_ws as for
 * srmi_bn_train_stats (srmi_train_workspace_bytes(Cout)).  Deterministic; the two forms sum in different orders, so mean /
 * rstd may differ in the last bit between a fused and an unfused shape. */
int srmi_conv2d_fwd_bn_stats(const srmi_conv_args* a, void* conv_ws, size_t conv_ws_bytes, float eps, float momentum,
                             float* mean, float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                             int32_t updates, void* stat_ws, void* stream);

/* y = store(act(bn(z))) + residual.  mean/rstd/gamma/beta are read only when d->bn; residual (shaped like y)
 * may be NULL.  Replaces BatchNorm2d + activation + PixelShuffle + `res += x` (P/ops.py:24-29, :47, :75-83). */
int srmi_epilogue_fwd(const srmi_epi_desc* d, const float* z, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, const float* alpha, const float* residual, float* y,
                      void* stream);

/* Backward of srmi_epilogue_fwd w.r.t. z and the parameters, given dy (shaped like y):
 *   dz      gradient w.r.t. the conv output (same shape as z);
 *   dsum1   [C]: d beta when d->bn, else the conv bias gradient (sum of dz over pixels; may be NULL when !d->bn and
 *           nobody wants it);
 *   dsum2   [C]: d gamma (d->bn only; may be NULL otherwise);
 *   dalpha  [1] or [C/r^2]: PReLU slope gradient (may be NULL when act != PRELU, or when the slope is frozen).
 * Without BN and with dsum1, dsum2, dalpha and dzero all NULL nothing is reduced: one streaming launch (dense layouts).
 * The residual's gradient is dy itself.  autograd of the same modules in the reference (P/mode.py:105, :160, :178). */
int srmi_epilogue_bwd(const srmi_epi_desc* d, const float* z, const float* dy, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, const float* alpha, float* dz, float* dsum1,
                      float* dsum2, float* dalpha, float* dzero, void* ws, void* stream);
/* (dzero, may be NULL: [C] floats set to 0 by the finalize launch - the gradient of a conv bias that feeds train-mode
 *  BatchNorm, which removes the mean: exactly zero.) */

/* Input gradient of the conv described by `d` (geometry fields only; NHWC f32): dx[N,H,W,Cin] from
 * dz[N,Ho,Wo,Cout].  The weights are packed once per parameter version by srmi_conv2d_dgrad_pack_weights.
 * `ws` (16-byte aligned, srmi_conv2d_dgrad_workspace_bytes(d) bytes; may be NULL) lets spatially small, deep layers
 * split K over the chip: the slices are stored and added in slice order.  Deterministic with or without it
 * (same inputs => same bits; autograd of the reference's nn.Conv2d, P/mode.py:105, :160, :178). */
int srmi_conv2d_dgrad_packed_weight_bytes(const srmi_conv_desc* d, size_t* bytes);
int srmi_conv2d_dgrad_pack_weights(const srmi_conv_desc* d, const float* w_oihw, void* w_packed, void* stream);
int srmi_conv2d_dgrad_workspace_bytes(const srmi_conv_desc* d, size_t* bytes);
int srmi_conv2d_dgrad(const srmi_conv_desc* d, const float* dz, const void* w_packed, float* dx, void* ws,
                      size_t ws_bytes, void* stream);

/* Weight gradient: dw (OIHW f32, the nn.Conv2d.weight.grad layout) = sum over pixels of dz (x) x; dw is
 * overwritten.  x is NHWC f32 with d->in_cstride channels per pixel (0 = Cin).  The pixel sum is split over
 * workgroups; with `ws` (16-byte aligned, srmi_conv2d_wgrad_workspace_bytes(d) bytes) each stores its partial tile
 * there and a second launch adds them in a fixed order; with ws == NULL one workgroup owns each tile (slow on large
 * maps).  Deterministic either way. */
int srmi_conv2d_wgrad_workspace_bytes(const srmi_conv_desc* d, size_t* bytes);
int srmi_conv2d_wgrad(const srmi_conv_desc* d, const float* x, const float* dz, float* dw, void* ws, size_t ws_bytes,
                      void* stream);

/* 2x2 stride-2 max pool, NHWC f32 (VGG19 features, P/vgg19.py:13-17); H, W are the INPUT size. */
int srmi_maxpool2x2_fwd(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int srmi_maxpool2x2_bwd(const float* x, const float* dy, float* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                        void* stream);

/* Gradients of srmi_mse_loss / srmi_bce_loss; gout is the upstream scalar gradient in device memory.
 * da or db may be NULL. */
int srmi_mse_loss_bwd(const float* a, const float* b, const float* gout, size_t n, float* da, float* db,
                      void* stream);
int srmi_bce_loss_bwd(const float* p, const float* t, const float* gout, size_t n, float* dp, void* stream);

/* Gradient of srmi_tv_loss w.r.t. x (NCHW f32), scaled by the upstream scalar gradient gout[0]. */
int srmi_tv_loss_bwd(const float* x, const float* gout, int32_t N, int32_t C, int32_t H, int32_t W, float weight,
                     float* dx, void* stream);

/* One Adam update of a flat f32 parameter buffer (torch.optim.Adam defaults: no amsgrad, no weight decay; P/mode.py:90,
 * :132); `step` is the 1-based step count used for the bias corrections.  The hyper-parameters are doubles, as Python
 * holds them: the scalars the update uses (1 - beta1, 1 - beta2, lr / (1 - beta1^step), sqrt(1 - beta2^step)) are formed
 * in double and rounded to f32 once, as torch.optim.Adam forms them - the update then follows torch's order
 * (lerp, mul + addcmul, sqrt / bc2 + eps, addcdiv) and agrees with it to an ulp.  One launch for a whole model when its
 * parameters, gradients and moments live in flat buffers (srmi/optim.py) instead of a dozen multi-tensor launches. */
int srmi_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                   double beta1, double beta2, double eps, int32_t step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SRMI_H */
