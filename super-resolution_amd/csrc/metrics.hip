// Loss and metric reductions (HBM-bound): MSE, BCE, TV loss, Y-channel PSNR and SSIM.
//
// Every reduction is two launches and DETERMINISTIC: stage 1 leaves one f64 partial per
// workgroup in the caller's workspace (fixed grid, fixed grid-stride order, wave shuffle tree,
// LDS across the four waves), stage 2 is one workgroup that adds the partials in index order
// and writes the scalar(s).  No atomics, no host synchronisation, nothing allocated.
#include <mutex>

#include "common.h"

namespace srmi {

constexpr int RED_BLOCK = 256;
constexpr int RED_MAX_BLOCKS = 1024;  // 2 sums x 1024 x 8 B = SRMI_REDUCE_WS_BYTES

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// sum over the workgroup; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* lds) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();  // lds may still be read from a previous call
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < RED_BLOCK / 64; ++w) r += lds[w];
    return r;
}

enum { OP_SQDIFF = 0, OP_BCE = 1 };

template <int OP>
__device__ __forceinline__ double pair_term(float a, float b) {
    if (OP == OP_SQDIFF) {
        const float d = a - b;
        return (double)d * (double)d;
    }
    // nn.BCELoss: -(t*log(p) + (1-t)*log(1-p)), each log clamped at -100
    const float lp = fmaxf(logf(a), -100.f), lq = fmaxf(log1pf(-a), -100.f);
    return -((double)b * (double)lp + (1.0 - (double)b) * (double)lq);
}

template <int OP>
__global__ __launch_bounds__(RED_BLOCK) void pair_reduce_kernel(const float* __restrict__ a,
                                                                const float* __restrict__ b, size_t n, int vec,
                                                                double* __restrict__ ws) {
    __shared__ double lds[RED_BLOCK / 64];
    double acc = 0.0;
    const size_t tid = (size_t)blockIdx.x * RED_BLOCK + threadIdx.x, nth = (size_t)gridDim.x * RED_BLOCK;
    if (vec) {  // both pointers 16-B aligned
        const size_t n4 = n / 4;
        for (size_t i = tid; i < n4; i += nth) {
            const f32x4 p = ((const f32x4*)a)[i], q = ((const f32x4*)b)[i];
            acc += pair_term<OP>(p[0], q[0]) + pair_term<OP>(p[1], q[1]) + pair_term<OP>(p[2], q[2]) +
                   pair_term<OP>(p[3], q[3]);
        }
        for (size_t i = n4 * 4 + tid; i < n; i += nth) acc += pair_term<OP>(a[i], b[i]);
    } else {
        for (size_t i = tid; i < n; i += nth) acc += pair_term<OP>(a[i], b[i]);
    }
    const double r = block_sum(acc, lds);
    if (threadIdx.x == 0) ws[blockIdx.x] = r;
}

// TVLoss (P/losses.py:46-55): sums of squared vertical / horizontal neighbour differences, NCHW f32
__global__ __launch_bounds__(RED_BLOCK) void tv_reduce_kernel(const float* __restrict__ x, size_t total, int H, int W,
                                                              double* __restrict__ ws) {
    __shared__ double lds[RED_BLOCK / 64];
    double ah = 0.0, aw = 0.0;
    const size_t tid = (size_t)blockIdx.x * RED_BLOCK + threadIdx.x, nth = (size_t)gridDim.x * RED_BLOCK;
    for (size_t i = tid; i < total; i += nth) {
        const int xc = (int)(i % W);
        const int yr = (int)((i / W) % H);
        const float v = x[i];
        if (yr > 0) { const float d = v - x[i - W]; ah += (double)d * (double)d; }
        if (xc > 0) { const float d = v - x[i - 1]; aw += (double)d * (double)d; }
    }
    const double rh = block_sum(ah, lds);
    const double rw = block_sum(aw, lds);
    if (threadIdx.x == 0) { ws[blockIdx.x] = rh; ws[RED_MAX_BLOCKS + blockIdx.x] = rw; }
}

// out[0] = c0 * sum(ws[0:nb]) + c1 * sum(ws[MAX:MAX+nb])
__global__ __launch_bounds__(RED_BLOCK) void finalize_f32_kernel(const double* __restrict__ ws, int nb, double c0,
                                                                 double c1, float* __restrict__ out) {
    __shared__ double lds[RED_BLOCK / 64];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nb; i += RED_BLOCK) {
        a += ws[i];
        if (c1 != 0.0) b += ws[RED_MAX_BLOCKS + i];
    }
    const double ra = block_sum(a, lds);
    const double rb = block_sum(b, lds);
    if (threadIdx.x == 0) out[0] = (float)(c0 * ra + c1 * rb);
}

// ---- Y-channel PSNR / SSIM (P/mode.py:24-45, :355-373) -----------------------------------------
// skimage.color.rgb2ycbcr on RGB in [0,1]: Y = 16 + 65.481 R + 128.553 G + 24.966 B
struct YImage {
    const float* p;   // CHW f32, values in [-1,1]
    int rs;           // row stride (elements)
    size_t ps;        // plane stride (elements)
};

__device__ __forceinline__ double luma(const YImage im, int y, int x, int clip) {
    const size_t o = (size_t)y * im.rs + x;
    float r = im.p[o], g = im.p[o + im.ps], b = im.p[o + 2 * im.ps];
    if (clip) {  // np.clip(output, -1, 1), P/mode.py:360
        r = fminf(fmaxf(r, -1.f), 1.f); g = fminf(fmaxf(g, -1.f), 1.f); b = fminf(fmaxf(b, -1.f), 1.f);
    }
    const double R = ((double)r + 1.0) * 0.5, G = ((double)g + 1.0) * 0.5, B = ((double)b + 1.0) * 0.5;
    return 16.0 + 65.481 * R + 128.553 * G + 24.966 * B;
}

// sum over the cropped region of ((Ya - Yb)/255)^2
// blockIdx.y = image of the batch (image strides 3 * plane stride); each image owns gridDim.x partial slots
__global__ __launch_bounds__(RED_BLOCK) void ypsnr_kernel(YImage a, YImage b, int y0, int x0, int h, int w, int clip_a,
                                                          double* __restrict__ ws) {
    __shared__ double lds[RED_BLOCK / 64];
    a.p += (size_t)blockIdx.y * 3 * a.ps;
    b.p += (size_t)blockIdx.y * 3 * b.ps;
    double acc = 0.0;
    const size_t total = (size_t)h * w;
    const size_t tid = (size_t)blockIdx.x * RED_BLOCK + threadIdx.x, nth = (size_t)gridDim.x * RED_BLOCK;
    for (size_t i = tid; i < total; i += nth) {
        const int yy = y0 + (int)(i / w), xx = x0 + (int)(i % w);
        const double d = (luma(a, yy, xx, clip_a) - luma(b, yy, xx, 0)) / 255.0;
        acc += d * d;
    }
    const double r = block_sum(acc, lds);
    if (threadIdx.x == 0) ws[blockIdx.y * gridDim.x + blockIdx.x] = r;
}

// skimage.metrics.structural_similarity, uniform win x win window, sample covariance, K1=.01 K2=.03,
// mean of the SSIM map over the region where the window fits.  16x16 outputs per workgroup, the two
// luma tiles (16+win-1)^2 staged in LDS as f64 (the moments of Y in [16,235] against C2 = 9e-4 need it).
constexpr int SS_T = 16;
constexpr int SS_MAXWIN = 7;
constexpr int SS_TILE = SS_T + SS_MAXWIN - 1;

__global__ __launch_bounds__(RED_BLOCK) void yssim_kernel(YImage a, YImage b, int y0, int x0, int h, int w, int win,
                                                          int clip_a, double c1, double c2, int tiles_x, int tiles,
                                                          double* __restrict__ ws) {
    __shared__ double lds[RED_BLOCK / 64];
    __shared__ double ta[SS_TILE * SS_TILE], tb[SS_TILE * SS_TILE];
    a.p += (size_t)blockIdx.y * 3 * a.ps;
    b.p += (size_t)blockIdx.y * 3 * b.ps;
    const int oh = h - win + 1, ow = w - win + 1;  // valid outputs
    const int span = SS_T + win - 1;
    const double inv_np = 1.0 / (double)(win * win);
    const double cov_norm = (double)(win * win) / (double)(win * win - 1);
    double acc = 0.0;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int ty = (t / tiles_x) * SS_T, tx = (t % tiles_x) * SS_T;
        __syncthreads();
        for (int i = threadIdx.x; i < span * span; i += RED_BLOCK) {
            const int ly = i / span, lx = i % span;
            const int yy = ty + ly, xx = tx + lx;
            double va = 0.0, vb = 0.0;
            if (yy < h && xx < w) { va = luma(a, y0 + yy, x0 + xx, clip_a); vb = luma(b, y0 + yy, x0 + xx, 0); }
            ta[ly * SS_TILE + lx] = va;
            tb[ly * SS_TILE + lx] = vb;
        }
        __syncthreads();
        const int ly = threadIdx.x / SS_T, lx = threadIdx.x % SS_T;
        if (ty + ly < oh && tx + lx < ow) {
            double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
            for (int dy = 0; dy < win; ++dy)
                for (int dx = 0; dx < win; ++dx) {
                    const double p = ta[(ly + dy) * SS_TILE + lx + dx], q = tb[(ly + dy) * SS_TILE + lx + dx];
                    sx += p; sy += q; sxx += p * p; syy += q * q; sxy += p * q;
                }
            const double ux = sx * inv_np, uy = sy * inv_np;
            const double vx = cov_norm * (sxx * inv_np - ux * ux), vy = cov_norm * (syy * inv_np - uy * uy);
            const double vxy = cov_norm * (sxy * inv_np - ux * uy);
            acc += ((2.0 * ux * uy + c1) * (2.0 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
        }
    }
    const double r = block_sum(acc, lds);
    if (threadIdx.x == 0) ws[RED_MAX_BLOCKS + blockIdx.y * gridDim.x + blockIdx.x] = r;
}

// out[0] = mse of Y/255, out[1] = PSNR (dB, data_range 1), out[2] = mean SSIM (NaN when not requested)
__global__ __launch_bounds__(RED_BLOCK) void finalize_metrics_kernel(const double* __restrict__ ws, int nb_psnr,
                                                                     int nb_ssim, double inv_px, double inv_ssim_px,
                                                                     double* __restrict__ out) {
    __shared__ double lds[RED_BLOCK / 64];
    double a = 0.0, b = 0.0;
    out += 3 * blockIdx.x;   // one workgroup per image
    for (int i = threadIdx.x; i < nb_psnr; i += RED_BLOCK) a += ws[blockIdx.x * nb_psnr + i];
    for (int i = threadIdx.x; i < nb_ssim; i += RED_BLOCK) b += ws[RED_MAX_BLOCKS + blockIdx.x * nb_ssim + i];
    const double ra = block_sum(a, lds);
    const double rb = block_sum(b, lds);
    if (threadIdx.x == 0) {
        const double mse = ra * inv_px;
        out[0] = mse;
        out[1] = 10.0 * log10(1.0 / mse);  // +inf for identical images, as skimage
        out[2] = nb_ssim > 0 ? rb * inv_ssim_px : __builtin_nan("");
    }
}

static int red_blocks(size_t n) {
    size_t b = (n + RED_BLOCK * 4 - 1) / (RED_BLOCK * 4);
    if (b < 1) b = 1;
    if (b > RED_MAX_BLOCKS) b = RED_MAX_BLOCKS;
    return (int)b;
}

int pair_loss(int op, const float* a, const float* b, size_t n, float* out, void* ws, hipStream_t s) {
    const int nb = red_blocks(n);
    const int vec = (((uintptr_t)a | (uintptr_t)b) % 16) == 0;
    if (op == OP_SQDIFF)
        hipLaunchKernelGGL(pair_reduce_kernel<OP_SQDIFF>, dim3(nb), dim3(RED_BLOCK), 0, s, a, b, n, vec, (double*)ws);
    else
        hipLaunchKernelGGL(pair_reduce_kernel<OP_BCE>, dim3(nb), dim3(RED_BLOCK), 0, s, a, b, n, vec, (double*)ws);
    int rc = hip_check(hipGetLastError(), "pair_reduce_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(finalize_f32_kernel, dim3(1), dim3(RED_BLOCK), 0, s, (const double*)ws, nb, 1.0 / (double)n, 0.0,
                       out);
    return hip_check(hipGetLastError(), "finalize_f32_kernel");
}

int tv_loss(const float* x, int N, int C, int H, int W, float weight, float* out, void* ws, hipStream_t s) {
    const size_t total = (size_t)N * C * H * W;
    const int nb = red_blocks(total);
    hipLaunchKernelGGL(tv_reduce_kernel, dim3(nb), dim3(RED_BLOCK), 0, s, x, total, H, W, (double*)ws);
    int rc = hip_check(hipGetLastError(), "tv_reduce_kernel");
    if (rc) return rc;
    // weight * 2 * (h_tv / count_h + w_tv / count_w) / batch; a zero count divides 0 by 0 in the
    // reference (NaN) and does so here
    const double count_h = (double)C * (H - 1) * W, count_w = (double)C * H * (W - 1);
    const double k = 2.0 * (double)weight / (double)N;
    hipLaunchKernelGGL(finalize_f32_kernel, dim3(1), dim3(RED_BLOCK), 0, s, (const double*)ws, nb, k / count_h,
                       k / count_w, out);
    return hip_check(hipGetLastError(), "finalize_f32_kernel");
}

// nimg image pairs (dense batches: sr [nimg,3,H,W], gt [nimg,3,gt_H,gt_W]) in three launches; the RED_MAX_BLOCKS partial
// slots of the workspace are divided evenly between the images
int y_metrics(const float* sr, const float* gt, int nimg, int H, int W, int gt_H, int gt_W, int border, int clip,
              int win, double* out, void* ws, hipStream_t s) {
    YImage a{sr, W, (size_t)H * W}, b{gt, gt_W, (size_t)gt_H * gt_W};
    const int h = H - 2 * border, w = W - 2 * border;
    const int slots = RED_MAX_BLOCKS / nimg;   // caller checked 1 <= nimg <= RED_MAX_BLOCKS
    int nbp = red_blocks((size_t)h * w);
    if (nbp > slots) nbp = slots;
    hipLaunchKernelGGL(ypsnr_kernel, dim3(nbp, nimg), dim3(RED_BLOCK), 0, s, a, b, border, border, h, w, clip,
                       (double*)ws);
    int rc = hip_check(hipGetLastError(), "ypsnr_kernel");
    if (rc) return rc;
    int nbs = 0;
    double inv_ssim = 0.0;
    if (win > 0) {
        const int oh = h - win + 1, ow = w - win + 1;
        const int tiles_x = (ow + SS_T - 1) / SS_T, tiles_y = (oh + SS_T - 1) / SS_T;
        const long long tiles = (long long)tiles_x * tiles_y;
        nbs = (int)(tiles < slots ? tiles : slots);
        inv_ssim = 1.0 / ((double)oh * ow);
        // data_range = 1.0 as the reference passes it (P/mode.py:45), K1 = 0.01, K2 = 0.03
        hipLaunchKernelGGL(yssim_kernel, dim3(nbs, nimg), dim3(RED_BLOCK), 0, s, a, b, border, border, h, w, win, clip,
                           0.01 * 0.01, 0.03 * 0.03, tiles_x, (int)tiles, (double*)ws);
        rc = hip_check(hipGetLastError(), "yssim_kernel");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(finalize_metrics_kernel, dim3(nimg), dim3(RED_BLOCK), 0, s, (const double*)ws, nbp, nbs,
                       1.0 / ((double)h * w), inv_ssim, out);
    return hip_check(hipGetLastError(), "finalize_metrics_kernel");
}

}  // namespace srmi

// ---- the one entry point that allocates nothing and takes no workspace (include/srmi.h) -----------------------------
// srmi_mse_loss_noalloc = srmi_mse_loss with the per-workgroup f64 partials kept in a file-scope device array instead of
// a caller workspace.  In host code the NAME `srmi_noalloc_scratch` is not a device address (passing it to a kernel
// compiles and then faults): the array's device address is resolved once per device with hipGetSymbolAddress (each
// GPU holds its own copy at its own address) and only that pointer is ever handed to a kernel.
static __device__ __align__(256) unsigned char srmi_noalloc_scratch[SRMI_REDUCE_WS_BYTES];

namespace srmi {
constexpr int NOALLOC_MAX_DEVICES = 64;
static void* g_noalloc_ptr[NOALLOC_MAX_DEVICES];   // per device; written once under the mutex, then read-only
static std::mutex g_noalloc_mutex;

// device pointer of this device's copy of the scratch array (resolved on the first call on each device)
static int noalloc_scratch(void** out) {
    int dev = -1;
    int rc = hip_check(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) {
        set_error("mse_loss_noalloc: device ordinal %d outside 0..%d", dev, NOALLOC_MAX_DEVICES - 1);
        return SRMI_EUNSUPPORTED;
    }
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (!g_noalloc_ptr[dev]) {
        void* p = nullptr;
        rc = hip_check(hipGetSymbolAddress(&p, HIP_SYMBOL(srmi_noalloc_scratch)), "hipGetSymbolAddress(scratch)");
        if (rc) return rc;
        if (!p) { set_error("mse_loss_noalloc: hipGetSymbolAddress returned NULL"); return SRMI_EHIP; }
        g_noalloc_ptr[dev] = p;
    }
    *out = g_noalloc_ptr[dev];
    return SRMI_OK;
}
}  // namespace srmi

extern "C" int srmi_mse_loss_noalloc(const float* a, const float* b, size_t n, float* out, void* stream) {
    using namespace srmi;
    if (!a || !b || !out) { set_error("mse_loss_noalloc: NULL argument"); return SRMI_EINVAL; }
    if (n > (size_t)SRMI_NOALLOC_MAX_ELEMS) {
        set_error("mse_loss_noalloc: n = %zu exceeds SRMI_NOALLOC_MAX_ELEMS = %zu", n, (size_t)SRMI_NOALLOC_MAX_ELEMS);
        return SRMI_EINVAL;
    }
    void* ws = nullptr;
    int rc = noalloc_scratch(&ws);
    if (rc) return rc;
    if (stream) {          // the scratch copy belongs to the CURRENT device: refuse a stream that lives on another GPU
        int sdev = -1, cdev = -1;
        if (hipStreamGetDevice((hipStream_t)stream, &sdev) == hipSuccess && hipGetDevice(&cdev) == hipSuccess && sdev != cdev) {
            set_error("mse_loss_noalloc: stream belongs to device %d, current device is %d", sdev, cdev);
            return SRMI_EINVAL;
        }
    }
    // nothing to clear: stage 1 overwrites slots 0..nb-1 and stage 2 reads exactly those slots
    return pair_loss(OP_SQDIFF, a, b, n, out, ws, (hipStream_t)stream);
}
