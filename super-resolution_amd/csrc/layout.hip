// Dense tensor re-layout / cast kernels (module-edge contract: NCHW f32 <-> NHWC f32|bf16).
#include "common.h"

namespace srmi {

// One thread per (n, y, x, c) element of the NHWC side; NHWC accesses are coalesced over c and x,
// the NCHW side is strided by H*W (C is 3 or 64 on this path, the tensors are L2-resident).
__global__ void layout_convert_kernel(const void* __restrict__ src, void* __restrict__ dst, int N, int C, int H,
                                      int W, int s_nchw, int s_bf16, int d_nchw, int d_bf16, size_t total) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        // idx enumerates the DESTINATION in its own memory order (coalesced writes)
        int n, c, y, x;
        size_t t = idx;
        if (d_nchw) { x = (int)(t % W); t /= W; y = (int)(t % H); t /= H; c = (int)(t % C); n = (int)(t / C); }
        else { c = (int)(t % C); t /= C; x = (int)(t % W); t /= W; y = (int)(t % H); n = (int)(t / H); }
        const size_t sidx = s_nchw ? (((size_t)n * C + c) * H + y) * W + x : (((size_t)n * H + y) * W + x) * C + c;
        const float v = s_bf16 ? bf2f(((const bf16_t*)src)[sidx]) : ((const float*)src)[sidx];
        if (d_bf16) ((bf16_t*)dst)[idx] = f2bf(v);
        else ((float*)dst)[idx] = v;
    }
}

// NHWC -> NCHW through an LDS transpose: 64 pixels x 64 channels per workgroup, both sides coalesced
// (the one-thread-per-element kernel reads 2-4 B at a stride of C elements on this direction).
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const void* __restrict__ src, void* __restrict__ dst,
                                                           int C, int HW, int tiles_per_img, int s_bf16, int d_bf16) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    const int cb = blockIdx.y * 64;
    const int n = blockIdx.x / tiles_per_img;
    const int p0 = (blockIdx.x - n * tiles_per_img) * 64;
    const size_t img = (size_t)n * HW;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + i * 256;
        const int px = e >> 6, c = e & 63;
        float v = 0.f;
        if (p0 + px < HW && cb + c < C) {
            const size_t si = (img + p0 + px) * C + cb + c;
            v = s_bf16 ? bf2f(((const bf16_t*)src)[si]) : ((const float*)src)[si];
        }
        tile[c][px] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + i * 256;
        const int c = e >> 6, px = e & 63;
        if (p0 + px < HW && cb + c < C) {
            const size_t di = ((size_t)n * C + cb + c) * HW + p0 + px;
            if (d_bf16) ((bf16_t*)dst)[di] = f2bf(tile[c][px]);
            else ((float*)dst)[di] = tile[c][px];
        }
    }
}

// uint8 HWC -> f32 NCHW in [-1,1]: v / 127.5 - 1 evaluated in f64 and rounded once, exactly as numpy does it in
// P/dataset.py:40-47 (`(LR / 127.5) - 1.0` on a float64 array, then .astype(np.float32)).
__global__ void u8hwc_to_f32nchw_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int C, int HW,
                                        size_t total) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const size_t px = idx % HW;
        const size_t t = idx / HW;
        const int c = (int)(t % C);
        const size_t n = t / C;
        const double v = (double)src[(n * HW + px) * C + c];
        dst[idx] = (float)(v / 127.5 - 1.0);
    }
}

// f32 NCHW (tanh range) -> uint8 HWC: [clip to [-1,1]] -> (v + 1) / 2 -> * 255 -> truncate, each step one f32
// rounding as in numpy (P/mode.py:360, :397-400); values outside [0,255] saturate (numpy's cast is undefined there).
__global__ void f32nchw_to_u8hwc_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, int C, int HW,
                                        int clip, size_t total) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const size_t t = idx / C;
        const size_t px = t % HW;
        const size_t n = t / HW;
        float v = src[(n * C + c) * HW + px];
        if (clip) v = fminf(fmaxf(v, -1.f), 1.f);
        v = __fmul_rn(__fdiv_rn(__fadd_rn(v, 1.f), 2.f), 255.f);
        v = fminf(fmaxf(v, 0.f), 255.f);
        dst[idx] = (unsigned char)(int)v;     // truncation toward zero
    }
}

int u8_to_f32(const unsigned char* src, float* dst, int N, int H, int W, int C, hipStream_t s) {
    const size_t total = (size_t)N * C * H * W;
    if (total == 0) return SRMI_OK;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(u8hwc_to_f32nchw_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, C, H * W, total);
    return hip_check(hipGetLastError(), "u8hwc_to_f32nchw_kernel");
}

int f32_to_u8(const float* src, unsigned char* dst, int N, int H, int W, int C, int clip, hipStream_t s) {
    const size_t total = (size_t)N * C * H * W;
    if (total == 0) return SRMI_OK;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(f32nchw_to_u8hwc_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, C, H * W, clip, total);
    return hip_check(hipGetLastError(), "f32nchw_to_u8hwc_kernel");
}

// ---- addressing ops of the U-Net / autoencoder generator bodies (inference, NHWC, f32 or bf16) -------------------
template <typename T>
__device__ __forceinline__ float pool_val(T v);
template <> __device__ __forceinline__ float pool_val<float>(float v) { return v; }
template <> __device__ __forceinline__ float pool_val<bf16_t>(bf16_t v) { return bf2f(v); }

template <typename T>
__global__ void maxpool2x2_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C, int in_cs, size_t total) {
    const int Ho = H / 2, Wo = W / 2;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        size_t t = idx / C;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const size_t n = t / Ho;
        const T* b = x + ((n * H + 2 * oy) * W + 2 * ox) * in_cs + c;
        const T v00 = b[0], v01 = b[in_cs], v10 = b[(size_t)W * in_cs], v11 = b[(size_t)W * in_cs + in_cs];
        T m = v00;                       // the stored value itself is returned (no re-rounding), first maximum wins
        if (pool_val(v01) > pool_val(m)) m = v01;
        if (pool_val(v10) > pool_val(m)) m = v10;
        if (pool_val(v11) > pool_val(m)) m = v11;
        y[idx] = m;
    }
}

template <typename T>
__global__ void upsample2x_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C, int out_cs, int out_co,
                                       size_t total) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        size_t t = idx / C;
        const int ox = (int)(t % (2 * W)); t /= 2 * W;
        const int oy = (int)(t % (2 * H));
        const size_t n = t / (2 * H);
        y[((n * 2 * H + oy) * 2 * W + ox) * out_cs + out_co + c] = x[((n * H + oy / 2) * W + ox / 2) * C + c];
    }
}

int pool_up_launch(int op, const void* x, void* y, int N, int H, int W, int C, int cs, int co, int dtype, hipStream_t s) {
    const size_t total = op == 0 ? (size_t)N * (H / 2) * (W / 2) * C : (size_t)N * 4 * H * W * C;
    if (total == 0) return SRMI_OK;
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    const dim3 g((unsigned)blocks), b(256);
    if (op == 0) {
        if (dtype == SRMI_F32) hipLaunchKernelGGL(maxpool2x2_nhwc_kernel<float>, g, b, 0, s, (const float*)x, (float*)y, H, W, C, cs, total);
        else hipLaunchKernelGGL(maxpool2x2_nhwc_kernel<bf16_t>, g, b, 0, s, (const bf16_t*)x, (bf16_t*)y, H, W, C, cs, total);
    } else {
        if (dtype == SRMI_F32) hipLaunchKernelGGL(upsample2x_nhwc_kernel<float>, g, b, 0, s, (const float*)x, (float*)y, H, W, C, cs, co, total);
        else hipLaunchKernelGGL(upsample2x_nhwc_kernel<bf16_t>, g, b, 0, s, (const bf16_t*)x, (bf16_t*)y, H, W, C, cs, co, total);
    }
    return hip_check(hipGetLastError(), op == 0 ? "maxpool2x2_nhwc_kernel" : "upsample2x_nhwc_kernel");
}

// crop + augmentation + normalisation + HWC->CHW of one training batch (P/dataset.py:27-49, :81-132).
// Output element (b, c, oy, ox) of the final patch = T(V(H(patch))): undo the transpose, then the vertical, then the
// horizontal flip to find the source pixel.  One launch covers the LR and the GT outputs (GT elements first).
__global__ void crop_augment_kernel(const unsigned char* __restrict__ lr_pack, const unsigned char* __restrict__ gt_pack,
                                    const srmi_crop_item* __restrict__ items, int C, int patch, int scale, int f32_arith,
                                    float* __restrict__ lr_out, float* __restrict__ gt_out, size_t n_gt, size_t total) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const bool is_gt = idx < n_gt;
        size_t e = is_gt ? idx : idx - n_gt;
        const int P = is_gt ? patch * scale : patch;
        const int ox = (int)(e % P); e /= P;
        const int oy = (int)(e % P); e /= P;
        const int c = (int)(e % C);
        const int b = (int)(e / C);
        const srmi_crop_item it = items[b];
        int y = oy, x = ox;
        if (it.flags & SRMI_AUG_TRANSPOSE) { y = ox; x = oy; }
        if (it.flags & SRMI_AUG_VFLIP) y = P - 1 - y;
        if (it.flags & SRMI_AUG_HFLIP) x = P - 1 - x;
        const int s = is_gt ? scale : 1;
        const size_t W = (size_t)it.lr_w * s;
        const unsigned char* img = is_gt ? gt_pack + it.gt_offset : lr_pack + it.lr_offset;
        const unsigned char u = img[(((size_t)it.iy * s + y) * W + (size_t)it.ix * s + x) * C + c];
        float v;
        if (f32_arith) {
            v = (float)u / 127.5f;      // two f32 roundings, like numpy on a float32 array
            v = v - 1.0f;
        } else {
            v = (float)((double)u / 127.5 - 1.0);
        }
        (is_gt ? gt_out : lr_out)[is_gt ? idx : idx - n_gt] = v;
    }
}

int crop_augment(const unsigned char* lr_pack, const unsigned char* gt_pack, const srmi_crop_item* items, int B, int patch,
                 int scale, int C, int f32_arith, float* lr_out, float* gt_out, hipStream_t s) {
    const size_t n_lr = (size_t)B * C * patch * patch;
    const size_t n_gt = n_lr * scale * scale;
    const size_t total = n_lr + n_gt;
    if (total == 0) return SRMI_OK;
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(crop_augment_kernel, dim3((unsigned)blocks), dim3(256), 0, s, lr_pack, gt_pack, items, C, patch, scale,
                       f32_arith, lr_out, gt_out, n_gt, total);
    return hip_check(hipGetLastError(), "crop_augment_kernel");
}

// The same transpose for the case the generator needs (`feat`: NHWC bf16 -> NCHW f32, C % 8 == 0, HW % 4 == 0),
// vectorised: 16-B loads of 8 channels, 16-B stores of 4 pixels; one element per thread and instruction left the
// 56 MB copy at 2 TB/s.
__global__ __launch_bounds__(256) void nhwc_bf16_to_nchw_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst,
                                                                    int C, int HW, int tiles_per_img) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    const int cb = blockIdx.y * 64;
    const int n = blockIdx.x / tiles_per_img;
    const int p0 = (blockIdx.x - n * tiles_per_img) * 64;
    const size_t img = (size_t)n * HW;
    u32x4 v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + i * 256;
        const int px = e >> 3, cc = (e & 7) * 8;
        v[i] = (u32x4){0u, 0u, 0u, 0u};
        if (p0 + px < HW && cb + cc < C) v[i] = *(const u32x4*)(src + (img + p0 + px) * C + cb + cc);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + i * 256;
        const int px = e >> 3, cc = (e & 7) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tile[cc + 2 * j][px] = bf2f((bf16_t)(v[i][j] & 0xffffu));
            tile[cc + 2 * j + 1][px] = bf2f((bf16_t)(v[i][j] >> 16));
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + i * 256;
        const int c = e >> 4, pq = (e & 15) * 4;
        if (cb + c < C && p0 + pq < HW) {      // HW % 4 == 0: a group of 4 pixels is inside or outside as a whole
            const f32x4 o = {tile[c][pq], tile[c][pq + 1], tile[c][pq + 2], tile[c][pq + 3]};
            *(f32x4*)(dst + ((size_t)n * C + cb + c) * HW + p0 + pq) = o;
        }
    }
}

int layout_convert(const void* src, void* dst, int N, int C, int H, int W, int sl, int sd, int dl, int dd,
                   hipStream_t s) {
    const size_t total = (size_t)N * C * H * W;
    if (total == 0) return SRMI_OK;
    if (sl == SRMI_NHWC && dl == SRMI_NCHW && sd == SRMI_BF16 && dd == SRMI_F32 && C % 8 == 0 && (H * W) % 4 == 0 &&
        ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0) {
        const int HW = H * W;
        const int tiles = (HW + 63) / 64;
        const long long gx = (long long)N * tiles;
        if (gx <= 0x7fffffffLL) {
            hipLaunchKernelGGL(nhwc_bf16_to_nchw_f32_kernel, dim3((unsigned)gx, (unsigned)((C + 63) / 64)), dim3(256), 0, s,
                               (const bf16_t*)src, (float*)dst, C, HW, tiles);
            return hip_check(hipGetLastError(), "nhwc_bf16_to_nchw_f32_kernel");
        }
    }
    if (sl == SRMI_NHWC && dl == SRMI_NCHW && C >= 8) {
        const int HW = H * W;
        const int tiles = (HW + 63) / 64;
        const long long gx = (long long)N * tiles;
        if (gx <= 0x7fffffffLL) {
            hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)gx, (unsigned)((C + 63) / 64)), dim3(256), 0, s, src, dst,
                               C, HW, tiles, sd == SRMI_BF16, dd == SRMI_BF16);
            return hip_check(hipGetLastError(), "nhwc_to_nchw_kernel");
        }
    }
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(layout_convert_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, N, C, H, W,
                       sl == SRMI_NCHW, sd == SRMI_BF16, dl == SRMI_NCHW, dd == SRMI_BF16, total);
    return hip_check(hipGetLastError(), "layout_convert_kernel");
}

}  // namespace srmi
