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

int layout_convert(const void* src, void* dst, int N, int C, int H, int W, int sl, int sd, int dl, int dd,
                   hipStream_t s) {
    const size_t total = (size_t)N * C * H * W;
    if (total == 0) return SRMI_OK;
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
