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

int layout_convert(const void* src, void* dst, int N, int C, int H, int W, int sl, int sd, int dl, int dd,
                   hipStream_t s) {
    const size_t total = (size_t)N * C * H * W;
    if (total == 0) return SRMI_OK;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(layout_convert_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, N, C, H, W,
                       sl == SRMI_NCHW, sd == SRMI_BF16, dl == SRMI_NCHW, dd == SRMI_BF16, total);
    return hip_check(hipGetLastError(), "layout_convert_kernel");
}

}  // namespace srmi
