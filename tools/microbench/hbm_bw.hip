// Microbenchmark: streaming read / write / copy bandwidth of this box (302 MB working set = the x4 activations of
// the cfg1 batch, larger than the 256 MB memory-side cache), as the denominator for the HBM-bound kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void k_read(const int4* __restrict__ p, size_t n, int* out) {
    int4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int4 v = p[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = 1;
}
__global__ __launch_bounds__(256) void k_write(int4* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = int4{1, 2, 3, 4};
}
__global__ __launch_bounds__(256) void k_copy(const int4* __restrict__ a, int4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

template <class F>
static float timeit(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); f();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) f();
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main() {
    const size_t bytes = 302ull << 20, n = bytes / 16;
    int4 *a, *b; int* o;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&o, 64);
    (void)hipMemset(a, 1, bytes);
    for (int grid : {1024, 4096, 16384}) {
        float r = timeit([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, o); });
        float w = timeit([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n); });
        float c = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
        printf("grid %5d: read %.2f TB/s (%.1f us)   write %.2f TB/s (%.1f us)   copy %.2f TB/s moved (%.1f us)\n", grid,
               bytes / (r * 1e-3) / 1e12, r * 1e3, bytes / (w * 1e-3) / 1e12, w * 1e3, 2.0 * bytes / (c * 1e-3) / 1e12, c * 1e3);
    }
    return 0;
}
