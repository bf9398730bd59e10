// Microbenchmark: what does a back-to-back launch of an (almost) empty kernel cost on this GPU, as a function of
// workgroup size, LDS allocation and register footprint?  (Explains the fixed part of the fused ResBlock launch.)
#include <hip/hip_runtime.h>
#include <stdio.h>

extern "C" __global__ __launch_bounds__(512, 1) void k_empty(int* p) {
    extern __shared__ unsigned char lds[];
    if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) lds[0] = 1, p[0] = lds[1];
}
extern "C" __global__ __launch_bounds__(512, 1) void k_regs(int* p) {   // touches a high VGPR: 250 registers allocated
    extern __shared__ unsigned char lds[];
    asm volatile("v_mov_b32 v249, 0" ::: "v249");
    if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) lds[0] = 1, p[0] = lds[1];
}
extern "C" __global__ __launch_bounds__(512, 1) void k_store(int* p) {   // every lane stores 32 B: 256 x 512 x 32 = 4 MiB
    int4* q = (int4*)p + ((size_t)blockIdx.x * 512 + threadIdx.x) * 2;
    q[0] = int4{1, 2, 3, 4}; q[1] = int4{1, 2, 3, 4};
}

static float run(void (*k)(int*), int grid, int block, int lds, int* buf, int n) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, 0, buf);
    hipEventRecord(a, 0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, 0, buf);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / n;
}

int main() {
    int* buf; hipMalloc(&buf, 64 << 20);
    const int n = 500;
    printf("empty  256 x 512, no LDS      %6.2f us\n", run(k_empty, 256, 512, 0, buf, n));
    printf("empty  256 x 512, 157 KiB LDS %6.2f us\n", run(k_empty, 256, 512, 157 * 1024, buf, n));
    printf("empty  256 x 256, 157 KiB LDS %6.2f us\n", run(k_empty, 256, 256, 157 * 1024, buf, n));
    printf("empty  256 x 64,  no LDS      %6.2f us\n", run(k_empty, 256, 64, 0, buf, n));
    printf("empty 2048 x 64,  no LDS      %6.2f us\n", run(k_empty, 2048, 64, 0, buf, n));
    printf("regs   256 x 512, 157 KiB LDS %6.2f us\n", run(k_regs, 256, 512, 157 * 1024, buf, n));
    printf("store  256 x 512 (4 MiB)      %6.2f us\n", run(k_store, 256, 512, 0, buf, n));
    printf("store 1024 x 512 (16 MiB)     %6.2f us\n", run(k_store, 1024, 512, 0, buf, n));
    hipFree(buf);
    return 0;
}
