// Microbenchmark: what do the matrix cores sustain on this box?  Back-to-back independent v_mfma_f32_16x16x32_bf16
// from registers only (no LDS, no memory), 1 or 2 waves per SIMD, all CUs.  The price list says 2.5 PFLOP/s dense bf16.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512, 1) void k_mfma(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC>
static void run(const char* name, int block, int iters, float* buf) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(256), dim3(block), 0, 0, buf, iters);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_mfma<NACC>, dim3(256), dim3(block), 0, 0, buf, iters);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * 256 * (block / 64) * (double)iters * NACC * 16 * 16 * 32 * 2;
    printf("%-40s %7.1f TFLOP/s  (%.3f ms per launch)\n", name, flops / (ms * 1e-3) / 1e12, ms / 5);
}

int main() {
    float* buf; (void)hipMalloc(&buf, 1024);
    run<8>("4 waves/CU, 8 independent accumulators", 256, 20000, buf);
    run<8>("8 waves/CU, 8 independent accumulators", 512, 20000, buf);
    run<4>("8 waves/CU, 4 independent accumulators", 512, 20000, buf);
    run<8>("8 waves/CU, short (30 us) launch", 512, 280, buf);
    run<8>("8 waves/CU, long (2 ms) launch", 512, 100000, buf);
    (void)hipFree(buf);
    return 0;
}
