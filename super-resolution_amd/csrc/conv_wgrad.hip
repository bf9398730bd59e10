// Weight gradient of a conv on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
//   dW[co][ci][ky][kx] = sum over (n, oy, ox) of dz[n,oy,ox,co] * x[n, oy*s-pad+ky, ox*s-pad+kx, ci]
//
// GEMM view: M = Cout, N = Cin, K = N*Ho*Wo output pixels, one GEMM per tap.  One wave owns a
// 32(co) x 32*NT(ci) tile of ONE tap over a slice of output rows (split-K), walking two pixels
// per MFMA (lanes 0-31 pixel ox, lanes 32-63 pixel ox+1); both operands are read with lanes along
// the channel axis (128-B coalesced rows of the NHWC tensors).  Partial tiles are added to dW
// (OIHW f32, zeroed by the caller of the C ABI entry) with f32 atomics: the gradient is the same set
// of products, summation order varies between runs at the 1e-7 level.
#include "common.h"

namespace srmi {

struct WgradParams {
    const float* x; const float* dz; float* dw;
    int N, H, W, Cin, Cout, kh, kw, stride, pad, Ho, Wo;
    int in_cs;        // channels per pixel of x
    int rows;         // N * Ho
    int rows_per_split;
    int co_tiles, ci_tiles;
};

template <int NT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    // blockIdx.x: (tap, co tile, ci tile); blockIdx.y * 4 + wave: split
    int t = blockIdx.x;
    const int cit = t % p.ci_tiles; t /= p.ci_tiles;
    const int cot = t % p.co_tiles; t /= p.co_tiles;
    const int tap = t;
    const int ky = tap / p.kw, kx = tap % p.kw;
    const int split = blockIdx.y * 4 + wave;
    const int r0 = min(split * p.rows_per_split, p.rows);   // a wave past the end runs an empty loop (it still joins the barrier)
    const int r1 = min(r0 + p.rows_per_split, p.rows);
    const int co = cot * 32 + i;
    const int ci0 = cit * 32 * NT + i;
    const bool co_ok = co < p.Cout;

    f32x16 acc[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[u][e] = 0.f;

    for (int r = r0; r < r1; ++r) {
        const int n = r / p.Ho, oy = r % p.Ho;
        const int iy = oy * p.stride - p.pad + ky;
        if (iy < 0 || iy >= p.H) continue;  // the whole row of taps falls in the padding
        const float* dzr = p.dz + (size_t)r * p.Wo * p.Cout;
        const float* xr = p.x + ((size_t)n * p.H + iy) * p.W * p.in_cs;
        for (int ox0 = 0; ox0 < p.Wo; ox0 += 2) {
            const int ox = ox0 + h;
            const int ix = ox * p.stride - p.pad + kx;
            const bool pv = ox < p.Wo;
            const float a = (pv && co_ok) ? dzr[(size_t)ox * p.Cout + co] : 0.f;
            const bool xv = pv && ix >= 0 && ix < p.W;
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int ci = ci0 + 32 * u;
                const float b = (xv && ci < p.Cin) ? xr[(size_t)ix * p.in_cs + ci] : 0.f;
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
            }
        }
    }

    // the four waves of a workgroup hold partial sums of the SAME tile (consecutive K slices): add them in LDS
    // so that one wave issues the atomics (4x fewer, and they are what bounds small layers)
    __shared__ float red[3][NT][16][64];
    if (wave > 0) {
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[wave - 1][u][e][lane] = acc[u][e];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[u][e] += red[0][u][e][lane] + red[1][u][e][lane] + red[2][u][e][lane];

    // D[row = co][col = ci]: acc[e] holds row (e&3) + 8*(e>>2) + 4*h, column i
    const size_t taps = (size_t)p.kh * p.kw;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int ci = ci0 + 32 * u;
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = cot * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (row < p.Cout) unsafeAtomicAdd(p.dw + ((size_t)row * p.Cin + ci) * taps + tap, acc[u][e]);
        }
    }
}

int wgrad_launch(const srmi_conv_desc& d, const ConvGeom& g, const float* x, const float* dz, float* dw,
                 hipStream_t s) {
    WgradParams p;
    p.x = x; p.dz = dz; p.dw = dw;
    p.N = d.N; p.H = d.H; p.W = d.W; p.Cin = d.Cin; p.Cout = d.Cout; p.kh = d.kh; p.kw = d.kw;
    p.stride = d.stride; p.pad = d.pad; p.Ho = g.Ho; p.Wo = g.Wo; p.in_cs = g.in_cs;
    const long long rows = (long long)d.N * g.Ho;
    if (rows > 0x7fffffffLL) { set_error("wgrad: N*Ho = %lld exceeds int32", rows); return SRMI_EINVAL; }
    p.rows = (int)rows;
    if (p.rows == 0 || g.Wo == 0) return SRMI_OK;
    const int NT = d.Cin > 32 ? 2 : 1;
    p.co_tiles = (d.Cout + 31) / 32;
    p.ci_tiles = (d.Cin + 32 * NT - 1) / (32 * NT);
    const long long tiles = (long long)d.kh * d.kw * p.co_tiles * p.ci_tiles;
    if (tiles > 0x7fffffffLL) { set_error("wgrad: too many tiles"); return SRMI_EINVAL; }
    // split K so that ~2048 waves are in flight (4 per workgroup), at least 64 output pixels per wave
    long long splits = (2048 / tiles + 3) / 4 * 4;
    if (splits < 4) splits = 4;
    const int min_rows = (64 + g.Wo - 1) / g.Wo;
    if (splits > (p.rows + min_rows - 1) / min_rows) splits = (p.rows + min_rows - 1) / min_rows;
    if (splits < 1) splits = 1;
    if (splits > 65535 * 4) splits = 65535 * 4;
    p.rows_per_split = (int)((p.rows + splits - 1) / splits);
    splits = (p.rows + p.rows_per_split - 1) / p.rows_per_split;
    dim3 grid((unsigned)tiles, (unsigned)((splits + 3) / 4));
    if (NT == 2) hipLaunchKernelGGL(conv_wgrad_kernel<2>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(256), 0, s, p);
    return hip_check(hipGetLastError(), "conv_wgrad_kernel");
}

}  // namespace srmi
