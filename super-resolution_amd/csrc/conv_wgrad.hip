// Weight gradient of a conv on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
//   dW[co][ci][ky][kx] = sum over (n, oy, ox) of dz[n,oy,ox,co] * x[n, oy*s-pad+ky, ox*s-pad+kx, ci]
//
// GEMM view: M = Cout, N = Cin, K = N*Ho*Wo output pixels, one GEMM per tap.  One wave owns a
// 32(co) x 32*NT(ci) tile of ONE tap over a slice of output rows (split-K), walking two pixels
// per MFMA (lanes 0-31 pixel ox, lanes 32-63 pixel ox+1); both operands are read with lanes along
// the channel axis (128-B coalesced rows of the NHWC tensors).  DETERMINISTIC, no atomics: the four
// waves of a workgroup add their K slices in LDS in wave order; with several workgroups per tile
// (grid.y > 1) each stores its partial tile in the caller's workspace and wgrad_reduce_kernel adds them
// in grid.y order and writes dW (OIHW f32); with one workgroup per tile it writes dW directly.
#include "common.h"

namespace srmi {

struct WgradParams {
    const float* x; const float* dz; float* dw;
    float* partial;   // non-NULL: workgroup (tile, y) stores its raw accumulator tile at ((y * tiles + tile) * NT*1024 floats)
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

    if (p.partial) {   // lane-linear (coalesced) image of the accumulators; wgrad_reduce_kernel knows the layout
        float* dst = p.partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (NT * 1024) + lane;
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) dst[(u * 16 + e) * 64] = acc[u][e];
        return;
    }
    // D[row = co][col = ci]: acc[e] holds row (e&3) + 8*(e>>2) + 4*h, column i
    const size_t taps = (size_t)p.kh * p.kw;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int ci = ci0 + 32 * u;
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = cot * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (row < p.Cout) p.dw[((size_t)row * p.Cin + ci) * taps + tap] = acc[u][e];
        }
    }
}

// dw = sum over y (in order) of the partial tiles; one thread per accumulator element of a tile
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradParams p, int NT, int tiles, int ysplits) {
    const size_t per_tile = (size_t)NT * 1024;
    const size_t total = per_tile * tiles;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int lane = (int)(idx & 63);
        const int ue = (int)((idx >> 6) % (NT * 16));
        int t = (int)(idx / per_tile);
        float v = p.partial[idx];
        for (int y = 1; y < ysplits; ++y) v += p.partial[(size_t)y * total + idx];
        const int cit = t % p.ci_tiles; t /= p.ci_tiles;
        const int cot = t % p.co_tiles; t /= p.co_tiles;
        const int tap = t;
        const int u = ue >> 4, e = ue & 15, i = lane & 31, h = lane >> 5;
        const int row = cot * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int ci = cit * 32 * NT + 32 * u + i;
        if (row < p.Cout && ci < p.Cin) p.dw[((size_t)row * p.Cin + ci) * ((size_t)p.kh * p.kw) + tap] = v;
    }
}

// tiling / split plan shared by the launch and the workspace query
struct WgradPlan { int NT, co_tiles, ci_tiles, rows, rows_per_split, gy; long long tiles; };

static int wgrad_plan(const srmi_conv_desc& d, const ConvGeom& g, bool allow_split, WgradPlan* w) {
    const long long rows = (long long)d.N * g.Ho;
    if (rows > 0x7fffffffLL) { set_error("wgrad: N*Ho = %lld exceeds int32", rows); return SRMI_EINVAL; }
    w->rows = (int)rows;
    w->NT = d.Cin > 32 ? 2 : 1;
    w->co_tiles = (d.Cout + 31) / 32;
    w->ci_tiles = (d.Cin + 32 * w->NT - 1) / (32 * w->NT);
    w->tiles = (long long)d.kh * d.kw * w->co_tiles * w->ci_tiles;
    if (w->tiles > 0x7fffffffLL) { set_error("wgrad: too many tiles"); return SRMI_EINVAL; }
    // split K so that ~2048 waves are in flight (4 per workgroup), at least 64 output pixels per wave
    long long splits = (2048 / w->tiles + 3) / 4 * 4;
    if (splits < 4) splits = 4;
    if (!allow_split) splits = 4;             // one workgroup per tile: its four waves share the rows
    const int min_rows = g.Wo > 0 ? (64 + g.Wo - 1) / g.Wo : 1;
    if (splits > (w->rows + min_rows - 1) / min_rows) splits = (w->rows + min_rows - 1) / min_rows;
    if (splits < 1) splits = 1;
    if (splits > 65535 * 4) splits = 65535 * 4;
    w->rows_per_split = (int)((w->rows + splits - 1) / splits);
    if (w->rows_per_split < 1) w->rows_per_split = 1;
    splits = (w->rows + w->rows_per_split - 1) / w->rows_per_split;
    w->gy = (int)((splits + 3) / 4);
    if (w->gy < 1) w->gy = 1;
    return SRMI_OK;
}

size_t wgrad_workspace_bytes(const srmi_conv_desc& d, const ConvGeom& g) {
    WgradPlan w;
    if (wgrad_plan(d, g, true, &w) || w.rows == 0 || g.Wo == 0) return 0;
    return w.gy > 1 ? (size_t)w.gy * w.tiles * w.NT * 1024 * sizeof(float) : 0;
}

int wgrad_launch(const srmi_conv_desc& d, const ConvGeom& g, const float* x, const float* dz, float* dw,
                 hipStream_t s, void* ws, size_t ws_bytes) {
    WgradParams p;
    p.x = x; p.dz = dz; p.dw = dw; p.partial = nullptr;
    p.N = d.N; p.H = d.H; p.W = d.W; p.Cin = d.Cin; p.Cout = d.Cout; p.kh = d.kh; p.kw = d.kw;
    p.stride = d.stride; p.pad = d.pad; p.Ho = g.Ho; p.Wo = g.Wo; p.in_cs = g.in_cs;
    WgradPlan w;
    int rc = wgrad_plan(d, g, true, &w);
    if (rc) return rc;
    if (w.rows == 0 || g.Wo == 0) return SRMI_OK;
    if (w.gy > 1) {
        const size_t need = (size_t)w.gy * w.tiles * w.NT * 1024 * sizeof(float);
        if (ws && ws_bytes >= need) p.partial = (float*)ws;
        else if ((rc = wgrad_plan(d, g, false, &w))) return rc;     // no workspace: one workgroup per tile (slow, same bits rule)
    }
    p.rows = w.rows; p.rows_per_split = w.rows_per_split; p.co_tiles = w.co_tiles; p.ci_tiles = w.ci_tiles;
    dim3 grid((unsigned)w.tiles, (unsigned)w.gy);
    if (w.NT == 2) hipLaunchKernelGGL(conv_wgrad_kernel<2>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(256), 0, s, p);
    rc = hip_check(hipGetLastError(), "conv_wgrad_kernel");
    if (rc || !p.partial) return rc;
    const size_t total = (size_t)w.tiles * w.NT * 1024;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, w.NT, (int)w.tiles, w.gy);
    return hip_check(hipGetLastError(), "wgrad_reduce_kernel");
}

}  // namespace srmi
