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
    unsigned wo_mul, ho_mul; int wo_sh, ho_sh;   // tiled kernel: multiply-shift division by Wo / Ho (f32t_magic)
};

void f32t_magic(int d, unsigned* mul, int* sh);
__device__ __forceinline__ int wg_fastdiv(int n, unsigned mul, int sh) {      // n / d for 0 <= n < 2^31
    return (int)(((unsigned long long)(unsigned)n * mul) >> sh);
}

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

// =====================================================================================================================
// LDS-tiled weight gradient (the roofline kernel; the kernel above stays as the fallback for kw > 9).
//
//   dW[co][ci][ky][kx] = sum over flattened output pixels m of dz[m][co] * x[pixel(m) * stride - pad + (ky, kx)][ci]
//
// GEMM view per kernel row ky: M = Cout, N = (kx, ci), K = output pixels.  One workgroup (4 waves) owns CO output
// channels x CT input channels x all KW taps of ONE ky and walks its slice of the pixels in chunks of 32: per chunk the
// dz rows [32 px][CO] and the im2col'd input rows [32 px][KW][CT] (each (pixel, kx) is CT*4 contiguous bytes of the NHWC
// tensor; zero for padding) go global -> registers -> LDS, double-buffered, one barrier per chunk.  Both LDS images are
// pixel-major = K-major for the MFMA: every operand read is a conflict-free ds_read_b32 per 32 lanes.  A wave keeps KW
// accumulator tiles of 32(co) x 32(ci): one dz read feeds KW MFMAs.  The 8 threads that share a pixel decode its
// coordinates once per chunk.  Deterministic like the kernel above: a fixed pixel order inside the workgroup, partial
// tiles through the caller's workspace when the pixels are split over workgroups, added in slice order.
template <int KW, int WCO, int WCI>
__global__ __launch_bounds__(256) void conv_wgrad2_kernel(WgradParams p, int chunks_per_slice, int M) {
    static_assert(WCO * WCI == 4, "four waves");
    constexpr int CO = WCO * 32, CT = WCI * 32;
    constexpr int A_LOADS = CO / 4 / 8;                  // 16-B pieces per thread: 8 threads share a pixel
    constexpr int B_LOADS = KW * CT / 4 / 8;
    __shared__ __attribute__((aligned(16))) float As[2][32][CO];
    __shared__ __attribute__((aligned(16))) float Bs[2][32][KW * CT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wco = wave / WCI, wci = wave % WCI;
    int t = blockIdx.x;
    const int cit = t % p.ci_tiles; t /= p.ci_tiles;
    const int cot = t % p.co_tiles; t /= p.co_tiles;
    const int ky = t;
    const int co0 = cot * CO, ci0 = cit * CT;
    const int kpx = tid >> 3, part = tid & 7;            // this thread's pixel of every chunk, and its share of that pixel's pieces
    const int c_first = blockIdx.y * chunks_per_slice;
    const int c_last = min(c_first + chunks_per_slice, (M + 31) / 32);
    const bool a_vec = (p.Cout % 4 == 0) && (((uintptr_t)p.dz) % 16 == 0);
    const bool b_vec = (p.Cin % 4 == 0) && (p.in_cs % 4 == 0) && (((uintptr_t)p.x) % 16 == 0);

    f32x4 ra[A_LOADS], rb[B_LOADS];
    auto load_chunk = [&](int c) {
        const int m = c * 32 + kpx;
        const bool mv = m < M;
        const float* dzr = p.dz + (size_t)(mv ? m : 0) * p.Cout;
#pragma unroll
        for (int l = 0; l < A_LOADS; ++l) {
            const int co = co0 + 4 * (part + 8 * l);
            f32x4 q = {0.f, 0.f, 0.f, 0.f};
            if (mv && co < p.Cout) {
                if (a_vec) q = *(const f32x4*)(dzr + co);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (co + j < p.Cout) q[j] = dzr[co + j];
                }
            }
            ra[l] = q;
        }
        const int mm = mv ? m : 0;                       // (run-time divisions cost ~40 instructions each: four per chunk were
        const int tq = wg_fastdiv(mm, p.wo_mul, p.wo_sh);   //  a fifth of the chunk's MFMA time)
        const int ox = mm - tq * p.Wo;
        const int n = wg_fastdiv(tq, p.ho_mul, p.ho_sh);
        const int oy = tq - n * p.Ho;
        const int iy = oy * p.stride - p.pad + ky;
        const bool rv = mv && iy >= 0 && iy < p.H;
        const float* xrow = p.x + ((size_t)n * p.H + (rv ? iy : 0)) * p.W * p.in_cs;
#pragma unroll
        for (int l = 0; l < B_LOADS; ++l) {
            const int q4 = part + 8 * l;                 // piece of this pixel's [KW][CT] row
            const int kx = q4 / (CT / 4), ci = ci0 + 4 * (q4 % (CT / 4));
            const int ix = ox * p.stride - p.pad + kx;
            f32x4 q = {0.f, 0.f, 0.f, 0.f};
            if (rv && ix >= 0 && ix < p.W && ci < p.Cin) {
                const float* src = xrow + (size_t)ix * p.in_cs + ci;
                if (b_vec) q = *(const f32x4*)src;
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (ci + j < p.Cin) q[j] = src[j];
                }
            }
            rb[l] = q;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int l = 0; l < A_LOADS; ++l) *(f32x4*)&As[buf][kpx][4 * (part + 8 * l)] = ra[l];
#pragma unroll
        for (int l = 0; l < B_LOADS; ++l) *(f32x4*)&Bs[buf][kpx][4 * (part + 8 * l)] = rb[l];
    };

    f32x16 acc[KW];
#pragma unroll
    for (int u = 0; u < KW; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[u][e] = 0.f;

    if (c_first < c_last) {
        load_chunk(c_first);
        store_chunk(0);
    }
    __syncthreads();
    int buf = 0;
    for (int c = c_first; c < c_last; ++c) {
        const bool more = c + 1 < c_last;
        if (more) load_chunk(c + 1);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int k = 2 * kk + h;
            const float a = As[buf][k][wco * 32 + i];
#pragma unroll
            for (int u = 0; u < KW; ++u)
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bs[buf][k][u * CT + wci * 32 + i], acc[u], 0, 0, 0);
        }
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    if (p.partial) {   // raw accumulators, lane-linear: [slice][tile][kx][wave][e][lane]
        float* dst = p.partial + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * KW) * 4096 + wave * 1024 + lane;
#pragma unroll
        for (int u = 0; u < KW; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) dst[(size_t)u * 4096 + e * 64] = acc[u][e];
        return;
    }
    const size_t taps = (size_t)p.kh * p.kw;
    const int ci = ci0 + wci * 32 + i;
    if (ci >= p.Cin) return;
#pragma unroll
    for (int u = 0; u < KW; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = co0 + wco * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (row < p.Cout) p.dw[((size_t)row * p.Cin + ci) * taps + (size_t)ky * p.kw + u] = acc[u][e];
        }
}

// dw = sum over slices of the partial tiles of conv_wgrad2_kernel, in a FIXED association (deterministic): a workgroup
// handles 32 elements, 8 threads per element take the slices y = g, g + 8, ... in order, and the 8 sub-sums are added
// in g order through LDS.
__global__ __launch_bounds__(256) void wgrad2_reduce_kernel(WgradParams p, int KW, int WCI, int tiles, int slices) {
    __shared__ float part[8][32];
    const size_t per_tile = (size_t)KW * 4096;
    const size_t total = per_tile * tiles;
    const int CO = (4 / WCI) * 32, CT = WCI * 32;
    const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
    for (size_t base = (size_t)blockIdx.x * 32; base < total; base += (size_t)gridDim.x * 32) {   // total % 32 == 0
        const size_t idx = base + el;
        float v = 0.f;
        for (int y = g; y < slices; y += 8) v += p.partial[(size_t)y * total + idx];
        __syncthreads();
        part[g][el] = v;
        __syncthreads();
        if (g != 0) continue;
        v = part[0][el];
#pragma unroll
        for (int r = 1; r < 8; ++r) v += part[r][el];
        const int lane = (int)(idx & 63), e = (int)((idx >> 6) & 15), wave = (int)((idx >> 10) & 3);
        const int u = (int)((idx >> 12) % KW);
        int t = (int)(idx / per_tile);
        const int cit = t % p.ci_tiles; t /= p.ci_tiles;
        const int cot = t % p.co_tiles; t /= p.co_tiles;
        const int ky = t;
        const int wco = wave / WCI, wci = wave % WCI, i = lane & 31, h = lane >> 5;
        const int row = cot * CO + wco * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int ci = cit * CT + wci * 32 + i;
        if (row < p.Cout && ci < p.Cin)
            p.dw[((size_t)row * p.Cin + ci) * ((size_t)p.kh * p.kw) + (size_t)ky * p.kw + u] = v;
    }
}

// =====================================================================================================================
// Full-map convs (the kernel covers the whole input, no padding: nn.Linear on a flattened feature map).  One output pixel
// per image, so the contraction runs over the batch only:  dW[co][c][tap] = sum over n of dz[n][co] * x[n][tap][c]  - an
// outer-product stream that is bound by WRITING dW (75 MB at the discriminator's first Linear).  Pass 1 transposes x to
// the weight's own order, xT[n][c*taps + tap], in the workspace (N * K floats); pass 2 gives a lane one k = (c, tap) and
// a workgroup 16 output channels: N coalesced reads of xT, 16 * N FMAs in ascending n (deterministic), 16 coalesced
// stores.  The tiled kernel above spent 191 us on it (K = 16 per MFMA tile column); this is HBM-bound.
__global__ __launch_bounds__(256) void wgrad_fullmap_transpose_kernel(const float* __restrict__ x, float* __restrict__ xT,
                                                                      int N, int Cin, int taps, int in_cs) {
    const size_t K = (size_t)Cin * taps, total = (size_t)N * K;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t n = idx / K;
        const int k = (int)(idx - n * K);                 // read order: (tap, c), coalesced
        const int tap = k / Cin, c = k - tap * Cin;
        xT[n * K + (size_t)c * taps + tap] = x[(n * taps + tap) * in_cs + c];
    }
}

constexpr int FM_CO = 16, FM_NMAX = 256;
__global__ __launch_bounds__(256) void wgrad_fullmap_kernel(const float* __restrict__ xT, const float* __restrict__ dz,
                                                            float* __restrict__ dw, int N, int K, int Cout) {
    __shared__ float dzs[FM_NMAX][FM_CO];
    const int co0 = blockIdx.y * FM_CO;
    for (int idx = threadIdx.x; idx < N * FM_CO; idx += 256) {
        const int n = idx / FM_CO, o = idx % FM_CO;
        dzs[n][o] = co0 + o < Cout ? dz[(size_t)n * Cout + co0 + o] : 0.f;
    }
    __syncthreads();
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    float acc[FM_CO];
#pragma unroll
    for (int o = 0; o < FM_CO; ++o) acc[o] = 0.f;
#pragma unroll 4
    for (int n = 0; n < N; ++n) {
        const float xv = xT[(size_t)n * K + k];
#pragma unroll
        for (int o = 0; o < FM_CO; ++o) acc[o] = fmaf(dzs[n][o], xv, acc[o]);
    }
#pragma unroll
    for (int o = 0; o < FM_CO; ++o)
        if (co0 + o < Cout) dw[(size_t)(co0 + o) * K + k] = acc[o];
}

static bool wgrad_fullmap(const srmi_conv_desc& d, const ConvGeom& g) {
    if (dbg_flags() & (1 << 12)) return false;
    return d.pad == 0 && d.kh == d.H && d.kw == d.W && g.Ho == 1 && g.Wo == 1 && d.N >= 1 && d.N <= FM_NMAX &&
           (long long)d.kh * d.kw * d.Cin < 0x7fffffffLL && (d.Cout + FM_CO - 1) / FM_CO <= 65535 &&
           (long long)d.kh * d.kw * d.Cin >= 4096;        // small ones stay on the MFMA kernel (one launch)
}

// tiling / split plan of the tiled kernel, shared by the launch and the workspace query
struct Wgrad2Plan { int KW, WCI, co_tiles, ci_tiles, slices, cps; long long tiles, M; };

static bool wgrad2_plan(const srmi_conv_desc& d, const ConvGeom& g, bool allow_split, Wgrad2Plan* w) {
    if (dbg_flags() & (1 << 25)) return false;
    if (d.kw != 1 && d.kw != 3 && d.kw != 6 && d.kw != 9) return false;
    w->KW = d.kw;
    w->WCI = (d.Cin > 32 && d.kw <= 6) ? 2 : 1;               // CT = 64 (CO = 64) or CT = 32 (CO = 128)
    const int CO = (4 / w->WCI) * 32, CT = w->WCI * 32;
    w->co_tiles = (d.Cout + CO - 1) / CO;
    w->ci_tiles = (d.Cin + CT - 1) / CT;
    w->tiles = (long long)d.kh * w->co_tiles * w->ci_tiles;
    w->M = (long long)d.N * g.Ho * g.Wo;
    if (w->tiles > 0x7fffffffLL || w->M > 0x7fffffffLL || w->M == 0) return false;
    const long long chunks = (w->M + 31) / 32;
    long long slices = allow_split ? (512 + w->tiles - 1) / w->tiles : 1;
    if (slices > (chunks + 3) / 4) slices = (chunks + 3) / 4;   // at least four chunks (128 pixels) per slice
    if (slices < 1) slices = 1;
    if (slices > 65535) slices = 65535;
    w->cps = (int)((chunks + slices - 1) / slices);
    w->slices = (int)((chunks + w->cps - 1) / w->cps);
    return true;
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

// <= 4 channels on one side (conv_thin_f32.hip)
bool thinf_wgrad_ok(const srmi_conv_desc& d, const ConvGeom& g);
size_t thinf_wgrad_workspace_bytes(const srmi_conv_desc& d, const ConvGeom& g);
int thinf_wgrad_launch(const srmi_conv_desc& d, const ConvGeom& g, const float* x, const float* dz, float* dw, hipStream_t s,
                       void* ws);

size_t wgrad_workspace_bytes(const srmi_conv_desc& d, const ConvGeom& g) {
    if (wgrad_fullmap(d, g)) return (size_t)d.N * d.kh * d.kw * d.Cin * sizeof(float);
    if (thinf_wgrad_ok(d, g)) return thinf_wgrad_workspace_bytes(d, g);
    Wgrad2Plan w2;
    if (wgrad2_plan(d, g, true, &w2)) return w2.slices > 1 ? (size_t)w2.slices * w2.tiles * w2.KW * 4096 * sizeof(float) : 0;
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
    f32t_magic(g.Wo > 0 ? g.Wo : 1, &p.wo_mul, &p.wo_sh);
    f32t_magic(g.Ho > 0 ? g.Ho : 1, &p.ho_mul, &p.ho_sh);
    if (wgrad_fullmap(d, g) && ws && ws_bytes >= (size_t)d.N * d.kh * d.kw * d.Cin * sizeof(float)) {
        const int taps = d.kh * d.kw, K = taps * d.Cin;
        size_t blocks = ((size_t)d.N * K + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(wgrad_fullmap_transpose_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, (float*)ws, d.N, d.Cin,
                           taps, g.in_cs);
        int rcf = hip_check(hipGetLastError(), "wgrad_fullmap_transpose_kernel");
        if (rcf) return rcf;
        hipLaunchKernelGGL(wgrad_fullmap_kernel, dim3((unsigned)((K + 255) / 256), (unsigned)((d.Cout + FM_CO - 1) / FM_CO)),
                           dim3(256), 0, s, (const float*)ws, dz, dw, d.N, K, d.Cout);
        return hip_check(hipGetLastError(), "wgrad_fullmap_kernel");
    }
    if (thinf_wgrad_ok(d, g) && ws && ws_bytes >= thinf_wgrad_workspace_bytes(d, g) && !(((uintptr_t)x | (uintptr_t)dz | (uintptr_t)ws) & 15))
        return thinf_wgrad_launch(d, g, x, dz, dw, s, ws);
    Wgrad2Plan w2;
    if (wgrad2_plan(d, g, true, &w2)) {
        if (w2.slices > 1) {
            const size_t need = (size_t)w2.slices * w2.tiles * w2.KW * 4096 * sizeof(float);
            if (ws && ws_bytes >= need) p.partial = (float*)ws;
            else wgrad2_plan(d, g, false, &w2);              // no workspace: one workgroup per tile walks every pixel
        }
        p.rows = 0; p.rows_per_split = 0; p.co_tiles = w2.co_tiles; p.ci_tiles = w2.ci_tiles;
        dim3 grid((unsigned)w2.tiles, (unsigned)w2.slices);
        const int M = (int)w2.M;
#define SRMI_WG2(KWv, WCOv, WCIv) hipLaunchKernelGGL((conv_wgrad2_kernel<KWv, WCOv, WCIv>), grid, dim3(256), 0, s, p, w2.cps, M)
        if (w2.WCI == 2) {
            if (w2.KW == 1) SRMI_WG2(1, 2, 2); else if (w2.KW == 3) SRMI_WG2(3, 2, 2); else SRMI_WG2(6, 2, 2);
        } else {
            if (w2.KW == 1) SRMI_WG2(1, 4, 1); else if (w2.KW == 3) SRMI_WG2(3, 4, 1);
            else if (w2.KW == 6) SRMI_WG2(6, 4, 1); else SRMI_WG2(9, 4, 1);
        }
#undef SRMI_WG2
        int rc2 = hip_check(hipGetLastError(), "conv_wgrad2_kernel");
        if (rc2 || !p.partial) return rc2;
        const size_t total = (size_t)w2.tiles * w2.KW * 4096;
        size_t blocks = total / 32;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(wgrad2_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, w2.KW, w2.WCI, (int)w2.tiles, w2.slices);
        return hip_check(hipGetLastError(), "wgrad2_reduce_kernel");
    }
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
