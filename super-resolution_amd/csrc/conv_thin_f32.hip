// f32 convs with <= 4 channels on one side (training path: D / VGG first conv 3 -> 64, G last conv 64 -> 3, their dgrads and
// weight gradients; P/srgan_model.py:25,47, P/vgg19.py:11 under P/mode.py:148-179).  With 3 channels on the contracted or
// the produced side an implicit GEMM has nothing for one MFMA axis: the exact-f32 MFMA kernels spent 40-150 us per launch on
// 0.5 GFLOP (3-12 TF/s).  These layers are HBM-bound (37.7 MB of 64-channel f32 at 96 x 96 x 16: ~7 us at 5.4 TB/s), so they
// run on the vector ALU as plain fmaf chains, one pass over the wide tensor:
//   thinf_in_kernel    contracted <= 4 channels -> 64 produced: a lane owns 4 produced channels of a pixel (16 lanes per pixel:
//                      every store instruction writes whole 256-byte pixel lines), its 9 x C x 4 weights live in registers,
//                      the narrow pixel values are broadcast loads;
//   thinf_out_kernel   64 contracted -> <= 4 produced: the mirror image (a lane owns 4 contracted channels of a pixel, the 16
//                      lanes of a pixel add their sums in a fixed butterfly);
//   thinf_wgrad_kernel dW of either: acc[tap][narrow channel][4 wide channels] per lane, fixed pixel -> workgroup assignment,
//                      per-workgroup partials in the caller's workspace, added in workgroup order (no atomics: same bits on
//                      every run).
// Forward and dgrad share the kernels through GenericParams (dgrad = the transposed gather of the kernels in
// conv_generic.hip: contracted / produced roles swapped, tap offsets negated); the operand is conv_generic.hip's
// [tap][Cin_pad 8][Cout_pad 32] f32 - no new packed format.
#include "generic_common.h"

namespace srmi {

typedef float thin_f2 __attribute__((ext_vector_type(2)));      // packed pairs: v_pk_fma_f32 retires two FMAs per lane and issue slot

void f32t_magic(int d, unsigned* mul, int* sh);          // conv_f32_tiled.hip: multiply-shift constants of n / d

// geometry shared by the three kernels (all wave-uniform: scalar registers)
struct ThinGeom {
    int H, W, pad, sign, cs;          // source tensor H x W, tap offset sign (+1 forward / -1 transposed), its channel stride
    unsigned w_mul, h_mul; int w_sh, h_sh;
};

// pixel m -> (n, oy, ox), then for each of the K x K taps the element offset of its source pixel - clamped to the pixel's own
// position when it falls outside the image, so every load is unconditional - and a validity bit.  The offset of a tap is the
// pixel's plus a wave-uniform constant; rows and columns are tested once each (K + K compares, not 4 K^2).
template <int K>
struct ThinTaps {
    int off[K * K];
    unsigned ok;
    __device__ __forceinline__ void set(int m, const ThinGeom& g) {
        const int tq = (int)(((unsigned long long)(unsigned)m * g.w_mul) >> g.w_sh);      // m / W
        const int ox = m - tq * g.W;
        const int n = (int)(((unsigned long long)(unsigned)tq * g.h_mul) >> g.h_sh);      // tq / H
        const int oy = tq - n * g.H;
        const int base = m * g.cs;
        unsigned rok = 0u, cok = 0u;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int iy = oy + g.sign * (k - g.pad), ix = ox + g.sign * (k - g.pad);
            rok |= (iy >= 0 && iy < g.H) ? 1u << k : 0u;
            cok |= (ix >= 0 && ix < g.W) ? 1u << k : 0u;
        }
        ok = 0u;
#pragma unroll
        for (int t = 0; t < K * K; ++t) {
            const int ky = t / K, kx = t - ky * K;
            const bool v = ((rok >> ky) & (cok >> kx) & 1u) != 0u;
            ok |= v ? 1u << t : 0u;
            const int delta = g.sign * ((ky - g.pad) * g.W + (kx - g.pad)) * g.cs;         // wave-uniform
            off[t] = v ? base + delta : base;
        }
    }
};

// ---- contracted <= 4 -> 64 produced --------------------------------------------------------------------------------------
template <int C, int K>
__global__ __launch_bounds__(256) void thinf_in_kernel(GenericParams p) {
    constexpr int T = K * K;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int pg = lane >> 4, cg = lane & 15;            // pixel slot of the wave, 4-channel group
    f32x4 w[T][C];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int c = 0; c < C; ++c) w[t][c] = *(const f32x4*)(p.w + ((size_t)t * p.Cin_pad + c) * p.Cout_pad + 4 * cg);
    f32x4 sh = {0.f, 0.f, 0.f, 0.f};
    if (p.shift) sh = (f32x4){p.shift[4 * cg], p.shift[4 * cg + 1], p.shift[4 * cg + 2], p.shift[4 * cg + 3]};   // (4-byte aligned only)
    // slope of the negative side, as generic_epilogue_flat applies it (1: no activation)
    const float sl0 = p.act == SRMI_ACT_LRELU ? p.act_param : p.act == SRMI_ACT_RELU ? 0.f : 1.f;
    f32x4 sl = {sl0, sl0, sl0, sl0};
    if (p.act == SRMI_ACT_PRELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sl[j] = p.alpha[p.alpha_pc ? 4 * cg + j : 0];
    }
    const float* x = (const float*)p.x;
    float* y = (float*)p.y;
    const ThinGeom geo = {p.H, p.W, p.pad, p.transposed ? -1 : 1, p.in_cs, p.wo_mul, p.ho_mul, p.wo_sh, p.ho_sh};
    // software-pipelined: the loads of pixel m + stride (unconditional: clamped addresses) are in flight under the chain of m
    const int step = gridDim.x * 16;
    int m = (blockIdx.x * 4 + wave) * 4 + pg;
    ThinTaps<K> tp;
    float v[T][C];
    tp.set(min(m, p.M - 1), geo);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int c = 0; c < C; ++c) v[t][c] = x[tp.off[t] + c];
    for (; m < p.M; m += step) {
        ThinTaps<K> tpn;
        float vn[T][C];
        tpn.set(min(m + step, p.M - 1), geo);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int c = 0; c < C; ++c) vn[t][c] = x[tpn.off[t] + c];
        thin_f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float q = (tp.ok >> t) & 1u ? v[t][c] : 0.f;
                const thin_f2 qq = {q, q};
                a01 = __builtin_elementwise_fma(qq, (thin_f2){w[t][c][0], w[t][c][1]}, a01);
                a23 = __builtin_elementwise_fma(qq, (thin_f2){w[t][c][2], w[t][c][3]}, a23);
            }
        f32x4 acc = (f32x4){a01[0], a01[1], a23[0], a23[1]} + sh;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = acc[j] >= 0.f ? acc[j] : acc[j] * sl[j];
        *(f32x4*)(y + (size_t)m * p.out_cs + p.out_co + 4 * cg) = acc;
        tp = tpn;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int c = 0; c < C; ++c) v[t][c] = vn[t][c];
    }
}

// ---- 64 contracted -> <= 4 produced: the mirror image - a lane owns 4 contracted channels of a pixel (coalesced 256-byte
// pixel reads), the 16 lanes of a pixel add their partial sums in a fixed butterfly ----
template <int P, int K>
__global__ __launch_bounds__(256) void thinf_out_kernel(GenericParams p) {
    constexpr int T = K * K;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int pg = lane >> 4, cg = lane & 15;
    float w[T][4][P];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int o = 0; o < P; ++o) w[t][j][o] = p.w[((size_t)t * p.Cin_pad + 4 * cg + j) * p.Cout_pad + o];
    const float* x = (const float*)p.x;
    float* y = (float*)p.y;
    const ThinGeom geo = {p.H, p.W, p.pad, p.transposed ? -1 : 1, p.in_cs, p.wo_mul, p.ho_mul, p.wo_sh, p.ho_sh};
    const int step = gridDim.x * 16;
    ThinTaps<K> tp;
    f32x4 v[T];
    tp.set(min((int)(blockIdx.x * 4 + wave) * 4 + pg, p.M - 1), geo);
#pragma unroll
    for (int t = 0; t < T; ++t) v[t] = *(const f32x4*)(x + tp.off[t] + 4 * cg);
    for (int m0 = (blockIdx.x * 4 + wave) * 4; m0 < p.M; m0 += step) {       // (wave-uniform trip count: the butterfly below)
        const int m = min(m0 + pg, p.M - 1);
        ThinTaps<K> tpn;
        f32x4 vn[T];
        tpn.set(min(m0 + step + pg, p.M - 1), geo);       // next pixel's loads under this one's chain
#pragma unroll
        for (int t = 0; t < T; ++t) vn[t] = *(const f32x4*)(x + tpn.off[t] + 4 * cg);
        float acc[P];
        thin_f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};          // produced channels in packed pairs (v_pk_fma_f32)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const bool ok = (tp.ok >> t) & 1u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float q = ok ? v[t][j] : 0.f;
                const thin_f2 qq = {q, q};
                if constexpr (P >= 2) a01 = __builtin_elementwise_fma(qq, (thin_f2){w[t][j][0], w[t][j][1]}, a01);
                else a01[0] = fmaf(q, w[t][j][0], a01[0]);
                if constexpr (P == 4) a23 = __builtin_elementwise_fma(qq, (thin_f2){w[t][j][2], w[t][j][3]}, a23);
                else if constexpr (P == 3) a23[0] = fmaf(q, w[t][j][2], a23[0]);
            }
        }
        acc[0] = a01[0];
        if constexpr (P >= 2) acc[1] = a01[1];
        if constexpr (P >= 3) acc[2] = a23[0];
        if constexpr (P == 4) acc[3] = a23[1];
#pragma unroll
        for (int o = 0; o < P; ++o) {
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) acc[o] += __shfl_xor(acc[o], d);      // every lane of the pixel ends with the same sum
        }
        if (m0 + pg < p.M && cg < P) {
            float r = acc[0];
#pragma unroll
            for (int o = 1; o < P; ++o) r = cg == o ? acc[o] : r;
            y[(size_t)m * p.out_cs + p.out_co + cg] = r + (p.shift ? p.shift[cg] : 0.f);
        }
        tp = tpn;
#pragma unroll
        for (int t = 0; t < T; ++t) v[t] = vn[t];
    }
}

static bool thinf_linear(const GenericParams& p) {
    return !p.scale && !p.res && p.r == 1 && !p.in_nchw && !p.out_nchw && !p.in_bf16 && !p.out_bf16 &&
           p.stride == 1 && p.kh == p.kw && (p.kh == 1 || p.kh == 3) && 2 * p.pad == p.kh - 1 && p.Ho == p.H && p.Wo == p.W && !p.b3 &&
           (long long)p.N * p.H * p.W * (p.in_cs > p.out_cs ? p.in_cs : p.out_cs) < 0x7fffffffLL;       // 32-bit element offsets
}

// 1: thinf_in_kernel, 2: thinf_out_kernel, 0: neither
int thinf_kind(const GenericParams& p) {
    if (dbg_flags() & (1 << 11)) return 0;
    if (!thinf_linear(p)) return 0;
    if (((uintptr_t)p.x | (uintptr_t)p.y | (uintptr_t)p.w) & 15) return 0;
    const bool slope = p.act == SRMI_ACT_RELU || p.act == SRMI_ACT_LRELU || p.act == SRMI_ACT_PRELU;   // (PReLU: the API checks alpha)
    if (p.Cin >= 1 && p.Cin <= 4 && p.Cout == 64 && p.Cout_pad == 64 && p.out_cs % 4 == 0 && p.out_co % 4 == 0 &&
        (p.act == SRMI_ACT_NONE || slope))
        return 1;
    if (p.Cout >= 1 && p.Cout <= 4 && p.Cin == 64 && p.in_cs % 4 == 0 && p.act == SRMI_ACT_NONE) return 2;
    return 0;
}

int thinf_launch(const GenericParams& p0) {
    GenericParams p = p0;
    f32t_magic(p.W > 0 ? p.W : 1, &p.wo_mul, &p.wo_sh);       // (source tensor = output size: stride 1, same padding)
    f32t_magic(p.H > 0 ? p.H : 1, &p.ho_mul, &p.ho_sh);
    hipStream_t s = p.stream;
    long long nb = ((long long)p.M + 15) / 16;
    if (nb > 2048) nb = 2048;
    const dim3 grid((unsigned)nb);
    const bool k3 = p.kh == 3;
#define SRMI_THIN(kern, n) do { if (k3) hipLaunchKernelGGL((kern<n, 3>), grid, dim3(256), 0, s, p); else hipLaunchKernelGGL((kern<n, 1>), grid, dim3(256), 0, s, p); } while (0)
    if (thinf_kind(p) == 1) {
        switch (p.Cin) {
        case 1: SRMI_THIN(thinf_in_kernel, 1); break;
        case 2: SRMI_THIN(thinf_in_kernel, 2); break;
        case 3: SRMI_THIN(thinf_in_kernel, 3); break;
        default: SRMI_THIN(thinf_in_kernel, 4); break;
        }
        return hip_check(hipGetLastError(), "thinf_in_kernel");
    }
    switch (p.Cout) {
    case 1: SRMI_THIN(thinf_out_kernel, 1); break;
    case 2: SRMI_THIN(thinf_out_kernel, 2); break;
    case 3: SRMI_THIN(thinf_out_kernel, 3); break;
    default: SRMI_THIN(thinf_out_kernel, 4); break;
    }
#undef SRMI_THIN
    return hip_check(hipGetLastError(), "thinf_out_kernel");
}

// ---- weight gradient -----------------------------------------------------------------------------------------------------
struct ThinWgradParams {
    const float* wide; const float* narrow;   // [M][64] (cs wide_cs) and [M][S] (cs narrow_cs), both over N x H x W pixels
    float* partial;                           // [workgroups][taps][S][64]
    int N, H, W, kh, kw, pad, wide_cs, narrow_cs, M;
    int sign;                                 // narrow pixel of (wide pixel m, tap) = m + sign * (tap offset - pad)
    unsigned w_mul, h_mul; int w_sh, h_sh;    // multiply-shift division by W and H
};

template <int S, int K>
__global__ __launch_bounds__(256) void thinf_wgrad_kernel(ThinWgradParams p) {
    constexpr int T = K * K;
    __shared__ f32x4 red[4][T * S][16];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int pg = lane >> 4, cg = lane & 15;
    f32x4 acc[T][S];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int c = 0; c < S; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const ThinGeom geo = {p.H, p.W, p.pad, p.sign, p.narrow_cs, p.w_mul, p.h_mul, p.w_sh, p.h_sh};
    const int step = gridDim.x * 16;
    int m = (blockIdx.x * 4 + wave) * 4 + pg;
    ThinTaps<K> tp;
    float v[T][S];
    tp.set(min(m, p.M - 1), geo);
    f32x4 wv = *(const f32x4*)(p.wide + (size_t)min(m, p.M - 1) * p.wide_cs + 4 * cg);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int c = 0; c < S; ++c) v[t][c] = p.narrow[tp.off[t] + c];
    for (; m < p.M; m += step) {
        ThinTaps<K> tpn;
        float vn[T][S];
        const int mn = min(m + step, p.M - 1);           // next pixel's loads under this one's chain
        tpn.set(mn, geo);
        const f32x4 wvn = *(const f32x4*)(p.wide + (size_t)mn * p.wide_cs + 4 * cg);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int c = 0; c < S; ++c) vn[t][c] = p.narrow[tpn.off[t] + c];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int c = 0; c < S; ++c) {
                const float q = (tp.ok >> t) & 1u ? v[t][c] : 0.f;
                const thin_f2 qq = {q, q};
                const thin_f2 lo = __builtin_elementwise_fma(qq, (thin_f2){wv[0], wv[1]}, (thin_f2){acc[t][c][0], acc[t][c][1]});
                const thin_f2 hi = __builtin_elementwise_fma(qq, (thin_f2){wv[2], wv[3]}, (thin_f2){acc[t][c][2], acc[t][c][3]});
                acc[t][c] = (f32x4){lo[0], lo[1], hi[0], hi[1]};
            }
        tp = tpn; wv = wvn;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int c = 0; c < S; ++c) v[t][c] = vn[t][c];
    }
    // the wave's four pixel slots (lanes cg, cg + 16, cg + 32, cg + 48) pairwise, then the four waves in wave order
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int c = 0; c < S; ++c) {
            f32x4 v = acc[t][c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {          // (slot 0 + slot 1) + (slot 2 + slot 3): the same association in every lane
                v[e] += __shfl_xor(v[e], 16);
                v[e] += __shfl_xor(v[e], 32);
            }
            if (pg == 0) red[wave][t * S + c][cg] = v;
        }
    __syncthreads();
    for (int i = tid; i < T * S * 16; i += 256) {
        const int r = i >> 4, g = i & 15;
        const f32x4 v = ((red[0][r][g] + red[1][r][g]) + red[2][r][g]) + red[3][r][g];
        *(f32x4*)(p.partial + ((size_t)blockIdx.x * T * S + r) * 64 + 4 * g) = v;
    }
}

// dW (OIHW) = sum of the workgroups' partials in workgroup order.  narrow_is_in: the narrow tensor is the conv's INPUT
// (Cin = S, Cout = 64: dW[co][ci][tap] = partial[tap][ci][co]); otherwise its output gradient (Cout = S, Cin = 64, taps
// mirrored: the wide pixel was the input pixel, the narrow one the output pixel it contributes to).
__global__ __launch_bounds__(256) void thinf_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int nwg,
                                                                 int taps, int S, int narrow_is_in) {
    // a workgroup owns 64 consecutive elements; its four quarters each add a quarter of the workgroups' partials in
    // workgroup order (16 loads in flight), then the quarters are added in quarter order
    __shared__ float q[4][64];
    const int e = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + e;
    const int total = taps * S * 64;
    const int per = (nwg + 3) / 4;
    const int g0 = part * per, g1 = min(nwg, g0 + per);
    float v = 0.f;
    if (i < total) {
        int g = g0;
        for (; g + 16 <= g1; g += 16) {
            float t[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) t[k] = partial[(size_t)(g + k) * total + i];
#pragma unroll
            for (int k = 0; k < 16; ++k) v += t[k];
        }
        for (; g < g1; ++g) v += partial[(size_t)g * total + i];
    }
    q[part][e] = v;
    __syncthreads();
    if (part == 0 && i < total) {
        v = ((q[0][e] + q[1][e]) + q[2][e]) + q[3][e];
        const int wc = i & 63, r = i >> 6;
        const int t = r / S, sc = r - t * S;
        if (narrow_is_in) dw[((size_t)wc * S + sc) * taps + t] = v;
        else dw[((size_t)sc * 64 + wc) * taps + t] = v;
    }
}

static int thinf_wgrad_groups(long long M) {
    long long g = (M + 15) / 16;
    if (g > 256) g = 256;
    return (int)(g < 1 ? 1 : g);
}

bool thinf_wgrad_ok(const srmi_conv_desc& d, const ConvGeom& g) {
    if (dbg_flags() & (1 << 11)) return false;
    if (d.stride != 1 || (d.kh != 1 && d.kh != 3) || g.Ho != d.H || g.Wo != d.W || d.kh != d.kw || 2 * d.pad != d.kh - 1) return false;
    if ((long long)d.N * d.H * d.W * (g.in_cs > 64 ? g.in_cs : 64) >= 0x7fffffffLL || (long long)d.N * d.H * d.W == 0) return false;    // 32-bit element offsets
    const bool in_thin = d.Cin >= 1 && d.Cin <= 4 && d.Cout == 64;
    const bool out_thin = d.Cout >= 1 && d.Cout <= 4 && d.Cin == 64 && g.in_cs % 4 == 0;
    return in_thin || out_thin;
}

size_t thinf_wgrad_workspace_bytes(const srmi_conv_desc& d, const ConvGeom& g) {
    const int S = d.Cin <= 4 ? d.Cin : d.Cout;
    return (size_t)thinf_wgrad_groups((long long)d.N * d.H * d.W) * d.kh * d.kw * S * 64 * sizeof(float);
}

int thinf_wgrad_launch(const srmi_conv_desc& d, const ConvGeom& g, const float* x, const float* dz, float* dw, hipStream_t s,
                       void* ws) {
    const bool in_thin = d.Cin <= 4;
    ThinWgradParams p;
    p.N = d.N; p.H = d.H; p.W = d.W; p.kh = d.kh; p.kw = d.kw; p.pad = d.pad; p.M = d.N * d.H * d.W;
    p.partial = (float*)ws;
    f32t_magic(d.W, &p.w_mul, &p.w_sh);
    f32t_magic(d.H, &p.h_mul, &p.h_sh);
    int S;
    if (in_thin) {       // dW[co][ci][tap] = sum dz[m][co] x[m + tap - pad][ci]
        p.wide = dz; p.wide_cs = d.Cout; p.narrow = x; p.narrow_cs = g.in_cs; p.sign = 1; S = d.Cin;
    } else {             // dW[co][ci][tap] = sum x[m'][ci] dz[m' - (tap - pad)][co]
        p.wide = x; p.wide_cs = g.in_cs; p.narrow = dz; p.narrow_cs = d.Cout; p.sign = -1; S = d.Cout;
    }
    if (((uintptr_t)p.wide | (uintptr_t)ws) & 15) { set_error("wgrad (thin): buffers must be 16-byte aligned"); return SRMI_EINVAL; }
    const int nwg = thinf_wgrad_groups(p.M);
    const bool k3 = d.kh == 3;
#define SRMI_THINW(n) do { if (k3) hipLaunchKernelGGL((thinf_wgrad_kernel<n, 3>), dim3(nwg), dim3(256), 0, s, p); else hipLaunchKernelGGL((thinf_wgrad_kernel<n, 1>), dim3(nwg), dim3(256), 0, s, p); } while (0)
    switch (S) {
    case 1: SRMI_THINW(1); break;
    case 2: SRMI_THINW(2); break;
    case 3: SRMI_THINW(3); break;
    default: SRMI_THINW(4); break;
    }
#undef SRMI_THINW
    int rc = hip_check(hipGetLastError(), "thinf_wgrad_kernel");
    if (rc) return rc;
    const int total = d.kh * d.kw * S * 64;
    hipLaunchKernelGGL(thinf_wgrad_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, s, (const float*)ws, dw, nwg, d.kh * d.kw,
                       S, in_thin ? 1 : 0);
    return hip_check(hipGetLastError(), "thinf_wgrad_reduce_kernel");
}

}  // namespace srmi
