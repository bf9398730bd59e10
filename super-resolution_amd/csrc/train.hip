// Training-step pointwise / per-channel kernels (HBM-bound): train-mode BatchNorm statistics,
// the conv epilogue (normalise -> activation -> sub-pixel store -> residual) forward and backward,
// 2x2 max-pool forward/backward, MSE/BCE loss gradients and the Adam update.
//
// Tensors are dense NHWC f32.  Per-channel sums use lanes along the channel axis (coalesced rows), four waves along
// pixels and an LDS tree across the waves; every workgroup then STORES its f64 partial sums in its own slot of the
// caller's workspace and a finalize launch adds the slots in index order.  No atomics: the same input gives the
// same bits on every run (tests/test_srgan_step_gpu.py::test_srgan_step_is_bit_reproducible).
#include <initializer_list>

#include "common.h"

namespace srmi {

constexpr int CH_BLOCK = 256;
constexpr int TRAIN_MAX_SLOTS = 512;      // most partial slots per channel in the workspace (srmi_train_workspace_bytes)

struct EpiParams {
    const float* z; const float* dy; const float* mean; const float* rstd; const float* gamma; const float* beta;
    const float* alpha; const float* residual;
    float* y; float* dz;
    double* acc;  // partial slots [gridDim.x][3][C] (S1, S2, per-conv-channel dalpha terms), then the totals [3][C]
    int nslots;   // gridDim.x of the reducing launch
    long long M;  // pixels N*H*W
    int H, W, C, Cf, r, dcr, act, alpha_pc, bn;
    float act_param;
};

// conv-order element (pixel m, channel c) -> (index into the final tensor, final channel)
// (m < 2^31 is checked on the host: 32-bit divisions, the 64-bit ones cost ~20x more)
__device__ __forceinline__ size_t final_index(const EpiParams& p, int m, int c, int* cf_out) {
    if (p.r == 1) { *cf_out = c; return (size_t)m * p.C + c; }
    const int rr = p.r * p.r;
    int cf, sidx;
    if (p.dcr) { cf = c % p.Cf; sidx = c / p.Cf; } else { cf = c / rr; sidx = c % rr; }
    const int si = sidx / p.r, sj = sidx % p.r;
    const int ox = m % p.W;
    const int t = m / p.W;
    const int oy = t % p.H;
    const int n = t / p.H;
    *cf_out = cf;
    return (((size_t)n * p.H * p.r + (size_t)oy * p.r + si) * ((size_t)p.W * p.r) + (size_t)ox * p.r + sj) * p.Cf + cf;
}

// four consecutive per-channel constants (4-byte aligned only: parameters may be views into a flat optimizer buffer)
__device__ __forceinline__ f32x4 ld4(const float* q) { return f32x4{q[0], q[1], q[2], q[3]}; }

__device__ __forceinline__ float slope_of(const EpiParams& p, int cf) {
    if (p.act == SRMI_ACT_PRELU) return p.alpha[p.alpha_pc ? cf : 0];
    if (p.act == SRMI_ACT_LRELU) return p.act_param;
    return 0.f;  // ReLU
}

// ---- per-channel sum / sum of squares (BatchNorm batch statistics) --------------------------------
__global__ __launch_bounds__(CH_BLOCK) void bn_stats_kernel(const float* __restrict__ z, long long M, int C,
                                                            double* __restrict__ acc) {
    __shared__ double l1[CH_BLOCK], l2[CH_BLOCK];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + tx;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (long long m = (long long)blockIdx.x * 4 + ty; m < M; m += (long long)gridDim.x * 4) {
            const double v = (double)z[(size_t)m * C + c];
            s1 += v; s2 += v * v;
        }
    l1[threadIdx.x] = s1; l2[threadIdx.x] = s2;
    __syncthreads();
    if (ty == 0 && c < C) {
        s1 = l1[tx] + l1[tx + 64] + l1[tx + 128] + l1[tx + 192];
        s2 = l2[tx] + l2[tx + 64] + l2[tx + 128] + l2[tx + 192];
        acc[((size_t)blockIdx.x * 2) * C + c] = s1;
        acc[((size_t)blockIdx.x * 2 + 1) * C + c] = s2;
    }
}

// The same sums for C % 4 == 0 (every layer of the SRGAN step): a lane owns FOUR channels (one 16-byte load per pixel),
// 16 lanes cover a 64-channel group, 16 pixel rows per workgroup pass; the 16 rows meet in LDS and are added in row
// order.  A wave instruction reads 4 whole pixels of 256 bytes instead of one 4-byte element per lane.
__device__ __forceinline__ void rows16_reduce(double (&t)[4], double (*sh)[64], int prow, int l16) {
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[prow][4 * l16 + j] = t[j];
    __syncthreads();
}

__global__ __launch_bounds__(CH_BLOCK) void bn_stats_v4_kernel(const float* __restrict__ z, long long M, int C,
                                                               double* __restrict__ acc) {
    __shared__ double l1[16][64], l2[16][64];
    const int l16 = threadIdx.x & 15, prow = threadIdx.x >> 4;
    const int c0 = blockIdx.y * 64 + 4 * l16;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    if (c0 < C) {
#pragma unroll 4
        for (long long m = (long long)blockIdx.x * 16 + prow; m < M; m += (long long)gridDim.x * 16) {
            const f32x4 v = *(const f32x4*)(z + (size_t)m * C + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const double d = (double)v[j]; s1[j] += d; s2[j] += d * d; }
        }
    }
    rows16_reduce(s1, l1, prow, l16);
    rows16_reduce(s2, l2, prow, l16);
    const int c = blockIdx.y * 64 + threadIdx.x;
    if (threadIdx.x < 64 && c < C) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) { a += l1[r][threadIdx.x]; b += l2[r][threadIdx.x]; }
        acc[((size_t)blockIdx.x * 2) * C + c] = a;
        acc[((size_t)blockIdx.x * 2 + 1) * C + c] = b;
    }
}

// sum of the partial slots of channel c, array k of KA: 32 threads per channel take the slots g, g + 32, ... in order
// and the 32 sub-sums are added in g order (a fixed association: deterministic).  Block = 8 channels x 32 groups.
constexpr int FIN_CH = 8, FIN_G = 32;
template <int KA>
__device__ __forceinline__ void slot_sums(const double* __restrict__ acc, int nslots, int C, int c, int g, double (&t)[KA],
                                          double (*sh)[FIN_G][FIN_CH]) {
#pragma unroll
    for (int k = 0; k < KA; ++k) t[k] = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int b = g; b < nslots; b += FIN_G)
#pragma unroll
            for (int k = 0; k < KA; ++k) t[k] += acc[((size_t)b * KA + k) * C + c];
    }
    const int cl = threadIdx.x % FIN_CH;
#pragma unroll
    for (int k = 0; k < KA; ++k) sh[k][g][cl] = t[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KA; ++k) {
        double a = 0.0;
#pragma unroll
        for (int r = 0; r < FIN_G; ++r) a += sh[k][r][cl];
        t[k] = a;
    }
}

// mean, 1/sqrt(var+eps) and the running-statistics update of nn.BatchNorm2d (momentum, unbiased running var)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ acc, int nslots, long long M, int C,
                                                          float eps, float momentum, float* __restrict__ mean,
                                                          float* __restrict__ rstd, float* __restrict__ rmean,
                                                          float* __restrict__ rvar, long long* __restrict__ nbt, int updates) {
    __shared__ double sh[2][FIN_G][FIN_CH];
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += updates;  // nn.BatchNorm2d: num_batches_tracked.add_(1), per forward
    const int c = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH, g = threadIdx.x / FIN_CH;
    double t[2];
    slot_sums<2>(acc, nslots, C, c, g, t, sh);
    if (c >= C || g != 0) return;
    const double t1 = t[0], t2 = t[1];
    const double mu = t1 / (double)M;
    double var = t2 / (double)M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    // `updates` forwards over the same batch (srmi_bn_train_stats_n): the running statistics move once per forward, each
    // time rounded to f32 as a forward of its own would leave them
    if (rmean) {
        float r = rmean[c];
        for (int u = 0; u < updates; ++u) r = (float)((1.0 - (double)momentum) * (double)r + (double)momentum * mu);
        rmean[c] = r;
    }
    if (rvar) {
        const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
        float r = rvar[c];
        for (int u = 0; u < updates; ++u) r = (float)((1.0 - (double)momentum) * (double)r + (double)momentum * unbiased);
        rvar[c] = r;
    }
}

// ---- epilogue forward: y = store(act(bn(z))) + residual -------------------------------------------
__global__ __launch_bounds__(CH_BLOCK) void epilogue_fwd_kernel(EpiParams p) {
    const size_t total = (size_t)p.M * p.C;
    for (size_t idx = (size_t)blockIdx.x * CH_BLOCK + threadIdx.x; idx < total; idx += (size_t)gridDim.x * CH_BLOCK) {
        const int m = (int)(idx / (unsigned)p.C);
        const int c = (int)(idx - (size_t)m * p.C);
        float v = p.z[idx];
        if (p.bn) v = (v - p.mean[c]) * p.rstd[c] * p.gamma[c] + p.beta[c];
        int cf;
        const size_t o = final_index(p, m, c, &cf);
        const float al = (p.act == SRMI_ACT_PRELU) ? p.alpha[p.alpha_pc ? cf : 0] : 0.f;
        v = apply_act(v, p.act, p.act_param, al);
        if (p.residual) v += p.residual[o];
        p.y[o] = v;
    }
}

// r == 1 and C / 4 dividing the workgroup size: a lane keeps the same four channels for the whole grid-stride loop (its
// per-channel constants are loaded once, no index division) and moves 16 bytes per access.  Same arithmetic, same bits.
__global__ __launch_bounds__(CH_BLOCK) void epilogue_fwd_v4_kernel(EpiParams p) {
    const size_t total4 = (size_t)p.M * p.C / 4;
    const int c0 = 4 * (int)(threadIdx.x % (unsigned)(p.C / 4));
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = mu, ga = mu, be = mu, al = mu;
    if (p.bn) { mu = ld4(p.mean + c0); rs = ld4(p.rstd + c0); ga = ld4(p.gamma + c0); be = ld4(p.beta + c0); }
    if (p.act == SRMI_ACT_PRELU) {
        if (p.alpha_pc) al = ld4(p.alpha + c0);
        else { const float a0 = p.alpha[0]; al = f32x4{a0, a0, a0, a0}; }
    }
    for (size_t i = (size_t)blockIdx.x * CH_BLOCK + threadIdx.x; i < total4; i += (size_t)gridDim.x * CH_BLOCK) {
        f32x4 v = *(const f32x4*)(p.z + 4 * i);
        f32x4 rsd = {0.f, 0.f, 0.f, 0.f};
        if (p.residual) rsd = *(const f32x4*)(p.residual + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float u = v[j];
            if (p.bn) u = (u - mu[j]) * rs[j] * ga[j] + be[j];
            u = apply_act(u, p.act, p.act_param, al[j]);
            if (p.residual) u += rsd[j];
            v[j] = u;
        }
        *(f32x4*)(p.y + 4 * i) = v;
    }
}

// ---- epilogue backward, pass 1: g = dy * act'(u) -> dz, per-channel S1 = sum g, S2 = sum g*zhat, dalpha --
__global__ __launch_bounds__(CH_BLOCK) void epilogue_bwd_kernel(EpiParams p) {
    __shared__ double l1[CH_BLOCK], l2[CH_BLOCK], l3[CH_BLOCK];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + tx;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < p.C) {
        float mu = 0.f, rs = 1.f, ga = 1.f, be = 0.f;
        if (p.bn) { mu = p.mean[c]; rs = p.rstd[c]; ga = p.gamma[c]; be = p.beta[c]; }
        const int M = (int)p.M;
        for (int m = blockIdx.x * 4 + ty; m < M; m += gridDim.x * 4) {
            const size_t idx = (size_t)m * p.C + c;
            const float zh = (p.z[idx] - mu) * rs;        // = z when !bn
            const float u = p.bn ? zh * ga + be : zh;
            int cf;
            const size_t o = final_index(p, m, c, &cf);
            const float g0 = p.dy[o];
            float g;
            switch (p.act) {
                case SRMI_ACT_PRELU: case SRMI_ACT_LRELU: case SRMI_ACT_RELU: {
                    const float sl = slope_of(p, cf);
                    g = u > 0.f ? g0 : g0 * sl;
                    if (p.act == SRMI_ACT_PRELU && !(u > 0.f)) s3 += (double)g0 * (double)u;
                    break;
                }
                case SRMI_ACT_TANH: { const float t = tanhf(u); g = g0 * (1.f - t * t); break; }
                case SRMI_ACT_SIGMOID: { const float sg = 1.f / (1.f + expf(-u)); g = g0 * sg * (1.f - sg); break; }
                default: g = g0;
            }
            p.dz[idx] = g;
            s1 += (double)g;
            s2 += (double)g * (double)zh;
        }
    }
    l1[threadIdx.x] = s1; l2[threadIdx.x] = s2; l3[threadIdx.x] = s3;
    __syncthreads();
    if (ty == 0 && c < p.C) {
        double* slot = p.acc + (size_t)blockIdx.x * 3 * p.C;
        slot[c] = l1[tx] + l1[tx + 64] + l1[tx + 128] + l1[tx + 192];
        slot[p.C + c] = l2[tx] + l2[tx + 64] + l2[tx + 128] + l2[tx + 192];
        slot[2 * p.C + c] = l3[tx] + l3[tx + 64] + l3[tx + 128] + l3[tx + 192];    // dalpha terms of CONV channel c
    }
}

// pass 1 for r == 1, C % 4 == 0: four channels per lane, 16 pixel rows per workgroup pass (see bn_stats_v4_kernel)
__global__ __launch_bounds__(CH_BLOCK) void epilogue_bwd_v4_kernel(EpiParams p) {
    __shared__ double l1[16][64], l2[16][64], l3[16][64];
    const int l16 = threadIdx.x & 15, prow = threadIdx.x >> 4;
    const int c0 = blockIdx.y * 64 + 4 * l16;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0}, s3[4] = {0.0, 0.0, 0.0, 0.0};
    if (c0 < p.C) {
        f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = {1.f, 1.f, 1.f, 1.f}, ga = rs, be = mu, sl = mu;
        if (p.bn) { mu = ld4(p.mean + c0); rs = ld4(p.rstd + c0); ga = ld4(p.gamma + c0); be = ld4(p.beta + c0); }
        if (p.act == SRMI_ACT_PRELU || p.act == SRMI_ACT_LRELU || p.act == SRMI_ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sl[j] = slope_of(p, c0 + j);
        }
        const int M = (int)p.M;
#pragma unroll 2
        for (int m = blockIdx.x * 16 + prow; m < M; m += gridDim.x * 16) {
            const size_t idx = (size_t)m * p.C + c0;
            const f32x4 zv = *(const f32x4*)(p.z + idx);
            const f32x4 gv = *(const f32x4*)(p.dy + idx);
            f32x4 out;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float zh = (zv[j] - mu[j]) * rs[j];        // = z when !bn
                const float u = p.bn ? zh * ga[j] + be[j] : zh;
                const float g0 = gv[j];
                float g;
                switch (p.act) {
                    case SRMI_ACT_PRELU: case SRMI_ACT_LRELU: case SRMI_ACT_RELU: {
                        g = u > 0.f ? g0 : g0 * sl[j];
                        if (p.act == SRMI_ACT_PRELU && !(u > 0.f)) s3[j] += (double)g0 * (double)u;
                        break;
                    }
                    case SRMI_ACT_TANH: { const float t = tanhf(u); g = g0 * (1.f - t * t); break; }
                    case SRMI_ACT_SIGMOID: { const float sg = 1.f / (1.f + expf(-u)); g = g0 * sg * (1.f - sg); break; }
                    default: g = g0;
                }
                out[j] = g;
                s1[j] += (double)g;
                s2[j] += (double)g * (double)zh;
            }
            *(f32x4*)(p.dz + idx) = out;
        }
    }
    rows16_reduce(s1, l1, prow, l16);
    rows16_reduce(s2, l2, prow, l16);
    rows16_reduce(s3, l3, prow, l16);
    const int c = blockIdx.y * 64 + threadIdx.x;
    if (threadIdx.x < 64 && c < p.C) {
        double a = 0.0, b = 0.0, d = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) { a += l1[r][threadIdx.x]; b += l2[r][threadIdx.x]; d += l3[r][threadIdx.x]; }
        double* slot = p.acc + (size_t)blockIdx.x * 3 * p.C;
        slot[c] = a; slot[p.C + c] = b; slot[2 * p.C + c] = d;
    }
}

// pass 1 when NO per-channel sum is wanted (no BN, no bias / slope gradient: the frozen VGG19 stack, the discriminator's
// plain layers in the generator step): dz = dy * act'(z), a pure stream with the arithmetic of epilogue_bwd_v4_kernel
__global__ __launch_bounds__(CH_BLOCK) void act_bwd_v4_kernel(EpiParams p) {
    const size_t total4 = (size_t)p.M * p.C / 4;
    const int c0 = 4 * (int)(threadIdx.x % (unsigned)(p.C / 4));
    f32x4 sl = {0.f, 0.f, 0.f, 0.f};
    if (p.act == SRMI_ACT_PRELU || p.act == SRMI_ACT_LRELU || p.act == SRMI_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sl[j] = slope_of(p, c0 + j);
    }
    for (size_t i = (size_t)blockIdx.x * CH_BLOCK + threadIdx.x; i < total4; i += (size_t)gridDim.x * CH_BLOCK) {
        const f32x4 zv = *(const f32x4*)(p.z + 4 * i);
        f32x4 g = *(const f32x4*)(p.dy + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float u = zv[j], g0 = g[j];
            switch (p.act) {
                case SRMI_ACT_PRELU: case SRMI_ACT_LRELU: case SRMI_ACT_RELU: g[j] = u > 0.f ? g0 : g0 * sl[j]; break;
                case SRMI_ACT_TANH: { const float t = tanhf(u); g[j] = g0 * (1.f - t * t); break; }
                case SRMI_ACT_SIGMOID: { const float sg = 1.f / (1.f + expf(-u)); g[j] = g0 * sg * (1.f - sg); break; }
                default: break;
            }
        }
        *(f32x4*)(p.dz + 4 * i) = g;
    }
}

// pass 2 (BN only): dz = gamma * rstd * (g - S1/M - zhat * S2/M)
__global__ __launch_bounds__(CH_BLOCK) void bn_bwd_apply_v4_kernel(EpiParams p) {
    const size_t total4 = (size_t)p.M * p.C / 4;
    const double invM = 1.0 / (double)p.M;
    const int c0 = 4 * (int)(threadIdx.x % (unsigned)(p.C / 4));
    const double* tot = p.acc + (size_t)p.nslots * 3 * p.C;           // totals, written by the finalize launch
    const f32x4 mu = ld4(p.mean + c0), rs = ld4(p.rstd + c0), ga = ld4(p.gamma + c0);
    f32x4 m1, m2;
#pragma unroll
    for (int j = 0; j < 4; ++j) { m1[j] = (float)(tot[c0 + j] * invM); m2[j] = (float)(tot[p.C + c0 + j] * invM); }
    for (size_t i = (size_t)blockIdx.x * CH_BLOCK + threadIdx.x; i < total4; i += (size_t)gridDim.x * CH_BLOCK) {
        const f32x4 zv = *(const f32x4*)(p.z + 4 * i);
        f32x4 g = *(const f32x4*)(p.dz + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float zh = (zv[j] - mu[j]) * rs[j];
            g[j] = ga[j] * rs[j] * (g[j] - m1[j] - zh * m2[j]);
        }
        *(f32x4*)(p.dz + 4 * i) = g;
    }
}

__global__ __launch_bounds__(CH_BLOCK) void bn_bwd_apply_kernel(EpiParams p) {
    const size_t total = (size_t)p.M * p.C;
    const double invM = 1.0 / (double)p.M;
    for (size_t idx = (size_t)blockIdx.x * CH_BLOCK + threadIdx.x; idx < total; idx += (size_t)gridDim.x * CH_BLOCK) {
        const int c = (int)(idx - (idx / (unsigned)p.C) * p.C);
        const float rs = p.rstd[c];
        const float zh = (p.z[idx] - p.mean[c]) * rs;
        const double* tot = p.acc + (size_t)p.nslots * 3 * p.C;       // totals, written by the finalize launch
        const float m1 = (float)(tot[c] * invM), m2 = (float)(tot[p.C + c] * invM);
        p.dz[idx] = p.gamma[c] * rs * (p.dz[idx] - m1 - zh * m2);
    }
}

// partial slots -> totals (kept for bn_bwd_apply_kernel) and f32 parameter gradients, all sums in a fixed order.
// Workgroups 0 .. ceil(C/8)-1: 8 channels each (slot_sums).  The LAST workgroup produces dalpha from the slots:
// per-channel slopes (n_alpha = Cf): thread cf adds the conv channels that store to final channel cf, slot by slot;
// one shared slope (n_alpha = 1): 256 threads each add a strided share of the channels in order, then thread 0 adds
// the 256 shares in order.
__global__ __launch_bounds__(256) void epilogue_bwd_finalize_kernel(double* __restrict__ acc, int nslots, int C,
                                                                    int n_alpha, int bn, int Cf, int rr, int dcr,
                                                                    float* __restrict__ dsum1, float* __restrict__ dsum2,
                                                                    float* __restrict__ dalpha, float* __restrict__ dzero) {
    __shared__ double sh3[3][FIN_G][FIN_CH];
    __shared__ double sh[256];
    double* tot = acc + (size_t)nslots * 3 * C;
    if (blockIdx.x + 1 < gridDim.x) {
        const int c = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH, g = threadIdx.x / FIN_CH;
        double t[3];
        slot_sums<3>(acc, nslots, C, c, g, t, sh3);
        if (c >= C || g != 0) return;
        tot[c] = t[0]; tot[C + c] = t[1]; tot[2 * C + c] = t[2];        // [2]: dalpha terms per conv channel
        if (dsum1) dsum1[c] = (float)t[0];
        if (bn && dsum2) dsum2[c] = (float)t[1];
        if (dzero) dzero[c] = 0.f;
        return;
    }
    if (!dalpha || n_alpha <= 0) return;
    if (n_alpha > 1) {          // per-channel slopes: final channel cf <- conv channels {cf*rr + s} (CRD) or {s*Cf + cf} (DCR)
        for (int cf = threadIdx.x; cf < n_alpha; cf += 256) {
            double t = 0.0;
            for (int s = 0; s < rr; ++s) {
                const int c = dcr ? s * Cf + cf : cf * rr + s;
                for (int b = 0; b < nslots; ++b) t += acc[(size_t)b * 3 * C + 2 * C + c];
            }
            dalpha[cf] = (float)t;
        }
        return;
    }
    double t = 0.0;
    const int items = nslots * C;                  // <= 512 slots x C: 32-bit (C < 2^22 is checked on the host)
#pragma unroll 4
    for (int it = threadIdx.x; it < items; it += 256) {      // item = (slot, channel), fixed order per thread
        const int b = it / C, c = it - b * C;
        t += acc[(size_t)b * 3 * C + 2 * C + c];
    }
    sh[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x < 16) {                        // 16 threads add 16 shares each in order, thread 0 the 16 results
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += sh[threadIdx.x * 16 + k];
        sh[threadIdx.x * 16] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += sh[k * 16];
        dalpha[0] = (float)a;
    }
}

// ---- 2x2 stride-2 max pool (VGG19, P/vgg19.py) NHWC f32 -------------------------------------------------
__global__ __launch_bounds__(CH_BLOCK) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)N * Ho * Wo * C;
    for (size_t idx = (size_t)blockIdx.x * CH_BLOCK + threadIdx.x; idx < total; idx += (size_t)gridDim.x * CH_BLOCK) {
        const int c = (int)(idx % C);
        size_t t = idx / C;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const size_t n = t / Ho;
        const float* b = x + ((n * H + 2 * oy) * W + 2 * ox) * C + c;
        y[idx] = fmaxf(fmaxf(b[0], b[C]), fmaxf(b[(size_t)W * C], b[(size_t)W * C + C]));
    }
}

// gradient goes to the FIRST maximum in (0,0),(0,1),(1,0),(1,1) order (ATen's max_pool2d tie rule);
// pixels outside the pooled area (odd H/W) get zero
__global__ __launch_bounds__(CH_BLOCK) void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ dx, int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)N * H * W * C;
    for (size_t idx = (size_t)blockIdx.x * CH_BLOCK + threadIdx.x; idx < total; idx += (size_t)gridDim.x * CH_BLOCK) {
        const int c = (int)(idx % C);
        size_t t = idx / C;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const size_t n = t / H;
        const int oy = iy / 2, ox = ix / 2;
        float g = 0.f;
        if (oy < Ho && ox < Wo) {
            const float* b = x + ((n * H + 2 * oy) * W + 2 * ox) * C + c;
            const float v[4] = {b[0], b[C], b[(size_t)W * C], b[(size_t)W * C + C]};
            int am = 0;
            for (int k = 1; k < 4; ++k)
                if (v[k] > v[am]) am = k;
            if (am == (iy & 1) * 2 + (ix & 1)) g = dy[((n * Ho + oy) * Wo + ox) * C + c];
        }
        dx[idx] = g;
    }
}

// ---- loss gradients --------------------------------------------------------------------------------
// d mean((a-b)^2) / da = 2 (a-b) / n * gout ; db = -da
__global__ __launch_bounds__(CH_BLOCK) void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ gout, size_t n,
                                                           float* __restrict__ da, float* __restrict__ db) {
    const float k = 2.f * gout[0] / (float)n;
    for (size_t i = (size_t)blockIdx.x * CH_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * CH_BLOCK) {
        const float g = (a[i] - b[i]) * k;
        if (da) da[i] = g;
        if (db) db[i] = -g;
    }
}

// d BCE / dp = (p - t) / (max(p (1-p), 1e-12)) / n * gout   (ATen binary_cross_entropy_backward)
__global__ __launch_bounds__(CH_BLOCK) void bce_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                           const float* __restrict__ gout, size_t n,
                                                           float* __restrict__ dp) {
    const float k = gout[0] / (float)n;
    for (size_t i = (size_t)blockIdx.x * CH_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * CH_BLOCK) {
        const float pv = p[i];
        dp[i] = k * (pv - t[i]) / fmaxf((1.f - pv) * pv, 1e-12f);
    }
}

// d TVLoss / dx (P/losses.py:46-55): every element takes part in up to two vertical and two horizontal
// squared differences.  kh = weight*2/(count_h*N) * 2, kw likewise; gout = upstream scalar gradient.
__global__ __launch_bounds__(CH_BLOCK) void tv_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout,
                                                          size_t total, int H, int W, float kh, float kw,
                                                          float* __restrict__ dx) {
    const float g = gout[0];
    for (size_t i = (size_t)blockIdx.x * CH_BLOCK + threadIdx.x; i < total; i += (size_t)gridDim.x * CH_BLOCK) {
        const int xc = (int)(i % W);
        const int yr = (int)((i / W) % H);
        const float v = x[i];
        float a = 0.f, b = 0.f;
        if (yr > 0) a += v - x[i - W];
        if (yr < H - 1) a -= x[i + W] - v;
        if (xc > 0) b += v - x[i - 1];
        if (xc < W - 1) b -= x[i + 1] - v;
        dx[i] = g * (kh * a + kw * b);
    }
}

// ---- Adam (torch.optim.Adam defaults: no amsgrad, weight_decay 0; P/mode.py:90, :138-139) -------------
__global__ __launch_bounds__(CH_BLOCK) void adam_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                        float* __restrict__ m, float* __restrict__ v, size_t n,
                                                        float w1, float beta2, float w2, float eps, float step_size,
                                                        float bc2_sqrt) {
    // torch.optim.Adam (foreach form): exp_avg.lerp_(g, 1 - b1); exp_avg_sq.mul_(b2).addcmul_(g, g, value = 1 - b2);
    // denom = (exp_avg_sq.sqrt() / bc2_sqrt).add_(eps); p.addcdiv_(exp_avg, denom, value = -step_size)
    for (size_t i = (size_t)blockIdx.x * CH_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * CH_BLOCK) {
        const float g = grad[i];
        const float mi = m[i] + w1 * (g - m[i]);                  // lerp, weight < 0.5
        const float vi = v[i] * beta2 + w2 * g * g;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        param[i] -= step_size * (mi / denom);
    }
}

// ---- host launchers ------------------------------------------------------------------------------------
static unsigned ew_blocks(size_t total) {
    size_t b = (total + CH_BLOCK - 1) / CH_BLOCK;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// grid of the per-channel reductions: x = pixel slices (each leaves one partial slot: at most max_slots(C)),
// y = 64-channel groups; ~1024 workgroups in total (four per CU: the loops are latency-bound streams)
static int max_slots(int C) {
    int cap = 1024 / ((C + 63) / 64);
    if (cap < 16) cap = 16;
    if (cap > TRAIN_MAX_SLOTS) cap = TRAIN_MAX_SLOTS;
    return cap;
}

static dim3 ch_grid(long long M, int C) {
    long long bx = (M + 63) / 64;  // >= 4 passes of 16 pixel rows per workgroup
    const long long cap = max_slots(C);
    if (bx > cap) bx = cap;
    if (bx < 1) bx = 1;
    return dim3((unsigned)bx, (unsigned)((C + 63) / 64));
}

static bool vec4_ptrs(std::initializer_list<const void*> ps) {
    for (const void* q : ps)
        if (q && ((uintptr_t)q & 15)) return false;
    return true;
}
// four channels per lane with loop-invariant channels: C / 4 divides the 256-thread workgroup
static bool lane_keeps_channels(int C) { return C % 4 == 0 && C >= 4 && 256 % (C / 4) == 0; }

int train_max_slots(int C) { return max_slots(C); }

// partial slots [nslots][2][C] (sum, sum of squares) -> mean / rstd / running statistics (the second half of bn_train_stats)
int bn_finalize_launch(const void* ws, int nslots, long long M, int C, float eps, float momentum, float* mean, float* rstd,
                       float* rmean, float* rvar, long long* nbt, int updates, hipStream_t s) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(256), 0, s, (const double*)ws, nslots, M, C,
                       eps, momentum, mean, rstd, rmean, rvar, nbt, updates);
    return hip_check(hipGetLastError(), "bn_finalize_kernel");
}

size_t train_workspace_bytes(int C) { return ((size_t)max_slots(C) * 3 + 3) * C * sizeof(double); }

int bn_train_stats(const float* z, long long M, int C, float eps, float momentum, float* mean, float* rstd,
                   float* rmean, float* rvar, long long* nbt, int updates, void* ws, hipStream_t s) {
    const dim3 grid = ch_grid(M, C);      // every slot of the grid is written: nothing to clear
    if (C % 4 == 0 && vec4_ptrs({z}))
        hipLaunchKernelGGL(bn_stats_v4_kernel, grid, dim3(CH_BLOCK), 0, s, z, M, C, (double*)ws);
    else
        hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(CH_BLOCK), 0, s, z, M, C, (double*)ws);
    int rc = hip_check(hipGetLastError(), "bn_stats_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(256), 0, s, (const double*)ws, (int)grid.x, M, C,
                       eps, momentum, mean, rstd, rmean, rvar, nbt, updates);
    return hip_check(hipGetLastError(), "bn_finalize_kernel");
}

static EpiParams epi_params(const srmi_epi_desc& d) {
    EpiParams p;
    memset(&p, 0, sizeof(p));
    p.M = (long long)d.N * d.H * d.W;
    p.H = d.H; p.W = d.W; p.C = d.C; p.r = d.shuffle; p.Cf = d.C / (d.shuffle * d.shuffle);
    p.dcr = d.shuffle_order == SRMI_SHUFFLE_DCR;
    p.act = d.act; p.act_param = d.act_param; p.alpha_pc = d.alpha_per_channel; p.bn = d.bn;
    return p;
}

int epilogue_fwd(const srmi_epi_desc& d, const float* z, const float* mean, const float* rstd, const float* gamma,
                 const float* beta, const float* alpha, const float* residual, float* y, hipStream_t s) {
    EpiParams p = epi_params(d);
    p.z = z; p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.beta = beta; p.alpha = alpha; p.residual = residual;
    p.y = y;
    if (p.M == 0) return SRMI_OK;
    if (p.r == 1 && lane_keeps_channels(p.C) && vec4_ptrs({z, y, residual}))
        hipLaunchKernelGGL(epilogue_fwd_v4_kernel, dim3(ew_blocks((size_t)p.M * p.C / 4)), dim3(CH_BLOCK), 0, s, p);
    else
        hipLaunchKernelGGL(epilogue_fwd_kernel, dim3(ew_blocks((size_t)p.M * p.C)), dim3(CH_BLOCK), 0, s, p);
    return hip_check(hipGetLastError(), "epilogue_fwd_kernel");
}

int epilogue_bwd(const srmi_epi_desc& d, const float* z, const float* dy, const float* mean, const float* rstd,
                 const float* gamma, const float* beta, const float* alpha, float* dz, float* dsum1, float* dsum2,
                 float* dalpha, float* dzero, void* ws, hipStream_t s) {
    EpiParams p = epi_params(d);
    p.z = z; p.dy = dy; p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.beta = beta; p.alpha = alpha;
    p.dz = dz; p.acc = (double*)ws;
    const int n_alpha = d.act == SRMI_ACT_PRELU ? (d.alpha_per_channel ? p.Cf : 1) : 0;
    if (!d.bn && !dsum1 && !dsum2 && !dalpha && !dzero && p.r == 1 && lane_keeps_channels(p.C) &&
        vec4_ptrs({z, dy, dz})) {      // nothing to reduce: one streaming launch
        if (p.M == 0) return SRMI_OK;
        hipLaunchKernelGGL(act_bwd_v4_kernel, dim3(ew_blocks((size_t)p.M * p.C / 4)), dim3(CH_BLOCK), 0, s, p);
        return hip_check(hipGetLastError(), "act_bwd_v4_kernel");
    }
    const dim3 grid = ch_grid(p.M > 0 ? p.M : 1, p.C);
    p.nslots = p.M > 0 ? (int)grid.x : 0;
    int rc;
    if (p.M > 0) {
        if (p.r == 1 && p.C % 4 == 0 && vec4_ptrs({z, dy, dz}))
            hipLaunchKernelGGL(epilogue_bwd_v4_kernel, grid, dim3(CH_BLOCK), 0, s, p);
        else
            hipLaunchKernelGGL(epilogue_bwd_kernel, grid, dim3(CH_BLOCK), 0, s, p);
        rc = hip_check(hipGetLastError(), "epilogue_bwd_kernel");
        if (rc) return rc;
    }
    // totals + parameter gradients (zero slots: all sums are zero)
    hipLaunchKernelGGL(epilogue_bwd_finalize_kernel, dim3((d.C + FIN_CH - 1) / FIN_CH + 1), dim3(256), 0, s, (double*)ws, p.nslots,
                       d.C, n_alpha, d.bn, p.Cf, d.shuffle * d.shuffle, p.dcr, dsum1, dsum2, dalpha, dzero);
    rc = hip_check(hipGetLastError(), "epilogue_bwd_finalize_kernel");
    if (rc) return rc;
    if (p.M > 0 && d.bn) {
        if (lane_keeps_channels(p.C) && vec4_ptrs({z, dz}))
            hipLaunchKernelGGL(bn_bwd_apply_v4_kernel, dim3(ew_blocks((size_t)p.M * p.C / 4)), dim3(CH_BLOCK), 0, s, p);
        else
            hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks((size_t)p.M * p.C)), dim3(CH_BLOCK), 0, s, p);
        return hip_check(hipGetLastError(), "bn_bwd_apply_kernel");
    }
    return SRMI_OK;
}

int maxpool_fwd(const float* x, float* y, int N, int H, int W, int C, hipStream_t s) {
    const size_t total = (size_t)N * (H / 2) * (W / 2) * C;
    if (total == 0) return SRMI_OK;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_blocks(total)), dim3(CH_BLOCK), 0, s, x, y, N, H, W, C);
    return hip_check(hipGetLastError(), "maxpool_fwd_kernel");
}

int maxpool_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, hipStream_t s) {
    const size_t total = (size_t)N * H * W * C;
    if (total == 0) return SRMI_OK;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_blocks(total)), dim3(CH_BLOCK), 0, s, x, dy, dx, N, H, W, C);
    return hip_check(hipGetLastError(), "maxpool_bwd_kernel");
}

int mse_bwd(const float* a, const float* b, const float* gout, size_t n, float* da, float* db, hipStream_t s) {
    if (n == 0) return SRMI_OK;
    hipLaunchKernelGGL(mse_bwd_kernel, dim3(ew_blocks(n)), dim3(CH_BLOCK), 0, s, a, b, gout, n, da, db);
    return hip_check(hipGetLastError(), "mse_bwd_kernel");
}

int bce_bwd(const float* p, const float* t, const float* gout, size_t n, float* dp, hipStream_t s) {
    if (n == 0) return SRMI_OK;
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(ew_blocks(n)), dim3(CH_BLOCK), 0, s, p, t, gout, n, dp);
    return hip_check(hipGetLastError(), "bce_bwd_kernel");
}

int tv_bwd(const float* x, const float* gout, int N, int C, int H, int W, float weight, float* dx, hipStream_t s) {
    const size_t total = (size_t)N * C * H * W;
    if (total == 0) return SRMI_OK;
    const double count_h = (double)C * (H - 1) * W, count_w = (double)C * H * (W - 1);
    const double k = 2.0 * (double)weight / (double)N * 2.0;
    hipLaunchKernelGGL(tv_bwd_kernel, dim3(ew_blocks(total)), dim3(CH_BLOCK), 0, s, x, gout, total, H, W,
                       (float)(k / count_h), (float)(k / count_w), dx);
    return hip_check(hipGetLastError(), "tv_bwd_kernel");
}

int adam_step(float* param, const float* grad, float* m, float* v, size_t n, double lr, double beta1, double beta2,
              double eps, int step, hipStream_t s) {
    if (n == 0) return SRMI_OK;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    // every scalar in double first, rounded to f32 once (what torch's Python-side scalars go through)
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n)), dim3(CH_BLOCK), 0, s, param, grad, m, v, n, (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), (float)eps, (float)(lr / bc1), (float)sqrt(bc2));
    return hip_check(hipGetLastError(), "adam_kernel");
}

}  // namespace srmi
