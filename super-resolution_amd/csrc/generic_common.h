// Shared by the two exact-f32 conv kernels (conv_generic.hip: no LDS, any input layout / dtype; conv_f32_tiled.hip:
// LDS-tiled, NHWC f32 input): the launch parameters and the fused epilogue.
#pragma once
#include "common.h"

namespace srmi {

struct GenericParams {
    const void* x; const float* w; const float* scale; const float* shift; const float* alpha;
    const void* res; void* y;
    int N, H, W, Cin, Cout, kh, kw, stride, pad;
    int Ho, Wo, Cf, Hf, Wf, in_cs, out_cs, out_co, res_cs;
    int in_bf16, out_bf16, res_bf16, in_nchw, out_nchw, vec_ok;
    int act; float act_param; int alpha_pc; int r; int dcr;
    int Cin_pad, Cout_pad, M;
    hipStream_t stream;
    int ksplit;      // > 1: grid.z slices of K (linear epilogue only: no scale / activation / residual)
    float* partial;  // ksplit > 1: slice z STORES its partial tile at partial + z * partial_stride (same indexing
    size_t partial_stride;  // as y) and generic_splitk_reduce_kernel adds the slices in index order: deterministic
    int transposed;  // dgrad: gather dz at (o + pad - k) / stride where divisible
    int dbg;         // ablation switches of the tiled kernel (SRMI_DBG bits 26..29; investigation only, 0 = product path)
    unsigned wo_mul, ho_mul; int wo_sh, ho_sh;   // tiled kernel: multiply-shift division by Wo / Ho (f32t_magic)
    int b3;          // the operand is in conv_b3_f32.hip's split-bf16 format: that kernel runs
    double* bn_stats;  // conv_b3h_kernel only: per-workgroup sum / sum of squares of the output (train-mode BN statistics)
};


// Fast epilogue for the common case - NHWC output without a sub-pixel shuffle (the flattened output pixel index IS the
// pixel of the output tensor) and an activation of the slope family (none / ReLU / LeakyReLU / PReLU:
// v >= 0 ? v : v * slope) - or the raw slice of a split-K launch.  Every wave-uniform decision is taken OUTSIDE the
// element loops: the general epilogue below, with all its branches replicated for each of the 16 * NT elements, is
// ~10,000 instructions (more than the instruction cache) and cost 11 us per launch on the small layers of the
// training step.
template <int NT, bool HAS_RES, bool OUT_BF16, bool RES_BF16, bool PARTIAL>
__device__ __forceinline__ void generic_epilogue_flat(const GenericParams& p, const f32x16 (&acc)[NT], int m0, int co0,
                                                      int lane, int zslice) {
    const int i = lane & 31, h = lane >> 5;
    float slope0 = 1.f;
    if (p.act == SRMI_ACT_RELU) slope0 = 0.f;
    else if (p.act == SRMI_ACT_LRELU) slope0 = p.act_param;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int co = co0 + 32 * t + i;
        const bool cvalid = co < p.Cout;
        const int coc = cvalid ? co : 0;
        const float sc = p.scale ? p.scale[coc] : 1.f;
        const float sh = (p.shift && (!PARTIAL || zslice == 0)) ? p.shift[coc] : 0.f;
        const float sl = p.act == SRMI_ACT_PRELU ? p.alpha[p.alpha_pc ? coc : 0] : slope0;
        float* part = PARTIAL ? p.partial + (size_t)zslice * p.partial_stride + p.out_co + coc : nullptr;
        const size_t ocol = (size_t)p.out_co + coc;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            float v = acc[t][e] * sc + sh;
            if (cvalid && row < p.M) {
                if constexpr (PARTIAL) {
                    part[(size_t)row * p.out_cs] = v;
                } else {
                    v = v >= 0.f ? v : v * sl;
                    if constexpr (HAS_RES) {
                        const size_t ridx = (size_t)row * p.res_cs + coc;
                        v += RES_BF16 ? bf2f(((const bf16_t*)p.res)[ridx]) : ((const float*)p.res)[ridx];
                    }
                    if constexpr (OUT_BF16) ((bf16_t*)p.y)[(size_t)row * p.out_cs + ocol] = f2bf(v);
                    else ((float*)p.y)[(size_t)row * p.out_cs + ocol] = v;
                }
            }
        }
    }
}

// Lean epilogue of the stride-2 dgrad in its sub-pixel form (conv_generic.hip: dgrad_subpixel): raw f32 sums stored
// through depth_to_space(2), NHWC.  Cf % 32 == 0, so a 32-channel tile lies inside ONE sub-pixel (si, sj) and the 32 lanes
// of a row write 128 contiguous bytes; the output pixel of each accumulator row comes from one cross-lane read of the
// owning lane's base pixel, shared by all channel tiles (the general epilogue below does three per element and tile).
// (the shape it serves; such a launch may also split K: the slices go to p.partial with the same indexing)
__host__ __device__ __forceinline__ bool generic_subpix2_shape(const GenericParams& p) {
    return p.r == 2 && p.dcr && !p.out_nchw && p.act == SRMI_ACT_NONE && !p.scale && !p.shift && !p.res && !p.out_bf16 &&
           p.Cf % 32 == 0 && p.Cout == 4 * p.Cf && p.out_cs == p.Cf && p.out_co == 0;
}

template <int NT>
__device__ __forceinline__ void generic_epilogue_subpix2(const GenericParams& p, const f32x16 (&acc)[NT], int m0, int co0,
                                                         int n, int oy, int ox, int lane, int zslice) {
    const int i = lane & 31, h = lane >> 5;
    const int base = (n * p.Hf + 2 * oy) * p.Wf + 2 * ox;        // sub-pixel (0, 0) of pixel m0 + i (fits int: checked on the host)
    int rowpx[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) rowpx[e] = __shfl(base, (e & 3) + 8 * (e >> 2) + 4 * h);
    float* y = p.ksplit > 1 ? p.partial + (size_t)zslice * p.partial_stride : (float*)p.y;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int cot = co0 + 32 * t;                            // wave-uniform
        if (cot >= p.Cout) continue;
        const int sub = cot / p.Cf, c = cot - sub * p.Cf + i;
        const int shift = (sub >> 1) * p.Wf + (sub & 1);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
            if (m0 + row < p.M) y[(size_t)(rowpx[e] + shift) * p.out_cs + p.out_co + c] = acc[t][e];
        }
    }
}

// Fused epilogue of NT 32(pixel) x 32(channel) accumulator tiles of v_mfma_f32_32x32x2_f32:
//   scale/shift -> act -> +residual -> (sub-pixel) store, or the raw slice of a split-K launch.
// acc[t][e] = pixel row (e&3) + 8*(e>>2) + 4*h of the tile starting at flattened output pixel m0, channel co0 + 32*t + i
// (i = lane & 31, h = lane >> 5).  n / oy / ox are THIS LANE's coordinates of pixel m0 + i (shuffled to the rows).
template <int NT>
__device__ __forceinline__ void generic_epilogue(const GenericParams& p, const f32x16 (&acc)[NT], int m0, int co0, int n,
                                                 int oy, int ox, int lane, int zslice) {
    if (generic_subpix2_shape(p)) {
        generic_epilogue_subpix2<NT>(p, acc, m0, co0, n, oy, ox, lane, zslice);
        return;
    }
    if (p.r == 1 && !p.out_nchw && p.act != SRMI_ACT_TANH && p.act != SRMI_ACT_SIGMOID && p.act != SRMI_ACT_ADD_RELU) {
        if (p.ksplit > 1) generic_epilogue_flat<NT, false, false, false, true>(p, acc, m0, co0, lane, zslice);
        else if (!p.res) {
            if (p.out_bf16) generic_epilogue_flat<NT, false, true, false, false>(p, acc, m0, co0, lane, zslice);
            else generic_epilogue_flat<NT, false, false, false, false>(p, acc, m0, co0, lane, zslice);
        } else if (p.out_bf16) {
            if (p.res_bf16) generic_epilogue_flat<NT, true, true, true, false>(p, acc, m0, co0, lane, zslice);
            else generic_epilogue_flat<NT, true, true, false, false>(p, acc, m0, co0, lane, zslice);
        } else {
            if (p.res_bf16) generic_epilogue_flat<NT, true, false, true, false>(p, acc, m0, co0, lane, zslice);
            else generic_epilogue_flat<NT, true, false, false, false>(p, acc, m0, co0, lane, zslice);
        }
        return;
    }
    const int i = lane & 31, h = lane >> 5;
    // ---- general path: any activation, sub-pixel stores, NCHW output ---------------
    const int rr = p.r * p.r;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int co = co0 + 32 * t + i;
        const bool cvalid = co < p.Cout;
        const int coc = cvalid ? co : 0;
        const float sc = p.scale ? p.scale[coc] : 1.f;
        const float sh = (p.shift && (p.ksplit <= 1 || zslice == 0)) ? p.shift[coc] : 0.f;
        const float al = (p.act == SRMI_ACT_PRELU) ? p.alpha[p.alpha_pc ? coc : 0] : 0.f;
        int c = coc, si = 0, sj = 0;
        if (p.r > 1) {
            if (p.dcr) { c = coc % p.Cf; const int s = coc / p.Cf; si = s / p.r; sj = s % p.r; }
            else { c = coc / rr; const int s = coc % rr; si = s / p.r; sj = s % p.r; }
        }
        // NHWC output without a sub-pixel shuffle: the flattened output pixel index IS the pixel of the output tensor -
        // no coordinate shuffles, no index arithmetic beyond one multiply (the training path and most inference layers)
        const bool flat = p.r == 1 && !p.out_nchw;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
            int n_r = 0, oy_r = 0, ox_r = 0;
            if (!flat) { n_r = __shfl(n, row); oy_r = __shfl(oy, row); ox_r = __shfl(ox, row); }
            if (!cvalid || m0 + row >= p.M) continue;
            const bool late = p.act == SRMI_ACT_ADD_RELU;   // the residual joins before the ReLU
            float val = late ? acc[t][e] * sc + sh : apply_act(acc[t][e] * sc + sh, p.act, p.act_param, al);
            size_t oidx, ridx;
            if (flat) {
                const size_t px = (size_t)(m0 + row);
                oidx = px * p.out_cs + p.out_co + c;
                ridx = px * p.res_cs + c;
            } else {
                const int Y = oy_r * p.r + si, X = ox_r * p.r + sj;
                if (p.out_nchw) {
                    oidx = (((size_t)n_r * p.Cf + c) * p.Hf + Y) * p.Wf + X;
                    ridx = oidx;
                } else {
                    const size_t px = ((size_t)n_r * p.Hf + Y) * p.Wf + X;
                    oidx = px * p.out_cs + p.out_co + c;
                    ridx = px * p.res_cs + c;
                }
            }
            if (p.ksplit > 1) {
                p.partial[(size_t)zslice * p.partial_stride + oidx] = val;
                continue;
            }
            if (p.res) val += p.res_bf16 ? bf2f(((const bf16_t*)p.res)[ridx]) : ((const float*)p.res)[ridx];
            if (late) val = fmaxf(val, 0.f);
            if (p.out_bf16) ((bf16_t*)p.y)[oidx] = f2bf(val);
            else ((float*)p.y)[oidx] = val;
        }
    }
}

}  // namespace srmi
