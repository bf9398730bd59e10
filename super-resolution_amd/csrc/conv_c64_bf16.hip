// conv3x3 / stride 1 / pad 1 / Cin = 64 / Cout = 64*NCHUNK, NHWC bf16, on v_mfma_f32_16x16x32_bf16.
//
// This is the kernel that carries 98 % of the SRResNet generator's FLOPs: the 33 body convs
// (P/ops.py:41-42, P/srgan_model.py:16) and the two Upsampler convs (P/ops.py:74) with
// PixelShuffle(2) (P/ops.py:75) folded into the store addressing.
//
// Design (MI355X-first; see DESIGN.md "c64 kernel"):
//   * one workgroup (4 waves, one per SIMD) owns a TH x TW spatial tile of one image and ALL
//     output channels; the (TH+2) x (TW+2) x 64 input halo tile is staged ONCE in LDS (bf16,
//     128 B per pixel, 16-B chunks XOR-swizzled by pixel pair so the MFMA operand reads are
//     bank-conflict free) and re-used by all 9 taps and all Cout chunks;
//   * GEMM orientation D[cout][pixel] = W[cout][k] * X[k][pixel]: the activations are the B
//     operand (one ds_read_b128 per 16 pixels x 32 channels), the weights are the A operand,
//     pre-packed on the host in MFMA fragment order so that each fragment is one perfectly
//     coalesced 1-KiB global load that all four waves share through L1/L2 (no LDS, no barrier
//     in the K loop);
//   * the packing also permutes output channels so that a lane's 16 accumulator values of a
//     pixel are 16 CONSECUTIVE channels: the epilogue (scale/shift = bias+BN folded, PReLU /
//     LeakyReLU, residual add) reads and writes 32 contiguous bytes per lane, and for the
//     sub-pixel modes one 64-channel chunk is exactly one 128-B output pixel;
//   * each wave keeps PT pixel tiles x 4 channel tiles of f32 accumulators (PT = 9 for the
//     6 x 96 tile: 13 fragment reads feed 36 MFMAs per K-step).
#include "common.h"

namespace srmi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct C64Params {
    const bf16_t* x; const bf16_t* w; const float* scale; const float* shift; const float* alpha;
    const bf16_t* res; bf16_t* y;
    int N, H, W, Cout, nchunk;
    int in_cs, out_cs, out_co;
    int act; float act_param; int alpha_pc;
    int mode;  // 0 plain, 1 PixelShuffle(2) CRD, 2 depth_to_space(2) DCR
    int tiles_x, tiles_y;
};


// conv-channel index of (chunk s, channel-within-chunk cl)
__device__ __forceinline__ int c64_conv_channel(int mode, int s, int cl) {
    return mode == 1 ? 4 * cl + s : 64 * s + cl;
}

template <int TH, int TWT>
__global__ __launch_bounds__(256, 1) void conv_c64_kernel(C64Params p) {
    constexpr int TW = TWT * 16;
    constexpr int NPT = TH * TWT;
    static_assert(NPT % 4 == 0, "pixel tiles must split evenly over 4 waves");
    constexpr int PT = NPT / 4;
    constexpr int ROWP = (TW + 2 + 15) / 16 * 16;  // LDS row pitch in pixels (multiple of 16: swizzle independent of dy)
    constexpr int LDS_BYTES = (TH + 2) * ROWP * 128;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15;  // pixel within a 16-pixel tile (MFMA column)
    const int q = lane >> 4;   // k-quarter for operands / channel group for results

    int b = blockIdx.x;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- stage the input halo tile: global (NHWC bf16) -> registers -> swizzled LDS ----------
    {
        constexpr int NCH = (TH + 2) * (TW + 2) * 8;  // 16-B chunks
        const bf16_t* xin = p.x + (size_t)n * p.H * p.W * p.in_cs;
#pragma unroll 4
        for (int idx = tid; idx < NCH; idx += 256) {
            const int pp = idx >> 3, ch = idx & 7;
            const int row = pp / (TW + 2), col = pp - row * (TW + 2);
            const int gy = y0 - 1 + row, gx = x0 - 1 + col;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                v = *(const u32x4*)(xin + ((size_t)gy * p.W + gx) * p.in_cs + ch * 8);
            const int P = row * ROWP + col;
            *(u32x4*)(lds + P * 128 + ((ch ^ ((P >> 1) & 7)) << 4)) = v;
        }
    }
    __syncthreads();

    // per-wave pixel tiles: tile index t = wave*PT + k -> (row, 16-col group)
    int prow[PT], pcol[PT];
#pragma unroll
    for (int k = 0; k < PT; ++k) {
        const int t = wave * PT + k;
        prow[k] = t / TWT;
        pcol[k] = (t - prow[k] * TWT) * 16;
    }

    const float alpha0 = (p.act == SRMI_ACT_PRELU && !p.alpha_pc) ? p.alpha[0] : 0.f;

    for (int s = 0; s < p.nchunk; ++s) {
        f32x4 acc[PT][4];
#pragma unroll
        for (int k = 0; k < PT; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[k][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

        const bf16x8* wfrag = (const bf16x8*)p.w + (size_t)s * 18 * 4 * 64 + lane;
        bf16x8 a_cur[4], a_nxt[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) a_cur[t] = wfrag[t * 64];

        // K loop: dy rolled (keeps the weight-fragment prefetch distance bounded), (dx, channel half)
        // unrolled: the swizzle term (P>>1)&7 does not depend on dy because ROWP % 16 == 0, so the
        // 6 LDS offsets per pixel tile are loop invariants and dy only adds a multiple of ROWP*128.
#pragma unroll 1
        for (int dy = 0; dy < 3; ++dy) {
            const unsigned char* lrow = lds + dy * (ROWP * 128);
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) {
                const int dx = kk >> 1;
                const int cb = (kk & 1) * 4;
                const int ks = dy * 6 + kk;
                const int ksn = ks + 1 < 18 ? ks + 1 : 17;  // last prefetch re-reads step 17 (harmless)
#pragma unroll
                for (int t = 0; t < 4; ++t) a_nxt[t] = wfrag[(ksn * 4 + t) * 64];
                bf16x8 bfr[PT];
#pragma unroll
                for (int k = 0; k < PT; ++k) {
                    const int P = prow[k] * ROWP + pcol[k] + pl + dx;
                    bfr[k] = *(const bf16x8*)(lrow + P * 128 + (((cb + q) ^ ((P >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int k = 0; k < PT; ++k)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_cur[t], bfr[k], acc[k][t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) a_cur[t] = a_nxt[t];
            }
        }

        // ---- epilogue: lane holds channels 16q .. 16q+15 of chunk s for pixel pl of each tile ----
        float sc[16], sh[16], al[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = c64_conv_channel(p.mode, s, 16 * q + e);
            sc[e] = p.scale[co];
            sh[e] = p.shift[co];
            al[e] = (p.act == SRMI_ACT_PRELU && p.alpha_pc) ? p.alpha[co] : alpha0;
        }
        const int si = s >> 1, sj = s & 1;
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            const int oy = y0 + prow[k], ox = x0 + pcol[k] + pl;
            if (oy >= p.H || ox >= p.W) continue;
            size_t pix, cofs;
            if (p.mode == 0) {
                pix = ((size_t)n * p.H + oy) * p.W + ox;
                cofs = (size_t)64 * s + 16 * q;
            } else {
                pix = ((size_t)n * 2 * p.H + 2 * oy + si) * (2 * p.W) + 2 * ox + sj;
                cofs = (size_t)16 * q;
            }
            float v[16];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 4 * t + e;
                    v[c] = apply_act(acc[k][t][e] * sc[c] + sh[c], p.act, p.act_param, al[c]);
                }
            if (p.res) {
                const int Cf = p.mode == 0 ? p.Cout : 64;
                const u32x4* rp = (const u32x4*)(p.res + pix * Cf + cofs);
                const u32x4 r0 = rp[0], r1 = rp[1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += bf2f((bf16_t)(r0[e] & 0xffffu));
                    v[2 * e + 1] += bf2f((bf16_t)(r0[e] >> 16));
                    v[8 + 2 * e] += bf2f((bf16_t)(r1[e] & 0xffffu));
                    v[8 + 2 * e + 1] += bf2f((bf16_t)(r1[e] >> 16));
                }
            }
            u32x4 o0, o1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o0[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
                o1[e] = pack_bf2(v[8 + 2 * e], v[8 + 2 * e + 1]);
            }
            u32x4* op = (u32x4*)(p.y + pix * p.out_cs + p.out_co + cofs);
            op[0] = o0;
            op[1] = o1;
        }
    }
}

// OIHW f32 -> bf16 MFMA A-fragment order:
//   packed[(((s*18 + ks)*4 + t)*64 + lane)*8 + j] = W[co][ci][ky][kx]
//   tap = ks>>1 (ky = tap/3, kx = tap%3), ci = (ks&1)*32 + 8*(lane>>4) + j,
//   A-row r = lane&15 of channel tile t  <->  channel-in-chunk cl = 16*(r>>2) + 4*t + (r&3),
//   co = conv channel of (chunk s, cl) for the store mode.
__global__ void c64_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int mode,
                                size_t total) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx & 7);
        const int lane = (int)((idx >> 3) & 63);
        const int t = (int)((idx >> 9) & 3);
        const size_t rest = idx >> 11;
        const int ks = (int)(rest % 18);
        const int s = (int)(rest / 18);
        const int r = lane & 15, qq = lane >> 4;
        const int cl = 16 * (r >> 2) + 4 * t + (r & 3);
        const int co = c64_conv_channel(mode, s, cl);
        const int ci = (ks & 1) * 32 + 8 * qq + j;
        const int tap = ks >> 1;
        out[idx] = f2bf(w[((size_t)co * 64 + ci) * 9 + tap]);
    }
}

bool c64_supported(const srmi_conv_desc& d) {
    if (d.compute != SRMI_COMPUTE_BF16) return false;
    if (d.kh != 3 || d.kw != 3 || d.stride != 1 || d.pad != 1 || d.Cin != 64) return false;
    if (d.Cout % 64 != 0 || d.Cout <= 0) return false;
    if (d.in_dtype != SRMI_BF16 || d.out_dtype != SRMI_BF16) return false;
    if (d.in_layout != SRMI_NHWC || d.out_layout != SRMI_NHWC) return false;
    if (d.shuffle != 1 && !(d.shuffle == 2 && d.Cout == 256)) return false;
    if (d.act == SRMI_ACT_TANH || d.act == SRMI_ACT_SIGMOID) return false;
    const int in_cs = d.in_cstride ? d.in_cstride : d.Cin;
    if (in_cs % 8 != 0) return false;
    const int Cf = d.Cout / (d.shuffle * d.shuffle);
    const int out_cs = d.out_cstride ? d.out_cstride : Cf;
    if (out_cs % 8 != 0 || d.out_coffset % 8 != 0) return false;
    return true;
}

size_t c64_packed_bytes(const srmi_conv_desc& d) { return (size_t)d.Cout * 64 * 9 * sizeof(bf16_t); }

static int c64_mode(const srmi_conv_desc& d) {
    if (d.shuffle == 1) return 0;
    return d.shuffle_order == SRMI_SHUFFLE_DCR ? 2 : 1;
}

int c64_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s) {
    const size_t total = (size_t)d.Cout * 64 * 9;
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(c64_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w_oihw, (bf16_t*)w_packed,
                       d.Cout, c64_mode(d), total);
    return hip_check(hipGetLastError(), "c64_pack_kernel");
}

int c64_launch(const srmi_conv_args& a, const ConvGeom& g, hipStream_t s) {
    const srmi_conv_desc& d = a.d;
    if (((uintptr_t)a.x | (uintptr_t)a.y | (uintptr_t)a.w | (uintptr_t)a.residual) & 15) {
        set_error("c64_bf16: x, w, y and residual must be 16-byte aligned");
        return SRMI_EINVAL;
    }
    if (a.residual && d.res_dtype != SRMI_BF16) {
        set_error("c64_bf16: residual must be bf16");
        return SRMI_EUNSUPPORTED;
    }
    C64Params p;
    p.x = (const bf16_t*)a.x; p.w = (const bf16_t*)a.w; p.scale = a.scale; p.shift = a.shift; p.alpha = a.alpha;
    p.res = (const bf16_t*)a.residual; p.y = (bf16_t*)a.y;
    p.N = d.N; p.H = d.H; p.W = d.W; p.Cout = d.Cout; p.nchunk = d.Cout / 64;
    p.in_cs = g.in_cs; p.out_cs = g.out_cs; p.out_co = d.out_coffset;
    p.act = d.act; p.act_param = d.act_param; p.alpha_pc = d.alpha_per_channel;
    p.mode = c64_mode(d);
    if (d.N == 0 || d.H == 0 || d.W == 0) return SRMI_OK;
    // tile choice: 6 x 96 (9 pixel tiles per wave) when the image is wide enough to fill it,
    // otherwise 4 x 32 (2 pixel tiles per wave) so small images still spread over the CUs.
    const bool big = d.W > 64 && d.H >= 6;
    if (big) {
        p.tiles_x = (d.W + 95) / 96; p.tiles_y = (d.H + 5) / 6;
        const long long nb = (long long)d.N * p.tiles_x * p.tiles_y;
        if (nb > 0x7fffffffLL) { set_error("c64_bf16: grid too large"); return SRMI_EINVAL; }
        hipLaunchKernelGGL((conv_c64_kernel<6, 6>), dim3((unsigned)nb), dim3(256), 0, s, p);
    } else {
        p.tiles_x = (d.W + 31) / 32; p.tiles_y = (d.H + 3) / 4;
        const long long nb = (long long)d.N * p.tiles_x * p.tiles_y;
        if (nb > 0x7fffffffLL) { set_error("c64_bf16: grid too large"); return SRMI_EINVAL; }
        hipLaunchKernelGGL((conv_c64_kernel<4, 2>), dim3((unsigned)nb), dim3(256), 0, s, p);
    }
    return hip_check(hipGetLastError(), "conv_c64_kernel");
}

}  // namespace srmi
