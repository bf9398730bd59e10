// conv3x3 / stride 1 / pad 1 / Cin = 64 / Cout <= 4, NHWC bf16 in -> NCHW|NHWC f32|bf16 out.
//
// The generator's `last_conv` (64 -> 3 + Tanh, P/srgan_model.py:25) at 4x resolution: 1.4 % of the
// FLOPs but it reads the largest tensor of the network (N x 4H x 4W x 64 bf16).  It is HBM-bound
// (128 B in, 12 B out per pixel), so the design goal is: read every input byte once per tile, keep
// many bytes in flight, and spend as little LDS bandwidth and issue slots as possible on the math.
//
//   * the (TH+2) x (TW+2) x 64 halo tile is staged in LDS exactly like the c64 kernel (same XOR
//     swizzle); tiles are small (6 x 64 -> 66 KiB) so two workgroups share a CU and one's staging
//     overlaps the other's math;
//   * math on v_mfma_f32_16x16x32_bf16 with the (zero-padded) output channels as the 16 MFMA rows:
//     only 3 of 16 rows are useful, which is affordable because the kernel is memory-bound, and it
//     keeps the activation operand a plain ds_read_b128 (no VALU dot products);
//   * LDS traffic is cut 3x by streaming INPUT rows: the fragment of input row r (per dx and channel
//     half) is read once and feeds the three output rows r-dy (3 MFMAs per read) - the 18 weight
//     fragments (9 taps x 2 halves) live in registers for the whole kernel.
#include "common.h"

namespace srmi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ThinParams {
    const bf16_t* x; const bf16_t* w; const float* scale; const float* shift; const float* alpha; void* y;
    int N, H, W, Cout, in_cs, out_cs, out_co;
    int act; float act_param; int alpha_pc;
    int out_nchw, out_bf16;
    int tiles_x, tiles_y;
};

template <int TH, int TWT>
__global__ __launch_bounds__(64 * TWT, 2) void conv_thin_kernel(ThinParams p) {
    constexpr int TW = TWT * 16;
    constexpr int NT = 64 * TWT;
    constexpr int ROWP = (TW + 2 + 15) / 16 * 16;
    constexpr int IN_BYTES = (TH + 2) * ROWP * 128;
    __shared__ __attribute__((aligned(16))) unsigned char lds[IN_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, q = lane >> 4;

    int b = blockIdx.x;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    // weights: 18 A fragments (tap-major, channel half minor), 1 KiB each, resident in registers
    bf16x8 afr[18];
#pragma unroll
    for (int f = 0; f < 18; ++f) afr[f] = ((const bf16x8*)p.w)[f * 64 + lane];

    // stage the halo tile (all loads in flight before the first LDS write; clamped address + select)
    {
        constexpr int NCH = (TH + 2) * (TW + 2) * 8;
        constexpr int NIT = (NCH + NT - 1) / NT;
        const bf16_t* xin = p.x + (size_t)n * p.H * p.W * p.in_cs;
        u32x4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = min(tid + it * NT, NCH - 1);
            const int pp = idx >> 3, ch = idx & 7;
            const int row = pp / (TW + 2), col = pp - row * (TW + 2);
            const int gy = min(max(y0 - 1 + row, 0), p.H - 1), gx = min(max(x0 - 1 + col, 0), p.W - 1);
            v[it] = *(const u32x4*)(xin + ((size_t)gy * p.W + gx) * p.in_cs + ch * 8);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT;
            const int pp = idx >> 3, ch = idx & 7;
            const int row = pp / (TW + 2), col = pp - row * (TW + 2);
            const int gy = y0 - 1 + row, gx = x0 - 1 + col;
            u32x4 z = v[it];
            if (!(gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)) z = (u32x4){0u, 0u, 0u, 0u};
            const int P = row * ROWP + col;
            if (idx < NCH) *(u32x4*)(lds + P * 128 + ((ch ^ ((P >> 1) & 7)) << 4)) = z;
        }
    }
    __syncthreads();

    // this wave's strip: columns [16*wave, 16*wave+16) of the tile
    int boff[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int P = wave * 16 + pl + dx;
        boff[dx] = P * 128 + ((q ^ ((P >> 1) & 7)) << 4);
    }
    f32x4 acc[TH];
#pragma unroll
    for (int y = 0; y < TH; ++y) acc[y] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int r = 0; r < TH + 2; ++r) {          // input row of the halo tile
        bf16x8 bfr[3][2];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                bfr[dx][ks] = *(const bf16x8*)(lds + ((boff[dx] ^ (ks << 6)) + r * (ROWP * 128)));
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int y = r - dy;                // output row fed by (input row r, tap row dy)
            if (y < 0 || y >= TH) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    acc[y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[(dy * 3 + dx) * 2 + ks], bfr[dx][ks], acc[y], 0, 0, 0);
        }
    }

    // epilogue: MFMA rows 0..3 (lanes q == 0) are output channels 0..3
    if (q == 0) {
        const int ox = x0 + wave * 16 + pl;
        const int Hf = p.H, Wf = p.W;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e >= p.Cout) break;
            const float sc = p.scale[e], sh = p.shift[e];
            const float al = p.act == SRMI_ACT_PRELU ? p.alpha[p.alpha_pc ? e : 0] : 0.f;
#pragma unroll
            for (int y = 0; y < TH; ++y) {
                const int oy = y0 + y;
                if (oy >= Hf || ox >= Wf) continue;
                const float val = apply_act(acc[y][e] * sc + sh, p.act, p.act_param, al);
                const size_t idx = p.out_nchw ? (((size_t)n * p.Cout + e) * Hf + oy) * Wf + ox
                                              : (((size_t)n * Hf + oy) * Wf + ox) * p.out_cs + p.out_co + e;
                if (p.out_bf16) ((bf16_t*)p.y)[idx] = f2bf(val);
                else ((float*)p.y)[idx] = val;
            }
        }
    }
}

// OIHW f32 -> [tap][half][lane][8] bf16; MFMA A-row r = output channel r (rows >= Cout are zero)
__global__ void thin_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 18 * 64 * 8) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, f = idx >> 9;
    const int tap = f >> 1, ks = f & 1;
    const int r = lane & 15, qq = lane >> 4;
    const int ci = ks * 32 + 8 * qq + j;
    out[idx] = r < Cout ? f2bf(w[((size_t)r * 64 + ci) * 9 + tap]) : (bf16_t)0;
}

bool thin_supported(const srmi_conv_desc& d) {
    if (d.compute != SRMI_COMPUTE_BF16) return false;
    if (d.kh != 3 || d.kw != 3 || d.stride != 1 || d.pad != 1 || d.Cin != 64) return false;
    if (d.Cout < 1 || d.Cout > 4 || d.shuffle != 1) return false;
    if (d.in_dtype != SRMI_BF16 || d.in_layout != SRMI_NHWC) return false;
    const int in_cs = d.in_cstride ? d.in_cstride : d.Cin;
    if (in_cs % 8 != 0) return false;
    return true;
}

size_t thin_packed_bytes(const srmi_conv_desc&) { return 18 * 64 * 8 * sizeof(bf16_t); }

int thin_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s) {
    hipLaunchKernelGGL(thin_pack_kernel, dim3(36), dim3(256), 0, s, w_oihw, (bf16_t*)w_packed, d.Cout);
    return hip_check(hipGetLastError(), "thin_pack_kernel");
}

int thin_launch(const srmi_conv_args& a, const ConvGeom& g, hipStream_t s) {
    const srmi_conv_desc& d = a.d;
    if (((uintptr_t)a.x | (uintptr_t)a.w) & 15) {
        set_error("c64_thin_bf16: x and w must be 16-byte aligned");
        return SRMI_EINVAL;
    }
    if (a.residual) { set_error("c64_thin_bf16: residual not supported"); return SRMI_EUNSUPPORTED; }
    ThinParams p;
    p.x = (const bf16_t*)a.x; p.w = (const bf16_t*)a.w; p.scale = a.scale; p.shift = a.shift; p.alpha = a.alpha;
    p.y = a.y;
    p.N = d.N; p.H = d.H; p.W = d.W; p.Cout = d.Cout; p.in_cs = g.in_cs; p.out_cs = g.out_cs; p.out_co = d.out_coffset;
    p.act = d.act; p.act_param = d.act_param; p.alpha_pc = d.alpha_per_channel;
    p.out_nchw = d.out_layout == SRMI_NCHW; p.out_bf16 = d.out_dtype == SRMI_BF16;
    if (d.N == 0 || d.H == 0 || d.W == 0) return SRMI_OK;
    p.tiles_x = (d.W + 63) / 64; p.tiles_y = (d.H + 5) / 6;
    const long long nb = (long long)d.N * p.tiles_x * p.tiles_y;
    if (nb > 0x7fffffffLL) { set_error("c64_thin_bf16: grid too large"); return SRMI_EINVAL; }
    hipLaunchKernelGGL((conv_thin_kernel<6, 4>), dim3((unsigned)nb), dim3(256), 0, s, p);
    return hip_check(hipGetLastError(), "conv_thin_kernel");
}

}  // namespace srmi
