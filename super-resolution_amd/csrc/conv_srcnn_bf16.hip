// The two HBM-bound layer shapes of SRCNN-style networks (SRCNN/srcnn.py:205-210; BASELINE config 2's 9-1-5 variant),
// NHWC bf16 activations, bf16 MFMA with f32 accumulation:
//
//   pw_bf16     1x1 conv, Cin in {32, 64, 128} -> Cout in {16, 32, 64}, + scale/shift + (none | ReLU | LeakyReLU | PReLU).
//               A streaming GEMM: a wave takes 16-pixel tiles, the pixels' channels go from global memory straight into
//               the MFMA B operand (16 bytes per lane, every fetched byte used), the weights live in registers as A
//               fragments, outputs leave as 8-byte bf16x4 stores.  No LDS.  Algorithmic bytes: 2 (Cin + Cout) per pixel.
//
//   c1out_bf16  k x k conv (k = 3 or 5), Cin in {32, 64} -> ONE channel, + scale/shift + any pointwise activation, f32 or
//               bf16 out.  With one output channel an implicit GEMM has nothing to put on the MFMA M axis, so the M axis
//               carries 16 - k + 1 neighbouring OUTPUT COLUMNS instead: for kernel row dy and input pixel column j of a
//               16-pixel block, A_dy,j[m][c] = w[dy][j - m][c] (zero outside the k taps) is a banded (Toeplitz) operand,
//               and D[m][n] += A_dy,j x B_j sums the k taps of a kernel row INSIDE the MFMA.  The N axis is 16 independent
//               column blocks of the same image row, so a wave covers 16 (16 - k + 1) output columns and streams down a
//               band of rows: every input row is loaded ONCE (global -> B operand, no LDS staging) and contributes to k
//               output rows held in k rotating accumulators; a finished row is stored, the others shift.  The banded
//               operands of all (dy, j) are one small LDS table (rows j - m + 15 of [dy][31][Cin], 10-20 KB) read with
//               a per-lane base and immediate offsets, conflict-free.  No atomics, no cross-lane reduction, every output
//               is one fixed-order f32 sum: same bits for every band / strip split.  Algorithmic bytes: 2 Cin + 4 per pixel.
//
// k/16 of the MFMA work is useful, which does not matter: both kernels are bound by the HBM stream, not by MFMA issue.
#include "c64_common.h"

namespace srmi {

// ======================================================= pw_bf16 =======================================================

struct PwParams {
    const bf16_t* x; const bf16_t* w; const float* scale; const float* shift; const float* alpha;
    const void* res; bf16_t* y;
    long long P;               // pixels (N * H * W)
    int in_cs, out_cs, out_co, res_cs, res_bf16;
    int st16;                  // y, out_cs, out_co allow 16-byte stores (else 8-byte pieces)
    int act; float act_param; int alpha_pc;
};

// Output channel of MFMA row r of M tile mt.  A lane (pixel pl, group q) gets rows 4 q .. 4 q + 3 of every M tile; the
// rows are permuted at pack time so that these values are CONSECUTIVE channels: 16 channels = one 8-byte store per lane
// and 32 contiguous bytes per pixel; 32 channels = 8 q .. 8 q + 7, one 16-byte store per lane and the pixel's whole 64
// bytes from its four lanes (a wave instruction writes 1 KB without a gap); 64 channels = 8 q .. 8 q + 7 and
// 32 + 8 q .. + 7, two 16-byte stores of 64 contiguous bytes per pixel each.
__host__ __device__ __forceinline__ int pw_row_to_ch(int MT, int mt, int r) {
    if (MT == 1) return r;
    if (MT == 2) return 8 * (r >> 2) + 4 * mt + (r & 3);
    return 32 * (mt >> 1) + 8 * (r >> 2) + 4 * (mt & 1) + (r & 3);
}

// weights -> A fragments [mt][ks][lane][8]: row (lane & 15) of M tile mt = channel pw_row_to_ch, k = 32 ks + 8 (lane >> 4) + j
__global__ void pw_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cin, int Cout) {
    const int KS = Cin / 32, MT = Cout / 16;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= MT * KS * 512) return;
    const int j = i & 7, lane = (i >> 3) & 63, f = i >> 9;
    const int ks = f % KS, mt = f / KS;
    out[i] = f2bf(w[(size_t)pw_row_to_ch(MT, mt, lane & 15) * Cin + 32 * ks + 8 * (lane >> 4) + j]);
}

template <int KS, int MT>
__global__ __launch_bounds__(256) void pw_kernel(PwParams p) {
    constexpr int U = (KS * MT > 8) ? 2 : 4;          // pixel tiles in flight per wave
    const int lane = threadIdx.x & 63;
    const int pl = lane & 15, q = lane >> 4;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * 4;
    const long long ntile = (p.P + 15) >> 4;

    bf16x8 a[MT][KS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[mt][ks] = *(const bf16x8*)(p.w + ((size_t)(mt * KS + ks) * 64 + lane) * 8);
    // this lane's output channels of M tile mt: pw_row_to_ch(MT, mt, 4 q) + e
    f32x4 sc[MT], sh[MT], sl[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int c = pw_row_to_ch(MT, mt, 4 * q);
        sc[mt] = *(const f32x4*)(p.scale + c);
        sh[mt] = *(const f32x4*)(p.shift + c);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            sl[mt][e] = p.act == SRMI_ACT_PRELU ? p.alpha[p.alpha_pc ? c + e : 0]
                      : p.act == SRMI_ACT_LRELU ? p.act_param : p.act == SRMI_ACT_RELU ? 0.f : 1.f;
    }

    for (long long t0 = wave * U; t0 < ntile; t0 += nwaves * U) {
        bf16x8 b[U][KS];
        long long pix[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pix[u] = (t0 + u) * 16 + pl;
            const long long pc = pix[u] < p.P ? pix[u] : p.P - 1;
            const bf16_t* src = p.x + pc * p.in_cs + 8 * q;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) b[u][ks] = __builtin_nontemporal_load((const bf16x8*)(src + 32 * ks));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][ks], b[u][ks], acc[mt], 0, 0, 0);
            }
            if (pix[u] < p.P) {
                bf16_t* dst = p.y + pix[u] * p.out_cs + p.out_co;
                u32x2 o[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int c = pw_row_to_ch(MT, mt, 4 * q);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float z = __builtin_fmaf(acc[mt][e], sc[mt][e], sh[mt][e]);
                        v[e] = z >= 0.f ? z : z * sl[mt][e];
                    }
                    if (p.res) {                       // the residual follows the activation (srmi.h)
                        const size_t ro = (size_t)pix[u] * p.res_cs + c;
                        if (p.res_bf16) {
                            const u32x2 r = *(const u32x2*)((const bf16_t*)p.res + ro);
                            v[0] += __builtin_bit_cast(float, r[0] << 16); v[1] += __builtin_bit_cast(float, r[0] & 0xffff0000u);
                            v[2] += __builtin_bit_cast(float, r[1] << 16); v[3] += __builtin_bit_cast(float, r[1] & 0xffff0000u);
                        } else {
                            v += *(const f32x4*)((const float*)p.res + ro);
                        }
                    }
                    o[mt] = (u32x2){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                }
                if constexpr (MT == 1) {
                    *(u32x2*)(dst + 4 * q) = o[0];
                } else {
#pragma unroll
                    for (int h = 0; h < MT / 2; ++h)       // channels 32 h + 8 q .. + 7: M tiles 2 h, 2 h + 1
                        if (p.st16) *(u32x4*)(dst + 32 * h + 8 * q) = (u32x4){o[2 * h][0], o[2 * h][1], o[2 * h + 1][0], o[2 * h + 1][1]};
                        else { *(u32x2*)(dst + 32 * h + 8 * q) = o[2 * h]; *(u32x2*)(dst + 32 * h + 8 * q + 4) = o[2 * h + 1]; }
                }
            }
        }
    }
}

bool pw_supported(const srmi_conv_desc& d) {
    if (dbg_flags() & (1 << 12)) return false;                       // A-B switch: the generic kernel
    if (d.compute != SRMI_COMPUTE_BF16) return false;
    if (d.kh != 1 || d.kw != 1 || d.stride != 1 || d.pad != 0 || d.shuffle != 1) return false;
    if (d.Cin != 32 && d.Cin != 64 && d.Cin != 128) return false;
    if (d.Cout != 16 && d.Cout != 32 && d.Cout != 64) return false;
    if (d.in_dtype != SRMI_BF16 || d.in_layout != SRMI_NHWC || d.out_dtype != SRMI_BF16 || d.out_layout != SRMI_NHWC) return false;
    if (d.act != SRMI_ACT_NONE && d.act != SRMI_ACT_RELU && d.act != SRMI_ACT_LRELU && d.act != SRMI_ACT_PRELU) return false;
    const int in_cs = d.in_cstride ? d.in_cstride : d.Cin, out_cs = d.out_cstride ? d.out_cstride : d.Cout;
    if (in_cs % 8 != 0 || out_cs % 4 != 0 || d.out_coffset % 4 != 0) return false;
    if (d.res_cstride % 4 != 0) return false;
    return true;
}

size_t pw_packed_bytes(const srmi_conv_desc& d) { return (size_t)d.Cin * d.Cout * 2; }

int pw_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s) {
    const int n = d.Cin * d.Cout;
    hipLaunchKernelGGL(pw_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w_oihw, (bf16_t*)w_packed, d.Cin, d.Cout);
    return hip_check(hipGetLastError(), "pw_pack_kernel");
}

int pw_launch(const srmi_conv_args& a, const ConvGeom& g, hipStream_t s) {
    const srmi_conv_desc& d = a.d;
    if ((((uintptr_t)a.x | (uintptr_t)a.w | (uintptr_t)a.scale | (uintptr_t)a.shift) & 15) || ((uintptr_t)a.y & 7)) {
        set_error("pw_bf16: x, w, scale, shift must be 16-byte aligned, y 8-byte aligned");
        return SRMI_EINVAL;
    }
    if (a.residual && ((uintptr_t)a.residual & (d.res_dtype == SRMI_BF16 ? 7 : 15))) {
        set_error("pw_bf16: residual must be 8-byte (bf16) / 16-byte (f32) aligned");
        return SRMI_EINVAL;
    }
    PwParams p;
    p.res = a.residual; p.res_cs = g.res_cs; p.res_bf16 = d.res_dtype == SRMI_BF16;
    p.x = (const bf16_t*)a.x; p.w = (const bf16_t*)a.w; p.scale = a.scale; p.shift = a.shift; p.alpha = a.alpha;
    p.y = (bf16_t*)a.y;
    p.P = (long long)d.N * d.H * d.W;
    p.in_cs = g.in_cs; p.out_cs = g.out_cs; p.out_co = d.out_coffset;
    p.act = d.act; p.act_param = d.act_param; p.alpha_pc = d.alpha_per_channel;
    p.st16 = (((uintptr_t)a.y & 15) == 0 && g.out_cs % 8 == 0 && d.out_coffset % 8 == 0) ? 1 : 0;
    if (p.P == 0) return SRMI_OK;
    const int KS = d.Cin / 32, MT = d.Cout / 16;
    const int U = (KS * MT > 8) ? 2 : 4;
    const long long ntile = (p.P + 15) / 16;
    long long nb = (ntile + 4 * U - 1) / (4 * U);
    if (nb > 4096) nb = 4096;                 // 16 workgroups per CU: enough loads in flight, the rest is the grid-stride loop
    const dim3 grid((unsigned)nb), blk(256);
#define SRMI_PW(KS_, MT_) if (KS == KS_ && MT == MT_) hipLaunchKernelGGL((pw_kernel<KS_, MT_>), grid, blk, 0, s, p)
    SRMI_PW(1, 1); SRMI_PW(1, 2); SRMI_PW(1, 4); SRMI_PW(2, 1); SRMI_PW(2, 2); SRMI_PW(2, 4); SRMI_PW(4, 1); SRMI_PW(4, 2); SRMI_PW(4, 4);
#undef SRMI_PW
    return hip_check(hipGetLastError(), "pw_kernel");
}

// ===================================================== c1out_bf16 ======================================================

struct C1Params {
    const bf16_t* x; const bf16_t* w; const float* scale; const float* shift; const float* alpha;
    void* y;
    int N, H, W, in_cs;
    int rb, bands, strips;     // rows per band, bands per image, column strips per row
    int act; float act_param;
    int out_bf16, vec4;
};

// table [dy][ks][31][32]: row 15 + d holds tap (dy, d) for d in [0, K), the other rows are zero
__global__ void c1_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cin, int K) {
    const int KS = Cin / 32;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * KS * 31 * 32) return;
    const int c = i & 31, row = (i >> 5) % 31, f = (i >> 5) / 31;
    const int ks = f % KS, dy = f / KS;
    const int dx = row - 15;
    out[i] = (dx >= 0 && dx < K) ? f2bf(w[((size_t)(32 * ks + c) * K + dy) * K + dx]) : (bf16_t)0;
}

template <int I, int N>
__device__ __forceinline__ void sched_pairs() {
    if constexpr (I < N) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        sched_pairs<I + 1, N>();
    }
}

template <int K, int KS>
__global__ __launch_bounds__(256) void c1_kernel(C1Params p) {
    constexpr int PAD = K / 2;
    constexpr int BO = 16 - K + 1;                     // output columns per 16-pixel block
    constexpr int SW = 16 * BO;                        // output columns per wave
    constexpr int TAB = K * KS * 31 * 64;              // bytes
    __shared__ __attribute__((aligned(16))) unsigned char tab[TAB];
    for (int i = threadIdx.x; i < TAB / 16; i += 256) ((u32x4*)tab)[i] = ((const u32x4*)p.w)[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int nl = lane & 15, q = lane >> 4;
    const long long task = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long ntask = (long long)p.N * p.bands * p.strips;
    if (task >= ntask) return;
    const int strip = (int)(task % p.strips);
    const int band = (int)((task / p.strips) % p.bands);
    const int n = (int)(task / ((long long)p.strips * p.bands));
    const int r0 = band * p.rb, r1 = min(r0 + p.rb, p.H);
    const int xo = strip * SW + BO * nl;               // first output column of this lane's block
    const int xb = xo - PAD;                           // its first input column (may be negative at the left edge)
    const int stride = p.in_cs * 2;                    // bytes per pixel
    const unsigned char* img = (const unsigned char*)(p.x + (size_t)n * p.H * p.W * p.in_cs);
    const unsigned row_bytes = (unsigned)p.W * stride;
    // A operand: row (j + 15 - m) of table (dy, ks); m = this lane's output column inside the block
    const unsigned char* abase = tab + (15 - nl) * 64 + q * 16;

    // A step = one input row, one PART of its 16 block columns: all of them with 32 channels; with 64 channels the
    // columns 0..7 or 8..15 with BOTH channel halves, so that the two 64-byte halves of a pixel's 128-byte line are
    // fetched back to back (by K half they would be fetched a whole step apart: every line twice through the L1).
    constexpr int JP = 16 / KS;                        // block columns per step; 16 operand registers either way
    auto load_row = [&](bf16x8 (&b)[16], int r, int part) {
        const bool ok = r >= 0 && r < p.H;
        const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(img + (size_t)(ok ? r : 0) * row_bytes), 0, ok ? row_bytes : 0u, 0x00020000);
#pragma unroll
        for (int jj = 0; jj < JP; ++jj)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int j = part * JP + jj;
                const int xin = xb + j;
                unsigned off = (unsigned)(xin * stride + 64 * ks + 16 * q);
                if (j < PAD && xin < 0) off = 0x80000000u;     // left of the image: zero (right of it: past num_records)
                b[jj * KS + ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(res, off, 0, 0));
            }
    };

    f32x4 acc[K];
#pragma unroll
    for (int dy = 0; dy < K; ++dy) acc[dy] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float sc0 = p.scale[0], sh0 = p.shift[0];
    const float al0 = p.act == SRMI_ACT_PRELU ? p.alpha[0] : 0.f;

    // one step: k x 16 banded MFMAs; after the last part of a row the oldest accumulator is a finished row
    auto compute = [&](const bf16x8 (&b)[16], int ir, int part, bool last) {
#pragma unroll
        for (int dy = 0; dy < K; ++dy)
#pragma unroll
            for (int jj = 0; jj < JP; ++jj)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int j = part * JP + jj;
                    const bf16x8 a = *(const bf16x8*)(abase + ((dy * KS + ks) * 31 + j) * 64);
                    acc[dy] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[jj * KS + ks], acc[dy], 0, 0, 0);
                }
        // keep only a few banded operands in flight: {4 LDS reads} {1 MFMA, 1 LDS read} x (16 K - 4) {4 MFMAs}
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        sched_pairs<0, 16 * K - 4>();
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        if (!last) return;
        const int o = r0 + ir - (K - 1);               // acc[K - 1] now holds output row o
        if (o >= r0 && o < r1) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act(acc[K - 1][e] * sc0 + sh0, p.act, p.act_param, al0);
            const int x = xo + 4 * q;
            const size_t idx = ((size_t)n * p.H + o) * p.W + x;
            if (p.out_bf16) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * q + e < BO && x + e < p.W) ((bf16_t*)p.y)[idx + e] = f2bf(v[e]);
            } else if (p.vec4 && BO % 4 == 0 && 4 * q + 3 < BO && x + 3 < p.W) {
                *(f32x4*)((float*)p.y + idx) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * q + e < BO && x + e < p.W) ((float*)p.y)[idx + e] = v[e];
            }
        }
#pragma unroll
        for (int dy = K - 1; dy > 0; --dy) acc[dy] = acc[dy - 1];
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };

    // steps = (input row, part), double-buffered by hand: the loads of step s + 1 are in flight during the MFMAs of step s
    const int nrow = (r1 - r0) + K - 1;                // input rows r0 - PAD .. r1 + PAD - 1
    const int S = nrow * KS;
    bf16x8 b0[16], b1[16];
    load_row(b0, r0 - PAD, 0);
    int s = 0;
    for (; s + 1 < S; s += 2) {
        // s is even: part 0 in b0, part KS - 1 in b1 (KS = 1: two rows)
        load_row(b1, r0 - PAD + (s + 1) / KS, KS - 1);
        compute(b0, s / KS, 0, KS == 1);
        if (s + 2 < S) load_row(b0, r0 - PAD + (s + 2) / KS, 0);
        compute(b1, (s + 1) / KS, KS - 1, true);
    }
    if (s < S) compute(b0, s / KS, 0, true);           // (odd step count: KS == 1 only)
}

bool c1_supported(const srmi_conv_desc& d) {
    if (dbg_flags() & (1 << 12)) return false;
    if (d.compute != SRMI_COMPUTE_BF16 || d.act == SRMI_ACT_ADD_RELU) return false;
    if (d.kh != d.kw || (d.kh != 3 && d.kh != 5) || d.stride != 1 || d.pad != d.kh / 2 || d.shuffle != 1) return false;
    if (d.Cout != 1 || (d.Cin != 32 && d.Cin != 64)) return false;
    if (d.in_dtype != SRMI_BF16 || d.in_layout != SRMI_NHWC) return false;
    if (d.out_cstride > 1 || d.out_coffset) return false;
    const int in_cs = d.in_cstride ? d.in_cstride : d.Cin;
    if (in_cs % 8 != 0) return false;
    if ((double)d.H * d.W * in_cs * 2.0 >= 2147483000.0) return false;      // 32-bit byte offsets inside one image
    return true;
}

size_t c1_packed_bytes(const srmi_conv_desc& d) { return (size_t)d.kh * (d.Cin / 32) * 31 * 64; }

int c1_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s) {
    const int n = d.kh * (d.Cin / 32) * 31 * 32;
    hipLaunchKernelGGL(c1_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w_oihw, (bf16_t*)w_packed, d.Cin, d.kh);
    return hip_check(hipGetLastError(), "c1_pack_kernel");
}

int c1_launch(const srmi_conv_args& a, const ConvGeom& g, hipStream_t s) {
    const srmi_conv_desc& d = a.d;
    if (((uintptr_t)a.x | (uintptr_t)a.w) & 15) { set_error("c1out_bf16: x and w must be 16-byte aligned"); return SRMI_EINVAL; }
    if (a.residual) { set_error("c1out_bf16: residual not supported"); return SRMI_EUNSUPPORTED; }
    C1Params p;
    p.x = (const bf16_t*)a.x; p.w = (const bf16_t*)a.w; p.scale = a.scale; p.shift = a.shift; p.alpha = a.alpha;
    p.y = a.y;
    p.N = d.N; p.H = d.H; p.W = d.W; p.in_cs = g.in_cs;
    p.act = d.act; p.act_param = d.act_param;
    p.out_bf16 = d.out_dtype == SRMI_BF16;
    p.vec4 = (!p.out_bf16 && ((uintptr_t)a.y & 15) == 0 && (d.W & 3) == 0) ? 1 : 0;
    if (d.N == 0 || d.H == 0 || d.W == 0) return SRMI_OK;
    const int SW = 16 * (16 - d.kh + 1);
    p.strips = (d.W + SW - 1) / SW;
    // rows per band: every band re-reads k - 1 halo rows (through L2: its neighbours stream them at the same time), so
    // long bands are cheaper per row, and the waves should fill the chip's 1024 SIMDs in whole rounds.  Measured at
    // 32 x 384 x 384 (64 column strips): 12 rows = 2048 waves 82 / 182 us (32 / 64 channels), 16 rows = 1536 waves 95 / 198,
    // 24 rows = 1024 waves 78 / 162, 48 rows = 512 waves 122 / 219: one wave per SIMD with the longer band wins.
    const long long cols = (long long)d.N * p.strips;
    long long nbands = (1024 + cols - 1) / cols;
    int rb = (int)((d.H + nbands - 1) / nbands);
    if (rb < 8) rb = 8;
    if (rb > d.H) rb = d.H;
    if (dbg_thin_rows() > 0) rb = dbg_thin_rows() > d.H ? d.H : dbg_thin_rows();      // (investigation: SRMI_THIN_RB)
    p.rb = rb;
    p.bands = (d.H + rb - 1) / rb;
    const long long ntask = cols * p.bands;
    const long long nb = (ntask + 3) / 4;
    if (nb > 0x7fffffffLL) { set_error("c1out_bf16: grid too large"); return SRMI_EINVAL; }
    const dim3 grid((unsigned)nb), blk(256);
    const int KS = d.Cin / 32;
    if (d.kh == 5 && KS == 1) hipLaunchKernelGGL((c1_kernel<5, 1>), grid, blk, 0, s, p);
    else if (d.kh == 5) hipLaunchKernelGGL((c1_kernel<5, 2>), grid, blk, 0, s, p);
    else if (KS == 1) hipLaunchKernelGGL((c1_kernel<3, 1>), grid, blk, 0, s, p);
    else hipLaunchKernelGGL((c1_kernel<3, 2>), grid, blk, 0, s, p);
    return hip_check(hipGetLastError(), "c1_kernel");
}

}  // namespace srmi
