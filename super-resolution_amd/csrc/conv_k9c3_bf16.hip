// conv9x9 / stride 1 / pad 4 / Cin <= 4 / Cout = 64 on v_mfma_f32_16x16x32_bf16: the generator's first
// layer `conv01` (3 -> 64, 9x9, PReLU; P/srgan_model.py:11).  Reads the caller's tensor as it comes
// (NCHW f32 - the module-edge contract, P/dataset.py:46-47 - or NHWC f32/bf16) and writes the NHWC bf16
// activation layout of the rest of the network, so there is no separate re-layout pass.
//
// Implicit GEMM with a padded K so that MFMA operands are plain aligned LDS reads:
//   K = 9 tap rows x 64, where one tap row is 16 tap columns (9 real + 7 zero-weight) x 4 channels
//   (3 real + 1 zero).  The input tile is staged in LDS as [row][col][4] bf16 (8 B per pixel), so the 8
//   consecutive k of a lane's B fragment are two neighbouring pixels = 16 contiguous bytes (read as
//   2 x ds_read_b64: the start is only 8-B aligned).  42 % of the MFMA work is padding, which costs
//   ~3 us on a layer that is 0.8 % of the network's FLOPs - the price of keeping the operand path
//   identical to the 64-channel kernels.
//   KS = 3 (round 3): the same kernel for 3x3 / pad 1 convs of a <= 3-channel image (the RRDB generator's head conv,
//   ESRGAN/untitled0.py:41; the first conv of D / VGG19 in bf16 inference): 3 tap rows, and only the lower half of a tap
//   row's 16 column slots can hold a real tap, so the K loop is 3 steps and the weights 12 KiB.
//   All 72 weight fragments (72 KiB, packed in MFMA order on the host) are brought into LDS once per
//   workgroup by LDS-DMA while the input tile is being staged; the K loop then has no barrier at all.
#include "c64_common.h"

namespace srmi {

struct K9Params {
    const void* x; const bf16_t* w;
    C64Epi epi;
    int N, H, W, Cin, in_cs, in_nchw, in_bf16;
    int tiles_x, tiles_y;
};

// per kernel size KS: K steps = KS tap rows x NH column halves (the upper half, tap columns 8..15, is all zero for KS <= 8)
__host__ __device__ constexpr int k9_nh(int KS) { return KS > 8 ? 2 : 1; }
__host__ __device__ constexpr int k9_w_bytes(int KS) { return KS * k9_nh(KS) * 4 * 1024; }

template <int TH, int TWT, int KS = 9>
__global__ __launch_bounds__(256, 1) void conv_k9c3_kernel(K9Params p) {
    constexpr int NH = k9_nh(KS), NS = KS * NH, PAD = KS / 2, WB = k9_w_bytes(KS);
    static_assert(WB % 4096 == 0, "weights split evenly over 4 waves in 1-KiB DMA instructions");
    constexpr int TW = TWT * 16;
    constexpr int NPT = TH * TWT;
    static_assert(NPT % 4 == 0, "pixel tiles must split evenly over 4 waves");
    constexpr int PT = NPT / 4;
    constexpr int ROWP = TW + 16;                 // tile cols 0 .. TW+15 (taps 9..15 read real-or-zero data x zero weight)
    constexpr int ROWS = TH + KS - 1;
    constexpr int IN_BYTES = ROWS * ROWP * 8;
    __shared__ __attribute__((aligned(16))) unsigned char lds[WB + IN_BYTES + 192 * 4];
    unsigned char* const wl = lds;
    unsigned char* const tile = lds + WB;
    float* const par = (float*)(lds + WB + IN_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, q = lane >> 4;

    int b = blockIdx.x;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- weights: 72 KiB (KS = 9) by LDS-DMA, 18 KiB per wave (inline asm: see conv_c64_bf16.hip) ----------
    {
        constexpr int WK = WB / 4096;                  // KiB per wave
        const unsigned char* src = (const unsigned char*)p.w + wave * WK * 1024 + lane * 16;
        const unsigned dst0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)wl + (unsigned)(wave * WK * 1024));
        if constexpr (WK % 2 == 0) {
#pragma unroll
            for (int i = 0; i < WK; i += 2) {
                unsigned keep;
                asm volatile(
                    "s_mov_b32 %0, m0\n\t"
                    "s_mov_b32 m0, %3\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %1, off\n\t"
                    "s_add_u32 m0, m0, 0x400\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %2, off\n\t"
                    "s_mov_b32 m0, %0"
                    : "=&s"(keep)
                    : "v"(src + i * 1024), "v"(src + (i + 1) * 1024), "s"(dst0 + (unsigned)(i * 1024))
                    : "memory", "scc");
            }
        } else {
#pragma unroll
            for (int i = 0; i < WK; ++i) {
                unsigned keep;
                asm volatile(
                    "s_mov_b32 %0, m0\n\t"
                    "s_mov_b32 m0, %2\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %1, off\n\t"
                    "s_mov_b32 m0, %0"
                    : "=&s"(keep)
                    : "v"(src + i * 1024), "s"(dst0 + (unsigned)(i * 1024))
                    : "memory");
            }
        }
    }

    C64ParRegs pregs;
    c64_params_load(pregs, p.epi, 1, tid);
    // ---- input tile: (TH+KS-1) x (TW+16) pixels -> [row][col][4] bf16, zero outside the image ---------
    {
        constexpr int NPIX = ROWS * ROWP;
        constexpr int NIT = (NPIX + 255) / 256;
        float v[NIT][3];
        const size_t HW = (size_t)p.H * p.W;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = min(tid + it * 256, NPIX - 1);
            const int row = idx / ROWP, col = idx - row * ROWP;
            const int gy = min(max(y0 - PAD + row, 0), p.H - 1), gx = min(max(x0 - PAD + col, 0), p.W - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int cc = c < p.Cin ? c : 0;
                const size_t gi = p.in_nchw ? ((size_t)n * p.Cin + cc) * HW + (size_t)gy * p.W + gx
                                            : (((size_t)n * p.H + gy) * p.W + gx) * p.in_cs + cc;
                v[it][c] = p.in_bf16 ? bf2f(((const bf16_t*)p.x)[gi]) : ((const float*)p.x)[gi];
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256;
            const int row = idx / ROWP, col = idx - row * ROWP;
            const int gy = y0 - PAD + row, gx = x0 - PAD + col;
            const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            u32x2 o;
            o[0] = ok ? pack_bf2(v[it][0], p.Cin > 1 ? v[it][1] : 0.f) : 0u;
            o[1] = ok ? pack_bf2(p.Cin > 2 ? v[it][2] : 0.f, 0.f) : 0u;
            if (idx < NPIX) *(u32x2*)(tile + idx * 8) = o;
        }
    }

    c64_params_store(par, pregs, 1, tid);
    int prow[PT], pcol[PT];
#pragma unroll
    for (int k = 0; k < PT; ++k) {
        const int t = wave * PT + k;
        prow[k] = t / TWT;
        pcol[k] = (t - prow[k] * TWT) * 16;
    }
    // weights landed (own DMA: vmcnt; others': barrier), tile and params written
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int boff[PT];   // byte offset of this lane's first pixel for (ky = 0, h = 0)
#pragma unroll
    for (int k = 0; k < PT; ++k) boff[k] = (prow[k] * ROWP + pcol[k] + pl + 2 * q) * 8;

    f32x4 acc[PT][4];
#pragma unroll
    for (int k = 0; k < PT; ++k)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[k][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    typedef union { u32x2 h[2]; bf16x8 v; } frag_t;
    bf16x8 afr[2][4];
    frag_t bfr[2][PT];
    const unsigned char* wlane = wl + lane * 16;
#pragma unroll
    for (int t = 0; t < 4; ++t) afr[0][t] = *(const bf16x8*)(wlane + t * 1024);
#pragma unroll
    for (int k = 0; k < PT; ++k) {
        bfr[0][k].h[0] = *(const u32x2*)(tile + boff[k]);
        bfr[0][k].h[1] = *(const u32x2*)(tile + boff[k] + 8);
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {     // K step j = (tap row ky = j / NH, tap-column half h = j % NH)
        if (j + 1 < NS) {
            const int kyn = (j + 1) / NH, hn = (j + 1) % NH;
#pragma unroll
            for (int t = 0; t < 4; ++t) afr[(j + 1) & 1][t] = *(const bf16x8*)(wlane + ((j + 1) * 4 + t) * 1024);
#pragma unroll
            for (int k = 0; k < PT; ++k) {
                const unsigned char* a = tile + boff[k] + (kyn * ROWP + 8 * hn) * 8;
                bfr[(j + 1) & 1][k].h[0] = *(const u32x2*)a;
                bfr[(j + 1) & 1][k].h[1] = *(const u32x2*)(a + 8);
            }
        }
#pragma unroll
        for (int k = 0; k < PT; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[j & 1][t], bfr[j & 1][k].v, acc[k][t], 0, 0, 0);
        if (j + 1 < NS) sched_interleave<0, 4 + 2 * PT, 4 * PT>();
    }

    c64_epilogue<PT>(acc, par, p.epi, 0, n, y0, x0, prow, pcol, pl, q);
}

// OIHW f32 -> [ky][h][t][lane][8] bf16:  k = 32h + 8(lane>>4) + j  ->  kx = 8h + 2(lane>>4) + (j>>2), ch = j&3
// (KS x KS taps; h runs over k9_nh(KS) column halves)
__global__ void k9c3_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cin, int KS) {
    const int NH = k9_nh(KS);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= KS * NH * 4 * 512) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, t = (idx >> 9) & 3;
    const int h = (idx >> 11) % NH, ky = (idx >> 11) / NH;
    const int r = lane & 15, qq = lane >> 4;
    const int co = c64_row_to_cl(t, r);
    const int kx = 8 * h + 2 * qq + (j >> 2), ch = j & 3;
    float v = 0.f;
    if (kx < KS && ch < Cin) v = w[(((size_t)co * Cin + ch) * KS + ky) * KS + kx];
    out[idx] = f2bf(v);
}

bool k9c3_supported(const srmi_conv_desc& d) {
    if (d.compute != SRMI_COMPUTE_BF16) return false;
    if (d.res_cstride && d.res_cstride != d.Cout / (d.shuffle * d.shuffle)) return false;   // dense residual only
    if (!((d.kh == 9 && d.kw == 9 && d.pad == 4) || (d.kh == 3 && d.kw == 3 && d.pad == 1)) || d.stride != 1) return false;
    if (d.Cin < 1 || d.Cin > 3 || d.Cout % 64 != 0 || d.Cout > 512 || d.shuffle != 1) return false;
    if (d.out_dtype != SRMI_BF16 || d.out_layout != SRMI_NHWC) return false;
    if (d.act == SRMI_ACT_TANH || d.act == SRMI_ACT_SIGMOID || d.act == SRMI_ACT_ADD_RELU) return false;
    const int out_cs = d.out_cstride ? d.out_cstride : d.Cout;
    if (out_cs % 8 != 0 || d.out_coffset % 8 != 0) return false;
    return true;
}

// Cout = 64 m runs as m launches of the 64-channel kernel (the input is 1-3 channels: re-reading it is free)
size_t k9c3_packed_bytes(const srmi_conv_desc& d) { return (size_t)(d.Cout / 64) * k9_w_bytes(d.kh); }

int k9c3_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s) {
    const int wb = k9_w_bytes(d.kh);
    for (int c = 0; c < d.Cout / 64; ++c)
        hipLaunchKernelGGL(k9c3_pack_kernel, dim3((wb / 2 + 255) / 256), dim3(256), 0, s,
                           w_oihw + (size_t)c * 64 * d.Cin * d.kh * d.kw, (bf16_t*)((unsigned char*)w_packed + (size_t)c * wb),
                           d.Cin, d.kh);
    return hip_check(hipGetLastError(), "k9c3_pack_kernel");
}

int k9c3_launch(const srmi_conv_args& a, const ConvGeom& g, hipStream_t s) {
    const srmi_conv_desc& d = a.d;
    if (((uintptr_t)a.y | (uintptr_t)a.w | (uintptr_t)a.residual) & 15) {
        set_error("k9c3_bf16: w, y and residual must be 16-byte aligned");
        return SRMI_EINVAL;
    }
    if (a.residual && d.res_dtype != SRMI_BF16) { set_error("k9c3_bf16: residual must be bf16"); return SRMI_EUNSUPPORTED; }
    for (int c = 0; c < d.Cout / 64; ++c) {
    K9Params p;
    p.x = a.x; p.w = (const bf16_t*)((const unsigned char*)a.w + (size_t)c * k9_w_bytes(d.kh));
    p.epi.scale = a.scale + 64 * c; p.epi.shift = a.shift + 64 * c;
    p.epi.alpha = (a.alpha && d.alpha_per_channel) ? a.alpha + 64 * c : a.alpha;
    p.epi.res = a.residual ? (const bf16_t*)a.residual + 64 * c : nullptr; p.epi.y = (bf16_t*)a.y;
    p.epi.H = d.H; p.epi.W = d.W; p.epi.Cout = d.Cout; p.epi.out_cs = g.out_cs; p.epi.out_co = d.out_coffset + 64 * c;
    p.epi.mode = C64_PLAIN;
    p.epi.act = d.act; p.epi.act_param = d.act_param; p.epi.alpha_pc = d.alpha_per_channel;
    p.N = d.N; p.H = d.H; p.W = d.W; p.Cin = d.Cin; p.in_cs = g.in_cs;
    p.in_nchw = d.in_layout == SRMI_NCHW; p.in_bf16 = d.in_dtype == SRMI_BF16;
    if (d.N == 0 || d.H == 0 || d.W == 0) return SRMI_OK;
    const bool big = d.W > 64 && d.H >= 6;
    if (big) {
        p.tiles_x = (d.W + 95) / 96; p.tiles_y = (d.H + 5) / 6;
        const long long nb = (long long)d.N * p.tiles_x * p.tiles_y;
        if (nb > 0x7fffffffLL) { set_error("k9c3_bf16: grid too large"); return SRMI_EINVAL; }
        if (d.kh == 9) hipLaunchKernelGGL((conv_k9c3_kernel<6, 6, 9>), dim3((unsigned)nb), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_k9c3_kernel<6, 6, 3>), dim3((unsigned)nb), dim3(256), 0, s, p);
    } else {
        p.tiles_x = (d.W + 31) / 32; p.tiles_y = (d.H + 3) / 4;
        const long long nb = (long long)d.N * p.tiles_x * p.tiles_y;
        if (nb > 0x7fffffffLL) { set_error("k9c3_bf16: grid too large"); return SRMI_EINVAL; }
        if (d.kh == 9) hipLaunchKernelGGL((conv_k9c3_kernel<4, 2, 9>), dim3((unsigned)nb), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_k9c3_kernel<4, 2, 3>), dim3((unsigned)nb), dim3(256), 0, s, p);
    }
    int rc = hip_check(hipGetLastError(), "conv_k9c3_kernel");
    if (rc) return rc;
    }
    return SRMI_OK;
}

}  // namespace srmi
