// 9x9 conv of a ONE-channel image to 64 g channels (SRCNN's first layer, SRCNN/srcnn.py:205-206): f32 plane in,
// NHWC bf16 out, + scale/shift + (none | ReLU | LeakyReLU | scalar PReLU).  bf16 MFMA, f32 accumulation.
//
// With one input channel the GEMM K axis is the taps themselves, packed densely: kernel row dy occupies K indices
// 10 dy .. 10 dy + 9 (nine taps and a zero), K = 90 of 96 (three 16 x 16 x 32 MFMA steps): 12 MFMAs per 16 pixels and 64
// output channels - the Cin <= 3 kernel (conv_k9c3_bf16.hip) pads every tap to four channels and spends 72.  The B
// operand (8 consecutive K indices of a pixel per lane = four PAIRS of horizontally adjacent pixels, never straddling a
// kernel row because rows take 10 slots) is gathered from a bf16 copy of the input halo tile in LDS with four 4-byte
// reads per MFMA step; the tile is kept twice, the second copy shifted by one pixel, so that every pair is
// dword-aligned in the copy that matches the parity of the lane's column.  The 12 read addresses are per-lane
// constants that only advance by one tile row per step.  The weights stay in registers as A
// fragments (48 VGPRs).  The M rows are PERMUTED at pack time so that a lane ends up with channels 8 q .. 8 q + 7
// and 32 + 8 q .. 32 + 8 q + 7 of its pixel: two 16-byte stores, each of which makes 64 contiguous bytes with the other
// three lanes of the pixel (whole 32-byte sectors; 16 x 8 tiles: more, shorter workgroups measured faster than 16 x 64 rows).
//
// The layer is bound by its output stream: 4 B in, 128 B out per pixel and 64 channels (algorithmic bytes).
//
// FUSED WITH A FOLLOWING 1x1 CONV (k9c1_kernel<2>, "k9c1pw_bf16": SRCNN 9-1-5, conv 9x9 1 -> 64, ReLU, conv 1x1 64 -> 32,
// ReLU): the 64-channel tensor - 128 B per pixel written and read back, 2/3 of the network's HBM traffic - never
// exists.  A lane's 16 activated, bf16-rounded channels of its pixel (8 q .. 8 q + 7 and 32 + 8 q .. + 7) ARE the B
// operand of the 1x1 conv's two K steps, as they stand in its registers; the 1x1 weights are pw_bf16's A fragments
// (conv_srcnn_bf16.hip: the same packed buffer), four more MFMAs per 16 pixels, and a lane stores channels 8 q .. 8 q + 7
// of the 32: one 16-byte store, a wave instruction writes 1 KiB without a gap.  Same values as the two launches, bit for
// bit (same bf16 rounding of the intermediate, same K order).  Algorithmic bytes: 4 B in, 64 B out per pixel.
#include <type_traits>

#include "c64_common.h"

namespace srmi {

struct K9c1Params {
    const float* x; const bf16_t* w; const float* scale; const float* shift; const float* alpha;
    bf16_t* y;
    int N, H, W, out_cs, out_co;
    int tiles_x, tiles_y;
    int act; float act_param;
    // the fused 1x1 conv (k9c1_kernel<2>): y above is unused, y2 [.., out_cs2] receives 32 channels at out_co2
    const bf16_t* w2; const float* scale2; const float* shift2; const float* alpha2;
    bf16_t* y2;
    int out_cs2, out_co2, act2, alpha_pc2; float act_param2;
};

constexpr int K9C1_TH = 8, K9C1_TW = 64, K9C1_LW = K9C1_TW + 10;      // (even; column +9 is read, weight 0)

// A fragments [group][mt][ks][lane][8]: lane = 16 kq + r, row r of M tile mt = channel 64 g + 32 (mt >> 1) + 8 (r >> 2) + 4 (mt & 1) + (r & 3),
// K index 32 ks + 8 kq + i = 10 dy + dx, zero for dx = 9 and from 90 on
__global__ void k9c1_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cout / 64 * 12 * 512) return;
    const int e = i & 7, lane = (i >> 3) & 63, f = i >> 9;
    const int ks = f % 3, mt = (f / 3) & 3, g = f / 12;
    const int r = lane & 15, kq = lane >> 4;
    const int ch = 64 * g + 32 * (mt >> 1) + 8 * (r >> 2) + 4 * (mt & 1) + (r & 3);
    const int kk = 32 * ks + 8 * kq + e;
    const int dy = kk / 10, dx = kk - 10 * dy;
    out[i] = (kk < 90 && dx < 9) ? f2bf(w[(size_t)ch * 81 + dy * 9 + dx]) : (bf16_t)0;
}

template <int MT2, int TH>      // MT2 = 0: the 9x9 conv alone; 2: + 1x1 conv 64 -> 32.  TH: rows per workgroup
__global__ __launch_bounds__(256, 3) void k9c1_kernel(K9c1Params p) {      // (3 workgroups per CU: 168 registers hold the fused form without spills; 2 -> 3 waves per SIMD: 138 -> 112 us; 4: 37 spills, 470 us)
    constexpr int TW = K9C1_TW, LH = TH + 8, LW = K9C1_LW;
    __shared__ __attribute__((aligned(16))) bf16_t tile[2 * LH * LW];      // [1]: the same tile shifted left by one pixel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 15, q = lane >> 4;
    int b = blockIdx.x;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int g = blockIdx.y;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- input halo tile -> bf16 in LDS (zero outside the image) ----
    const float* img = p.x + (size_t)n * p.H * p.W;
    for (int i = tid; i < LH * LW; i += 256) {
        const int r = i / LW, c = i - r * LW;
        const int gy = y0 - 4 + r, gx = x0 - 4 + c;
        float v = 0.f;
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) v = img[(size_t)gy * p.W + gx];
        const bf16_t h = f2bf(v);
        tile[i] = h;
        if (i > 0) tile[LH * LW + i - 1] = h;
    }
    if (tid == 0) tile[2 * LH * LW - 1] = 0;

    bf16x8 a[4][3];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) a[mt][ks] = *(const bf16x8*)(p.w + ((size_t)((g * 4 + mt) * 3 + ks) * 64 + lane) * 8);
    // this lane's 16 output channels: 64 g + 8 q + 0..7 (M tiles 0, 1) and 64 g + 32 + 8 q + 0..7 (M tiles 2, 3)
    f32x4 sc[4], sh[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        sc[mt] = *(const f32x4*)(p.scale + 64 * g + 32 * (mt >> 1) + 8 * q + 4 * (mt & 1));
        sh[mt] = *(const f32x4*)(p.shift + 64 * g + 32 * (mt >> 1) + 8 * q + 4 * (mt & 1));
    }
    const float slope = p.act == SRMI_ACT_PRELU ? p.alpha[0] : p.act == SRMI_ACT_LRELU ? p.act_param
                      : p.act == SRMI_ACT_RELU ? 0.f : 1.f;
    // the 1x1 conv: A fragments [mt][ks][lane][8] as pw_pack_kernel lays them out (Cin = 64: two K steps), and the
    // parameters of this lane's 8 output channels 8 q + 4 mt + e
    constexpr int M2 = MT2 ? MT2 : 1;
    bf16x8 a2[M2][2];
    f32x4 sc2[M2], sh2[M2], sl2v[M2];
    if constexpr (MT2 > 0) {
        static_assert(MT2 == 2, "32 output channels: a lane's 8 channels are one 16-byte store");
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a2[mt][ks] = *(const bf16x8*)(p.w2 + ((size_t)(mt * 2 + ks) * 64 + lane) * 8);
            const int c = 8 * q + 4 * mt;
            sc2[mt] = *(const f32x4*)(p.scale2 + c);
            sh2[mt] = *(const f32x4*)(p.shift2 + c);
            const float s0 = p.act2 == SRMI_ACT_LRELU ? p.act_param2 : p.act2 == SRMI_ACT_RELU ? 0.f : 1.f;
            sl2v[mt] = (f32x4){s0, s0, s0, s0};
            if (p.act2 == SRMI_ACT_PRELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) sl2v[mt][e] = p.alpha2[p.alpha_pc2 ? c + e : 0];
            }
        }
    }

    // ---- LDS byte address of the pixel pair at K indices 32 ks + 8 q + 2 i, + 1, for this lane's pixel (tile row 0,
    // column 16 wave + pl).  Element index e of the pair is even in copy 0 or odd (-> copy 1, where it sits at e - 1) ----
    unsigned addr[3][4];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = 32 * ks + 8 * q + 2 * i;
            const int dy = kk < 90 ? kk / 10 : 0, dx = kk < 90 ? kk - 10 * dy : 0;      // (zero weights: any valid address)
            const int e = dy * LW + 16 * wave + pl + dx;
            addr[ks][i] = (unsigned)((e & 1) ? (LH * LW + e - 1) * 2 : e * 2);
        }
    __syncthreads();

    const f32x2 sl2 = {slope, slope};
    const bool maxform = slope >= 0.f && slope <= 1.f;       // max(z, z * slope) == PReLU(z) there
    const int x = x0 + 16 * wave + pl;
    const int rows = min(TH, p.H - y0);
    const unsigned char* lds = (const unsigned char*)tile;
    auto body = [&](auto MAXF) {
    for (int k = 0; k < rows; ++k) {
        bf16x8 bfr[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            u32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *(const unsigned*)(lds + addr[ks][i] + k * (LW * 2));
            bfr[ks] = __builtin_bit_cast(bf16x8, v);
        }
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][ks], bfr[ks], acc[mt], 0, 0, 0);
        }
        u32x4 o[2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2 ac = {acc[mt][2 * h], acc[mt][2 * h + 1]};
                const f32x2 s2 = {sc[mt][2 * h], sc[mt][2 * h + 1]}, h2 = {sh[mt][2 * h], sh[mt][2 * h + 1]};
                const f32x2 z = decltype(MAXF)::value ? slope_act2_max(ac, s2, h2, sl2) : slope_act2(ac, s2, h2, sl2);
                o[mt >> 1][2 * (mt & 1) + h] = pack_bf2(z[0], z[1]);
            }
        if constexpr (MT2 > 0) {
            // o[0] / o[1] = channels 8 q .. + 7 and 32 + 8 q .. + 7 of this lane's pixel: the 1x1 conv's B operand, K steps 0 / 1
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, o[0]), b1 = __builtin_bit_cast(bf16x8, o[1]);
            u32x4 o2;
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) {
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[mt][0], b0, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[mt][1], b1, d, 0, 0, 0);
                f32x4 v;       // pw_kernel's epilogue arithmetic, to the bit
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float z = __builtin_fmaf(d[e], sc2[mt][e], sh2[mt][e]);
                    v[e] = z >= 0.f ? z : z * sl2v[mt][e];
                }
                o2[2 * mt] = pack_bf2(v[0], v[1]);
                o2[2 * mt + 1] = pack_bf2(v[2], v[3]);
            }
            if (x < p.W)
                *(u32x4*)(p.y2 + ((size_t)((size_t)n * p.H + y0 + k) * p.W + x) * p.out_cs2 + p.out_co2 + 8 * q) = o2;
        } else {
        // whole 128-byte lines per store instruction (store_lines_x8: lanes pl and pl ^ 8 swap a half)
        bf16_t* line = p.y + ((size_t)((size_t)n * p.H + y0 + k) * p.W + x) * p.out_cs + p.out_co + 64 * g;
        const int xa = x0 + 16 * wave + (pl & 7);
        store_lines_x8<true>(line, p.out_cs, pl, q, o[0], o[1], xa < p.W, xa + 8 < p.W);
        }
    }
    };
    if (maxform) body(std::true_type{}); else body(std::false_type{});
}

bool k9c1_supported(const srmi_conv_desc& d) {
    if (dbg_flags() & (1 << 12)) return false;                       // A-B switch: the Cin <= 3 kernel
    if (d.compute != SRMI_COMPUTE_BF16) return false;
    if (d.kh != 9 || d.kw != 9 || d.stride != 1 || d.pad != 4 || d.shuffle != 1) return false;
    if (d.Cin != 1 || d.Cout % 64 != 0 || d.Cout < 64 || d.Cout > 512) return false;
    if (d.in_dtype != SRMI_F32 || d.in_cstride > 1) return false;                 // one channel: NCHW == NHWC
    if (d.out_dtype != SRMI_BF16 || d.out_layout != SRMI_NHWC) return false;
    if (d.act != SRMI_ACT_NONE && d.act != SRMI_ACT_RELU && d.act != SRMI_ACT_LRELU && d.act != SRMI_ACT_PRELU) return false;
    if (d.act == SRMI_ACT_PRELU && d.alpha_per_channel) return false;
    const int out_cs = d.out_cstride ? d.out_cstride : d.Cout;
    if (out_cs % 8 != 0 || d.out_coffset % 8 != 0) return false;
    return true;
}

size_t k9c1_packed_bytes(const srmi_conv_desc& d) { return (size_t)d.Cout / 64 * 12 * 512 * 2; }

int k9c1_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s) {
    const int n = d.Cout / 64 * 12 * 512;
    hipLaunchKernelGGL(k9c1_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w_oihw, (bf16_t*)w_packed, d.Cout);
    return hip_check(hipGetLastError(), "k9c1_pack_kernel");
}

int k9c1_launch(const srmi_conv_args& a, const ConvGeom& g, hipStream_t s) {
    const srmi_conv_desc& d = a.d;
    if (((uintptr_t)a.w | (uintptr_t)a.y | (uintptr_t)a.scale | (uintptr_t)a.shift) & 15) {
        set_error("k9c1_bf16: w, y, scale, shift must be 16-byte aligned");
        return SRMI_EINVAL;
    }
    if (a.residual) { set_error("k9c1_bf16: residual not supported"); return SRMI_EUNSUPPORTED; }
    K9c1Params p;
    p.x = (const float*)a.x; p.w = (const bf16_t*)a.w; p.scale = a.scale; p.shift = a.shift; p.alpha = a.alpha;
    p.y = (bf16_t*)a.y;
    p.N = d.N; p.H = d.H; p.W = d.W; p.out_cs = g.out_cs; p.out_co = d.out_coffset;
    p.act = d.act; p.act_param = d.act_param;
    if (d.N == 0 || d.H == 0 || d.W == 0) return SRMI_OK;
    p.tiles_x = (d.W + K9C1_TW - 1) / K9C1_TW; p.tiles_y = (d.H + K9C1_TH - 1) / K9C1_TH;
    const long long nb = (long long)d.N * p.tiles_x * p.tiles_y;
    if (nb > 0x7fffffffLL) { set_error("k9c1_bf16: grid too large"); return SRMI_EINVAL; }
    hipLaunchKernelGGL((k9c1_kernel<0, K9C1_TH>), dim3((unsigned)nb, (unsigned)(d.Cout / 64)), dim3(256), 0, s, p);
    return hip_check(hipGetLastError(), "k9c1_kernel");
}

// conv1: 9x9, 1 -> 64 ("k9c1_bf16"); conv2: 1x1, 64 -> 32 on conv1's dense output ("pw_bf16"), no residuals
bool k9pw_supported(const srmi_conv_desc& d1, const srmi_conv_desc& d2) {
    if (dbg_flags() & (1 << 16)) return false;                       // ablation / A-B switch: two launches
    if (!k9c1_supported(d1) || !pw_supported(d2)) return false;
    if (d1.Cout != 64 || (d1.out_cstride && d1.out_cstride != 64) || d1.out_coffset) return false;
    if (d2.Cin != 64 || d2.Cout != 32 || (d2.in_cstride && d2.in_cstride != 64)) return false;
    if (d2.N != d1.N || d2.H != d1.H || d2.W != d1.W) return false;
    if (d2.act != SRMI_ACT_NONE && d2.act != SRMI_ACT_RELU && d2.act != SRMI_ACT_LRELU && d2.act != SRMI_ACT_PRELU) return false;
    const int out_cs = d2.out_cstride ? d2.out_cstride : d2.Cout;
    if (out_cs % 8 != 0 || d2.out_coffset % 8 != 0) return false;
    return true;
}

int k9pw_launch(const srmi_conv_args& a1, const srmi_conv_args& a2, hipStream_t s) {
    const srmi_conv_desc& d = a1.d;
    const srmi_conv_desc& d2 = a2.d;
    if (((uintptr_t)a1.w | (uintptr_t)a1.scale | (uintptr_t)a1.shift | (uintptr_t)a2.w | (uintptr_t)a2.y | (uintptr_t)a2.scale |
         (uintptr_t)a2.shift) & 15) {
        set_error("k9c1pw_bf16: w, scale, shift of both convs and y must be 16-byte aligned");
        return SRMI_EINVAL;
    }
    if (a1.residual || a2.residual) { set_error("k9c1pw_bf16: residuals are not supported"); return SRMI_EUNSUPPORTED; }
    K9c1Params p;
    memset(&p, 0, sizeof(p));
    p.x = (const float*)a1.x; p.w = (const bf16_t*)a1.w; p.scale = a1.scale; p.shift = a1.shift; p.alpha = a1.alpha;
    p.y = nullptr;
    p.N = d.N; p.H = d.H; p.W = d.W; p.out_cs = 64; p.out_co = 0;
    p.act = d.act; p.act_param = d.act_param;
    p.w2 = (const bf16_t*)a2.w; p.scale2 = a2.scale; p.shift2 = a2.shift; p.alpha2 = a2.alpha; p.y2 = (bf16_t*)a2.y;
    p.out_cs2 = d2.out_cstride ? d2.out_cstride : d2.Cout; p.out_co2 = d2.out_coffset;
    p.act2 = d2.act; p.alpha_pc2 = d2.alpha_per_channel; p.act_param2 = d2.act_param;
    if (d.N == 0 || d.H == 0 || d.W == 0) return SRMI_OK;
    // The fused pair is bound by MFMA issue, not by its output stream: taller tiles amortise the workgroup's prologue
    // (input tile staged twice, 64 registers of weights) and the 8 halo rows
    const int th = exp_flags() & 3 ? 4 << (exp_flags() & 3) : 32;       // 32 rows: 161 (8 rows) -> 138 us at 32x384x384; SRMI_EXP 1..3 (ablation build): 8 / 16 / 32... rows
    p.tiles_x = (d.W + K9C1_TW - 1) / K9C1_TW; p.tiles_y = (d.H + th - 1) / th;
    const long long nb = (long long)d.N * p.tiles_x * p.tiles_y;
    if (nb > 0x7fffffffLL) { set_error("k9c1pw_bf16: grid too large"); return SRMI_EINVAL; }
    if (th == 8) hipLaunchKernelGGL((k9c1_kernel<2, 8>), dim3((unsigned)nb, 1u), dim3(256), 0, s, p);
    else if (th == 16) hipLaunchKernelGGL((k9c1_kernel<2, 16>), dim3((unsigned)nb, 1u), dim3(256), 0, s, p);
    else if (th == 32) hipLaunchKernelGGL((k9c1_kernel<2, 32>), dim3((unsigned)nb, 1u), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k9c1_kernel<2, 32>), dim3((unsigned)nb, 1u), dim3(256), 0, s, p);
    return hip_check(hipGetLastError(), "k9c1pw_kernel");
}

}  // namespace srmi
