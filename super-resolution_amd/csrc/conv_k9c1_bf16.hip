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
#include <type_traits>

#include "c64_common.h"

namespace srmi {

struct K9c1Params {
    const float* x; const bf16_t* w; const float* scale; const float* shift; const float* alpha;
    bf16_t* y;
    int N, H, W, out_cs, out_co;
    int tiles_x, tiles_y;
    int act; float act_param;
};

constexpr int K9C1_TH = 8, K9C1_TW = 64, K9C1_LH = K9C1_TH + 8, K9C1_LW = K9C1_TW + 10;      // (even; column +9 is read, weight 0)

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

__global__ __launch_bounds__(256) void k9c1_kernel(K9c1Params p) {
    constexpr int TH = K9C1_TH, TW = K9C1_TW, LH = K9C1_LH, LW = K9C1_LW;
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
        // whole 128-byte lines per store instruction (store_lines_x8: lanes pl and pl ^ 8 swap a half)
        bf16_t* line = p.y + ((size_t)((size_t)n * p.H + y0 + k) * p.W + x) * p.out_cs + p.out_co + 64 * g;
        const int xa = x0 + 16 * wave + (pl & 7);
        store_lines_x8<true>(line, p.out_cs, pl, q, o[0], o[1], xa < p.W, xa + 8 < p.W);
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
    hipLaunchKernelGGL(k9c1_kernel, dim3((unsigned)nb, (unsigned)(d.Cout / 64)), dim3(256), 0, s, p);
    return hip_check(hipGetLastError(), "k9c1_kernel");
}

}  // namespace srmi
