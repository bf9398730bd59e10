// Split-bf16 f32 conv (see conv_b3_f32.hip for the arithmetic) for THE shape of the SRGAN generator's body: 3x3, stride 1,
// pad 1, 64 contracted channels, 64 g produced channels, NHWC f32, epilogue scale, shift, slope activation, f32 residual -
// forward, or the stride-1 dgrad of such a layer (transposed = 1).
//
// The general kernel re-loads and re-splits a pixel tile for each of the 9 taps and hands every 32/64-channel chunk over
// through two barriers: on the body's 24x24 maps (9216 pixels, one workgroup per CU) that chain is 13 of a launch's 20 us
// for 2 us of MFMA work, and the training step runs 99 such launches.  Here a workgroup owns an 8x8 output tile:
//   activations  the 10x10 halo tile is loaded ONCE, split into the three bf16 planes and left in LDS (43 KB);
//   weights      packed in the MFMA's own B-operand order (b3h_pack_kernel): per tap, 16-k slab and plane a wave reads one
//                contiguous kilobyte straight into registers, one tap ahead of the MFMAs - no LDS, no vector ALU;
//   K loop       9 taps x 4 slabs x 6 products, no barrier inside; two accumulators so consecutive MFMAs are independent.
// 4 waves = 2 pixel halves (32 pixels: MFMA rows) x 2 channel halves (32 channels: MFMA columns).
#include <stdlib.h>

#include "generic_common.h"

namespace srmi {

typedef __bf16 b3h_bf16x8 __attribute__((ext_vector_type(8)));
typedef float b3h_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 b3h_bf16x2 __attribute__((ext_vector_type(2)));

constexpr int B3H_T = 8;                     // output tile side
constexpr int B3H_HS = B3H_T + 2;            // halo tile side
constexpr int B3H_PITCH = 72;                // bf16 per LDS pixel (64 channels + 8: 144 bytes = 9 chunks of 16 B)
// bf16 per LDS halo ROW: 104 chunks of 16 B (10 pixels use 90).  A wave's A-operand read (ds_read_b128) is serviced in
// groups of 16 lanes, each group = four 4-pixel half rows of four consecutive tile rows; with 9 chunks per pixel and 104
// per row the 16 pixels of every group - for every tap - land on 16 different 16-byte slots of the 256-byte bank row.
// With rows packed at 90 chunks two of the four half rows coincided (SQ_LDS_BANK_CONFLICT 5 cycles per LDS-active cycle).
constexpr int B3H_ROWP = 104 * 8;

__device__ __forceinline__ void b3h_split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(b3h_f32x2{x0, x1}, b3h_bf16x2));
    const float r0 = x0 - __builtin_bit_cast(float, hi << 16), r1 = x1 - __builtin_bit_cast(float, hi & 0xffff0000u);
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(b3h_f32x2{r0, r1}, b3h_bf16x2));
    const float q0 = r0 - __builtin_bit_cast(float, mid << 16), q1 = r1 - __builtin_bit_cast(float, mid & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(b3h_f32x2{q0, q1}, b3h_bf16x2));
}

struct B3hParams {
    const float* x; const u32x4* w; const float* scale; const float* shift; const float* res; float* y;
    int N, H, W, Cout;                       // output = input size (stride 1, pad 1)
    int in_cs, out_cs, out_co, res_cs;
    int tiles_x, tiles_y;
    int transposed;
    const float* alpha; int act, alpha_pc; float act_param;      // slope activations (none / ReLU / LeakyReLU / PReLU)
    double* stats;      // NULL, or per-workgroup partial sums of the OUTPUT for train-mode BatchNorm (train.hip's slot
                        // layout: slot = tile index, [slot][2][Cout]: sum, sum of squares): the statistics pass that
                        // would re-read the whole output comes for free in the epilogue (act must be NONE)
};

__global__ __launch_bounds__(256) void conv_b3h_kernel(B3hParams p) {
    __shared__ __attribute__((aligned(16))) unsigned short As[3][B3H_HS * B3H_ROWP];
    __shared__ double bnsum[2][2][2][32];                   // [wm][wn][sum, sum of squares][channel of the wave's tile]
    auto at = [](int hy, int hx) { return hy * B3H_ROWP + hx * B3H_PITCH; };      // element offset of halo pixel (hy, hx) in a plane
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    int b = blockIdx.x;
    const int tx0 = (b % p.tiles_x) * B3H_T; b /= p.tiles_x;
    const int ty0 = (b % p.tiles_y) * B3H_T;
    const int n = b / p.tiles_y;
    const int NT = p.Cout >> 5;                             // 32-channel tiles of the produced channels
    const int nt = blockIdx.y * 2 + wn;
    const size_t w_plane16 = (size_t)9 * NT * 4 * 64;       // 16-byte pieces per weight plane ([tap][nt][4 slabs][64 lanes])

    // ---- weights of tap 0 go out first: they do not depend on anything staged below ----------------------------------
    u32x4 bq[2][4][3];                                      // [buffer][slab][plane]
    auto load_b = [&](int tap, u32x4 (&dst)[4][3]) {
        const u32x4* src = p.w + ((size_t)tap * NT + nt) * 4 * 64 + lane;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) dst[s][pl] = src[pl * w_plane16 + s * 64];
    };
    load_b(0, bq[0]);

    // ---- halo tile: 100 pixels x 16 pieces of 4 channels, loaded once, split, planes to LDS ---------------------------
    constexpr int PIECES = B3H_HS * B3H_HS * 16;
    constexpr int A_LOADS = (PIECES + 255) / 256;
    f32x4 ra[A_LOADS];
#pragma unroll
    for (int l = 0; l < A_LOADS; ++l) {
        const int idx = tid + 256 * l;
        const int hp = idx >> 4, piece = idx & 15;
        const int hy = hp / B3H_HS, hx = hp - hy * B3H_HS;
        const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
        f32x4 q = {0.f, 0.f, 0.f, 0.f};
        if (idx < PIECES && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
            q = *(const f32x4*)(p.x + ((size_t)(n * p.H + iy) * p.W + ix) * p.in_cs + 4 * piece);
        ra[l] = q;
    }
#pragma unroll
    for (int l = 0; l < A_LOADS; ++l) {
        const int idx = tid + 256 * l;
        if (idx < PIECES) {
            const int hp = idx >> 4, k = 4 * (idx & 15);
            const int hy = hp / B3H_HS, hx = hp - hy * B3H_HS;
            const int o = at(hy, hx) + k;
            unsigned hi[2], mid[2], lo[2];
            b3h_split2(ra[l][0], ra[l][1], hi[0], mid[0], lo[0]);
            b3h_split2(ra[l][2], ra[l][3], hi[1], mid[1], lo[1]);
            *(u32x2*)&As[0][o] = u32x2{hi[0], hi[1]};
            *(u32x2*)&As[1][o] = u32x2{mid[0], mid[1]};
            *(u32x2*)&As[2][o] = u32x2{lo[0], lo[1]};
        }
    }
    __syncthreads();

    // this lane's MFMA row: tile pixel 32 wm + i
    const int pidx = 32 * wm + i;
    const int py = pidx >> 3, px = pidx & 7;

    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }

    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};     // the six products, smallest first
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) load_b(tap + 1, bq[(tap + 1) & 1]);
        const int ky = tap / 3, kx = tap - 3 * ky;
        // forward: input pixel = o - 1 + k; dgrad: o + 1 - k  (halo origin = o - 1 in both)
        const int hp = p.transposed ? at(py + 2 - ky, px + 2 - kx) : at(py + ky, px + kx);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            b3h_bf16x8 av[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) av[pl] = *(const b3h_bf16x8*)&As[pl][hp + 16 * s + 8 * h];
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const b3h_bf16x8 bv = __builtin_bit_cast(b3h_bf16x8, bq[tap & 1][s][PB[t]]);
                if ((t + s) & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[t]], bv, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[t]], bv, acc0, 0, 0, 0);
            }
        }
    }

    // ---- epilogue: acc[e] = pixel row (e&3) + 8 (e>>2) + 4 h of this wave's 32 pixels, channel 32 nt + i -------------
    const int co = 32 * nt + i;
    const float sc = p.scale ? p.scale[co] : 1.f;
    const float sh = p.shift ? p.shift[co] : 0.f;
    // slope of the negative side, as generic_epilogue_flat applies it (1: no activation)
    const float sl = p.act == SRMI_ACT_PRELU ? p.alpha[p.alpha_pc ? co : 0] : p.act == SRMI_ACT_LRELU ? p.act_param
                   : p.act == SRMI_ACT_RELU ? 0.f : 1.f;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int q = 32 * wm + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int oy = ty0 + (q >> 3), ox = tx0 + (q & 7);
        if (oy < p.H && ox < p.W) {
            const size_t pix = (size_t)(n * p.H + oy) * p.W + ox;
            float v = (acc0[e] + acc1[e]) * sc + sh;
            v = v >= 0.f ? v : v * sl;
            if (p.res) v += p.res[pix * p.res_cs + co];
            p.y[pix * p.out_cs + p.out_co + co] = v;
            if (p.stats) { s1 += (double)v; s2 += (double)v * (double)v; }
        }
    }
    if (p.stats) {      // (wave-uniform) sums of this workgroup's 64 pixels per channel, in a fixed order: rows, h, wm
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (h == 0) { bnsum[wm][wn][0][i] = s1; bnsum[wm][wn][1][i] = s2; }
        __syncthreads();
        if (tid < 128) {
            const int k = tid >> 6, w2 = (tid >> 5) & 1, c = tid & 31;
            const int ch = 32 * (blockIdx.y * 2 + w2) + c;
            p.stats[((size_t)blockIdx.x * 2 + k) * p.Cout + ch] = bnsum[0][w2][k][c] + bnsum[1][w2][k][c];
        }
    }
}

// ---- operand pack: OIHW f32 -> [plane][tap][tile nt][slab ks][lane][8] bf16 -------------------------------------------
// lane = 32 h + i holds, for produced channel n = 32 nt + i, the contracted channels k = 16 ks + 8 h .. + 7: exactly the B
// operand of one v_mfma_f32_32x32x16_bf16.  forward: n = co, k = ci;  dgrad (transposed): n = ci, k = co.
__global__ __launch_bounds__(256) void b3h_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cin,
                                                       int Cout, int transposed, size_t plane) {
    const int Nn = transposed ? Cin : Cout, Kk = transposed ? Cout : Cin;
    const int NT = Nn >> 5, KS = Kk >> 4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < plane; idx += (size_t)gridDim.x * 256) {
        const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        size_t t1 = idx >> 9;
        const int ks = (int)(t1 % KS); t1 /= KS;
        const int nt = (int)(t1 % NT);
        const int tap = (int)(t1 / NT);
        const int n = 32 * nt + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
        const int co = transposed ? k : n, ci = transposed ? n : k;
        const float x = w[((size_t)co * Cin + ci) * 9 + tap];
        const __bf16 hb = (__bf16)x;
        const float r1 = x - (float)hb;
        const __bf16 mb = (__bf16)r1;
        const __bf16 lb = (__bf16)(r1 - (float)mb);
        out[idx] = __builtin_bit_cast(unsigned short, hb);
        out[plane + idx] = __builtin_bit_cast(unsigned short, mb);
        out[2 * plane + idx] = __builtin_bit_cast(unsigned short, lb);
    }
}

static bool b3h_enabled() {
    static const bool v = [] {
        const char* e = getenv("SRMI_F32_EXACT");
        return !(e && atoi(e)) && !(dbg_flags() & (1 << 9));       // SRMI_DBG bit 9: the general split kernel instead
    }();
    return v;
}

// From the descriptor alone (operand format!): the forward conv of `d` / the dgrad of `d`.
bool b3h_fwd_ok(const srmi_conv_desc& d) {
    return b3h_enabled() && d.compute == SRMI_COMPUTE_F32 && d.in_layout == SRMI_NHWC && d.in_dtype == SRMI_F32 &&
           d.out_layout == SRMI_NHWC && d.out_dtype == SRMI_F32 && d.res_dtype == SRMI_F32 && d.kh == 3 && d.kw == 3 &&
           d.stride == 1 && d.pad == 1 && d.Cin == 64 && d.Cout % 64 == 0 && d.Cout / 64 <= 65535 &&
           (d.act == SRMI_ACT_NONE || d.act == SRMI_ACT_RELU || d.act == SRMI_ACT_LRELU || d.act == SRMI_ACT_PRELU) &&
           d.shuffle == 1;
}
bool b3h_dgrad_ok(const srmi_conv_desc& d) {
    return b3h_enabled() && d.kh == 3 && d.kw == 3 && d.stride == 1 && d.pad == 1 && d.Cout == 64 && d.Cin % 64 == 0 &&
           d.Cin / 64 <= 65535;
}

// workgroups along x (= partial-sum slots of the fused BatchNorm statistics) for the forward conv of `d`
long long b3h_tiles(const srmi_conv_desc& d) {
    return (long long)d.N * ((d.H + B3H_T - 1) / B3H_T) * ((d.W + B3H_T - 1) / B3H_T);
}

size_t b3h_packed_bytes(const srmi_conv_desc& d) { return (size_t)3 * 9 * d.Cin * d.Cout * sizeof(unsigned short); }

int b3h_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, int transposed, hipStream_t s) {
    const size_t plane = (size_t)9 * d.Cin * d.Cout;
    size_t blocks = (plane + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(b3h_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w_oihw, (unsigned short*)w_packed, d.Cin, d.Cout,
                       transposed, plane);
    return hip_check(hipGetLastError(), "b3h_pack_kernel");
}

// GenericParams of a conv whose operand is in this file's format (p.b3 == 2).  Anything the kernel cannot run is refused
// loudly: no other kernel can read this operand.
int b3h_launch(const GenericParams& g) {
    if (g.in_bf16 || g.in_nchw || g.out_bf16 || g.out_nchw || g.res_bf16 || g.in_cs % 4 != 0 || ((uintptr_t)g.x & 15) ||
        !(g.act == SRMI_ACT_NONE || g.act == SRMI_ACT_RELU || g.act == SRMI_ACT_LRELU || g.act == SRMI_ACT_PRELU) ||
        (g.act == SRMI_ACT_PRELU && !g.alpha) || g.r != 1 || g.Cin != 64 || g.Cout % 64 != 0 || g.kh != 3 || g.kw != 3 || g.stride != 1 ||
        g.pad != 1 || g.H != g.Ho || g.W != g.Wo) {
        set_error("conv (split-bf16 3x3 / 64-channel kernel): NHWC f32 in / out with 16-byte aligned rows, slope activations only");
        return SRMI_EUNSUPPORTED;
    }
    B3hParams p;
    p.x = (const float*)g.x; p.w = (const u32x4*)g.w; p.scale = g.scale; p.shift = g.shift; p.res = (const float*)g.res;
    p.y = (float*)g.y;
    p.N = g.N; p.H = g.Ho; p.W = g.Wo; p.Cout = g.Cout;
    p.in_cs = g.in_cs; p.out_cs = g.out_cs; p.out_co = g.out_co; p.res_cs = g.res_cs;
    p.tiles_x = (p.W + B3H_T - 1) / B3H_T; p.tiles_y = (p.H + B3H_T - 1) / B3H_T;
    p.transposed = g.transposed;
    p.alpha = g.alpha; p.act = g.act; p.alpha_pc = g.alpha_pc; p.act_param = g.act_param;
    p.stats = g.bn_stats;
    const long long tiles = (long long)p.tiles_x * p.tiles_y * p.N;
    if (tiles <= 0) return SRMI_OK;
    if (tiles > 0x7fffffffLL) { set_error("conv (split-bf16 3x3 kernel): too many tiles"); return SRMI_EINVAL; }
    hipLaunchKernelGGL(conv_b3h_kernel, dim3((unsigned)tiles, (unsigned)(p.Cout / 64)), dim3(256), 0, g.stream, p);
    return hip_check(hipGetLastError(), "conv_b3h_kernel");
}

}  // namespace srmi
