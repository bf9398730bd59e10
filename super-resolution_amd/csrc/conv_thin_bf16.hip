// conv3x3 / stride 1 / pad 1 / Cin = 64 / Cout <= 4, NHWC bf16 in -> NCHW|NHWC f32|bf16 out.
//
// The generator's `last_conv` (64 -> 3 + Tanh, P/srgan_model.py:25) at 4x resolution: 1.4 % of the
// FLOPs but it reads the largest tensor of the network (N x 4H x 4W x 64 bf16 = 302 MB at config 1).
// It is HBM-bound (128 B in, 12 B out per pixel), so it is built as a STREAMING stencil:
//
//   * a workgroup (4 waves) owns a strip of 64 columns x RB rows and walks DOWN the strip; input rows
//     (66 pixels x 128 B) arrive through an 8-slot LDS ring filled by LDS-DMA, 7 rows (59 KB per
//     workgroup, two workgroups per CU) in flight at any time; every input byte is fetched once per
//     strip (halo overhead 1.03 x 1.08 instead of 1.38 for a halo tile) and nothing waits on a load;
//     the XOR swizzle of the 16-B chunks is applied on the per-lane SOURCE address (the LDS side of an
//     LDS-DMA is lane-linear); out-of-image pixels are fetched from a 64-B zero block that the weight
//     packer appends to the packed weights (conv zero padding without branches or masked lanes);
//   * each input row is consumed exactly once: its 6 operand fragments (3 dx x 2 channel halves) feed the
//     three output rows r-1, r, r+1 (18 MFMAs per wave, output channels zero-padded to the 16 MFMA rows:
//     affordable because the kernel is memory-bound) held in three rolling accumulators; the finished
//     output row is activated and stored while later rows are still in flight;
//   * hand-off per row = counted vmcnt + one s_barrier.  Every iteration issues exactly D DMA (3 on wave 0, 2 on the others) + 3 store
//     instructions per wave (stores of invalid lanes are buffer stores with an out-of-range offset, which
//     the hardware drops) so the vmcnt arithmetic is static.
#include <stdlib.h>

#include <type_traits>

#include "c64_common.h"

namespace srmi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lptr_t;

struct ThinParams {
    const bf16_t* x; const bf16_t* w; const float* scale; const float* shift; const float* alpha; void* y;
    int N, H, W, Cout, in_cs, out_cs, out_co;
    int act; float act_param; int alpha_pc;
    int out_nchw, out_bf16;
    int strips, bands, band_rows;
    unsigned y_bytes;
    int dbg;   // ablation (SRMI_DBG, investigation only): 2 skip MFMAs, 4 skip activation math, 8 skip the output stores
};

constexpr int THIN_NS = 8;               // ring slots (input rows): 7 rows = 59 KB in flight per workgroup
constexpr int THIN_SLOT_PIX = 72;        // 9 DMA instructions x 8 pixels (66 used): wave w issues #w, #w+4; wave 0 also #8
constexpr int THIN_SLOT_BYTES = THIN_SLOT_PIX * 128;
constexpr int THIN_W_BYTES = 18 * 1024;  // 18 A fragments; a 64-B zero block follows

// tanh(v) = 1 - 2/(e^{2v}+1): two transcendental ops, abs error ~1e-7 (saturates correctly at +-inf)
__device__ __forceinline__ float thin_tanh(float v) {
    const float e = __builtin_amdgcn_exp2f(v * 2.885390082f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

__device__ __forceinline__ float thin_act(float v, int act, float param, float alpha) {
    switch (act) {
        case SRMI_ACT_PRELU: return v >= 0.f ? v : v * alpha;
        case SRMI_ACT_LRELU: return v >= 0.f ? v : v * param;
        case SRMI_ACT_RELU: return v > 0.f ? v : 0.f;
        case SRMI_ACT_TANH: return thin_tanh(v);
        case SRMI_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.442695041f));
        default: return v;
    }
}

#define THIN_DMA3(MOD)                                                                                         \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" MOD "\n\t"       \
                 "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off" MOD "\n\t"                   \
                 "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off" MOD "\n\ts_mov_b32 m0, %0"   \
                 : "=&s"(keep) : "v"(s0), "v"(s1), "v"(s2), "s"(dst) : "memory", "scc")
#define THIN_DMA2(MOD)                                                                                         \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" MOD "\n\t"       \
                 "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off" MOD "\n\ts_mov_b32 m0, %0"   \
                 : "=&s"(keep) : "v"(s0), "v"(s1), "s"(dst) : "memory", "scc")

// TANH: the activation is known at compile time (no switch in the per-row epilogue).  CH: cache policy of the
// streaming row loads (low two bits: 0 default, 1 nt, 2 sc1, 3 sc0 sc1) and of the output stores (bit 2: nt).
// The launcher uses nt loads: the 302 MB input is read once, and keeping it out of the caches is worth 8 us here
// and 7 us to the up conv that runs before it in the next forward; SRMI_DBG bits 23-25 select
// the other policies for measurement.
template <bool TANH, int CH = 0>
__global__ __launch_bounds__(256, 2) void conv_thin_kernel(ThinParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[THIN_NS * THIN_SLOT_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, q = lane >> 4;

    int b = blockIdx.x;
    const int sx = b % p.strips; b /= p.strips;
    const int bd = b % p.bands;
    const int n = b / p.bands;
    const int x0 = sx * 64, y0 = bd * p.band_rows;
    const int nrows = min(p.band_rows, p.H - y0);       // output rows of this band
    const int niter = nrows + 2;                          // input rows y0-1 .. y0+nrows

    // weights: 18 A fragments resident in registers (row r of the MFMA = output channel r, rows >= Cout zero)
    bf16x8 afr[18];
#pragma unroll
    for (int f = 0; f < 18; ++f) afr[f] = ((const bf16x8*)p.w)[f * 64 + lane];
    const unsigned char* zero_blk = (const unsigned char*)p.w + THIN_W_BYTES;

    // ---- per-lane DMA source columns: instruction i of this wave covers slot pixels 32*i + 8*wave + (lane>>3),
    //      physical chunk lane&7 holds logical chunk (lane&7) ^ swizzle(pixel)
    const unsigned char* xin = (const unsigned char*)(p.x + (size_t)n * p.H * p.W * p.in_cs);
    long long coloff[3];     // byte offset inside an image row, or -1 for "outside the image" (zero block)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int pix = 8 * (wave + 4 * i) + (lane >> 3);        // DMA instruction wave+4i covers slot pixels 8*(wave+4i)..+7
        const int ch = (lane & 7) ^ c64_key(pix);
        const int gx = x0 - 1 + pix;
        coloff[i] = (pix < 66 && gx >= 0 && gx < p.W) ? ((long long)gx * p.in_cs + ch * 8) * 2 : -1;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)lds;
    auto issue_row = [&](int r) {       // input row r of the band (image row y0-1+r) -> slot r % NS
        const int gy = y0 - 1 + r;
        const bool rowok = gy >= 0 && gy < p.H;
        const unsigned char* rowp = xin + (size_t)(rowok ? gy : 0) * p.W * p.in_cs * 2;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((r % THIN_NS) * THIN_SLOT_BYTES + wave * 1024));
        const unsigned char* s0 = (rowok && coloff[0] >= 0) ? rowp + coloff[0] : zero_blk + (lane & 3) * 16;
        const unsigned char* s1 = (rowok && coloff[1] >= 0) ? rowp + coloff[1] : zero_blk + (lane & 3) * 16;
        unsigned keep;
        if (wave == 0) {
            const unsigned char* s2 = (rowok && coloff[2] >= 0) ? rowp + coloff[2] : zero_blk + (lane & 3) * 16;
            if constexpr ((CH & 3) == 1) THIN_DMA3(" nt");
            else if constexpr (CH == 2) THIN_DMA3(" sc1");
            else if constexpr (CH == 3) THIN_DMA3(" sc0 sc1");
            else THIN_DMA3("");
        } else {
            if constexpr ((CH & 3) == 1) THIN_DMA2(" nt");
            else if constexpr (CH == 2) THIN_DMA2(" sc1");
            else if constexpr (CH == 3) THIN_DMA2(" sc0 sc1");
            else THIN_DMA2("");
        }
    };
#pragma unroll
    for (int r = 0; r < THIN_NS - 1; ++r) issue_row(r);

    // B-fragment byte offsets inside a slot for this wave's 16 columns: [dx], channel half 0
    int boff[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int P = wave * 16 + pl + dx;
        boff[dx] = P * 128 + ((q ^ c64_key(P)) << 4);
    }

    // epilogue constants: MFMA rows 0..3 (lanes q == 0) are output channels 0..3
    float sc[4], sh[4], al[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ce = e < p.Cout ? e : 0;
        sc[e] = p.scale[ce];
        sh[e] = p.shift[ce];
        al[e] = p.act == SRMI_ACT_PRELU ? p.alpha[p.alpha_pc ? ce : 0] : 0.f;
    }
    const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const int ox = x0 + wave * 16 + pl;
    const bool lane_out = q == 0 && ox < p.W;
    const unsigned esz = p.out_bf16 ? 2u : 4u;
    const unsigned cstep = p.out_nchw ? (unsigned)(p.H * p.W) * esz : esz;   // byte step between output channels

    f32x4 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // one input row: r = band-local input row index; R3 = r % 3 (compile-time) selects the rolling accumulators
    auto row_step = [&](int r, auto R3c) {
        constexpr int R3 = decltype(R3c)::value;
        // row r landed: own copies by counted vmcnt, the other waves' by the barrier.
        // Ops per iteration and wave: D DMAs (3 on wave 0, else 2) + 3 stores.  Younger than row r's DMAs are at least
        // 6 later rows' DMAs (r < 7: 6D) resp. 6 whole iterations (r >= 7: 6(D+3)): leave exactly those in flight.
        if (wave == 0) {
            if (r < THIN_NS - 1) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
        } else {
            if (r < THIN_NS - 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();     // also: every wave is done with row r-1, whose slot the next DMA refills
        issue_row(r + THIN_NS - 1);
        const unsigned char* slot = lds + (r % THIN_NS) * THIN_SLOT_BYTES;
        bf16x8 bfr[3][2];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bfr[dx][ks] = *(const bf16x8*)(slot + (boff[dx] ^ (ks << 6)));
        // input row r feeds output row r-dy (band-local output row index = input index - 1 - ... see below):
        // output row o (image row y0+o) uses input rows o, o+1, o+2 (band-local) with tap rows dy = 0, 1, 2,
        // so input row r contributes tap row dy to output row o = r - dy, accumulated in acc[(r - dy) mod 3].
        if (!(p.dbg & 2))
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int ai = (R3 - dy + 3) % 3;
            if (dy == 0) acc[ai] = (f32x4){0.f, 0.f, 0.f, 0.f};     // a new output row starts with its tap row 0
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    acc[ai] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[(dy * 3 + dx) * 2 + ks], bfr[dx][ks], acc[ai], 0, 0, 0);
        }
        // output row o = r - 2 is complete (its tap row 2 was just added): acc[(R3 - 2 + 3) % 3]
        const int o = r - 2;
        const int oy = y0 + o;
        const bool ok = lane_out && o >= 0 && o < nrows && !(p.dbg & 8);     // (dbg 8: timing experiment, no output stores)
        const f32x4 a = acc[(R3 + 1) % 3];
        // byte offset of (channel 0, oy, ox); tensors are < 4 GiB (thin_supported), so 32-bit arithmetic
        const unsigned off0 = p.out_nchw ? (unsigned)(((n * p.Cout) * p.H + oy) * p.W + ox) * esz
                                         : (unsigned)(((n * p.H + oy) * p.W + ox) * p.out_cs + p.out_co) * esz;
#pragma unroll
        for (int e = 0; e < 3; ++e) {      // exactly 3 store instructions per iteration (channel 3 below)
            const float z = a[e] * sc[e] + sh[e];
            const float val = (p.dbg & 4) ? z : (TANH ? thin_tanh(z) : thin_act(z, p.act, p.act_param, al[e]));
            const unsigned off = (ok && e < p.Cout) ? off0 + (unsigned)e * cstep : 0xfffffff0u;   // out of range: dropped
            constexpr int AUX = (CH & 4) ? 2 : 0;      // 2 = nt
            if (p.out_bf16) __builtin_amdgcn_raw_buffer_store_b16((short)f2bf(val), yres, off, 0, AUX);
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), yres, off, 0, AUX);
        }
        if (p.Cout == 4) {                 // rare 4-channel case: drain first so the static vmcnt arithmetic holds
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float val = thin_act(a[3] * sc[3] + sh[3], p.act, p.act_param, al[3]);
            const unsigned off = ok ? off0 + 3u * cstep : 0xfffffff0u;
            if (p.out_bf16) __builtin_amdgcn_raw_buffer_store_b16((short)f2bf(val), yres, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), yres, off, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };

    int r = 0;
    for (; r + 2 < niter; r += 3) {
        row_step(r, std::integral_constant<int, 0>{});
        row_step(r + 1, std::integral_constant<int, 1>{});
        row_step(r + 2, std::integral_constant<int, 2>{});
    }
    if (r < niter) { row_step(r, std::integral_constant<int, 0>{}); ++r; }
    if (r < niter) { row_step(r, std::integral_constant<int, 1>{}); ++r; }
    // never leave an LDS-DMA in flight into LDS that the next workgroup may own
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// OIHW f32 -> [tap][half][lane][8] bf16 (MFMA A-row r = output channel r; rows >= Cout are zero) + 64 zero bytes
__global__ void thin_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 18 * 64 * 8 + 32) return;
    if (idx >= 18 * 64 * 8) { out[idx] = 0; return; }
    const int j = idx & 7, lane = (idx >> 3) & 63, f = idx >> 9;
    const int tap = f >> 1, ks = f & 1;
    const int r = lane & 15, qq = lane >> 4;
    const int ci = ks * 32 + 8 * qq + j;
    out[idx] = r < Cout ? f2bf(w[((size_t)r * 64 + ci) * 9 + tap]) : (bf16_t)0;
}

bool thin_supported(const srmi_conv_desc& d) {
    if (d.compute != SRMI_COMPUTE_BF16 || d.act == SRMI_ACT_ADD_RELU) return false;
    if (d.kh != 3 || d.kw != 3 || d.stride != 1 || d.pad != 1 || d.Cin != 64) return false;
    if (d.Cout < 1 || d.Cout > 4 || d.shuffle != 1) return false;
    if (d.in_dtype != SRMI_BF16 || d.in_layout != SRMI_NHWC) return false;
    const int in_cs = d.in_cstride ? d.in_cstride : d.Cin;
    if (in_cs % 8 != 0) return false;
    // 32-bit store offsets: the output tensor must stay below 4 GiB
    const int out_cs = d.out_cstride ? d.out_cstride : d.Cout;
    const double out_bytes = (double)d.N * d.H * d.W * (d.out_layout == SRMI_NCHW ? d.Cout : out_cs) * 4.0;
    if (out_bytes >= 4294967000.0) return false;
    return true;
}

size_t thin_packed_bytes(const srmi_conv_desc&) { return THIN_W_BYTES + 64; }

int thin_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s) {
    hipLaunchKernelGGL(thin_pack_kernel, dim3(37), dim3(256), 0, s, w_oihw, (bf16_t*)w_packed, d.Cout);
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
    const size_t esz = p.out_bf16 ? 2 : 4;
    p.y_bytes = (unsigned)((size_t)d.N * d.H * d.W * (p.out_nchw ? d.Cout : g.out_cs) * esz);
    p.strips = (d.W + 63) / 64;
    // rows per band: long bands amortise the ring fill (measured: 24 rows 99 us, 96 rows 84 us at config 1), but the
    // grid should still fill the 512 resident slots (2 workgroups per CU) in ONE round: bands = 512 / (N * strips).
    const long long cols = (long long)d.N * p.strips;
    long long nbands = cols >= 512 ? 1 : 512 / cols;
    int rb = (int)((d.H + nbands - 1) / nbands);
    if (rb < 8) rb = 8;
    if (rb > d.H) rb = d.H;
    if (dbg_thin_rows() > 0) rb = dbg_thin_rows() > d.H ? d.H : dbg_thin_rows();
    p.dbg = dbg_flags();
    p.band_rows = rb;
    p.bands = (d.H + rb - 1) / rb;
    const long long nb = (long long)d.N * p.strips * p.bands;
    if (nb > 0x7fffffffLL) { set_error("c64_thin_bf16: grid too large"); return SRMI_EINVAL; }
    int ch = ((p.dbg >> 23) & 3) ^ 1;     // cache policy of the streaming loads: nt unless SRMI_DBG bits 23-24 say otherwise
    if (ch == 1 && (p.dbg & (1 << 25))) ch = 5;     // experiment: nt output stores too
    if (d.act == SRMI_ACT_TANH && ch == 5) hipLaunchKernelGGL((conv_thin_kernel<true, 5>), dim3((unsigned)nb), dim3(256), 0, s, p); else
    if (d.act == SRMI_ACT_TANH && ch == 1) hipLaunchKernelGGL((conv_thin_kernel<true, 1>), dim3((unsigned)nb), dim3(256), 0, s, p);
    else if (d.act == SRMI_ACT_TANH && ch == 2) hipLaunchKernelGGL((conv_thin_kernel<true, 2>), dim3((unsigned)nb), dim3(256), 0, s, p);
    else if (d.act == SRMI_ACT_TANH && ch == 3) hipLaunchKernelGGL((conv_thin_kernel<true, 3>), dim3((unsigned)nb), dim3(256), 0, s, p);
    else if (d.act == SRMI_ACT_TANH) hipLaunchKernelGGL(conv_thin_kernel<true>, dim3((unsigned)nb), dim3(256), 0, s, p);
    else if (ch & 1) hipLaunchKernelGGL((conv_thin_kernel<false, 1>), dim3((unsigned)nb), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_thin_kernel<false>, dim3((unsigned)nb), dim3(256), 0, s, p);
    return hip_check(hipGetLastError(), "conv_thin_kernel");
}

}  // namespace srmi
