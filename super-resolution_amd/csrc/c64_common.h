// Pieces shared by the bf16 MFMA conv kernels that produce 64-channel chunks (conv_c64_bf16.hip,
// conv_k9c3_bf16.hip): output-channel permutation, epilogue-parameter staging, fused epilogue.
#pragma once
#include <type_traits>

#include "common.h"

namespace srmi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

typedef float f32x2 __attribute__((ext_vector_type(2)));

// LDS swizzle key of the pixel tiles (128 B = 8 chunks of 16 B per pixel): chunk c of pixel P lives at chunk position
// c ^ c64_key(P) of the pixel's line.  ds_read_b128 is serviced in four groups of 16 lanes ({0-3,12-15,20-27}, ...,
// MI355X_MICROARCH.md "LDS"), each a mix of two neighbouring channel groups q: a group is conflict-free when its 8 even and
// its 8 odd pixels each hit 8 different chunk positions.  With the key 2 * (pair & 3) bit 0 of the position is bit 0 of q
// (the two q of a group never meet) and any four consecutive pixel pairs get four different keys, so a run of 16
// consecutive pixels is conflict-free from ANY start column - the tap dx = 1 included.  Round 2's key (pair & 7) was
// conflict-free only for even starts with q ^ key landing apart: the model (tools/lds_conflict_sim.py) counts 4680
// conflict cycles per fused-ResBlock workgroup with it (measured: SQ_LDS_BANK_CONFLICT 5860 per CU) and 936 with this
// one (what is left are the conv-1 tiles that wrap around a row end of the flattened T region).
// Pixel pitches are multiples of 8 (or the key is taken from the column), so the key does not change with the row.
__host__ __device__ constexpr int c64_key(int P) { return ((P >> 1) & 3) << 1; }

// Slope activation on a channel pair:  z = a*sc + sh;  z >= 0 ? z : z*sl  with the multiply-adds as packed-f32
// instructions (v_pk_fma_f32 / v_pk_mul_f32): three VALU instructions per value instead of four.
__device__ __forceinline__ f32x2 slope_act2(f32x2 a, f32x2 sc, f32x2 sh, f32x2 sl) {
    const f32x2 z = __builtin_elementwise_fma(a, sc, sh);
    const f32x2 zs = z * sl;
    return (f32x2){z[0] >= 0.f ? z[0] : zs[0], z[1] >= 0.f ? z[1] : zs[1]};
}

// The same for slopes known to lie in [0, 1] (ReLU, LeakyReLU, a scalar PReLU alpha in that range, none):
// max(z, z*sl) is then the identical value and costs one instruction instead of a compare + select.
__device__ __forceinline__ f32x2 slope_act2_max(f32x2 a, f32x2 sc, f32x2 sh, f32x2 sl) {
    const f32x2 z = __builtin_elementwise_fma(a, sc, sh);
    const f32x2 zs = z * sl;
    return (f32x2){__builtin_fmaxf(z[0], zs[0]), __builtin_fmaxf(z[1], zs[1])};
}

// Whole-line stores.  In the accumulator layout a lane (pixel pl, group q) ends up with bytes [16 q, 16 q + 16) (o0) and
// [64 + 16 q, 64 + 16 q + 16) (o1) of its pixel's 128-byte line, so a plain "every lane stores its own o0, then its own
// o1" writes HALF lines per instruction - measured at 3.1 TB/s where whole lines reach 5.1 TB/s (604 MB store-only
// variant of k9c1_kernel, same box: 195 vs 119 us).  Lanes pl and pl ^ 8 therefore swap one half (DPP row_ror:8, no
// LDS): instruction A writes the whole lines of pixels 0..7 (lanes pl < 8: own o0; lanes pl >= 8: o1 of pixel pl - 8),
// instruction B those of pixels 8..15.  `base` = this lane's OWN pixel line, `pxstride` = elements between
// neighbouring pixels of the tile, okA / okB = pixel (pl & 7) / (pl & 7) + 8 may be written.
template <bool NT>
__device__ __forceinline__ void store_lines_x8(bf16_t* base, long pxstride, int pl, int q, u32x4 o0, u32x4 o1, bool okA, bool okB) {
    const bool lo = pl < 8;
    u32x4 recv;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned send = lo ? o1[i] : o0[i];
        recv[i] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0x128, 0xf, 0xf, false);     // row_ror:8 = lane pl ^ 8
    }
    const u32x4 dA = lo ? o0 : recv, dB = lo ? recv : o1;
    bf16_t* pA = base - (lo ? 0 : 8 * pxstride) + (lo ? 0 : 32) + 8 * q;
    bf16_t* pB = pA + 8 * pxstride;
    if (okA) { if constexpr (NT) __builtin_nontemporal_store(dA, (u32x4*)pA); else *(u32x4*)pA = dA; }
    if (okB) { if constexpr (NT) __builtin_nontemporal_store(dB, (u32x4*)pB); else *(u32x4*)pB = dB; }
}

// store modes
enum { C64_PLAIN = 0, C64_CRD = 1, C64_DCR = 2 };

// conv-channel index of (chunk s, channel-within-chunk cl)
__host__ __device__ __forceinline__ int c64_conv_channel(int mode, int s, int cl) {
    return mode == C64_CRD ? 4 * cl + s : 64 * s + cl;
}

// MFMA A-row r (0..15) of channel tile t (0..3)  ->  channel-within-chunk.
// The accumulator layout of v_mfma_f32_16x16x32 gives lane (pixel pl, group q) rows 4q..4q+3 of every
// tile; with this map those 16 values are channels 8q..8q+7 and 32+8q..32+8q+7, so the four lanes of a
// pixel write 64 contiguous bytes per 16-B store instruction (two instructions fill the 128-B pixel).
__host__ __device__ __forceinline__ int c64_row_to_cl(int t, int r) {
    return 32 * (t >> 1) + 8 * (r >> 2) + 4 * (t & 1) + (r & 3);
}

struct C64Epi {
    const float* scale; const float* shift; const float* alpha;
    const bf16_t* res; bf16_t* y;
    int H, W, Cout, out_cs, out_co, mode;
    int act; float act_param; int alpha_pc;
    int res_cs = 0;      // channels per pixel of the residual buffer (0 = dense = final channel count): a channel slice of a wider buffer
};

// 0: no activation, 1: one slope in [0, 1] for every channel (max form), 2: general (compare + select).  Wave-uniform.
__device__ __forceinline__ int c64_slope_kind(const C64Epi& e) {
    if (e.act == SRMI_ACT_NONE) return 0;
    if (e.act == SRMI_ACT_PRELU && e.alpha_pc) return 2;
    float s = 1.f;
    if (e.act == SRMI_ACT_RELU) s = 0.f;
    else if (e.act == SRMI_ACT_LRELU) s = e.act_param;
    else if (e.act == SRMI_ACT_PRELU) s = e.alpha[0];
    s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s)));
    return (s >= 0.f && s <= 1.f) ? 1 : 2;
}

constexpr int C64_MAX_CHUNKS = 16;                       // Cout <= 1024
constexpr int C64_PAR_BYTES = C64_MAX_CHUNKS * 3 * 64 * 4;  // per chunk: scale[64] | shift[64] | slope[64]

// Every activation these kernels serve is  v >= 0 ? v : v * slope  (none: 1, ReLU: 0, LeakyReLU: param,
// PReLU: alpha): branch-free, which keeps the unrolled epilogue small (instruction cache).
// scale / shift / slope of all chunks are staged into LDS in kernel channel order, once per workgroup,
// in two halves so that the global loads are IN FLIGHT TOGETHER with the input-tile loads (one memory
// round trip in the kernel prologue, not two): c64_params_load before the tile loads, c64_params_store after.
struct C64ParRegs { float sc[C64_MAX_CHUNKS / 4], sh[C64_MAX_CHUNKS / 4], sl[C64_MAX_CHUNKS / 4]; };

__device__ __forceinline__ void c64_params_load(C64ParRegs& r, const C64Epi& e, int nchunk, int tid) {
    float slope0 = 1.f;
    if (e.act == SRMI_ACT_RELU) slope0 = 0.f;
    else if (e.act == SRMI_ACT_LRELU) slope0 = e.act_param;
    else if (e.act == SRMI_ACT_PRELU && !e.alpha_pc) slope0 = e.alpha[0];
    const bool pc = e.act == SRMI_ACT_PRELU && e.alpha_pc;
#pragma unroll
    for (int k = 0; k < C64_MAX_CHUNKS / 4; ++k) {
        const int i = tid + 256 * k;
        r.sc[k] = 0.f; r.sh[k] = 0.f; r.sl[k] = slope0;
        if (i < nchunk * 64) {
            const int co = c64_conv_channel(e.mode, i >> 6, i & 63);
            r.sc[k] = e.scale[co];
            r.sh[k] = e.shift[co];
            if (pc) r.sl[k] = e.alpha[co];
        }
    }
}

__device__ __forceinline__ void c64_params_store(float* par, const C64ParRegs& r, int nchunk, int tid) {
#pragma unroll
    for (int k = 0; k < C64_MAX_CHUNKS / 4; ++k) {
        const int i = tid + 256 * k;
        if (i < nchunk * 64) {
            float* ps = par + (i >> 6) * 192;
            ps[i & 63] = r.sc[k];
            ps[64 + (i & 63)] = r.sh[k];
            ps[128 + (i & 63)] = r.sl[k];
        }
    }
}

// Fused epilogue of one 64-channel chunk: scale/shift -> slope activation -> +residual -> bf16 store
// (plain NHWC channel slice, or one sub-pixel of a PixelShuffle(2)/depth_to_space(2) output).
// acc[k][t][e] = channel 32*(t>>1) + 8q + 4*(t&1) + e of pixel pl of this wave's pixel tile k.
// PRE: the residual of every pixel tile is already in registers (rpre), e.g. taken from an LDS-resident
// input tile by the fused ResBlock kernel; otherwise it is loaded from o.res (if any).
template <int PT, bool PRE = false>
__device__ __forceinline__ void c64_epilogue(const f32x4 (&acc)[PT][4], const float* par_s, const C64Epi& o, int s,
                                             int n, int y0, int x0, const int (&prow)[PT], const int (&pcol)[PT],
                                             int pl, int q, const u32x4 (*rpre)[2] = nullptr) {
    f32x2 sc[8], sh[8], sl[8];     // channel pairs, in the order of this lane's 16 values
#pragma unroll
    for (int hi = 0; hi < 2; ++hi)
#pragma unroll
        for (int v4 = 0; v4 < 2; ++v4) {
            const int c = 32 * hi + 8 * q + 4 * v4;
            const f32x4 a = *(const f32x4*)(par_s + c);
            const f32x4 bsh = *(const f32x4*)(par_s + 64 + c);
            const f32x4 csl = *(const f32x4*)(par_s + 128 + c);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                sc[4 * hi + 2 * v4 + e] = (f32x2){a[2 * e], a[2 * e + 1]};
                sh[4 * hi + 2 * v4 + e] = (f32x2){bsh[2 * e], bsh[2 * e + 1]};
                sl[4 * hi + 2 * v4 + e] = (f32x2){csl[2 * e], csl[2 * e + 1]};
            }
        }
    const int si = s >> 1, sj = s & 1;
    const bool plain = o.mode == C64_PLAIN;
    const int Cf = plain ? o.Cout : 64;
    const int cbase = (plain ? 64 * s : 0) + 8 * q;
    const int Hs = plain ? o.H : 2 * o.H, Ws = plain ? o.W : 2 * o.W;      // output image
    // A pixel tile is one image row segment: its row is wave-uniform, so the 64-bit part of every address (image, row)
    // is scalar arithmetic and a lane adds a 32-bit in-row offset.
    // opaque zero: keeps hipcc from hoisting the lane arithmetic above the K loop (it would be spilled to scratch
    // there: the loops run at the register limit)
    int opaque0;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque0));
    size_t rowpix[PT];
    int ocol[PT];
    bool ok[PT];
#pragma unroll
    for (int k = 0; k < PT; ++k) {
        const int oy = y0 + prow[k], ox = x0 + pcol[k] + pl + opaque0;
        ok[k] = oy < o.H && ox < o.W;
        rowpix[k] = ((size_t)n * Hs + (plain ? oy : 2 * oy + si)) * Ws;
        ocol[k] = plain ? ox : 2 * ox + sj;
    }
    u32x4 r[PT][2];
    const bool have_res = PRE || o.res != nullptr;
    if constexpr (PRE) {
#pragma unroll
        for (int k = 0; k < PT; ++k) { r[k][0] = rpre[k][0]; r[k][1] = rpre[k][1]; }
    } else if (o.res) {   // all residual loads in flight before the first use
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            r[k][0] = (u32x4){0u, 0u, 0u, 0u};
            r[k][1] = (u32x4){0u, 0u, 0u, 0u};
            if (ok[k]) {
                const int rcs = o.res_cs ? o.res_cs : Cf;
                const bf16_t* rp = o.res + rowpix[k] * rcs + (unsigned)(ocol[k] * rcs + cbase);
                r[k][0] = *(const u32x4*)rp;
                r[k][1] = *(const u32x4*)(rp + 32);
            }
        }
    }
    // KIND (c64_slope_kind): 0 no activation (e.g. the second conv of a ResBlock), 1 max form, 2 compare + select
    // RES: a residual is added (compile-time per variant: as a run-time flag hipcc turns the add into 40 selects per tile)
    auto body = [&](auto KIND, auto RES) {
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            u32x4 o0, o1;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int j = 2 * t + e;                     // channel pair of this lane's 16 values
                    const f32x2 a = {acc[k][t][2 * e], acc[k][t][2 * e + 1]};
                    f32x2 v;
                    if constexpr (decltype(KIND)::value == 2) v = slope_act2(a, sc[j], sh[j], sl[j]);
                    else if constexpr (decltype(KIND)::value == 1) v = slope_act2_max(a, sc[j], sh[j], sl[j]);
                    else v = __builtin_elementwise_fma(a, sc[j], sh[j]);
                    if constexpr (decltype(RES)::value) {
                        const unsigned w = r[k][j >> 2][j & 3];
                        v += (f32x2){__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
                    }
                    const unsigned w = pack_bf2(v[0], v[1]);
                    if (t < 2) o0[j & 3] = w; else o1[j & 3] = w;
                }
            if (ok[k]) {
                // (whole-line stores via store_lines_x8 were measured here: the fused ResBlock got 4 % slower - its 18.9 MB
                //  land in L2 / the memory-side cache, the swap's registers spill - and the 75 MB up conv did not change)
                bf16_t* op = o.y + rowpix[k] * o.out_cs + (unsigned)(ocol[k] * o.out_cs + o.out_co + cbase);
                *(u32x4*)op = o0;
                *(u32x4*)(op + 32) = o1;
            }
        }
    };
    const int kind = c64_slope_kind(o);
    if (have_res) {
        if (kind == 0) body(std::integral_constant<int, 0>{}, std::true_type{});
        else if (kind == 1) body(std::integral_constant<int, 1>{}, std::true_type{});
        else body(std::integral_constant<int, 2>{}, std::true_type{});
    } else {
        if (kind == 0) body(std::integral_constant<int, 0>{}, std::false_type{});
        else if (kind == 1) body(std::integral_constant<int, 1>{}, std::false_type{});
        else body(std::integral_constant<int, 2>{}, std::false_type{});
    }
}

// scheduling directive: NR groups of {1 LDS read, ~NM/NR MFMAs} (compile-time only, no code emitted)
template <int I, int NR, int NM>
__device__ __forceinline__ void sched_interleave() {
    if constexpr (I < NR) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NM / NR + (I < NM % NR ? 1 : 0), 0);
        sched_interleave<I + 1, NR, NM>();
    }
}

}  // namespace srmi
