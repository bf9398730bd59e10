// conv3x3 / stride 1 / pad 1, NHWC bf16, ANY Cin that is a multiple of 32 and Cout a multiple of 32, on
// v_mfma_f32_16x16x32_bf16: the wide-input sibling of conv_c64_bf16.hip.
//
// Serves the layers the Cin = 64 kernel cannot: the ESRGAN-style dense block (ESRGAN/untitled0.py:18-35:
// 64/96/128 -> 32 with the input a channel slice of the 160-channel concat buffer, 160 -> 64 with the block
// input as a channel-slice residual), SRCNN's 128 -> 64 (SRCNN/srcnn.py:205) and VGG / discriminator widths.
//
// Same machinery as conv_c64_bf16.hip (one workgroup = 4 waves = one TH x TW tile; activations are the MFMA B
// operand read from an XOR-swizzled LDS tile; weights are pre-packed A fragments streamed through a 4-slot
// LDS-DMA ring with a counted-vmcnt hand-off per tap), with two differences:
//   * the contraction runs over K-CHUNKS of 64 input channels: the halo tile of one chunk is staged, its
//     9 taps x 2 k-steps are accumulated, the next chunk is staged over it; accumulators live across chunks
//     (Cin = 96 / 160: the last chunk's upper 32 channels are zero-filled, their packed weights are zero);
//   * NCT = 4 or 2 channel tiles per workgroup (64 or 32 output channels); more output channels = more
//     workgroups along grid.y, each re-reading the tile (L2-resident).
// The chunk hand-off is synchronous (stage -> barrier -> compute): ~60 % of the Cin = 64 kernel's rate, 8-10x the
// generic f32 kernel's.  Launches with many tiles run the persistent, wave-specialised form further down
// (conv_cinp_kernel): same arithmetic, same bits.
#include <stdlib.h>

#include <atomic>

#include <type_traits>

#include "c64_common.h"

namespace srmi {

struct CinParams {
    const bf16_t* x; const bf16_t* w;
    const float* scale; const float* shift; const float* alpha;
    const bf16_t* res; bf16_t* y;
    int N, H, W, Cin, Cout, kchunks, in_cs, out_cs, out_co, res_cs;
    int act; float act_param; int alpha_pc;
    int tiles_x, tiles_y;
    int exp;           // ablation build only (SRMI_EXP): 0 in the product library
};

constexpr int CIN_WRING = 4;

// SKIPH (8-wave, 64-channel variants, launched for Cin % 64 == 32): in a chunk that holds only 32 real channels the
// MFMAs of the upper half-steps are branched around (their operand reads stay: one loop body, no second unrolled copy)
template <int TH, int TWT, int NCT, int NW, int WGPC = 1, bool SKIPH = false>
__global__ __launch_bounds__(NW * 64, WGPC) void conv_cin_kernel(CinParams p) {
    constexpr int NTH = NW * 64;
    constexpr int TW = TWT * 16;
    constexpr int NPT = TH * TWT;
    constexpr int PT = (NPT + NW - 1) / NW;            // waves past the end recompute the last tile and store nothing
    constexpr int ROWP = (TW + 2 + 15) / 16 * 16;
    constexpr int IN_BYTES = (TH + 2) * ROWP * 128;
    constexpr int TAP_BYTES = 2 * NCT * 1024;          // 2 k-steps x NCT channel tiles x 64 lanes x 16 B
    constexpr int NDW = TAP_BYTES / 1024 < NW ? TAP_BYTES / 1024 : NW;   // waves that move weights (1 KiB DMA granularity)
    constexpr int WAVE_BYTES = TAP_BYTES / NDW;
    constexpr int NDMA = WAVE_BYTES / 1024;            // LDS-DMA instructions per moving wave and tap (2 or 1)
    constexpr int CG = NCT * 16;                       // output channels of this workgroup
    __shared__ __attribute__((aligned(16))) unsigned char lds[IN_BYTES + CIN_WRING * TAP_BYTES + 3 * 64 * 4];
    unsigned char* const wring = lds + IN_BYTES;
    float* const par = (float*)(lds + IN_BYTES + CIN_WRING * TAP_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, q = lane >> 4;

    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);   // XCD-aware tile order
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int cg = blockIdx.y;                         // output-channel group
    const int co0 = cg * CG;

    // ---- weight stream: tap index g = kc * 9 + tap of this channel group ------------------------------
    const int ntap = p.kchunks * 9;
    const unsigned char* wsrc = (const unsigned char*)p.w + (size_t)cg * ntap * TAP_BYTES + wave * WAVE_BYTES + lane * 16;
    const unsigned wring_lds = (unsigned)(uintptr_t)(lptr_t)wring;
    auto issue_tap = [&](int g) {
        const unsigned char* src = wsrc + (size_t)g * TAP_BYTES;
        const unsigned dst = __builtin_amdgcn_readfirstlane(
            wring_lds + (unsigned)((g & (CIN_WRING - 1)) * TAP_BYTES + wave * WAVE_BYTES));
        unsigned keep;
        if (wave >= NDW) return;
        if constexpr (NDMA == 2) {
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
                : "v"(src), "v"(src + 1024), "s"(dst)
                : "memory", "scc");
        } else {
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %2\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, off\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(src), "s"(dst)
                : "memory");
        }
    };
    issue_tap(0);
    issue_tap(1);
    issue_tap(2);

    // epilogue parameters of this group's CG channels, in kernel channel order (= conv channel order here)
    if (tid < CG) {
        const int co = co0 + tid;
        float sl = 1.f;
        if (p.act == SRMI_ACT_RELU) sl = 0.f;
        else if (p.act == SRMI_ACT_LRELU) sl = p.act_param;
        else if (p.act == SRMI_ACT_PRELU) sl = p.alpha[p.alpha_pc ? co : 0];
        par[tid] = p.scale[co];
        par[64 + tid] = p.shift[co];
        par[128 + tid] = sl;
    }

    int prow[PT], pcol[PT];
#pragma unroll
    for (int k = 0; k < PT; ++k) {
        const int t = min(wave * PT + k, NPT - 1);
        prow[k] = t / TWT;
        pcol[k] = (t - prow[k] * TWT) * 16;
    }
    int boff[PT][3];
#pragma unroll
    for (int k = 0; k < PT; ++k)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int P = prow[k] * ROWP + pcol[k] + pl + dx;
            boff[k][dx] = P * 128 + ((q ^ c64_key(P)) << 4);
        }

    f32x4 acc[PT][NCT];
#pragma unroll
    for (int k = 0; k < PT; ++k)
#pragma unroll
        for (int t = 0; t < NCT; ++t) acc[k][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bf16_t* xin = p.x + (size_t)n * p.H * p.W * p.in_cs;
    for (int kc = 0; kc < p.kchunks; ++kc) {
        // ---- stage the halo tile of input channels [64 kc, 64 kc + 64) -----------------------------
        if (kc > 0) __builtin_amdgcn_s_barrier();      // every wave has consumed the previous chunk's tile
        {
            constexpr int NCH = (TH + 2) * (TW + 2) * 8;
            constexpr int NIT = (NCH + NTH - 1) / NTH;
            const int cvalid = min(p.Cin - kc * 64, 64) >> 3;      // 16-B channel chunks that exist (8 or 4)
            int tid_o;       // opaque copy of tid: keeps hipcc from hoisting the per-load addresses out of the chunk
            asm volatile("v_mov_b32 %0, %1" : "=v"(tid_o) : "v"(tid));   // loop (they would be spilled across the K loop)
            u32x4 v[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = min(tid_o + it * NTH, NCH - 1);
                const int pp = idx >> 3, ch = idx & 7;
                const int row = pp / (TW + 2), col = pp - row * (TW + 2);
                const int gy = min(max(y0 - 1 + row, 0), p.H - 1), gx = min(max(x0 - 1 + col, 0), p.W - 1);
                const int chc = ch < cvalid ? ch : 0;              // clamped: never reads past the pixel's channels
                v[it] = *(const u32x4*)(xin + ((size_t)gy * p.W + gx) * p.in_cs + kc * 64 + chc * 8);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid_o + it * NTH;
                const int pp = idx >> 3, ch = idx & 7;
                const int row = pp / (TW + 2), col = pp - row * (TW + 2);
                const int gy = y0 - 1 + row, gx = x0 - 1 + col;
                const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W && ch < cvalid;
                u32x4 z = v[it];
                if (!ok) z = (u32x4){0u, 0u, 0u, 0u};
                const int P = row * ROWP + col;
                if (idx < NCH) *(u32x4*)(lds + P * 128 + ((ch ^ c64_key(P)) << 4)) = z;
            }
        }
        // tile written; the loads' vmcnt(0) also retired every weight tap issued so far
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        // K steps of this chunk: 9 taps x 2 halves of 32 channels - or, when the chunk holds only 32 real channels (the last
        // chunk of Cin = 96 / 160 / 32: its upper half is zero-filled, its packed weights there are zero), the 9 lower
        // halves alone: the skipped MFMAs would add exact zeros (same bits), 25 % of a 96-channel layer's matrix work.
        const bool halfchunk = p.Cin - kc * 64 <= 32;
        auto ksteps = [&](auto HALFC) {
            constexpr bool HALF = decltype(HALFC)::value;
            constexpr int NS = HALF ? 9 : 18;
            bf16x8 afr[2][NCT], bfr[2][PT];
            {
                const unsigned char* wslot = wring + ((kc * 9) & (CIN_WRING - 1)) * TAP_BYTES + lane * 16;
#pragma unroll
                for (int t = 0; t < NCT; ++t) afr[0][t] = *(const bf16x8*)(wslot + t * 1024);
#pragma unroll
                for (int k = 0; k < PT; ++k) bfr[0][k] = *(const bf16x8*)(lds + boff[k][0]);
            }
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int tap = HALF ? j : j >> 1, half = HALF ? 0 : j & 1;
                const int g = kc * 9 + tap;
                if (HALF || half == 1) {
                    // tap g+1 landed (own copy: counted vmcnt, tap g+2 may stay in flight; others': barrier);
                    // the barrier also retires every wave's reads of tap g-1, whose slot tap g+3 re-uses
                    if (g + 2 < ntap) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (g + 3 < ntap) issue_tap(g + 3);
                }
                if (j + 1 < NS) {
                    const int tapn = HALF ? j + 1 : (j + 1) >> 1, halfn = HALF ? 0 : (j + 1) & 1;
                    const int dyn = tapn / 3, dxn = tapn - 3 * dyn;
                    const unsigned char* wslot =
                        wring + ((kc * 9 + tapn) & (CIN_WRING - 1)) * TAP_BYTES + lane * 16 + halfn * (NCT * 1024);
#pragma unroll
                    for (int t = 0; t < NCT; ++t) afr[(j + 1) & 1][t] = *(const bf16x8*)(wslot + t * 1024);
#pragma unroll
                    for (int k = 0; k < PT; ++k)
                        bfr[(j + 1) & 1][k] =
                            *(const bf16x8*)(lds + ((boff[k][dxn] ^ (halfn << 6)) + dyn * (ROWP * 128)));
                }
                if (!(SKIPH && half == 1 && halfchunk)) {
#pragma unroll
                    for (int k = 0; k < PT; ++k)
#pragma unroll
                        for (int t = 0; t < NCT; ++t)
                            acc[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[j & 1][t], bfr[j & 1][k], acc[k][t], 0, 0, 0);
                }
                if (j + 1 < NS) sched_interleave<0, NCT + PT, NCT * PT>();
            }
        };
        // (the 8-wave, 64-channel variants run at the 256-register budget: a second unrolled loop costs them 23-75 spills)
        constexpr bool HALF_OK = !(NCT == 4 && NW == 8);
        if (HALF_OK && halfchunk) {
            if constexpr (HALF_OK) ksteps(std::true_type{});
        } else {
            ksteps(std::false_type{});
        }
    }

    // ---- epilogue: scale/shift -> slope activation -> + residual (channel-slice strides) -> bf16 store ----
    // SRMI_ACT_ADD_RELU: the slope stage is the identity (slope 1) and the ReLU follows the residual add.
    const bool late_relu = p.act == SRMI_ACT_ADD_RELU;
    // channel map (cin_row_to_cl): lane (pl, q) holds, of channel tile t, rows 4q..4q+3; with NCT = 4 these are
    // channels 8q..8q+7 and 32+8q..32+8q+7 of the group (two 16-B stores), with NCT = 2 channels 8q..8q+7 (one).
    constexpr int NH = NCT / 2;                        // 8-channel runs per lane
    int opaque0;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque0));
#pragma unroll
    for (int k = 0; k < PT; ++k) {
        const int oy = y0 + prow[k], ox = x0 + pcol[k] + pl + opaque0;
        if (!(oy < p.H && ox < p.W) || wave * PT + k >= NPT) continue;
        const size_t pix = ((size_t)n * p.H + oy) * p.W + ox;
#pragma unroll
        for (int hi = 0; hi < NH; ++hi) {
            // parameters of this lane's 8 channels, re-read from LDS per run (keeps the epilogue under the
            // 256-register budget of the 8-wave variants)
            const float* pc = par + 32 * hi + 8 * q;
            const f32x4 sc0 = *(const f32x4*)pc, sc1 = *(const f32x4*)(pc + 4);
            const f32x4 sh0 = *(const f32x4*)(pc + 64), sh1 = *(const f32x4*)(pc + 68);
            const f32x4 sl0 = *(const f32x4*)(pc + 128), sl1 = *(const f32x4*)(pc + 132);
            float v[8];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float z = acc[k][2 * hi + tt][e] * (tt ? sc1[e] : sc0[e]) + (tt ? sh1[e] : sh0[e]);
                    v[4 * tt + e] = z >= 0.f ? z : z * (tt ? sl1[e] : sl0[e]);
                }
            const int cch = co0 + 32 * hi + 8 * q;
            if (p.res) {
                const u32x4 r = *(const u32x4*)(p.res + pix * p.res_cs + cch);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += bf2f((bf16_t)(r[e] & 0xffffu));
                    v[2 * e + 1] += bf2f((bf16_t)(r[e] >> 16));
                }
            }
            if (late_relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
            *(u32x4*)(p.y + pix * p.out_cs + p.out_co + cch) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Persistent, software-pipelined form for layers with many tiles (the RRDB dense / compress convs at 8 x 256 x 256,
// SRCNN's 128 -> 64 at 32 x 384 x 384).  In the kernel above every workgroup stages a chunk, waits for it, computes, stages
// the next: all workgroups of a round do so in step, so HBM idles while the matrix cores run and the other way round
// (rocprofv3: 2.4-3.1 TB/s of traffic AND 0.20-0.35 MFMA-pipe busy, profiles/r3_cfg4_layers.json).  Here one workgroup
// per CU walks through its tiles with the work split over SPECIALISED waves:
//   waves 0-7   (consumers)   : LDS reads, MFMAs, the epilogue and its stores - no vector-memory wait in the K loops;
//   waves 8-9   (tile movers) : the NEXT stage's halo tile (stage = one 64-channel chunk of one 8 x 32 tile: 10 rows of
//                               5 pieces of 1 KiB) by `buffer_load_dwordx4 ... lds` into the other tile buffer, five rows
//                               each, while the consumers work; the swizzle is applied on the source side, border pixels
//                               and the padded channels of a half chunk carry an offset beyond the buffer resource and
//                               read zeros; vmcnt(0) once per stage;
//   waves 10-11 (tap movers)  : the weight ring, in UNITS of three taps (one tap row, 12 / 24 KiB), 3 or 2 slots, half a
//                               unit each (unit g + RU - 1 issued when unit g - 1 is retired, a counted vmcnt per unit),
//                               as their own in-order streams.
// Three streams because vmcnt retires in order per wave: taps (needed three steps later) queued behind a tile (needed a
// stage later, a full HBM latency) would wait for it.  Every wave executes the same barriers: three per stage, one in front
// of every unit (the first one also says: the tile of the stage has landed, the previous stage's buffer is free).  All loops are counted:
// no wave waits on another workgroup.  Arithmetic and order of summation are those of conv_cin_kernel: same bits.
constexpr int CINP_TH = 8, CINP_TW = 32, CINP_ROWP = 40;
constexpr bool CINP_HALF_STEPS = true;      // a 32-channel tail chunk runs its 9 lower half-steps alone (false: 18 steps on the zero-filled chunk; measured 2-4 % slower)
constexpr int CINP_IN_BYTES = (CINP_TH + 2) * CINP_ROWP * 128;      // 51200 = 10 rows x 5 DMA pieces of 1 KiB
constexpr int cinp_ring(int nct) { return nct == 2 ? 3 : 2; }      // weight ring slots of three taps each (LDS: 140 KB / 152 KB per workgroup)

// NCW consumer waves (4: one per SIMD, 4 pixel tiles each; 8: two per SIMD, 2 pixel tiles each), then 2 tile movers, 2 tap movers
template <int NCT, int NCW>
__global__ __launch_bounds__((NCW + 4) * 64, 1) void conv_cinp_kernel(CinParams p, int ntiles) {
    constexpr int TH = CINP_TH, TW = CINP_TW, TWT = 2, ROWP = CINP_ROWP, PT = 16 / NCW;
    constexpr int IN_BYTES = CINP_IN_BYTES;
    constexpr int TAP_BYTES = 2 * NCT * 1024;
    // the weight ring moves in UNITS of three taps (one tap row): one barrier per unit instead of one per tap - at 8-16 MFMAs
    // per wave and tap a barrier per tap re-aligned the two waves of a SIMD every 128-256 cycles
    constexpr int UNIT_BYTES = 3 * TAP_BYTES;
    constexpr int NDU = UNIT_BYTES / 2048;             // LDS-DMA instructions per unit and tap mover (two waves move a unit)
    constexpr int RU = cinp_ring(NCT);                 // ring slots (units)
    constexpr int CG = NCT * 16;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * IN_BYTES + RU * UNIT_BYTES + 3 * 64 * 4];
    unsigned char* const wring = lds + 2 * IN_BYTES;
    float* const par = (float*)(lds + 2 * IN_BYTES + RU * UNIT_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // tiles of this workgroup: XCD x = blockIdx.x & 7 owns a contiguous range of tiles, its CUs take them round robin,
    // so the 32 CUs of an XCD work on 32 neighbouring tiles at any time (halos meet in that XCD's L2)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int per_xcd = (ntiles + 7) >> 3;
    const int mine = min(per_xcd, ntiles - xcd * per_xcd) - slot;       // tiles xcd * per_xcd + slot + it * nslot, it >= 0
    if (mine <= 0) return;
    const int ntl = (mine + nslot - 1) / nslot;
    const int KC = p.kchunks;
    const int nstage = ntl * KC;
    const int cg = blockIdx.y;
    const int tile0 = xcd * per_xcd + slot;

    if (wave == NCW || wave == NCW + 1) {
        // ================================ tile movers (rows 0-4 / 5-9 of the halo tile) ================================
        const int r0 = (wave - NCW) * 5;
        // buffer loads with the LDS as destination: lanes outside the image (or past the chunk's real channels) carry an
        // offset beyond the resource and read zeros - the padding of the conv and of a half chunk
        const int pos = lane & 7, sub = lane >> 3;
        const int ch = pos ^ c64_key(sub);              // c64_key(8 * piece + sub) = c64_key(sub): position pos holds chunk ch
        __amdgpu_buffer_rsrc_t rs;
        unsigned lanebase[5];
        int y0 = 0, rowpitch = p.W * p.in_cs * 2;
        unsigned char* dstbuf = lds;
        auto setup = [&](int s) {
            const int it = s / KC, kc = s - it * KC;
            int b = tile0 + it * nslot;
            const int tx = b % p.tiles_x; b /= p.tiles_x;
            const int ty = b % p.tiles_y;
            const int n = b / p.tiles_y;
            y0 = ty * TH - 1;
            const int x0 = tx * TW - 1;
            const int cvalid = min(p.Cin - kc * 64, 64) >> 3;
            rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)n * p.H * p.W * p.in_cs + kc * 64), 0,
                                                   p.H * rowpitch - kc * 128, 0x00020000);
            dstbuf = lds + (s & 1) * IN_BYTES;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int col = 8 * j + sub, gx = x0 + col;
                const bool ok = col < TW + 2 && gx >= 0 && gx < p.W && ch < cvalid;
                lanebase[j] = ok ? (unsigned)((gx * p.in_cs + ch * 8) * 2) : 0x80000000u;
            }
        };
        auto issue_row = [&](int r) {
#ifdef SRMI_CINP_EXP_FEWER_ROWS      // experiment build (wrong results): 2 of every 5 tile rows are not loaded
            if (r % 5 >= 3) return;
#endif
#ifdef SRMI_ABLATE
            if (p.exp & (1 << 21)) return;
#endif
            const int gy = y0 + r;
            const bool rowok = gy >= 0 && gy < p.H;
            const unsigned rowoff = rowok ? (unsigned)(gy * rowpitch) : 0x80000000u;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const unsigned voff = rowok ? lanebase[j] + rowoff : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dstbuf + (r * 5 + j) * 1024), 16, voff, 0, 0, 0);
            }
        };
        setup(0);
        for (int r = 0; r < 5; ++r) issue_row(r0 + r);
        for (int s = 0; s < nstage; ++s) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // stage s has landed
            __builtin_amdgcn_s_barrier();                            // ... and the consumers are done with stage s - 1
            const bool more = s + 1 < nstage;
            // the whole next tile is requested within the first three tap steps (a stage of lead: what is in flight per CU
            // over the memory latency is the bandwidth the pipeline can draw), in slices so as not to be late at a barrier
            if (more) { setup(s + 1); issue_row(r0); issue_row(r0 + 1); issue_row(r0 + 2); }
            __builtin_amdgcn_s_barrier();                            // unit barriers 1 and 2 of the stage
            if (more) { issue_row(r0 + 3); issue_row(r0 + 4); }
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    if (wave >= NCW + 2) {
        // ================================ tap movers (each one half of every unit) ================================
        const int nunit = nstage * 3;
        const unsigned char* wbase = (const unsigned char*)p.w + (size_t)cg * KC * 9 * TAP_BYTES + (wave - NCW - 2) * (UNIT_BYTES / 2) + lane * 16;
        const unsigned wring_lds = (unsigned)(uintptr_t)(lptr_t)wring + (unsigned)((wave - NCW - 2) * (UNIT_BYTES / 2));
        int islot = 0, is = 0, iu = 0;                               // ring slot / stage / unit-in-stage of the next unit to issue
        auto issue_next = [&]() {
#ifdef SRMI_ABLATE
            if (p.exp & (1 << 22)) return;
#endif
            const int kc = is % KC;
            const unsigned char* src = wbase + (size_t)(kc * 9 + 3 * iu) * TAP_BYTES;      // taps 3 iu .. 3 iu + 2: contiguous
            const unsigned dst = __builtin_amdgcn_readfirstlane(wring_lds + (unsigned)(islot * UNIT_BYTES));
#pragma unroll
            for (int i = 0; i < NDU; ++i) {
                unsigned keep;
                asm volatile(
                    "s_mov_b32 %0, m0\n\t"
                    "s_mov_b32 m0, %2\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %1, off\n\t"
                    "s_mov_b32 m0, %0"
                    : "=&s"(keep)
                    : "v"(src + i * 1024), "s"(dst + (unsigned)(i * 1024))
                    : "memory");
            }
            if (++islot == RU) islot = 0;
            if (++iu == 3) { iu = 0; ++is; }
        };
        // units 0 .. RU-2 up front (nunit >= 3 > RU - 2); unit g + RU - 1 follows when unit g - 1 is retired
        for (int i = 0; i < RU - 1; ++i) issue_next();
        for (int g = 0; g < nunit; ++g) {
            // unit g landed; the units issued after it (RU - 2, none at the end of the stream) may stay in flight
            if (RU > 2 && g + RU - 2 < nunit) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RU - 2) * NDU) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                            // publishes unit g; the consumers' reads of unit g - 1 are complete
            if (g + RU - 1 < nunit) issue_next();
        }
        return;
    }

    // ================================ consumers (waves 0 .. NCW-1) ================================
    const int pl = lane & 15, q = lane >> 4;
    const int co0 = cg * CG;
    if (tid < CG) {
        const int co = co0 + tid;
        float sl = 1.f;
        if (p.act == SRMI_ACT_RELU) sl = 0.f;
        else if (p.act == SRMI_ACT_LRELU) sl = p.act_param;
        else if (p.act == SRMI_ACT_PRELU) sl = p.alpha[p.alpha_pc ? co : 0];
        par[tid] = p.scale[co];
        par[64 + tid] = p.shift[co];
        par[128 + tid] = sl;
    }
    int prow[PT], pcol[PT];
#pragma unroll
    for (int k = 0; k < PT; ++k) {
        const int t = wave * PT + k;
        prow[k] = t / TWT;
        pcol[k] = (t - prow[k] * TWT) * 16;
    }
    int boff[PT][3];
#pragma unroll
    for (int k = 0; k < PT; ++k)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int P = prow[k] * ROWP + pcol[k] + pl + dx;
            boff[k][dx] = P * 128 + ((q ^ c64_key(P)) << 4);
        }
    f32x4 acc[PT][NCT];
#pragma unroll
    for (int k = 0; k < PT; ++k)
#pragma unroll
        for (int t = 0; t < NCT; ++t) acc[k][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool late_relu = p.act == SRMI_ACT_ADD_RELU;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // par[] written before the first barrier

    int s = 0, slot0 = 0;                                          // slot0: ring slot of the stage's first unit
    for (int it = 0; it < ntl; ++it) {
        for (int kc = 0; kc < KC; ++kc, ++s) {
            __builtin_amdgcn_s_barrier();                            // the tile of stage s is in buffer s & 1
            // B-operand addresses of this stage; the opaque buffer offset keeps hipcc from hoisting the 18 x PT per-step
            // addresses out of the stage loop (they would live - and spill - across everything)
            int tb;
            asm volatile("v_mov_b32 %0, %1" : "=v"(tb) : "s"((s & 1) * IN_BYTES));
            int bo[PT][3][2];
#pragma unroll
            for (int k = 0; k < PT; ++k)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    bo[k][dx][0] = boff[k][dx] + tb;
                    bo[k][dx][1] = (boff[k][dx] ^ 64) + tb;
                }
            const bool halfchunk = p.Cin - kc * 64 <= 32;
            auto ksteps = [&](auto HALFC) {
                constexpr bool HALF = decltype(HALFC)::value;
                constexpr int NS = HALF ? 9 : 18;
                bf16x8 afr[2][NCT], bfr[2][PT];
                {
                    const unsigned char* wslot = wring + slot0 * UNIT_BYTES + lane * 16;
#pragma unroll
                    for (int t = 0; t < NCT; ++t) afr[0][t] = *(const bf16x8*)(wslot + t * 1024);
#pragma unroll
                    for (int k = 0; k < PT; ++k) bfr[0][k] = *(const bf16x8*)(lds + bo[k][0][0]);
                }
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    {   // the step after this one opens a new unit (tap 3 or 6): every read of the old one complete, then the barrier
                        constexpr int SPU = HALF ? 3 : 6;                      // steps per unit
                        if (j + 1 < NS && (j + 1) % SPU == 0) {
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();
                        }
                    }
                    if (j + 1 < NS
#ifdef SRMI_ABLATE
                        && !(p.exp & (1 << 24))
#endif
                    ) {
                        const int tapn = HALF ? j + 1 : (j + 1) >> 1, halfn = HALF ? 0 : (j + 1) & 1;
                        const int dyn = tapn / 3, dxn = tapn - 3 * dyn;
                        const unsigned char* wslot =
                            wring + ((slot0 + tapn / 3) % RU) * UNIT_BYTES + (tapn % 3) * TAP_BYTES + lane * 16 + halfn * (NCT * 1024);
#pragma unroll
                        for (int t = 0; t < NCT; ++t) afr[(j + 1) & 1][t] = *(const bf16x8*)(wslot + t * 1024);
#pragma unroll
                        for (int k = 0; k < PT; ++k)
                            bfr[(j + 1) & 1][k] =
                                *(const bf16x8*)(lds + (bo[k][dxn][halfn] + dyn * (ROWP * 128)));
                    }
#ifdef SRMI_ABLATE
                    if (!(p.exp & (1 << 20)))
#endif
                    {
#pragma unroll
                    for (int k = 0; k < PT; ++k)
#pragma unroll
                        for (int t = 0; t < NCT; ++t)
                            acc[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[j & 1][t], bfr[j & 1][k], acc[k][t], 0, 0, 0);
                    }
                    if (j + 1 < NS) sched_interleave<0, NCT + PT, NCT * PT>();
                }
            };
            if (halfchunk && CINP_HALF_STEPS) ksteps(std::true_type{});
            else ksteps(std::false_type{});
            slot0 = (slot0 + 3) % RU;
        }
        // ---- epilogue of this tile (as conv_cin_kernel's); the movers are already at work on the next one ----
        int b = tile0 + it * nslot;
        const int tx = b % p.tiles_x; b /= p.tiles_x;
        const int ty = b % p.tiles_y;
        const int n = b / p.tiles_y;
        const int y0 = ty * TH, x0 = tx * TW;
        constexpr int NH = NCT / 2;
        // every residual line of the tile in flight before the first use (one memory round trip per tile, not one per line)
        u32x4 rres[PT][NH];
        if (p.res) {
#pragma unroll
            for (int k = 0; k < PT; ++k) {
                const int oy = min(y0 + prow[k], p.H - 1), ox = min(x0 + pcol[k] + pl, p.W - 1);
                const size_t pix = ((size_t)n * p.H + oy) * p.W + ox;
#pragma unroll
                for (int hi = 0; hi < NH; ++hi) rres[k][hi] = *(const u32x4*)(p.res + pix * p.res_cs + co0 + 32 * hi + 8 * q);
            }
        }
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            const int oy = y0 + prow[k], ox = x0 + pcol[k] + pl;
            if (oy < p.H && ox < p.W
#ifdef SRMI_ABLATE
                && !(p.exp & (1 << 23))
#endif
            ) {
                const size_t pix = ((size_t)n * p.H + oy) * p.W + ox;
#pragma unroll
                for (int hi = 0; hi < NH; ++hi) {
                    const float* pc = par + 32 * hi + 8 * q;
                    const f32x4 sc0 = *(const f32x4*)pc, sc1 = *(const f32x4*)(pc + 4);
                    const f32x4 sh0 = *(const f32x4*)(pc + 64), sh1 = *(const f32x4*)(pc + 68);
                    const f32x4 sl0 = *(const f32x4*)(pc + 128), sl1 = *(const f32x4*)(pc + 132);
                    float v[8];
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float z = acc[k][2 * hi + tt][e] * (tt ? sc1[e] : sc0[e]) + (tt ? sh1[e] : sh0[e]);
                            v[4 * tt + e] = z >= 0.f ? z : z * (tt ? sl1[e] : sl0[e]);
                        }
                    const int cch = co0 + 32 * hi + 8 * q;
                    if (p.res) {
                        const u32x4 r = rres[k][hi];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[2 * e] += bf2f((bf16_t)(r[e] & 0xffffu));
                            v[2 * e + 1] += bf2f((bf16_t)(r[e] >> 16));
                        }
                    }
                    if (late_relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
                    *(u32x4*)(p.y + pix * p.out_cs + p.out_co + cch) = o;
                }
            }
#pragma unroll
            for (int t = 0; t < NCT; ++t) acc[k][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
}

// A-row r (0..15) of channel tile t (0..NCT-1) -> channel within the group
__host__ __device__ __forceinline__ int cin_row_to_cl(int t, int r) { return 32 * (t >> 1) + 8 * (r >> 2) + 4 * (t & 1) + (r & 3); }

// OIHW f32 -> bf16 A fragments:  packed[((((cg*KC + kc)*18 + ks)*NCT + t)*64 + lane)*8 + j] = W[co][ci][ky][kx],
//   tap = ks>>1, ci = 64 kc + 32 (ks&1) + 8 (lane>>4) + j (zero when >= Cin), co = cg*16*NCT + cin_row_to_cl(t, lane&15)
__global__ void cin_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cin, int KC, int NCT,
                                size_t total) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx & 7);
        const int lane = (int)((idx >> 3) & 63);
        size_t rest = idx >> 9;
        const int t = (int)(rest % NCT); rest /= NCT;
        const int ks = (int)(rest % 18); rest /= 18;
        const int kc = (int)(rest % KC);
        const int cg = (int)(rest / KC);
        const int co = cg * 16 * NCT + cin_row_to_cl(t, lane & 15);
        const int ci = 64 * kc + 32 * (ks & 1) + 8 * (lane >> 4) + j;
        const int tap = ks >> 1;
        out[idx] = ci < Cin ? f2bf(w[((size_t)co * Cin + ci) * 9 + tap]) : (bf16_t)0;
    }
}

static int cin_nct(const srmi_conv_desc& d) { return d.Cout % 64 == 0 ? 4 : 2; }
static int cin_kchunks(const srmi_conv_desc& d) { return (d.Cin + 63) / 64; }

bool cin_supported(const srmi_conv_desc& d) {
    if (d.compute != SRMI_COMPUTE_BF16) return false;
    if (d.kh != 3 || d.kw != 3 || d.stride != 1 || d.pad != 1 || d.shuffle != 1) return false;
    if (d.Cin % 32 != 0 || d.Cin < 32 || d.Cin > 1024 || d.Cout % 32 != 0 || d.Cout > 1024) return false;
    if (d.in_dtype != SRMI_BF16 || d.out_dtype != SRMI_BF16) return false;
    if (d.in_layout != SRMI_NHWC || d.out_layout != SRMI_NHWC) return false;
    if (d.act == SRMI_ACT_TANH || d.act == SRMI_ACT_SIGMOID) return false;
    const int in_cs = d.in_cstride ? d.in_cstride : d.Cin;
    const int out_cs = d.out_cstride ? d.out_cstride : d.Cout;
    const int res_cs = d.res_cstride ? d.res_cstride : d.Cout;
    if (in_cs % 8 != 0 || out_cs % 8 != 0 || d.out_coffset % 8 != 0 || res_cs % 8 != 0) return false;
    return true;
}

size_t cin_packed_bytes(const srmi_conv_desc& d) {
    return (size_t)d.Cout * cin_kchunks(d) * 64 * 9 * sizeof(bf16_t);
}

int cin_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s) {
    const size_t total = (size_t)d.Cout * cin_kchunks(d) * 64 * 9;
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cin_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w_oihw, (bf16_t*)w_packed, d.Cin,
                       cin_kchunks(d), cin_nct(d), total);
    return hip_check(hipGetLastError(), "cin_pack_kernel");
}

// CUs of the current device (cached per device index)
static int cin_cus() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int c = cached[dev].load(std::memory_order_relaxed);
    if (!c) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cached[dev].store(c, std::memory_order_relaxed);
    }
    return c;
}

int cin_launch(const srmi_conv_args& a, const ConvGeom& g, hipStream_t s) {
    const srmi_conv_desc& d = a.d;
    if (((uintptr_t)a.x | (uintptr_t)a.y | (uintptr_t)a.w | (uintptr_t)a.residual) & 15) {
        set_error("cin_bf16: x, w, y and residual must be 16-byte aligned");
        return SRMI_EINVAL;
    }
    if (a.residual && d.res_dtype != SRMI_BF16) { set_error("cin_bf16: residual must be bf16"); return SRMI_EUNSUPPORTED; }
    CinParams p;
    p.x = (const bf16_t*)a.x; p.w = (const bf16_t*)a.w;
    p.scale = a.scale; p.shift = a.shift; p.alpha = a.alpha;
    p.res = (const bf16_t*)a.residual; p.y = (bf16_t*)a.y;
    p.N = d.N; p.H = d.H; p.W = d.W; p.Cin = d.Cin; p.Cout = d.Cout; p.kchunks = cin_kchunks(d);
    p.in_cs = g.in_cs; p.out_cs = g.out_cs; p.out_co = d.out_coffset; p.res_cs = g.res_cs;
    p.act = d.act; p.act_param = d.act_param; p.alpha_pc = d.alpha_per_channel;
    p.exp = exp_flags();
    if (d.N == 0 || d.H == 0 || d.W == 0) return SRMI_OK;
    const int nct = cin_nct(d);
    const unsigned gy = (unsigned)(d.Cout / (16 * nct));
    // many tiles: the persistent, pipelined form (one workgroup per CU; off: SRMI_DBG bit 17 of the ablation build, or a
    // library built with -DSRMI_NO_CINP, which is what tools/ab_cinp.sh compares against)
    {
        const long long tx8 = (d.W + CINP_TW - 1) / CINP_TW, ty8 = (d.H + CINP_TH - 1) / CINP_TH;
        const long long nt = (long long)d.N * tx8 * ty8;
        const int cus = cin_cus();
#ifndef SRMI_NO_CINP
        if (cus >= 8 && nt >= 4LL * cus && nt <= 0x7fffffffLL && gy <= 65535u && !(dbg_flags() & (1 << 17)) &&
            (long long)d.H * d.W * g.in_cs * 2 < 0x7fffffffLL) {      // the tile mover's 32-bit buffer offsets
            p.tiles_x = (int)tx8; p.tiles_y = (int)ty8;
            const dim3 grid((unsigned)(cus & ~7), gy);
            // eight consumer waves (two per SIMD, 2 pixel tiles each: one fills the other's barrier / LDS-latency bubbles) beat four
            // (4 pixel tiles each) on every layer measured: 64->32 36.1 -> 31.9 us, 128->32 56.4 -> 51.1, 160->64 105 -> 100.7,
            // 128->64 81.5 -> 75.5 at 8 x 256 x 256 (A/B build: -DSRMI_CINP_NCW4)
#ifdef SRMI_CINP_NCW4
            if (nct == 4) hipLaunchKernelGGL((conv_cinp_kernel<4, 4>), grid, dim3(512), 0, s, p, (int)nt);
            else hipLaunchKernelGGL((conv_cinp_kernel<2, 4>), grid, dim3(512), 0, s, p, (int)nt);
#else
            if (nct == 4) hipLaunchKernelGGL((conv_cinp_kernel<4, 8>), grid, dim3(768), 0, s, p, (int)nt);
            else hipLaunchKernelGGL((conv_cinp_kernel<2, 8>), grid, dim3(768), 0, s, p, (int)nt);
#endif
            return hip_check(hipGetLastError(), "conv_cinp_kernel");
        }
#endif
    }
    int tile = pick_tile3x3(d.N, d.H, d.W);
    static const int th[4] = {6, 8, 4, 8}, tw[4] = {96, 64, 32, 32};
    // 32 output channels per workgroup and several rounds of tiles: 8 x 32 tiles with 4 waves need 77 KB of LDS, so TWO
    // workgroups share a CU and one's synchronous chunk staging runs under the other's MFMAs (cfg4 RRDB-23: 8.13 ->
    // 7.71 ms; the same idea at 64 channels per workgroup - 8 x 16 tiles - was slower, 7.89 -> 8.09 ms).  SRMI_DBG bit 15: off.
    if (nct == 2 && tile != 2 && !(dbg_flags() & (1 << 15)) &&
        (long long)d.N * ((d.W + 31) / 32) * ((d.H + 7) / 8) >= 1024) tile = 3;
    p.tiles_x = (d.W + tw[tile] - 1) / tw[tile]; p.tiles_y = (d.H + th[tile] - 1) / th[tile];
    const long long nb = (long long)d.N * p.tiles_x * p.tiles_y;
    if (nb > 0x7fffffffLL || gy > 65535u) { set_error("cin_bf16: grid too large"); return SRMI_EINVAL; }
    const dim3 grid((unsigned)nb, gy);
    // the two big tiles run with 8 waves (two per SIMD hide each other's LDS / barrier / staging bubbles)
    if (tile == 3) hipLaunchKernelGGL((conv_cin_kernel<8, 2, 2, 4, 2>), grid, dim3(256), 0, s, p);
    else if (tile == 0 && nct == 4 && d.Cin % 64 == 32) hipLaunchKernelGGL((conv_cin_kernel<6, 6, 4, 8, 1, true>), grid, dim3(512), 0, s, p);
    else if (tile == 0 && nct == 4) hipLaunchKernelGGL((conv_cin_kernel<6, 6, 4, 8>), grid, dim3(512), 0, s, p);
    else if (tile == 0) hipLaunchKernelGGL((conv_cin_kernel<6, 6, 2, 8>), grid, dim3(512), 0, s, p);
    else if (tile == 1 && nct == 4 && d.Cin % 64 == 32) hipLaunchKernelGGL((conv_cin_kernel<8, 4, 4, 8, 1, true>), grid, dim3(512), 0, s, p);
    else if (tile == 1 && nct == 4) hipLaunchKernelGGL((conv_cin_kernel<8, 4, 4, 8>), grid, dim3(512), 0, s, p);
    else if (tile == 1) hipLaunchKernelGGL((conv_cin_kernel<8, 4, 2, 8>), grid, dim3(512), 0, s, p);
    else if (nct == 4) hipLaunchKernelGGL((conv_cin_kernel<4, 2, 4, 4>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv_cin_kernel<4, 2, 2, 4>), grid, dim3(256), 0, s, p);
    return hip_check(hipGetLastError(), "conv_cin_kernel");
}

}  // namespace srmi
