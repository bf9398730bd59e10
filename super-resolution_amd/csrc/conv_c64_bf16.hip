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
#include <stdlib.h>

#include <type_traits>

#include "c64_common.h"

namespace srmi {


struct C64Params {
    const bf16_t* x; const bf16_t* w;
    C64Epi epi;   // scale/shift/alpha/residual/output, store mode (0 plain, 1 PixelShuffle(2) CRD, 2 depth_to_space(2) DCR)
    int N, H, W, nchunk, in_cs;
    int tiles_x, tiles_y;
    int dbg;  // ablation switches (SRMI_DBG env, investigation only): 1 skip staging, 2 skip K loop, 4 skip epilogue
    float* xcopy;   // NULL, or [N, 64, H, W] f32: NCHW copy of the input, written from the staged tile (srmi_conv_args.x_copy)
};



constexpr int C64_WRING = 4;          // weight ring slots (one 3x3 tap = 8 KiB of A fragments per slot)
constexpr int C64_TAP_BYTES = 8192;   // 2 k-steps x 4 channel tiles x 64 lanes x 16 B

// NW = waves per workgroup: 4 (one per SIMD, 512 registers each) or 8 (two per SIMD, 256 each: the second wave of a
// SIMD fills the first one's LDS / barrier / epilogue bubbles; same tile and LDS, half the pixel tiles per wave).
// SLOTS = ring depth (taps), WGPC = workgroups per CU the register budget is cut for, PITCH = LDS row pitch granule
// (16: padded, the swizzle key can be taken from the pixel index; 2: unpadded, column-keyed), MAXCH = output chunks
// the parameter area holds.  <6, 3, 4, 3, 2, 2, 4>: a 6 x 48 tile in 77 KiB of LDS and 256 registers - TWO workgroups
// per CU, so that one's staging / epilogue / store-acknowledge waits run under the other's MFMA phases (multi-round
// launches, where the two drift out of phase).
template <int TH, int TWT, int NW, int SLOTS = C64_WRING, int WGPC = 1, int PITCH = 16, int MAXCH = C64_MAX_CHUNKS>
__global__ __launch_bounds__(NW * 64, WGPC) void conv_c64_kernel(C64Params p) {
    constexpr int NTH = NW * 64;
    constexpr int TW = TWT * 16;
    constexpr int NPT = TH * TWT;
    constexpr int PT = (NPT + NW - 1) / NW;            // waves past the end recompute the last tile and store nothing
    constexpr int WAVE_BYTES = C64_TAP_BYTES / NW;     // each wave moves 1/NW of every tap
    constexpr int NDMA = WAVE_BYTES / 1024;            // LDS-DMA instructions per wave and tap (2 or 1)
    constexpr int ROWP = (TW + 2 + PITCH - 1) / PITCH * PITCH;  // LDS row pitch in pixels: even, so a row starts on a 256-B bank row
    static_assert(ROWP % 2 == 0, "pixel pairs must not straddle rows");
    auto swz = [](int P, int col) { return ROWP % 8 == 0 ? c64_key(P) : c64_key(col); };   // swizzle key: the same for every row
    constexpr int IN_BYTES = (TH + 2) * ROWP * 128;
    // ONE shared array (input tile | weight ring): a second __shared__ object next to an LDS-DMA
    // target makes hipcc drain vmcnt before every ds_read (cdna guide, "three .s-level traps").
    __shared__ __attribute__((aligned(16))) unsigned char lds[IN_BYTES + SLOTS * C64_TAP_BYTES + MAXCH * 768];
    unsigned char* const wring = lds + IN_BYTES;
    float* const par = (float*)(lds + IN_BYTES + SLOTS * C64_TAP_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15;  // pixel within a 16-pixel tile (MFMA column)
    const int q = lane >> 4;   // k-quarter for operands / channel group for results

    // XCD-aware tile order: consecutive workgroup ids go round-robin to the 8 XCDs (one L2 each), so give each
    // XCD a CONTIGUOUS run of tiles - vertically adjacent tiles then share their halo rows through one L2.
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0 && !(p.dbg & 32)) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- weight stream: packed A fragments, one tap (8 KiB) per ring slot, moved by LDS-DMA ------
    // wave w copies KiB 2w and 2w+1 of each tap; the LDS image is lane-linear (= fragment order).
    const int ntap = p.nchunk * 9;
    const unsigned char* wsrc = (const unsigned char*)p.w + wave * WAVE_BYTES + lane * 16;
    // The DMA is issued from inline asm on purpose: hipcc tracks a builtin LDS-DMA as an LDS write that
    // may alias every later ds_read of the same array and drains vmcnt(0) in front of it, which would
    // serialise the ring.  Here the hand-off is ordered by hand (counted vmcnt + s_barrier below).
    const unsigned wring_lds = (unsigned)(uintptr_t)(lptr_t)wring;
    auto issue_tap = [&](int g) {
        const unsigned char* src = wsrc + (size_t)g * C64_TAP_BYTES;
        const unsigned dst = __builtin_amdgcn_readfirstlane(
            wring_lds + (unsigned)((g % SLOTS) * C64_TAP_BYTES + wave * WAVE_BYTES));
        unsigned keep;
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
#pragma unroll
    for (int g = 0; g < SLOTS - 1; ++g)
        if (g < ntap) issue_tap(g);

    C64ParRegs pregs;
    c64_params_load(pregs, p.epi, p.nchunk, tid);
    // ---- stage the input halo tile: global (NHWC bf16) -> registers -> swizzled LDS ----------
    // All loads are issued before the first is consumed: one HBM/L2 round trip for the whole tile.  Row-oriented so
    // that almost all index arithmetic is wave-uniform or an immediate: part A gives a thread one (column, 16-B chunk)
    // and walks the rows, part B spreads the left-over columns.  Pixels outside the image are read through a buffer
    // descriptor at an out-of-range offset: zeros without touching memory (the conv's zero padding), no select.
    if (!(p.dbg & 1)) {
        constexpr int XR = TH + 2, XC = TW + 2;
        constexpr int CPT = NTH / 8;                  // columns per pass
        constexpr int NA = XC / CPT, RB = XC - NA * CPT;
        constexpr int NB = (XR * RB * 8 + NTH - 1) / NTH;
        constexpr unsigned OOB_COL = 0x80000000u, OOB_ROW = 0x40000000u;   // image bytes <= 2^30 (checked on the host)
        const unsigned img_bytes = (unsigned)p.H * p.W * p.in_cs * 2;
        const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.x + (size_t)n * p.H * p.W * p.in_cs), 0, img_bytes, 0x00020000);
        const unsigned row_bytes = (unsigned)p.W * p.in_cs * 2;
        const int ch = tid & 7, c0 = tid >> 3;
        u32x4 va[NA > 0 ? NA * XR : 1], vb[NB > 0 ? NB : 1];
        unsigned la[NA > 0 ? NA : 1], lb[NB > 0 ? NB : 1];
#pragma unroll
        for (int g = 0; g < NA; ++g) {
            const int col = g * CPT + c0, gx = x0 - 1 + col;
            const unsigned coff = (gx >= 0 && gx < p.W) ? (unsigned)gx * p.in_cs * 2 + ch * 16 : OOB_COL;
            la[g] = col * 128 + ((ch ^ c64_key(col)) << 4);     // ROWP % 16 == 0: the swizzle key depends on the column only
#pragma unroll
            for (int row = 0; row < XR; ++row) {
                const int gy = y0 - 1 + row;                               // wave-uniform
                const unsigned roff = (gy >= 0 && gy < p.H) ? (unsigned)gy * row_bytes : OOB_ROW;
                va[g * XR + row] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, coff + roff, 0, 0));
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int sidx = tid + i * NTH;
            const int row = sidx / (RB * 8), rem = sidx - row * (RB * 8);
            const int col = NA * CPT + (rem >> 3), chb = rem & 7;
            const int gy = y0 - 1 + row, gx = x0 - 1 + col;
            const bool ok = sidx < XR * RB * 8 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const unsigned off = ok ? (unsigned)gy * row_bytes + (unsigned)gx * p.in_cs * 2 + chb * 16 : OOB_COL;
            vb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, off, 0, 0));
            lb[i] = (row * ROWP + col) * 128 + ((chb ^ c64_key(col)) << 4);
        }
#pragma unroll
        for (int g = 0; g < NA; ++g)
#pragma unroll
            for (int row = 0; row < XR; ++row) *(u32x4*)(lds + la[g] + row * (ROWP * 128)) = va[g * XR + row];
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (tid + i * NTH < XR * RB * 8) *(u32x4*)(lds + lb[i]) = vb[i];
    }

    c64_params_store(par, pregs, p.nchunk, tid);
    // input tile written (own ds_writes: lgkmcnt; the staging loads' vmcnt(0) also retired taps 0..2)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- optional NCHW f32 copy of the INPUT (srmi_conv_args.x_copy): the tile's interior is in LDS anyway ----
    // A unit = (4 consecutive pixels of a tile row, 8 channels): four 16-byte LDS reads, eight 16-byte stores (one per
    // channel plane).  The octet index runs fastest over the lanes: 8 lanes read one pixel's whole 128-byte line (two
    // pixels of equal parity per 16-lane group: 2-way conflicts at worst) and a store instruction writes whole 128-byte
    // lines of 8 channel planes.  ~150 instructions per thread, once per workgroup, under the first weight taps' DMA.
    if (p.xcopy) {
        constexpr int QPR = TW / 4;
        constexpr int UNITS = 8 * TH * QPR;
        const size_t plane = (size_t)p.H * p.W;
        float* const dstn = p.xcopy + (size_t)n * 64 * plane;
        for (int u = tid; u < UNITS; u += NTH) {
            const int oct = u & 7, t = u >> 3;
            const int r = t / QPR, qd = t - r * QPR;
            const int gy = y0 + r, gx = x0 + 4 * qd;
            if (gy < p.H && gx < p.W) {                    // W % 4 == 0 (checked on the host): whole quads
                u32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = 1 + 4 * qd + j;
                    const int P = (r + 1) * ROWP + col;
                    v[j] = *(const u32x4*)(lds + P * 128 + ((oct ^ swz(P, col)) << 4));
                }
                float* const d0 = dstn + (size_t)(8 * oct) * plane + (size_t)gy * p.W + gx;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned w = v[j][c >> 1];
                        o[j] = __builtin_bit_cast(float, (c & 1) ? (w & 0xffff0000u) : (w << 16));
                    }
                    *(f32x4*)(d0 + c * plane) = o;
                }
            }
        }
    }

    // Pixel tiles per wave.  With two waves per SIMD (w and w + NW/2 share one) the SIMD's MFMA time per K-step is the SUM
    // of its two waves' tile counts: 36 tiles = 4 x (5 + 4) fill every slot where 8 x 5 left four of 40 recomputing a
    // neighbour's tile (see conv_resblock_bf16.hip).  Two instantiations behind a wave-uniform branch, same barriers.
    constexpr bool SPLIT = NW == 8 && NPT <= (NW / 2) * (2 * PT - 1);
    constexpr int PTB = SPLIT ? PT - 1 : PT;
    auto body = [&](auto PC, const int tbase) {
    constexpr int PTc = decltype(PC)::value;
    // per-wave pixel tiles: tile index t = tbase + k -> (row, 16-col group)
    int prow[PTc], pcol[PTc];
#pragma unroll
    for (int k = 0; k < PTc; ++k) {
        const int t = min(tbase + k, NPT - 1);
        prow[k] = t / TWT;
        pcol[k] = (t - prow[k] * TWT) * 16;
    }
    // loop-invariant LDS byte offsets of this lane's B fragments: [pixel tile][dx], channel half 0
    // (half 1 flips bit 6: chunk index cb = 4 only toggles bit 2 of the swizzled chunk).
    int boff[PTc][3];
#pragma unroll
    for (int k = 0; k < PTc; ++k)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int col = pcol[k] + pl + dx;
            const int P = prow[k] * ROWP + col;
            boff[k][dx] = P * 128 + ((q ^ swz(P, col)) << 4);
        }
#pragma unroll
    for (int k = 0; k < PTc; ++k)
        if (tbase + k >= NPT) prow[k] = 1 << 20;   // padding tile: the epilogue's bounds check drops it

    for (int s = 0; s < p.nchunk; ++s) {
        f32x4 acc[PTc][4];
#pragma unroll
        for (int k = 0; k < PTc; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[k][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // K loop = 18 steps (9 taps x 2 channel halves), fully unrolled and software-pipelined by hand:
        // the 13 operand fragments of step j+1 are read from LDS while the 36 MFMAs of step j run
        // (one wave per SIMD: nothing else hides the ds_read latency).  Weight taps arrive through a
        // 4-slot LDS-DMA ring; the hand-off for tap g+1 (counted vmcnt + barrier) sits in the MIDDLE of
        // tap g so that the first fragments of the next tap can be prefetched like any others.
        bf16x8 afr[2][4], bfr[2][PTc];
        {
            const unsigned char* wslot = wring + ((s * 9) % SLOTS) * C64_TAP_BYTES + lane * 16;
#pragma unroll
            for (int t = 0; t < 4; ++t) afr[0][t] = *(const bf16x8*)(wslot + t * 1024);
#pragma unroll
            for (int k = 0; k < PTc; ++k) bfr[0][k] = *(const bf16x8*)(lds + boff[k][0]);
        }
        if (!(p.dbg & 2))
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            const int tap = j >> 1, half = j & 1;
            const int g = s * 9 + tap;
            if (half == 1) {
                // tap g+1 landed: own copies by counted vmcnt (tap g+2 stays in flight), the other waves'
                // by the barrier, which also retires every wave's reads of tap g-1 -> its slot is free.
                // (taps g+1 .. g+SLOTS-2 are in flight; all but g+1 may stay so)
                if (SLOTS >= 4 && g + 2 < ntap) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (SLOTS - 3)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (g + SLOTS - 1 < ntap) issue_tap(g + SLOTS - 1);
            }
            if (j + 1 < 18) {
                const int tapn = (j + 1) >> 1, halfn = (j + 1) & 1;
                const int dyn = tapn / 3, dxn = tapn - 3 * dyn;
                const unsigned char* wslot =
                    wring + ((s * 9 + tapn) % SLOTS) * C64_TAP_BYTES + lane * 16 + halfn * 4096;
#pragma unroll
                for (int t = 0; t < 4; ++t) afr[(j + 1) & 1][t] = *(const bf16x8*)(wslot + t * 1024);
#pragma unroll
                for (int k = 0; k < PTc; ++k)
                    bfr[(j + 1) & 1][k] =
                        *(const bf16x8*)(lds + ((boff[k][dxn] ^ (halfn << 6)) + dyn * (ROWP * 128)));
            }
#pragma unroll
            for (int k = 0; k < PTc; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[j & 1][t], bfr[j & 1][k], acc[k][t], 0, 0, 0);
            if (j + 1 < 18) {
                // interleave: one LDS read per ~3 MFMAs (13 reads under 36 MFMAs)
                sched_interleave<0, 4 + PTc, 4 * PTc>();
            }
        }

        if (!(p.dbg & 4)) c64_epilogue<PTc>(acc, par + s * 192, p.epi, s, n, y0, x0, prow, pcol, pl, q);
    }
    };
    if constexpr (SPLIT) {
        if (wave < NW / 2) body(std::integral_constant<int, PT>{}, wave * PT);
        else body(std::integral_constant<int, PTB>{}, (NW / 2) * PT + (wave - NW / 2) * PTB);
    } else {
        body(std::integral_constant<int, PT>{}, wave * PT);
    }
}

// OIHW f32 -> bf16 MFMA A-fragment order:
//   packed[(((s*18 + ks)*4 + t)*64 + lane)*8 + j] = W[co][ci][ky][kx]
//   tap = ks>>1 (ky = tap/3, kx = tap%3), ci = (ks&1)*32 + 8*(lane>>4) + j,
//   A-row r = lane&15 of channel tile t  <->  channel-in-chunk cl = c64_row_to_cl(t, r),
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
        const int cl = c64_row_to_cl(t, r);
        const int co = c64_conv_channel(mode, s, cl);
        const int ci = (ks & 1) * 32 + 8 * qq + j;
        const int tap = ks >> 1;
        out[idx] = f2bf(w[((size_t)co * 64 + ci) * 9 + tap]);
    }
}

bool c64_supported(const srmi_conv_desc& d) {
    if (d.compute != SRMI_COMPUTE_BF16) return false;
    if (d.res_cstride && (d.res_cstride % 8 != 0 || d.res_cstride < d.Cout / (d.shuffle * d.shuffle))) return false;   // (16-byte residual reads)
    if (d.kh != 3 || d.kw != 3 || d.stride != 1 || d.pad != 1 || d.Cin != 64) return false;
    if (d.Cout % 64 != 0 || d.Cout <= 0 || d.Cout > 64 * C64_MAX_CHUNKS) return false;
    if (d.in_dtype != SRMI_BF16 || d.out_dtype != SRMI_BF16) return false;
    if (d.in_layout != SRMI_NHWC || d.out_layout != SRMI_NHWC) return false;
    if (d.shuffle != 1 && !(d.shuffle == 2 && d.Cout == 256)) return false;
    if (d.act == SRMI_ACT_TANH || d.act == SRMI_ACT_SIGMOID || d.act == SRMI_ACT_ADD_RELU) return false;
    const int in_cs = d.in_cstride ? d.in_cstride : d.Cin;
    if (in_cs % 8 != 0) return false;
    // the staging addresses one image through a 32-bit buffer descriptor with two out-of-range markers
    if ((unsigned long long)d.H * d.W * in_cs * 2 > (1ull << 30)) return false;
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
    p.x = (const bf16_t*)a.x; p.w = (const bf16_t*)a.w;
    p.epi.scale = a.scale; p.epi.shift = a.shift; p.epi.alpha = a.alpha;
    p.epi.res = (const bf16_t*)a.residual; p.epi.y = (bf16_t*)a.y; p.epi.res_cs = d.res_cstride;
    p.epi.H = d.H; p.epi.W = d.W; p.epi.Cout = d.Cout; p.epi.out_cs = g.out_cs; p.epi.out_co = d.out_coffset;
    p.epi.mode = c64_mode(d);
    p.epi.act = d.act; p.epi.act_param = d.act_param; p.epi.alpha_pc = d.alpha_per_channel;
    p.N = d.N; p.H = d.H; p.W = d.W; p.nchunk = d.Cout / 64; p.in_cs = g.in_cs;
    p.dbg = dbg_flags();
    p.xcopy = a.x_copy;
    if (p.xcopy && (g.in_cs != 64 || (d.W & 3) || ((uintptr_t)p.xcopy & 15))) {
        set_error("c64_bf16: x_copy needs a dense 64-channel input, W %% 4 == 0 and a 16-byte aligned buffer");
        return SRMI_EUNSUPPORTED;
    }
    if (p.dbg & 1024) p.epi.H = 0;      // timing experiment: epilogue arithmetic only (every pixel fails the bounds check: no loads, no stores)
    if (d.N == 0 || d.H == 0 || d.W == 0) return SRMI_OK;
    // tile choice (pick_tile3x3): 6 x 96 (9 pixel tiles per wave), 8 x 64 (8 per wave) or 4 x 32 (2 per wave),
    // whichever wastes the fewest workgroup-rounds on ragged edges and a partly filled last round.
    const int tile = pick_tile3x3(d.N, d.H, d.W);
    static const int th[3] = {6, 8, 4}, tw[3] = {96, 64, 32};
    p.tiles_x = (d.W + tw[tile] - 1) / tw[tile]; p.tiles_y = (d.H + th[tile] - 1) / th[tile];
    const long long nb = (long long)d.N * p.tiles_x * p.tiles_y;
    if (nb > 0x7fffffffLL) { set_error("c64_bf16: grid too large"); return SRMI_EINVAL; }
    const bool w4 = (p.dbg & 256) != 0;    // ablation: the 4-wave variants
    // Two workgroups per CU (6 x 48 tiles) once the launch has several rounds of them: they drift out of phase and
    // hide each other's staging, epilogue and store waits.  (One round runs in lockstep: it gains nothing with several
    // channel chunks - 64->256 at 96x96: 47.5 -> 48.8 us - and 1 us with one - 64->64 at 96x96: 22.5 -> 21.5 us.)
    const long long nb2 = (long long)d.N * ((d.W + 47) / 48) * ((d.H + 5) / 6);
    const bool two = tile == 0 && p.nchunk <= 4 && !w4 && ((p.dbg & (1 << 21)) ? false : ((p.dbg & (1 << 22)) ? true : nb2 >= (p.nchunk == 1 ? 512 : 1024)));
    if (two) {
        p.tiles_x = (d.W + 47) / 48; p.tiles_y = (d.H + 5) / 6;
        hipLaunchKernelGGL((conv_c64_kernel<6, 3, 4, 3, 2, 2, 4>), dim3((unsigned)nb2), dim3(256), 0, s, p);
    } else if (tile == 0 && w4) hipLaunchKernelGGL((conv_c64_kernel<6, 6, 4>), dim3((unsigned)nb), dim3(256), 0, s, p);
    else if (tile == 0) hipLaunchKernelGGL((conv_c64_kernel<6, 6, 8>), dim3((unsigned)nb), dim3(512), 0, s, p);
    else if (tile == 1 && w4) hipLaunchKernelGGL((conv_c64_kernel<8, 4, 4>), dim3((unsigned)nb), dim3(256), 0, s, p);
    else if (tile == 1) hipLaunchKernelGGL((conv_c64_kernel<8, 4, 8>), dim3((unsigned)nb), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((conv_c64_kernel<4, 2, 4>), dim3((unsigned)nb), dim3(256), 0, s, p);
    return hip_check(hipGetLastError(), "conv_c64_kernel");
}

}  // namespace srmi
