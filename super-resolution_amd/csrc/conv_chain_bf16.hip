// A CHAIN of fused ResBlocks (P/ops.py:37-48, the 16-block trunk of Generator.forward, P/srgan_model.py:32-33) in ONE
// persistent launch: every workgroup keeps its 12 x 48 tile of the activation in LDS from block to block and exchanges
// only the two-pixel halo ring with its (up to eight) neighbouring tiles through global memory.
//
// Why: as sixteen launches of resblock_c64_kernel each block pays ~4 us of launch boundary (store drain, L2 write-back,
// dispatch) and ~3 us of staging a tile that the same CU has just written - a quarter of its 24 us.  Here a block ends
// with   write-through (sc1) stores of the tile -> every wave drains -> barrier -> ONE flag store per workgroup
// and the next one starts with   wave 0 polls the neighbours' flags -> barrier -> sc1 loads of 256 halo pixels,
// the hand-off form of MI355X_MICROARCH.md "inter-workgroup visibility" (Valid forms, table row 1: one lane signals for
// the whole workgroup behind a barrier, all payload stores and EVERY load of activation bytes in this kernel are sc1, so
// no L1 line can go stale and no fence is needed).  Results do not depend on placement or timing; the arithmetic is
// resblock_c64_kernel<12, 3, 8, 3, 1, 2>'s, instruction for instruction (bit-identical outputs, tests/test_chain_gpu.py).
//
// Residency: the grid must be co-resident (one workgroup per CU: 157 KB of LDS) - chain_launch checks it against the
// device's CU count and serialises chain launches of different streams.  Every spin is bounded: a workgroup that gives
// up (another process holds CUs, a lost neighbour) writes a sticky status word in host-visible memory, stops waiting and
// runs to the end, so the launch always terminates; the next srmi call on that device fails with SRMI_ETIMEOUT.
#include <algorithm>
#include <mutex>
#include <type_traits>

#include "c64_common.h"

namespace srmi {

constexpr int CHAIN_MAX_BLOCKS = 24;
constexpr int CHAIN_FLAG_WORDS = 1024;                  // 4 KiB per device: tile flags + the two words below
constexpr int CHAIN_EPOCH = CHAIN_FLAG_WORDS - 2, CHAIN_DONE = CHAIN_FLAG_WORDS - 1;
constexpr int CHAIN_MAX_TILES = CHAIN_FLAG_WORDS - 64;
constexpr unsigned CHAIN_SPIN_LIMIT = 1u << 19;   // polls of ~1-2 us (load round trip + s_sleep) before a workgroup gives up: ~1 s

struct ChainBlk {      // what changes from block to block
    const bf16_t* w1; const bf16_t* w2;
    const float* scale1; const float* shift1; const float* alpha1;
    const float* scale2; const float* shift2; const float* alpha2;
    bf16_t* y;
    int act1; float act_param1; int alpha_pc1;
    int act2; float act_param2; int alpha_pc2;
    int halo_only;      // y is read by the next block of this launch only: store just the two-pixel border of every tile
    int single;         // 1: ONE conv (w2, scale2, ...) on the block's input + a residual from global memory (a tail link);
                        // 2: ONE conv to 64 * nchunk channels with any c64 store mode and plain stores: the chain's LAST link
    const bf16_t* res;  // single == 1: the residual tensor (never written in this launch)
    int nchunk, mode;   // single == 2: output chunks of 64 channels, C64_PLAIN / C64_CRD / C64_DCR
    float* xcopy;       // single == 2: NULL, or the link's input as [N, 64, H, W] f32 (srmi_conv_args.x_copy)
};

struct ChainHead {     // optional first link: the 9x9 conv from <= 3 image channels (conv01 of Generator.forward, P/srgan_model.py:29)
    const void* x; const bf16_t* w;        // the caller's image tensor; weights as k9c3_pack lays them out (72 KiB)
    const float* scale; const float* shift; const float* alpha;
    bf16_t* y;                             // its output (block 0's input), written in full
    int on, Cin, in_cs, in_nchw, in_bf16;
    int act; float act_param; int alpha_pc;
};

struct ChainParams {
    ChainHead head;
    const bf16_t* x;       // input of block 0 (no head link)
    unsigned* flags;       // [tiles] links published by each tile, counted from 64 x (launch epoch): never reset.  flags[CHAIN_EPOCH]
                           // = the epoch, flags[CHAIN_DONE] = workgroups of the running launch that have finished
    unsigned* status;      // sticky give-up word (host-mapped)
    int N, H, W, tiles_x, tiles_y, nblk;
    ChainBlk blk[CHAIN_MAX_BLOCKS];
};

typedef unsigned int v4u_t __attribute__((__vector_size__(16)));
constexpr int CH_STEP_BYTES = 4096;   // one K-step: 4 channel tiles x 64 lanes x 16 B
constexpr int SC1 = 16;               // aux bits of the raw buffer builtins: write-through store / L1-bypassing load

__global__ __launch_bounds__(512, 1) void resblock_chain_kernel(ChainParams p) {
    constexpr int TH = 12, TWT = 3, NW = 8, SLOTS = 3, KS = 2;
    constexpr int NTH = NW * 64, TW = TWT * 16;
    constexpr int XR = TH + 4, XC = TW + 4;          // x halo tile
    constexpr int TR = TH + 2, TC = TW + 2;          // T region
    constexpr int ROWP = 64;                         // LDS row pitch in pixels
    constexpr int IN_BYTES = XR * ROWP * 128;
    constexpr int NT1 = (TR * TC + 15) / 16, NPT2 = TH * TWT;
    constexpr int PT1 = 6, PT1B = 5, PT2 = 5, PT2B = 4;     // tiles per wave: waves 0-3 / waves 4-7 (SIMD partners)
    static_assert(NT1 == (NW / 2) * (PT1 + PT1B) && NPT2 == (NW / 2) * (PT2 + PT2B), "balanced split fills every slot");
    constexpr int SLOT_BYTES = KS * CH_STEP_BYTES, NS = 36 / KS;
    constexpr int MAXCH = 4;                         // output chunks of the last link
    __shared__ __attribute__((aligned(16))) unsigned char lds[IN_BYTES + SLOTS * SLOT_BYTES + (1 + MAXCH) * 768];
    unsigned char* const wring = lds + IN_BYTES;
    float* const par = (float*)(lds + IN_BYTES + SLOTS * SLOT_BYTES);

    const int tid0 = threadIdx.x;
    const int lane0 = tid0 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);

    // XCD-aware tile order (as resblock_c64_kernel): a contiguous run of tiles per XCD, so most halo traffic stays in one L2
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const int tile_id = b;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    // wave 0, lanes 0..7: the flag word of one neighbouring tile each (-1: no such tile)
    int nb_idx = -1;
    if (lane0 < 8) {
        const int j = lane0 < 4 ? lane0 : lane0 + 1;
        const int nty = ty + j / 3 - 1, ntx = tx + j % 3 - 1;
        if (nty >= 0 && nty < p.tiles_y && ntx >= 0 && ntx < p.tiles_x) nb_idx = (n * p.tiles_y + nty) * p.tiles_x + ntx;
    }
    bool dead = false;      // this workgroup gave up waiting once: it waits no more (wave-uniform, wave 0 only)
    // Flag values count from 64 x epoch, and the epoch lives in device memory: the workgroup that finishes last raises it
    // for the next launch (launches on one device are ordered, chain_launch sees to it), so no word has to be zeroed per
    // launch and nothing depends on a host-side launch count (a captured graph replays correctly).
    const unsigned fepoch = __hip_atomic_load(p.flags + CHAIN_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 6;

    const unsigned img_bytes = (unsigned)p.H * p.W * 128u;
    const unsigned row_bytes = (unsigned)p.W * 128u;
    constexpr unsigned OOB = 0x80000000u;        // image bytes <= 2^30 (checked on the host)

    // ---- weight stream: slot S of 18 = one 3x3 tap (two K-steps, 8 KiB), conv1 taps 0..8 then conv2 taps; wave w
    // copies KiB w of each slot by LDS-DMA
    const unsigned char* wb1 = nullptr;
    const unsigned char* wb2 = nullptr;
    const unsigned wring_lds = (unsigned)(uintptr_t)(lptr_t)wring;
    auto issue_slot = [&](int S) {
        const int g0 = S * KS;
        const unsigned char* base = g0 < 18 ? wb1 : wb2;
        const unsigned char* src = base + (size_t)(g0 < 18 ? g0 : g0 - 18) * CH_STEP_BYTES + wave * 1024 + lane0 * 16;
        const unsigned dst = __builtin_amdgcn_readfirstlane(wring_lds + (unsigned)((S % SLOTS) * SLOT_BYTES + wave * 1024));
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, off\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(src), "s"(dst)
            : "memory");
    };
    // hand-off at the top of K-step g when step g+1 opens slot S: slot S landed (own KiB: vmcnt, the others': barrier);
    // slot S-2's fragments were consumed by MFMAs every wave has issued, so slot S+1 is issued into its ring position
    // (ns = slots in the block's weight stream: NS, or 9 + 9 per output chunk in the last link)
    auto handoff = [&](int g, int ns) {
        if ((g + 1) % KS != 0) return;
        const int S = (g + 1) / KS;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (S + 1 < ns) issue_slot(S + 1);
    };
    auto frag = [&](int g) { return wring + ((g / KS) % SLOTS) * SLOT_BYTES + (g % KS) * CH_STEP_BYTES + lane0 * 16; };

    const int fbase = p.head.on ? 1 : 0;     // flag value = links published: the head link counts as one
    if (p.head.on) {
        // ================= head link: conv 9x9 from <= 3 channels on this workgroup's 12 x 48 pixels (conv_k9c3_kernel's
        // arithmetic: K = 9 tap rows x 16 tap columns x 4 channels, zero-weight padding), its halo comes from the
        // neighbours like every later block's.  LDS during this phase: [72 KiB weights | 20 x 64 x 4 bf16 input tile],
        // all inside the (still empty) x tile region ==========================================================
        const ChainHead& Hd = p.head;
        int tid;
        asm volatile("v_mov_b32 %0, %1" : "=v"(tid) : "v"(tid0));
        const int lane = tid & 63;
        const int pl = lane & 15, q = lane >> 4;
        constexpr int K9W = 72 * 1024, ROWP9 = TW + 16, ROWS9 = TH + 8;
        static_assert(K9W + ROWS9 * ROWP9 * 8 <= IN_BYTES, "head link's weights and input tile fit the x tile region");
        unsigned char* const wl = lds;
        unsigned char* const tile = lds + K9W;
        {
            const unsigned char* wsrc = (const unsigned char*)Hd.w + wave * 9 * 1024 + lane * 16;
            const unsigned dst0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)wl + (unsigned)(wave * 9 * 1024));
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                unsigned keep;
                asm volatile(
                    "s_mov_b32 %0, m0\n\t"
                    "s_mov_b32 m0, %2\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %1, off\n\t"
                    "s_mov_b32 m0, %0"
                    : "=&s"(keep)
                    : "v"(wsrc + i * 1024), "s"(dst0 + (unsigned)(i * 1024))
                    : "memory");
            }
        }
        C64Epi eh;
        eh.scale = Hd.scale; eh.shift = Hd.shift; eh.alpha = Hd.alpha; eh.res = nullptr; eh.y = nullptr;
        eh.H = p.H; eh.W = p.W; eh.Cout = 64; eh.out_cs = 64; eh.out_co = 0; eh.mode = C64_PLAIN;
        eh.act = Hd.act; eh.act_param = Hd.act_param; eh.alpha_pc = Hd.alpha_pc;
        C64ParRegs pregs;
        c64_params_load(pregs, eh, 1, tid);
        {
            constexpr int NPIX = ROWS9 * ROWP9;
            constexpr int NIT = (NPIX + NTH - 1) / NTH;
            float v[NIT][3];
            const size_t HW = (size_t)p.H * p.W;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = min(tid + it * NTH, NPIX - 1);
                const int row = idx / ROWP9, col = idx - row * ROWP9;
                const int gy = min(max(y0 - 4 + row, 0), p.H - 1), gx = min(max(x0 - 4 + col, 0), p.W - 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int cc = c < Hd.Cin ? c : 0;
                    const size_t gi = Hd.in_nchw ? ((size_t)n * Hd.Cin + cc) * HW + (size_t)gy * p.W + gx
                                                 : (((size_t)n * p.H + gy) * p.W + gx) * Hd.in_cs + cc;
                    v[it][c] = Hd.in_bf16 ? bf2f(((const bf16_t*)Hd.x)[gi]) : ((const float*)Hd.x)[gi];
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * NTH;
                const int row = idx / ROWP9, col = idx - row * ROWP9;
                const int gy = y0 - 4 + row, gx = x0 - 4 + col;
                const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                u32x2 o;
                o[0] = ok ? pack_bf2(v[it][0], Hd.Cin > 1 ? v[it][1] : 0.f) : 0u;
                o[1] = ok ? pack_bf2(Hd.Cin > 2 ? v[it][2] : 0.f, 0.f) : 0u;
                if (idx < NPIX) *(u32x2*)(tile + idx * 8) = o;
            }
        }
        c64_params_store(par, pregs, 1, tid);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const __amdgpu_buffer_rsrc_t yres =
            __builtin_amdgcn_make_buffer_rsrc((void*)(Hd.y + (size_t)n * p.H * p.W * 64), 0, img_bytes, 0x00020000);
        auto head = [&](auto P2, const int t2base) {
            constexpr int PTc = decltype(P2)::value;
            int prow[PTc], pcol[PTc], boff[PTc];
#pragma unroll
            for (int k = 0; k < PTc; ++k) {
                const int t = t2base + k;
                prow[k] = t / TWT;
                pcol[k] = (t - prow[k] * TWT) * 16;
                boff[k] = (prow[k] * ROWP9 + pcol[k] + pl + 2 * q) * 8;
            }
            f32x4 acc[PTc][4];
#pragma unroll
            for (int k = 0; k < PTc; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[k][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            typedef union { u32x2 h[2]; bf16x8 v; } frag_t;
            bf16x8 afr[2][4];
            frag_t bfr[2][PTc];
            const unsigned char* wlane = wl + lane * 16;
#pragma unroll
            for (int t = 0; t < 4; ++t) afr[0][t] = *(const bf16x8*)(wlane + t * 1024);
#pragma unroll
            for (int k = 0; k < PTc; ++k) {
                bfr[0][k].h[0] = *(const u32x2*)(tile + boff[k]);
                bfr[0][k].h[1] = *(const u32x2*)(tile + boff[k] + 8);
            }
#pragma unroll
            for (int j = 0; j < 18; ++j) {     // K step j = (tap row ky = j >> 1, tap-column half h = j & 1)
                if (j + 1 < 18) {
                    const int kyn = (j + 1) >> 1, hn = (j + 1) & 1;
#pragma unroll
                    for (int t = 0; t < 4; ++t) afr[(j + 1) & 1][t] = *(const bf16x8*)(wlane + ((j + 1) * 4 + t) * 1024);
#pragma unroll
                    for (int k = 0; k < PTc; ++k) {
                        const unsigned char* a = tile + boff[k] + (kyn * ROWP9 + 8 * hn) * 8;
                        bfr[(j + 1) & 1][k].h[0] = *(const u32x2*)a;
                        bfr[(j + 1) & 1][k].h[1] = *(const u32x2*)(a + 8);
                    }
                }
#pragma unroll
                for (int k = 0; k < PTc; ++k)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[j & 1][t], bfr[j & 1][k].v, acc[k][t], 0, 0, 0);
                if (j + 1 < 18) sched_interleave<0, 4 + 2 * PTc, 4 * PTc>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // every wave has finished reading the weights and the image tile
            f32x2 sc[8], sh[8], sl[8];
#pragma unroll
            for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                for (int v4 = 0; v4 < 2; ++v4) {
                    const int c = 32 * hi + 8 * q + 4 * v4;
                    const f32x4 a = *(const f32x4*)(par + c);
                    const f32x4 bsh = *(const f32x4*)(par + 64 + c);
                    const f32x4 csl = *(const f32x4*)(par + 128 + c);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        sc[4 * hi + 2 * v4 + e] = (f32x2){a[2 * e], a[2 * e + 1]};
                        sh[4 * hi + 2 * v4 + e] = (f32x2){bsh[2 * e], bsh[2 * e + 1]};
                        sl[4 * hi + 2 * v4 + e] = (f32x2){csl[2 * e], csl[2 * e + 1]};
                    }
                }
            auto epi = [&](auto KIND) {
#pragma unroll
                for (int k = 0; k < PTc; ++k) {
                    u32x4 o0, o1;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int j = 2 * t + e;
                            const f32x2 a = {acc[k][t][2 * e], acc[k][t][2 * e + 1]};
                            f32x2 v;
                            if constexpr (decltype(KIND)::value == 2) v = slope_act2(a, sc[j], sh[j], sl[j]);
                            else if constexpr (decltype(KIND)::value == 1) v = slope_act2_max(a, sc[j], sh[j], sl[j]);
                            else v = __builtin_elementwise_fma(a, sc[j], sh[j]);
                            const unsigned o = pack_bf2(v[0], v[1]);
                            if (t < 2) o0[j & 3] = o; else o1[j & 3] = o;
                        }
                    const int oy = y0 + prow[k], ox = x0 + pcol[k] + pl;
                    const bool ok = oy < p.H && ox < p.W;
                    if (!ok) { o0 = (u32x4){0u, 0u, 0u, 0u}; o1 = o0; }
                    const int P = (prow[k] + 2) * ROWP + pcol[k] + pl + 2;
                    const int ra = P * 128 + ((q ^ c64_key(P)) << 4);
                    *(u32x4*)(lds + ra) = o0;
                    *(u32x4*)(lds + (ra ^ 64)) = o1;
                    const unsigned off = ok ? (unsigned)oy * row_bytes + (unsigned)ox * 128u + 16u * q : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, o0), yres, off, 0, SC1);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, o1), yres, off + 64, 0, SC1);
                }
            };
            const int kind = c64_slope_kind(eh);
            if (kind == 0) epi(std::integral_constant<int, 0>{});
            else if (kind == 1) epi(std::integral_constant<int, 1>{});
            else epi(std::integral_constant<int, 2>{});
        };
        if (wave < NW / 2) head(std::integral_constant<int, PT2>{}, wave * PT2);
        else head(std::integral_constant<int, PT2B>{}, (NW / 2) * PT2 + (wave - NW / 2) * PT2B);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tid0 == 0) __hip_atomic_store(p.flags + tile_id, fepoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // the per-block table is read from the kernel-argument segment by scalar loads (indexing the by-value struct with the
    // loop counter would make hipcc copy it to scratch)
    typedef __attribute__((address_space(4))) const ChainParams* kparams_t;
    const kparams_t kp = (kparams_t)__builtin_amdgcn_kernarg_segment_ptr();
    for (int kb = 0; kb < p.nblk; ++kb) {
        const __attribute__((address_space(4))) ChainBlk& B = kp->blk[kb];
        // an opaque copy of the thread index per block: without it hipcc hoists every block-invariant address (operand
        // offsets, staging offsets, masks: ~130 registers) out of this loop and parks them in scratch
        int tid;
        asm volatile("v_mov_b32 %0, %1" : "=v"(tid) : "v"(tid0));
        const int lane = tid & 63;
        const int pl = lane & 15, q = lane >> 4;
        wb1 = (const unsigned char*)B.w1;
        wb2 = (const unsigned char*)B.w2;
        asm volatile("" : "+s"(wb1), "+s"(wb2));
        const int linkkind = B.single;
        const bool single = linkkind != 0;
        const int nchunk = linkkind == 2 ? B.nchunk : 1;
        if (single) { issue_slot(9); issue_slot(10); }      // (every wave is past the previous block's K loops: barrier A below)
        else { issue_slot(0); issue_slot(1); }
        C64Epi e1, e2;
        e1.scale = B.scale1; e1.shift = B.shift1; e1.alpha = B.alpha1; e1.res = nullptr; e1.y = nullptr;
        e1.H = p.H; e1.W = p.W; e1.Cout = 64; e1.out_cs = 64; e1.out_co = 0; e1.mode = C64_PLAIN;
        e1.act = B.act1; e1.act_param = B.act_param1; e1.alpha_pc = B.alpha_pc1;
        e2 = e1;
        e2.scale = B.scale2; e2.shift = B.shift2; e2.alpha = B.alpha2;
        e2.act = B.act2; e2.act_param = B.act_param2; e2.alpha_pc = B.alpha_pc2;
        if (linkkind == 2) { e2.mode = B.mode; e2.Cout = 64 * nchunk; e2.y = B.y; e2.out_cs = B.mode == C64_PLAIN ? 64 * nchunk : 64; }
        C64ParRegs pregs1, pregs2;
        c64_params_load(pregs1, e1, 1, tid);
        c64_params_load(pregs2, e2, nchunk, tid);
        const bf16_t* src = kb == 0 ? (fbase ? (const bf16_t*)p.head.y : p.x) : (const bf16_t*)kp->blk[kb - 1].y;
        const bool halo_only = B.halo_only != 0;
        const __amdgpu_buffer_rsrc_t xres =
            __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)n * p.H * p.W * 64), 0, img_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t yres =
            __builtin_amdgcn_make_buffer_rsrc((void*)(B.y + (size_t)n * p.H * p.W * 64), 0, img_bytes, 0x00020000);
        if (kb + fbase == 0) {
            // ---- block 0: stage the whole x halo tile; pixels outside the image read as zeros through the descriptor ----
            constexpr int NB = (XR * XC * 8 + NTH - 1) / NTH;
            u32x4 vb[NB];
            unsigned lb[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int sidx = tid + i * NTH;
                const int row = sidx / (XC * 8), rem = sidx - row * (XC * 8);
                const int col = rem >> 3, chb = rem & 7;
                const int gy = y0 - 2 + row, gx = x0 - 2 + col;
                const bool ok = sidx < XR * XC * 8 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                const unsigned off = ok ? (unsigned)gy * row_bytes + (unsigned)gx * 128u + chb * 16 : OOB;
                vb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, off, 0, SC1));
                lb[i] = (row * ROWP + col) * 128 + ((chb ^ c64_key(col)) << 4);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i)
                if (tid + i * NTH < XR * XC * 8) *(u32x4*)(lds + lb[i]) = vb[i];
        } else {
            // ---- later blocks: the centre of the tile is in LDS already (this workgroup's own output); wait for the
            // neighbours to have published block kb-1, then fetch the 256 pixels of the halo ring ----
            if (wave == 0 && !dead) {
                const unsigned* f = p.flags + (nb_idx >= 0 ? nb_idx : tile_id);
                unsigned spins = 0;
                for (;;) {
                    unsigned v = fepoch + (unsigned)(kb + fbase);
                    if (nb_idx >= 0) v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // ready = published in THIS epoch, at least kb + fbase links (a word left over from another epoch - a tile
                    // that earlier, smaller grids did not use - never matches)
                    if (__all((int)((v ^ fepoch) < 64u && (v & 63u) >= (unsigned)(kb + fbase)))) break;
                    if (++spins > CHAIN_SPIN_LIMIT) {          // wave-uniform
                        dead = true;
                        if (lane == 0) __hip_atomic_store(p.status, 0x100u + (unsigned)kb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // (no instruction: keeps the loads below the poll)
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();                // B: the other waves load only after wave 0's poll has matched
            asm volatile("" ::: "memory");
            constexpr int NHP = 2 * 2 * XC + TH * 4;     // 256 halo pixels: rows 0, 1, XR-2, XR-1 whole; columns 0, 1, XC-2, XC-1 of the rest
            constexpr int NH = (NHP * 8 + NTH - 1) / NTH;
            u32x4 vh[NH];
            unsigned lh[NH];
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                const int sidx = tid + i * NTH;
                const int hp = sidx >> 3, chb = sidx & 7;
                int row, col;
                if (hp < 2 * XC) { row = hp / XC; col = hp - row * XC; }
                else if (hp < 4 * XC) { const int h2 = hp - 2 * XC; row = h2 / XC; col = h2 - row * XC; row += XR - 2; }
                else { const int s4 = hp - 4 * XC; row = 2 + (s4 >> 2); const int c4 = s4 & 3; col = c4 < 2 ? c4 : XC - 4 + c4; }
                const int gy = y0 - 2 + row, gx = x0 - 2 + col;
                const bool ok = hp < NHP && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                const unsigned off = ok ? (unsigned)gy * row_bytes + (unsigned)gx * 128u + chb * 16 : OOB;
                vh[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, off, 0, SC1));
                lh[i] = (row * ROWP + col) * 128 + ((chb ^ c64_key(col)) << 4);
            }
#pragma unroll
            for (int i = 0; i < NH; ++i)
                if (((tid + i * NTH) >> 3) < NHP) *(u32x4*)(lds + lh[i]) = vh[i];
        }
        c64_params_store(par, pregs1, 1, tid);          // (the previous block's readers are past barrier A)
        c64_params_store(par + 192, pregs2, nchunk, tid);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // also retires weight slots 0, 1
        __builtin_amdgcn_s_barrier();                    // C: tile, parameters and the first two taps are in LDS

        if (linkkind == 2) {
            // ---- the LAST link: one conv 64 -> 64 * nchunk on the tile (the first Upsampler conv of Generator.forward, P/ops.py
            // :74-78, with PixelShuffle as store addressing), conv_c64_kernel's chunk loop on the LDS-resident tile; plain stores:
            // nobody reads them in this launch ----
            if (B.xcopy) {
                // NCHW f32 copy of the link's INPUT from the tile's centre (srmi_conv_args.x_copy; see conv_c64_kernel): a unit =
                // (4 consecutive pixels of a tile row, 8 channels), four 16-byte LDS reads and eight 16-byte stores
                constexpr int QPR = TW / 4;
                constexpr int UNITS = 8 * TH * QPR;
                const size_t plane = (size_t)p.H * p.W;
                float* const dstn = B.xcopy + (size_t)n * 64 * plane;
                for (int u = tid; u < UNITS; u += NTH) {
                    const int oct = u & 7, t = u >> 3;
                    const int r = t / QPR, qd = t - r * QPR;
                    const int gy = y0 + r, gx = x0 + 4 * qd;
                    if (gy < p.H && gx < p.W) {                    // W % 4 == 0 (checked on the host): whole quads
                        u32x4 v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int P = (r + 2) * ROWP + 2 + 4 * qd + j;
                            v[j] = *(const u32x4*)(lds + P * 128 + ((oct ^ c64_key(P)) << 4));
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
            auto wide = [&](auto P2, const int t2base) {
                constexpr int PT2c = decltype(P2)::value;
                int prow[PT2c], pcol[PT2c];
                int boff2[PT2c][3];
#pragma unroll
                for (int k = 0; k < PT2c; ++k) {
                    const int t = t2base + k;
                    prow[k] = t / TWT;
                    pcol[k] = (t - prow[k] * TWT) * 16;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int P = (prow[k] + 1) * ROWP + pcol[k] + pl + dx + 1;      // (x lies one pixel row and column inside the T window)
                        boff2[k][dx] = P * 128 + ((q ^ c64_key(P)) << 4);
                    }
                }
                bf16x8 afr[2][4], bfr2[2][PT2c];
#pragma unroll
                for (int t = 0; t < 4; ++t) afr[0][t] = *(const bf16x8*)(frag(18) + t * 1024);
#pragma unroll
                for (int k = 0; k < PT2c; ++k) bfr2[0][k] = *(const bf16x8*)(lds + boff2[k][0]);
#pragma unroll 1
                for (int c = 0; c < nchunk; ++c) {
                    f32x4 acc2[PT2c][4];
#pragma unroll
                    for (int k = 0; k < PT2c; ++k)
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc2[k][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    const int gb = 18 + 18 * c;
                    const bool last = c + 1 == nchunk;
#pragma unroll
                    for (int j = 0; j < 18; ++j) {
                        const int g = gb + j;
                        if (j + 1 < 18 || !last) {      // (step 17 of a chunk prefetches step 0 of the next one: same pixels, tap 0)
                            if ((j + 1) % KS == 0) handoff(g, 9 + 9 * nchunk);
                            const unsigned char* wslot = frag(g + 1);
#pragma unroll
                            for (int t = 0; t < 4; ++t) afr[(j + 1) & 1][t] = *(const bf16x8*)(wslot + t * 1024);
                            const int jn = (j + 1) % 18;
                            const int tapn = jn >> 1, halfn = jn & 1;
                            const int dyn = tapn / 3, dxn = tapn - 3 * dyn;
#pragma unroll
                            for (int k = 0; k < PT2c; ++k)
                                bfr2[(j + 1) & 1][k] =
                                    *(const bf16x8*)(lds + ((boff2[k][dxn] ^ (halfn << 6)) + dyn * (ROWP * 128)));
                        }
#pragma unroll
                        for (int k = 0; k < PT2c; ++k)
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                acc2[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[j & 1][t], bfr2[j & 1][k], acc2[k][t], 0, 0, 0);
                        if (j + 1 < 18) sched_interleave<0, 4 + PT2c, 4 * PT2c>();
                    }
                    c64_epilogue<PT2c>(acc2, par + 192 * (1 + c), e2, c, n, y0, x0, prow, pcol, pl, q);
                }
            };
            if (wave < NW / 2) wide(std::integral_constant<int, PT2>{}, wave * PT2);
            else wide(std::integral_constant<int, PT2B>{}, (NW / 2) * PT2 + (wave - NW / 2) * PT2B);
            break;          // (the last link: nothing to publish)
        }

        auto body = [&](auto P1, auto P2, const int t1base, const int t2base) {
            constexpr int PT1c = decltype(P1)::value, PT2c = decltype(P2)::value;
            bf16x8 afr[2][4];
            u32x4 rpre[PT2c][2];     // the residual of this wave's output tiles
            int ra[PT2c];            // LDS byte address of (this lane's pixel of output tile k, chunk q): the output goes back there
            if (!single) {
            // ================= phase 1: conv1 over the flattened T region =================================
            int boff1[PT1c][3];
            unsigned inside = 0;     // bit k: this lane's pixel of tile k lies inside the image (else T = 0: conv2's zero padding)
#pragma unroll
            for (int k = 0; k < PT1c; ++k) {
                const int pix = min((t1base + k) * 16 + pl, TR * TC - 1);   // the last tile's spare lanes recompute the last pixel
                const int ti = pix / TC, tj = pix - ti * TC;
                const int gy = y0 - 1 + ti, gx = x0 - 1 + tj;
                inside |= (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? 1u << k : 0u;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int P = ti * ROWP + tj + dx;
                    boff1[k][dx] = P * 128 + ((q ^ c64_key(P)) << 4);
                }
            }
            asm volatile("" : "+v"(inside));
            f32x4 acc[PT1c][4];
#pragma unroll
            for (int k = 0; k < PT1c; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[k][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            bf16x8 bfr[2][PT1c];
#pragma unroll
            for (int t = 0; t < 4; ++t) afr[0][t] = *(const bf16x8*)(wring + lane * 16 + t * 1024);
#pragma unroll
            for (int k = 0; k < PT1c; ++k) bfr[0][k] = *(const bf16x8*)(lds + boff1[k][0]);
#pragma unroll
            for (int j = 0; j < 18; ++j) {
                handoff(j, NS);
                {   // prefetch the fragments of global step j+1 (step 18 = first step of conv2: weights only)
                    const unsigned char* wslot = frag(j + 1);
#pragma unroll
                    for (int t = 0; t < 4; ++t) afr[(j + 1) & 1][t] = *(const bf16x8*)(wslot + t * 1024);
                    if (j + 1 < 18) {
                        const int tapn = (j + 1) >> 1, halfn = (j + 1) & 1;
                        const int dyn = tapn / 3, dxn = tapn - 3 * dyn;
#pragma unroll
                        for (int k = 0; k < PT1c; ++k)
                            bfr[(j + 1) & 1][k] =
                                *(const bf16x8*)(lds + ((boff1[k][dxn] ^ (halfn << 6)) + dyn * (ROWP * 128)));
                    }
                }
#pragma unroll
                for (int k = 0; k < PT1c; ++k)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[j & 1][t], bfr[j & 1][k], acc[k][t], 0, 0, 0);
                if (j + 1 < 18) sched_interleave<0, 4 + PT1c, 4 * PT1c>();
            }

            // ---- phase 2: the residual (= centre of the x tile) into registers, then T = act(affine1(acc)) as bf16 over
            // the x tile (zero outside the image) ----
            {
                int opr;      // (opaque zero in the address: keeps these loads where they are written)
                asm volatile("v_mov_b32 %0, 0" : "=v"(opr));
#pragma unroll
                for (int k = 0; k < PT2c; ++k) {
                    const int t = t2base + k;
                    const int col = (t % TWT) * 16 + pl + 2 + opr;
                    const int P = (t / TWT + 2) * ROWP + col;
                    ra[k] = P * 128 + ((q ^ c64_key(P)) << 4);
                    rpre[k][0] = *(const u32x4*)(lds + ra[k]);
                    rpre[k][1] = *(const u32x4*)(lds + (ra[k] ^ 64));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // every wave has finished reading the x tile
            auto twrite = [&](auto KIND) {
#pragma unroll
                for (int hi = 0; hi < 2; ++hi) {
                    f32x2 sc[4], sh[4], sl[4];
#pragma unroll
                    for (int v4 = 0; v4 < 2; ++v4) {
                        const int c = 32 * hi + 8 * q + 4 * v4;
                        const f32x4 a = *(const f32x4*)(par + c);
                        const f32x4 bsh = *(const f32x4*)(par + 64 + c);
                        const f32x4 csl = *(const f32x4*)(par + 128 + c);
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            sc[2 * v4 + e] = (f32x2){a[2 * e], a[2 * e + 1]};
                            sh[2 * v4 + e] = (f32x2){bsh[2 * e], bsh[2 * e + 1]};
                            sl[2 * v4 + e] = (f32x2){csl[2 * e], csl[2 * e + 1]};
                        }
                    }
#pragma unroll
                    for (int k = 0; k < PT1c; ++k) {
                        const unsigned keep = (unsigned)((int)(inside << (31 - k)) >> 31);
                        u32x4 o;
#pragma unroll
                        for (int v4 = 0; v4 < 2; ++v4)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const int t = 2 * hi + v4;
                                const f32x2 a = {acc[k][t][2 * e], acc[k][t][2 * e + 1]};
                                const f32x2 v = decltype(KIND)::value == 1 ? slope_act2_max(a, sc[2 * v4 + e], sh[2 * v4 + e], sl[2 * v4 + e])
                                                                           : slope_act2(a, sc[2 * v4 + e], sh[2 * v4 + e], sl[2 * v4 + e]);
                                o[2 * v4 + e] = pack_bf2(v[0], v[1]) & keep;
                            }
                        *(u32x4*)(lds + (boff1[k][0] ^ (hi << 6))) = o;
                    }
                }
            };
            if (c64_slope_kind(e1) <= 1) twrite(std::integral_constant<int, 1>{});
            else twrite(std::integral_constant<int, 2>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // T complete
            } else {
                // ---- the tail link: ONE conv on the x tile itself (conv02 of Generator.forward + the long skip add).  Its
                // weights stream as conv "2"'s; its operand window is the T window moved one pixel row and column into the x
                // tile; the residual is a tensor in global memory that nobody writes in this launch (plain loads) ----
#pragma unroll
                for (int t = 0; t < 4; ++t) afr[0][t] = *(const bf16x8*)(frag(18) + t * 1024);
                const __amdgpu_buffer_rsrc_t rres =
                    __builtin_amdgcn_make_buffer_rsrc((void*)(B.res + (size_t)n * p.H * p.W * 64), 0, img_bytes, 0x00020000);
                int opr;
                asm volatile("v_mov_b32 %0, 0" : "=v"(opr));
#pragma unroll
                for (int k = 0; k < PT2c; ++k) {
                    const int t = t2base + k;
                    const int trow = t / TWT, colp = (t - trow * TWT) * 16 + pl + opr;
                    const int P = (trow + 2) * ROWP + colp + 2;
                    ra[k] = P * 128 + ((q ^ c64_key(P)) << 4);
                    const int oy = y0 + trow, ox = x0 + colp;
                    const unsigned off = (oy < p.H && ox < p.W) ? (unsigned)oy * row_bytes + (unsigned)ox * 128u + 16u * q : OOB;   // (out of range: zeros)
                    rpre[k][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, off, 0, SC1));      // (sc1: this workgroup's
                    rpre[k][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, off + 64, 0, SC1)); //  head link may have stored it)
                }
            }
            // ================= phase 3: conv2 from T ==================================================
            int prow[PT2c], pcol[PT2c];
            int boff2[PT2c][3];
            int opaque0;       // keeps this phase's address arithmetic below conv1's K loop (which runs at the register limit)
            asm volatile("v_mov_b32 %0, 0" : "=v"(opaque0));
            const int shp = single ? ROWP + 1 : 0;      // (the tail link reads x, which lies one pixel row and column further in)
#pragma unroll
            for (int k = 0; k < PT2c; ++k) {
                const int t = t2base + k;
                prow[k] = t / TWT;
                pcol[k] = (t - prow[k] * TWT) * 16;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int col = pcol[k] + pl + opaque0 + dx;
                    const int P = prow[k] * ROWP + col + shp;
                    boff2[k][dx] = P * 128 + ((q ^ c64_key(P)) << 4);
                }
            }
            f32x4 acc2[PT2c][4];
#pragma unroll
            for (int k = 0; k < PT2c; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc2[k][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            bf16x8 bfr2[2][PT2c];
#pragma unroll
            for (int k = 0; k < PT2c; ++k) bfr2[0][k] = *(const bf16x8*)(lds + boff2[k][0]);
#pragma unroll
            for (int j = 0; j < 18; ++j) {
                const int g = 18 + j;
                if (g + 1 < 36) {
                    handoff(g, NS);
                    const unsigned char* wslot = frag(g + 1);
#pragma unroll
                    for (int t = 0; t < 4; ++t) afr[(j + 1) & 1][t] = *(const bf16x8*)(wslot + t * 1024);
                    const int tapn = (j + 1) >> 1, halfn = (j + 1) & 1;
                    const int dyn = tapn / 3, dxn = tapn - 3 * dyn;
#pragma unroll
                    for (int k = 0; k < PT2c; ++k)
                        bfr2[(j + 1) & 1][k] =
                            *(const bf16x8*)(lds + ((boff2[k][dxn] ^ (halfn << 6)) + dyn * (ROWP * 128)));
                }
#pragma unroll
                for (int k = 0; k < PT2c; ++k)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc2[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[j & 1][t], bfr2[j & 1][k], acc2[k][t], 0, 0, 0);
                if (j + 1 < 18) sched_interleave<0, 4 + PT2c, 4 * PT2c>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // D: every wave has finished reading T: the tile may take the outputs
            // ---- epilogue: y = affine2(acc2) (+ activation) + x  ->  bf16, back into the LDS tile (the next block's x) and
            // write-through to global memory (the neighbours' halo; the chain's result) ----
            {
                f32x2 sc[8], sh[8], sl[8];
#pragma unroll
                for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                    for (int v4 = 0; v4 < 2; ++v4) {
                        const int c = 32 * hi + 8 * q + 4 * v4;
                        const f32x4 a = *(const f32x4*)(par + 192 + c);
                        const f32x4 bsh = *(const f32x4*)(par + 192 + 64 + c);
                        const f32x4 csl = *(const f32x4*)(par + 192 + 128 + c);
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            sc[4 * hi + 2 * v4 + e] = (f32x2){a[2 * e], a[2 * e + 1]};
                            sh[4 * hi + 2 * v4 + e] = (f32x2){bsh[2 * e], bsh[2 * e + 1]};
                            sl[4 * hi + 2 * v4 + e] = (f32x2){csl[2 * e], csl[2 * e + 1]};
                        }
                    }
                int opq;
                asm volatile("v_mov_b32 %0, 0" : "=v"(opq));
                auto epi = [&](auto KIND) {
#pragma unroll
                    for (int k = 0; k < PT2c; ++k) {
                        u32x4 o0, o1;
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const int j = 2 * t + e;
                                const f32x2 a = {acc2[k][t][2 * e], acc2[k][t][2 * e + 1]};
                                f32x2 v;
                                if constexpr (decltype(KIND)::value == 2) v = slope_act2(a, sc[j], sh[j], sl[j]);
                                else if constexpr (decltype(KIND)::value == 1) v = slope_act2_max(a, sc[j], sh[j], sl[j]);
                                else v = __builtin_elementwise_fma(a, sc[j], sh[j]);
                                const unsigned w = rpre[k][j >> 2][j & 3];
                                v += (f32x2){__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
                                const unsigned o = pack_bf2(v[0], v[1]);
                                if (t < 2) o0[j & 3] = o; else o1[j & 3] = o;
                            }
                        const int oy = y0 + prow[k], ox = x0 + pcol[k] + pl + opq;
                        const bool ok = oy < p.H && ox < p.W;
                        if (!ok) { o0 = (u32x4){0u, 0u, 0u, 0u}; o1 = o0; }     // outside the image: the next conv's zero padding
                        *(u32x4*)(lds + ra[k]) = o0;
                        *(u32x4*)(lds + (ra[k] ^ 64)) = o1;
                        // (an intermediate block of a chain whose y nobody else reads publishes only what the neighbours' halo
                        //  rings take: the two outermost pixel rows and columns of the tile)
                        const int oxl = pcol[k] + pl + opq;
                        const bool wanted = !halo_only || prow[k] < 2 || prow[k] >= TH - 2 || oxl < 2 || oxl >= TW - 2;
                        const unsigned off = ok && wanted ? (unsigned)oy * row_bytes + (unsigned)ox * 128u + 16u * q : OOB;   // (out of range: dropped)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, o0), yres, off, 0, SC1);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, o1), yres, off + 64, 0, SC1);
                    }
                };
                const int kind = c64_slope_kind(e2);
                if (kind == 0) epi(std::integral_constant<int, 0>{});
                else if (kind == 1) epi(std::integral_constant<int, 1>{});
                else epi(std::integral_constant<int, 2>{});
            }
        };
        if (wave < NW / 2) body(std::integral_constant<int, PT1>{}, std::integral_constant<int, PT2>{}, wave * PT1, wave * PT2);
        else body(std::integral_constant<int, PT1B>{}, std::integral_constant<int, PT2B>{},
                  (NW / 2) * PT1 + (wave - NW / 2) * PT1B, (NW / 2) * PT2 + (wave - NW / 2) * PT2B);

        // ---- publish: every storing wave drains its write-through stores, barrier, ONE lane raises the tile's flag ----
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                // A (also: the LDS tile holds the whole new centre)
        asm volatile("" ::: "memory");
        if (tid0 == 0 && kb + 1 < p.nblk)
            __hip_atomic_store(p.flags + tile_id, fepoch + (unsigned)(kb + 1 + fbase), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the last workgroup to finish opens the next epoch ----
    if (tid0 == 0) {
        const unsigned old = __hip_atomic_fetch_add(p.flags + CHAIN_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {
            __hip_atomic_store(p.flags + CHAIN_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.flags + CHAIN_EPOCH, (fepoch >> 6) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------

struct ChainDev {
    unsigned* flags = nullptr;        // device, 4 KiB
    unsigned* status_host = nullptr;  // pinned, mapped
    unsigned* status_dev = nullptr;
    hipStream_t last_stream = nullptr;   // stream of the most recent chain launch
    bool launched = false;
    int cus = 0;
};
static std::mutex g_chain_mu;
static ChainDev g_chain_dev[64];

static ChainDev* chain_dev() {       // call with g_chain_mu held
    int dev = 0;
    if (hip_check(hipGetDevice(&dev), "hipGetDevice")) return nullptr;
    if (dev < 0 || dev >= 64) { set_error("resblock_chain: device index %d out of range", dev); return nullptr; }
    ChainDev& d = g_chain_dev[dev];
    if (!d.flags) {
        hipDeviceProp_t prop;
        if (hip_check(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties")) return nullptr;
        void* st = nullptr;
        if (hip_check(hipHostMalloc(&st, 64, hipHostMallocMapped), "hipHostMalloc(chain status)")) return nullptr;
        memset(st, 0, 64);
        void* std_ = nullptr;
        if (hip_check(hipHostGetDevicePointer(&std_, st, 0), "hipHostGetDevicePointer")) return nullptr;
        void* fl = nullptr;
        if (hip_check(hipMalloc(&fl, CHAIN_FLAG_WORDS * 4), "hipMalloc(chain flags)")) return nullptr;
        if (hip_check(hipMemset(fl, 0, CHAIN_FLAG_WORDS * 4), "hipMemset(chain flags)")) return nullptr;     // once: epoch 0, nothing published
        d.status_host = (unsigned*)st; d.status_dev = (unsigned*)std_;
        d.cus = prop.multiProcessorCount;
        d.flags = (unsigned*)fl;
    }
    return &d;
}

// sticky: a chained launch on the current device gave up waiting for a neighbouring workgroup (its outputs are garbage)
int chain_status() {
    std::lock_guard<std::mutex> lk(g_chain_mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SRMI_OK;
    const ChainDev& d = g_chain_dev[dev];
    if (d.status_host && *(volatile unsigned*)d.status_host) {
        set_error("resblock_chain: a persistent launch on device %d gave up waiting for a neighbouring workgroup (code 0x%x): "
                  "its workgroups were not co-resident (another process or kernel holds CUs); its outputs are invalid",
                  dev, *(volatile unsigned*)d.status_host);
        return SRMI_ETIMEOUT;
    }
    return SRMI_OK;
}

// the pair (a1, a2) - already known to be a fusable ResBlock - can be a link of a chain
static bool chain_link_ok(const srmi_conv_args& a1, const srmi_conv_args& a2) {
    const srmi_conv_desc& d1 = a1.d;
    const srmi_conv_desc& d2 = a2.d;
    if (d1.in_cstride && d1.in_cstride != 64) return false;
    if (d2.out_cstride && d2.out_cstride != 64) return false;
    if (d2.out_coffset) return false;
    if (!a2.residual || a2.residual != a1.x || d2.res_dtype != SRMI_BF16) return false;     // y = x + ...: the residual comes from the LDS tile
    if (d1.W <= 64 || d1.H < 12) return false;                                              // shapes the 12 x 48 tile serves
    if (((uintptr_t)a1.x | (uintptr_t)a1.w | (uintptr_t)a2.w | (uintptr_t)a2.y) & 15) return false;
    return true;
}

// a single conv right behind the chained pairs that can run as a tail link: 3x3 64 -> 64 on the previous link's output,
// plus a dense bf16 residual that no block of the chain writes (conv02 + the long skip of Generator.forward,
// P/srgan_model.py:33-34).  nprev = convs of the chain before it.
static bool chain_tail_ok(const srmi_conv_args* a, int nprev, const srmi_conv_args& t) {
    const srmi_conv_desc& d = t.d;
    if (!(t.flags & SRMI_CONV_FUSE_CHAIN) || (t.flags & SRMI_CONV_FUSE_NEXT) || t.x_copy) return false;
    if (!c64_supported(d) || d.Cout != 64 || d.shuffle != 1) return false;
    if (d.N != a[0].d.N || d.H != a[0].d.H || d.W != a[0].d.W) return false;
    if ((d.in_cstride && d.in_cstride != 64) || (d.out_cstride && d.out_cstride != 64) || d.out_coffset) return false;
    if (d.res_cstride && d.res_cstride != 64) return false;
    if (!t.x || t.x != a[nprev - 1].y || !t.w || !t.y || !t.scale || !t.shift) return false;
    if (!t.residual || d.res_dtype != SRMI_BF16) return false;
    if (d.act == SRMI_ACT_PRELU && !t.alpha) return false;
    if (((uintptr_t)t.w | (uintptr_t)t.y | (uintptr_t)t.residual) & 15) return false;
    if (t.y == t.x || t.y == t.residual) return false;
    for (int i = 1; i < nprev; i += 2)
        if (a[i].y == t.residual || a[i].y == t.y) return false;   // (the residual is read by plain loads; y must not be a live halo source)
    return true;
}

// the LAST link: a single conv 3x3 64 -> 64 * (1..4) on the previous link's output, any c64 store mode (PixelShuffle),
// no residual, optionally with srmi_conv_args.x_copy (the first Upsampler conv of Generator.forward, P/ops.py:74-78)
static bool chain_wide_ok(const srmi_conv_args* a, int nprev, const srmi_conv_args& t) {
    const srmi_conv_desc& d = t.d;
    if (!(t.flags & SRMI_CONV_FUSE_CHAIN) || (t.flags & SRMI_CONV_FUSE_NEXT)) return false;
    if (!c64_supported(d) || d.Cout > 256) return false;
    if (d.N != a[0].d.N || d.H != a[0].d.H || d.W != a[0].d.W) return false;
    if (d.in_cstride && d.in_cstride != 64) return false;
    const int Cf = d.Cout / (d.shuffle * d.shuffle);
    if ((d.out_cstride && d.out_cstride != Cf) || d.out_coffset) return false;
    if (!t.x || t.x != a[nprev - 1].y || !t.w || !t.y || !t.scale || !t.shift || t.residual) return false;
    if (d.act == SRMI_ACT_PRELU && !t.alpha) return false;
    if (((uintptr_t)t.w | (uintptr_t)t.y) & 15) return false;
    if (t.x_copy && ((d.W & 3) || ((uintptr_t)t.x_copy & 15))) return false;
    if (t.y == t.x) return false;
    for (int i = 1; i < nprev; i += 2)
        if (a[i].y == t.y) return false;
    if (a[nprev - 1].y == t.y) return false;
    return true;
}

// the conv right in front of the first pair as the chain's head link: 9x9 from <= 3 channels to 64 (conv01 of
// Generator.forward, P/srgan_model.py:29), any input form "k9c3_bf16" reads, dense 64-channel bf16 output = the first pair's x
static bool chain_head_ok(const srmi_conv_args& h, const srmi_conv_args& first) {
    const srmi_conv_desc& d = h.d;
    if (!(h.flags & SRMI_CONV_FUSE_CHAIN) || (h.flags & SRMI_CONV_FUSE_NEXT) || h.x_copy || h.residual) return false;
    if (!k9c3_supported(d) || select_kernel(d) != K_K9C3_BF16 || d.Cout != 64 || d.kh != 9) return false;      // (packed-weight format = the 9x9 kernel's)
    if (d.N != first.d.N || d.H != first.d.H || d.W != first.d.W) return false;
    if ((d.out_cstride && d.out_cstride != 64) || d.out_coffset) return false;
    if (!h.x || !h.w || !h.y || !h.scale || !h.shift || h.y != first.x) return false;
    if (d.act == SRMI_ACT_PRELU && !h.alpha) return false;
    if (((uintptr_t)h.w | (uintptr_t)h.y) & 15) return false;
    ConvGeom g;
    if (conv_geom(d, &g)) return false;
    return true;
}

// Every workgroup of a chained launch must be resident at once (one per CU).  A batch with more tiles than CUs runs as
// several chained launches of k whole images each when that still beats one launch per block: the chain saves ~1/6 of a
// block's time (20 against 24 us at cfg1), so its fill N tpi / (launches * CUs) times 1.2 has to reach the fill of the
// launch-per-block form, N tpi / (rounds * CUs).  Returns k, or 0 for "do not chain".
static int chain_images_per_launch(long long N, long long tpi, long long cus) {
    if (N <= 0 || tpi <= 0 || tpi > cus || tpi > CHAIN_MAX_TILES) return 0;
    long long k = cus / tpi;
    if (k * tpi > CHAIN_MAX_TILES) k = CHAIN_MAX_TILES / tpi;
    if (k >= N) return (int)N;
    const long long launches = (N + k - 1) / k, rounds = (N * tpi + cus - 1) / cus;
    return 6 * rounds >= 5 * launches ? (int)k : 0;           // 1.2 / launches >= 1 / rounds
}

// Convs of a[0 .. n) that can run as one chain starting at a[0]: L >= 2 ResBlock pairs (a[0], a[1]), (a[2], a[3]), ..., then
// up to two single-conv tail links (*ntail: conv + residual, then the wide last link; either may be missing): 2 L + ntail,
// or 0.  `fusable(a1, a2)` is api.hip's test for the fused ResBlock pair.
int chain_run_length(const srmi_conv_args* a0, int n0, bool (*fusable)(const srmi_conv_args*, const srmi_conv_args*), int* npairs,
                     int* kinds, int* head) {
    // an optional head link in front (*head = 1): the pairs then start at a0[1]
    const bool hd = n0 > 1 && (a0[0].flags & SRMI_CONV_FUSE_CHAIN) && !(a0[0].flags & SRMI_CONV_FUSE_NEXT) && chain_head_ok(a0[0], a0[1]);
    const srmi_conv_args* a = a0 + (hd ? 1 : 0);
    const int n = n0 - (hd ? 1 : 0);
    if (head) *head = hd ? 1 : 0;
    if (n < 4) return 0;
    {   // cheap rejections first (this runs for every flagged conv of every forward): the shape and the grid
        const srmi_conv_desc& d = a[0].d;
        if (d.W <= 64 || d.H < 12 || d.N <= 0) return 0;
        const long long tpi = (long long)((d.W + 47) / 48) * ((d.H + 11) / 12);     // tiles per image
        std::lock_guard<std::mutex> lk(g_chain_mu);
        ChainDev* cd = chain_dev();
        if (!cd) return 0;
        if (chain_images_per_launch(d.N, tpi, cd->cus) == 0) return 0;
    }
    int L = 0;
    while (L < CHAIN_MAX_BLOCKS - 2 && 2 * L + 1 < n) {
        const srmi_conv_args& a1 = a[2 * L];
        const srmi_conv_args& a2 = a[2 * L + 1];
        if (!(a1.flags & SRMI_CONV_FUSE_NEXT) || !(a1.flags & SRMI_CONV_FUSE_CHAIN)) break;
        if (!fusable(&a1, &a2) || !chain_link_ok(a1, a2)) break;
        if (L > 0) {
            const srmi_conv_args& p1 = a[2 * L - 2];
            const srmi_conv_args& p2 = a[2 * L - 1];
            if (a1.x != p2.y || a1.d.N != p1.d.N || a1.d.H != p1.d.H || a1.d.W != p1.d.W) break;
        }
        ++L;
    }
    if (L < 2) return 0;
    int nc = 2 * L, nt = 0;
    int k[2] = {0, 0};
    if (nc < n && chain_tail_ok(a, nc, a[nc])) { k[nt++] = 1; ++nc; }
    if (nc < n && chain_wide_ok(a, nc, a[nc])) { k[nt++] = 2; ++nc; }
    if (npairs) *npairs = L;
    if (kinds) { kinds[0] = k[0]; kinds[1] = k[1]; }
    return nc + (hd ? 1 : 0);
}

// a0 = first conv of the chain (the head link if head != 0, else the first pair's first conv)
int chain_launch(const srmi_conv_args* a0, int head, int npairs, const int* kinds, hipStream_t s) {
    const srmi_conv_args* a = a0 + (head ? 1 : 0);
    const int L = npairs;
    const int ntail = (kinds[0] ? 1 : 0) + (kinds[1] ? 1 : 0);
    const srmi_conv_desc& d = a[0].d;
    if (d.N == 0 || d.H == 0 || d.W == 0) return SRMI_OK;
    std::lock_guard<std::mutex> lk(g_chain_mu);
    ChainDev* cd = chain_dev();
    if (!cd) return SRMI_EHIP;
    if (*(volatile unsigned*)cd->status_host) {
        set_error("resblock_chain: an earlier persistent launch on this device gave up waiting (code 0x%x); see srmi_chain_status",
                  *(volatile unsigned*)cd->status_host);
        return SRMI_ETIMEOUT;
    }
    ChainParams p;
    memset(&p, 0, sizeof p);
    p.x = (const bf16_t*)a[0].x;
    if (head) {
        const srmi_conv_args& h = a0[0];
        ConvGeom g;
        int rc = conv_geom(h.d, &g);
        if (rc) return rc;
        p.head.on = 1;
        p.head.x = h.x; p.head.w = (const bf16_t*)h.w;
        p.head.scale = h.scale; p.head.shift = h.shift; p.head.alpha = h.alpha;
        p.head.y = (bf16_t*)h.y;
        p.head.Cin = h.d.Cin; p.head.in_cs = g.in_cs;
        p.head.in_nchw = h.d.in_layout == SRMI_NCHW; p.head.in_bf16 = h.d.in_dtype == SRMI_BF16;
        p.head.act = h.d.act; p.head.act_param = h.d.act_param; p.head.alpha_pc = h.d.alpha_per_channel;
    }
    p.flags = cd->flags; p.status = cd->status_dev;
    p.N = d.N; p.H = d.H; p.W = d.W;
    p.tiles_x = (d.W + 47) / 48; p.tiles_y = (d.H + 11) / 12;
    p.nblk = L + ntail;
    for (int i = 0; i < L; ++i) {
        const srmi_conv_args& a1 = a[2 * i];
        const srmi_conv_args& a2 = a[2 * i + 1];
        ChainBlk& B = p.blk[i];
        B.w1 = (const bf16_t*)a1.w; B.w2 = (const bf16_t*)a2.w;
        B.scale1 = a1.scale; B.shift1 = a1.shift; B.alpha1 = a1.alpha;
        B.scale2 = a2.scale; B.shift2 = a2.shift; B.alpha2 = a2.alpha;
        B.y = (bf16_t*)a2.y;
        B.act1 = a1.d.act; B.act_param1 = a1.d.act_param; B.alpha_pc1 = a1.d.alpha_per_channel;
        B.act2 = a2.d.act; B.act_param2 = a2.d.act_param; B.alpha_pc2 = a2.d.alpha_per_channel;
        B.halo_only = (i + 1 < L + ntail) && (a1.flags & SRMI_CONV_Y_TRANSIENT) ? 1 : 0;
    }
    for (int j = 0; j < ntail; ++j) {
        const srmi_conv_args& t = a[2 * L + j];
        ChainBlk& B = p.blk[L + j];
        B.single = kinds[j];
        B.w2 = (const bf16_t*)t.w;
        B.scale2 = t.scale; B.shift2 = t.shift; B.alpha2 = t.alpha;
        B.y = (bf16_t*)t.y;
        B.act2 = t.d.act; B.act_param2 = t.d.act_param; B.alpha_pc2 = t.d.alpha_per_channel;
        // (conv "1"'s parameters are loaded but never used: point them at valid memory)
        B.scale1 = t.scale; B.shift1 = t.shift; B.alpha1 = t.alpha; B.act1 = SRMI_ACT_NONE;
        if (kinds[j] == 1) {
            B.res = (const bf16_t*)t.residual;
            B.halo_only = (j + 1 < ntail) && (t.flags & SRMI_CONV_Y_TRANSIENT) ? 1 : 0;
        } else {
            B.nchunk = t.d.Cout / 64;
            B.mode = t.d.shuffle == 1 ? C64_PLAIN : (t.d.shuffle_order == SRMI_SHUFFLE_DCR ? C64_DCR : C64_CRD);
            B.xcopy = t.x_copy;
        }
    }
    // two persistent grids at once would each hold half the CUs and wait for the other half: a chain launch on another
    // stream than the previous one waits (on the host: rare, and it costs the one-stream case nothing) for that stream
    if (cd->launched && cd->last_stream != s) {
        (void)hipStreamSynchronize(cd->last_stream);      // (a destroyed stream has drained: errors are not ours)
        (void)hipGetLastError();
    }
    // a batch with more tiles than CUs: k whole images per launch (chain_images_per_launch), every per-image pointer advanced
    const int tpi = p.tiles_x * p.tiles_y;
    const int kimg = chain_images_per_launch(d.N, tpi, cd->cus);
    if (kimg <= 0) { set_error("resblock_chain: %d images of %d tiles do not fit %d CUs", d.N, tpi, cd->cus); return SRMI_EINVAL; }
    const size_t HW = (size_t)d.H * d.W;
    const ChainParams p0 = p;
    for (int n0 = 0; n0 < d.N; n0 += kimg) {
        p = p0;
        p.N = std::min(kimg, d.N - n0);
        p.x = p0.x + (size_t)n0 * HW * 64;
        if (head) {
            const size_t esz = p0.head.in_bf16 ? 2 : 4;
            const size_t img = p0.head.in_nchw ? (size_t)p0.head.Cin * HW : HW * p0.head.in_cs;
            p.head.x = (const char*)p0.head.x + (size_t)n0 * img * esz;
            p.head.y = p0.head.y + (size_t)n0 * HW * 64;
        }
        for (int i = 0; i < p0.nblk; ++i) {
            const ChainBlk& B0 = p0.blk[i];
            ChainBlk& B = p.blk[i];
            const size_t cout = B0.single == 2 ? (size_t)B0.nchunk * 64 : 64;      // the wide link writes H r x W r x Cout / r^2
            B.y = B0.y + (size_t)n0 * HW * cout;
            if (B0.res) B.res = B0.res + (size_t)n0 * HW * 64;
            if (B0.xcopy) B.xcopy = B0.xcopy + (size_t)n0 * HW * 64;
        }
        hipLaunchKernelGGL(resblock_chain_kernel, dim3((unsigned)(p.N * tpi)), dim3(512), 0, s, p);
        int rc = hip_check(hipGetLastError(), "resblock_chain_kernel");
        if (rc) return rc;
        cd->launched = true;
        cd->last_stream = s;
    }
    return SRMI_OK;
}

}  // namespace srmi
