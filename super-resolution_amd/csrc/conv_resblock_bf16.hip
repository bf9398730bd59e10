// Fused ResBlock: y = x + affine2(conv3x3(act(affine1(conv3x3(x)))))  for 64 channels, NHWC bf16,
// in ONE launch (P/ops.py:37-48: conv-BN-PReLU, conv-BN, `res += x`; eval-mode BN folded into affine).
//
// Why: between two launches every activation goes through the memory side (per-XCD L2s are written
// back at a kernel boundary), which is bound at ~6 TB/s; an unfused 64->64 conv moves 64 MB per
// 10.9 GFLOP and is memory-bound.  Keeping the intermediate tensor T in LDS halves the traffic of
// the network's 32 body convs at the price of re-computing T on a one-pixel halo ring.
//
//   LDS (one array, 157 KiB for the 6 x 96 tile):
//     [ x halo tile (TH+4) x (TW+4) px, later overwritten by T (TH+2) x (TW+2) px | weight ring | params ]
//   phase 1  conv1 on the (TH+2) x (TW+2) T region, FLATTENED into 16-pixel MFMA tiles (49 tiles for
//            8 x 98 pixels, 7 per wave with 8 waves) so the halo ring costs 1.36x, not 1.56x, of a tile's MFMAs;
//   phase 2  barrier; T = act(affine1(acc)) -> bf16, zero outside the image (conv2's zero padding),
//            written over the x tile in the same swizzled layout; barrier;
//   phase 3  conv2 from T exactly like conv_c64_kernel; epilogue adds x and stores.  With 8 waves (5 pixel tiles
//            per wave) x is lifted out of the LDS tile into 40 registers just before T overwrites it; with 4 waves
//            (9 tiles, 72 registers: hipcc spills them at the 512-register limit) it is re-read from L2.
//   Weights: both convs' packed A fragments stream through a 4-slot x 4 KiB LDS-DMA ring, one K-step
//   (4 fragments) per slot, three steps ahead, across the phase boundary; hand-off per step =
//   counted vmcnt + s_barrier (slot j-1 is re-filled at barrier j: its readers are provably done).
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <type_traits>

#include "c64_common.h"

namespace srmi {

struct ResBlockVar {      // what changes from block to block
    const bf16_t* x; const bf16_t* w1; const bf16_t* w2;
    C64Epi epi1;   // scale/shift/act of conv1 (res, y unused)
    C64Epi epi2;   // scale/shift/act of conv2, residual, output
};

struct ResBlockParams {
    ResBlockVar v;
    int N, H, W, in_cs;
    int tiles_x, tiles_y;
    int res_is_x;   // the residual is the block's own input (same pointer, 64-channel pixels): it can come from the LDS tile
    int dbg;  // ablation (SRMI_DBG env, investigation only): 1 skip staging, 2 skip conv1 K loop, 8 skip conv2 K loop, 4 skip epilogue, 16 skip T write
};

// Persistent chain: nblocks consecutive ResBlocks (block i+1 reads what block i wrote) in ONE launch, one workgroup per
// tile for the whole chain (see the kernel comment, CHAIN = true).
constexpr int RB_CHAIN_MAX = 16;          // blocks per launch (the table travels as kernel arguments: 16 x 184 B)
constexpr int RB_CHAIN_SPIN = 1 << 18;    // polls of a neighbour flag before giving up (~0.3 s): a wait every wave leaves
struct ResChainParams {
    ResBlockVar blk[RB_CHAIN_MAX];
    unsigned* flags;      // one word per tile (this launch's flag set): target0 + number of blocks the tile has published
    unsigned* timeouts;   // counts waits that gave up (results are then wrong; tests and srmi_chain_status read it)
    unsigned long long* trace;   // investigation only (SRMI_DBG bit 16384): 8 phase time stamps per tile of block nblocks/2
    unsigned target0;
    int nblocks;
    int N, H, W, in_cs;
    int tiles_x, tiles_y;
    int res_is_x;
    int dbg;
};

constexpr int RB_STEP_BYTES = 4096;   // one K-step: 4 channel tiles x 64 lanes x 16 B

// NW = waves per workgroup: 4 (one per SIMD, 512 registers each) or 8 (two per SIMD, 256 each: the second wave
// of a SIMD fills the first one's LDS / barrier / VALU-epilogue bubbles; same tile, same LDS, half the pixel
// tiles per wave).
//
// SLOTS = depth of the weight ring (K-steps), WGPC = workgroups per CU the register budget is cut for.  The 6 x 48 tile
// with 4 waves and a 3-slot ring needs 78.5 KiB of LDS and 256 registers: TWO workgroups share a CU, and while one
// sits in a phase that leaves the matrix cores idle (launch, staging, T write-back, epilogue, store drain) the other
// one computes - provided they are out of phase, which two streams (or several rounds) give.
//
// CHAIN = true: the persistent chain.  The workgroup keeps its tile for all blocks: the epilogue of block i writes the
// centre of block i+1's x tile straight into LDS (and the whole tile to memory with agent-scope stores, for the
// neighbours and for the caller), publishes a per-tile flag once those stores have completed, waits for the flags of
// its (up to 8) neighbour tiles and re-stages only the 2-pixel halo ring with agent-scope loads.  No launch floor, no
// cold prologue, 42 % of the staging and no end-of-kernel write drain between blocks.  Needs every tile resident at
// once (grid <= CUs at one workgroup per CU, checked on the host); every wait is bounded (RB_CHAIN_SPIN).
template <int TH, int TWT, int NW, int SLOTS, int WGPC, bool CHAIN, int KS = 1, int PITCH = 16>
__global__ __launch_bounds__(NW * 64, WGPC) void resblock_c64_kernel(
    typename std::conditional<CHAIN, ResChainParams, ResBlockParams>::type p) {
    constexpr int NTH = NW * 64;
    constexpr int TW = TWT * 16;
    constexpr int XR = TH + 4, XC = TW + 4;          // x halo tile
    constexpr int TR = TH + 2, TC = TW + 2;          // T region
    constexpr int ROWP = (XC + PITCH - 1) / PITCH * PITCH;   // PITCH: 16 (padded), 8, or 2 (= XC, unpadded)   // LDS row pitch in pixels: even, so a row starts on a 256-B bank row and
                                                     // the swizzle key (col >> 1) & 7 is the same for every row (tap dy = +pitch)
    static_assert(ROWP % 2 == 0, "pixel pairs must not straddle rows");
    // swizzle key of LDS pixel P = row * ROWP + col (the P form keeps fewer values alive when the pitch allows it)
    auto swz = [](int P, int col) { return ROWP % 16 == 0 ? (P >> 1) & 7 : (col >> 1) & 7; };
    constexpr int IN_BYTES = XR * ROWP * 128;
    constexpr int NT1 = (TR * TC + 15) / 16;         // flattened 16-pixel tiles of the T region
    constexpr int PT1 = (NT1 + NW - 1) / NW;
    constexpr int NPT2 = TH * TWT;
    constexpr int PT2 = (NPT2 + NW - 1) / NW;          // waves past the end recompute the last tile and store nothing
    // KS = K-steps per ring slot: 2 (one 3x3 tap, 8 KiB) halves the hand-off barriers of the K loops
    constexpr int SLOT_BYTES = KS * RB_STEP_BYTES;
    constexpr int NS = 36 / KS;                       // slots in the weight stream of one block
    static_assert(36 % KS == 0 && 18 % KS == 0, "a slot must not straddle the two convs' weight arrays");
    __shared__ __attribute__((aligned(16))) unsigned char lds[IN_BYTES + SLOTS * SLOT_BYTES + 2 * 768];
    unsigned char* const wring = lds + IN_BYTES;
    float* const par = (float*)(lds + IN_BYTES + SLOTS * SLOT_BYTES);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // XCD-aware tile order: consecutive workgroup ids go round-robin to the 8 XCDs (one L2 each), so give each
    // XCD a CONTIGUOUS run of tiles - vertically adjacent tiles then share their halo rows through one L2.
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0 && !(p.dbg & 32)) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int tile_id = (n * p.tiles_y + ty) * p.tiles_x + tx;
    const int kdbg = CHAIN ? 0 : p.dbg;       // phase-ablation switches (never in the chain: they cost registers)
    auto stamp = [&](int blk, int i) {
        if constexpr (CHAIN) {
            if (p.trace && blk == p.nblocks / 2 && threadIdx.x == 0) p.trace[tile_id * 8 + i] = __builtin_amdgcn_s_memrealtime();
        }
    };

    int nblk = 1;
    if constexpr (CHAIN) nblk = p.nblocks;
#pragma nounroll
    for (int blk = 0; blk < nblk; ++blk) {
    const ResBlockVar& V = [&]() -> const ResBlockVar& {
        if constexpr (CHAIN) return p.blk[blk]; else return p.v;
    }();
    const bool ring = CHAIN && blk > 0;      // the centre of the x tile is already in LDS: stage the halo ring only
    stamp(blk, 0);
    // (chain: the lane ids are re-read through an opaque move in every block, or hipcc hoists all per-lane address
    // arithmetic out of the block loop and spills it)
    int tid = threadIdx.x;
    if constexpr (CHAIN) asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int pl = lane & 15, q = lane >> 4;

    // (pinned in scalar registers: in the chain hipcc would otherwise re-load the pointers from the kernel-argument table in
    // every K-step, and the s_waitcnt lgkmcnt(0) that goes with the scalar load drains the LDS operand reads)
    const unsigned char* wb1 = (const unsigned char*)V.w1;
    const unsigned char* wb2 = (const unsigned char*)V.w2;
    asm volatile("" : "+s"(wb1), "+s"(wb2));
    // ---- weight stream: K-step g of 36 (conv1: 0..17, conv2: 18..35); wave w copies KiB w of each ----
    const unsigned wring_lds = (unsigned)(uintptr_t)(lptr_t)wring;
    // slot S of the stream = K-steps S*KS .. S*KS+KS-1; the waves share the copy in 1-KiB DMA instructions
    constexpr int WAVE_BYTES = SLOT_BYTES / NW >= 1024 ? SLOT_BYTES / NW : 1024;
    constexpr int NDMA = WAVE_BYTES / 1024;            // DMA instructions per wave and slot (0 for the waves past SLOT_BYTES / 1024)
    constexpr int NCOPY = SLOT_BYTES / WAVE_BYTES;     // waves that copy
    auto issue_slot = [&](int S) {
        const int g0 = S * KS;
        const unsigned char* base = g0 < 18 ? wb1 : wb2;
        const unsigned char* src = base + (size_t)(g0 < 18 ? g0 : g0 - 18) * RB_STEP_BYTES + wave * WAVE_BYTES + lane * 16;
        const unsigned dst = __builtin_amdgcn_readfirstlane(
            wring_lds + (unsigned)((S % SLOTS) * SLOT_BYTES + wave * WAVE_BYTES));
        unsigned keep;
        if (NCOPY == NW || wave < NCOPY) {
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
                static_assert(NDMA == 1, "1 or 2 KiB per wave and slot");
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
        }
    };
#pragma unroll
    for (int S = 0; S < SLOTS - 1; ++S) issue_slot(S);

    C64ParRegs pregs1, pregs2;
    c64_params_load(pregs1, V.epi1, 1, tid);
    c64_params_load(pregs2, V.epi2, 1, tid);
    if constexpr (CHAIN) {
        if (ring) {
            // the neighbours' block blk-1 outputs must be complete in memory: lanes 0..7 of wave 0 poll one flag each
            if (wave == 0) {
                const int l = lane & 7;
                const int dy = l < 3 ? -1 : (l < 5 ? 0 : 1);
                const int dx = l < 3 ? l - 1 : (l < 5 ? (l == 3 ? -1 : 1) : l - 6);
                const bool have = lane < 8 && ty + dy >= 0 && ty + dy < p.tiles_y && tx + dx >= 0 && tx + dx < p.tiles_x;
                const unsigned* f = p.flags + (have ? tile_id + dy * p.tiles_x + dx : tile_id);
                const unsigned target = p.target0 + (unsigned)blk;
                bool done = !have;
                int it = 0;
                for (; it < RB_CHAIN_SPIN; ++it) {
                    if (!done) done = (int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0;
                    if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (it == RB_CHAIN_SPIN && lane == 0) atomicAdd(p.timeouts, 1u);
            }
            __builtin_amdgcn_s_barrier();
            stamp(blk, 1);
        }
    }
    // ---- stage the x halo tile (2-pixel ring) ----------------------------------------------------
    // Row-oriented so that almost all index arithmetic is wave-uniform (SALU) or a compile-time immediate: part A
    // gives a thread one (column, 16-B chunk) and walks the XR rows; part B spreads the XC % (NTH/8) left-over
    // columns over the threads.  Pixels outside the image are read through a buffer descriptor at an out-of-range
    // offset, which returns zeros without touching memory (conv1's zero padding) - no per-load select.
    if (!(kdbg & 1)) {
        constexpr int CPT = NTH / 8;                  // columns per pass
        constexpr int NA = XC / CPT, RB = XC - NA * CPT;
        constexpr int NB = (XR * RB * 8 + NTH - 1) / NTH;
        constexpr unsigned OOB_COL = 0x80000000u, OOB_ROW = 0x40000000u;   // image bytes <= 2^30 (checked on the host)
        const unsigned img_bytes = (unsigned)p.H * p.W * p.in_cs * 2;
        const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(V.x + (size_t)n * p.H * p.W * p.in_cs), 0, img_bytes, 0x00020000);
        constexpr int AUX = CHAIN ? 16 : 0;          // chain: agent-scope (sc1) loads - the data was written by other CUs in this launch
        const unsigned row_bytes = (unsigned)p.W * p.in_cs * 2;
        const int ch = tid & 7, c0 = tid >> 3;
        u32x4 va[NA > 0 ? NA * XR : 1], vb[NB > 0 ? NB : 1];
        unsigned la[NA > 0 ? NA : 1], lb[NB > 0 ? NB : 1];
#pragma unroll
        for (int g = 0; g < NA; ++g) {
            const int col = g * CPT + c0, gx = x0 - 2 + col;
            const unsigned coff = (gx >= 0 && gx < p.W) ? (unsigned)gx * p.in_cs * 2 + ch * 16 : OOB_COL;
            la[g] = col * 128 + ((ch ^ ((col >> 1) & 7)) << 4);
#pragma unroll
            for (int row = 0; row < XR; ++row) {
                const int gy = y0 - 2 + row;                               // wave-uniform
                const unsigned roff = (gy >= 0 && gy < p.H) ? (unsigned)gy * row_bytes : OOB_ROW;
                const bool need = (!ring || row < 2 || row >= XR - 2 || col < 2 || col >= XC - 2) && !(CHAIN && ring && (p.dbg & 65536));
                va[g * XR + row] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, need ? coff + roff : OOB_COL, 0, AUX));
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int sidx = tid + i * NTH;
            const int row = sidx / (RB * 8), rem = sidx - row * (RB * 8);
            const int col = NA * CPT + (rem >> 3), chb = rem & 7;
            const int gy = y0 - 2 + row, gx = x0 - 2 + col;
            const bool need = !ring || row < 2 || row >= XR - 2 || col < 2 || col >= XC - 2;
            const bool ok = sidx < XR * RB * 8 && need && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const unsigned off = ok ? (unsigned)gy * row_bytes + (unsigned)gx * p.in_cs * 2 + chb * 16 : OOB_COL;
            vb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, off, 0, AUX));
            lb[i] = (row * ROWP + col) * 128 + ((chb ^ ((col >> 1) & 7)) << 4);
            if (!need) lb[i] = 0xffffffffu;
        }
#pragma unroll
        for (int g = 0; g < NA; ++g)
#pragma unroll
            for (int row = 0; row < XR; ++row) {
                const int col = g * CPT + c0;
                const bool need = !ring || row < 2 || row >= XR - 2 || col < 2 || col >= XC - 2;
                if (need) *(u32x4*)(lds + la[g] + row * (ROWP * 128)) = va[g * XR + row];
            }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (tid + i * NTH < XR * RB * 8 && lb[i] != 0xffffffffu) *(u32x4*)(lds + lb[i]) = vb[i];
    }
    c64_params_store(par, pregs1, 1, tid);
    c64_params_store(par + 192, pregs2, 1, tid);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // also retires weight steps 0..2
    __builtin_amdgcn_s_barrier();
    stamp(blk, 2);

    // one K-step hand-off: step g+1 landed (own copy: counted vmcnt, others': barrier); slot of step g-1
    // is free (its fragments were consumed by the MFMAs of step g-1, which every wave has issued).
    // In slot units: called at the top of step g when step g+1 opens slot S = (g+1)/KS.  Slots up to S+SLOTS-3 are in
    // flight; all but S may stay so.  Then slot S+SLOTS-2 is issued into the ring position of slot S-2, whose fragments
    // were consumed by MFMAs every wave has issued.
    // `extra`: that many younger vector loads (the chain's residual prefetch) are allowed to stay in flight as well
    constexpr int NRES = 2 * ((TH * TWT + NW - 1) / NW);
    constexpr int S_LAST_ISSUE = NS - SLOTS + 1;       // the hand-off that issues the last slot of the block
    auto handoff = [&](int g) {
        if ((g + 1) % KS != 0) return;
        const int S = (g + 1) / KS;
        const bool extra = CHAIN && S > S_LAST_ISSUE;
        const bool more = SLOTS >= 4 && S + 1 < NS;         // slot S+1 exists and may stay in flight
        if (extra) {
            if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (SLOTS - 3) + NRES) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NRES) : "memory");
        } else if (!(kdbg & (1 << 17))) {       // (timing experiment: no wait for the weights)
            if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (SLOTS - 3)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!(kdbg & (1 << 18))) __builtin_amdgcn_s_barrier();      // (timing experiment: no hand-off barrier)
        if (S + SLOTS - 2 < NS && !(kdbg & (1 << 19))) issue_slot(S + SLOTS - 2);     // (timing experiment: no weight stream)
    };
    // LDS address of the fragments of K-step g
    auto frag = [&](int g) { return wring + ((g / KS) % SLOTS) * SLOT_BYTES + (g % KS) * RB_STEP_BYTES + lane * 16; };

    // ================= phase 1: conv1 over the flattened T region =================================
    int boff1[PT1][3];
    unsigned inside = 0;     // bit k: this lane's pixel of tile k lies inside the image (else T = 0: conv2's zero padding)
#pragma unroll
    for (int k = 0; k < PT1; ++k) {
        const int pix = min((wave * PT1 + k) * 16 + pl, TR * TC - 1);   // padding tiles recompute the last pixel
        const int ti = pix / TC, tj = pix - ti * TC;
        const int gy = y0 - 1 + ti, gx = x0 - 1 + tj;
        inside |= (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? 1u << k : 0u;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int P = ti * ROWP + tj + dx;
            boff1[k][dx] = P * 128 + ((q ^ swz(P, tj + dx)) << 4);
        }
    }
    asm volatile("" : "+v"(inside));      // materialise the mask HERE (hipcc would sink it below the K loop and keep its 2*PT1 inputs alive)
    {
        f32x4 acc[PT1][4];
#pragma unroll
        for (int k = 0; k < PT1; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[k][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 afr[2][4], bfr[2][PT1];
#pragma unroll
        for (int t = 0; t < 4; ++t) afr[0][t] = *(const bf16x8*)(wring + lane * 16 + t * 1024);
#pragma unroll
        for (int k = 0; k < PT1; ++k) bfr[0][k] = *(const bf16x8*)(lds + boff1[k][0]);
        if (!(kdbg & 2))
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            handoff(j);
            {   // prefetch the fragments of global step j+1 (step 18 = first step of conv2: weights only)
                const unsigned char* wslot = frag(j + 1);
#pragma unroll
                for (int t = 0; t < 4; ++t) afr[(j + 1) & 1][t] = *(const bf16x8*)(wslot + t * 1024);
                if (j + 1 < 18) {
                    const int tapn = (j + 1) >> 1, halfn = (j + 1) & 1;
                    const int dyn = tapn / 3, dxn = tapn - 3 * dyn;
#pragma unroll
                    for (int k = 0; k < PT1; ++k)
                        bfr[(j + 1) & 1][k] =
                            *(const bf16x8*)(lds + ((boff1[k][dxn] ^ (halfn << 6)) + dyn * (ROWP * 128)));
                }
            }
#pragma unroll
            for (int k = 0; k < PT1; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[j & 1][t], bfr[j & 1][k], acc[k][t], 0, 0, 0);
            if (j + 1 < 18) sched_interleave<0, 4 + PT1, 4 * PT1>();
        }

        // ---- phase 2: T = act(affine1(acc)) as bf16 over the x tile (zero outside the image) ----
        // The residual IS the centre of the x tile: take it from LDS now (the operand fragments are dead, T is about
        // to overwrite the tile) instead of re-reading it from L2 in the epilogue.  PT2 * 8 registers, kept across
        // conv2; worth it while they fit (8-wave variant, small tile) - at 9 tiles per wave hipcc spills them.
        // (not in the chain: inside the block loop hipcc spills these 40 registers over conv2 whatever is done; there the
        // residual - this workgroup's own output of the previous block - is re-read from memory under the epilogue)
        constexpr bool RES_LDS = PT2 * 8 <= 40 && !CHAIN;
        u32x4 rpre[RES_LDS ? PT2 : 1][2];
        if constexpr (RES_LDS) {
            if (p.res_is_x) {
                // (opaque zero in the address: in the chain hipcc would otherwise forward the previous block's epilogue
                // stores to these loads through the block loop - 40 registers alive across conv1, spilled)
                int opr;
                asm volatile("v_mov_b32 %0, 0" : "=v"(opr));
#pragma unroll
                for (int k = 0; k < PT2; ++k) {
                    const int t = min(wave * PT2 + k, NPT2 - 1);
                    const int col = (t % TWT) * 16 + pl + 2 + opr;
                    const int P = (t / TWT + 2) * ROWP + col;
                    const int a0 = P * 128 + ((q ^ swz(P, col)) << 4);
                    rpre[k][0] = *(const u32x4*)(lds + a0);
                    rpre[k][1] = *(const u32x4*)(lds + (a0 ^ 64));
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();           // every wave has finished reading the x tile
        stamp(blk, 3);
        // T pixel (ti, tj) lives where x pixel (ti, tj) lived: boff1[k][0] IS the address of its chunk q, and chunk
        // 4+q is 64 B further (q ^ 4 under the swizzle).  Lanes / tiles past the end of the T region were clamped to
        // its last pixel: they hold the same values and write them to the same place.
        // Two passes (channels 8q.., then 32+8q..): 24 parameter registers live at a time instead of 48.
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
            for (int k = 0; k < PT1; ++k) {
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
                        o[2 * v4 + e] = pack_bf2(v[0], v[1]) & keep;      // channels 32hi + 8q + 4v4 + 2e, +1
                    }
                *(u32x4*)(lds + (boff1[k][0] ^ (hi << 6))) = o;           // chunk q (hi = 0) or 4+q
            }
        }
        };
        if (!(kdbg & 16)) {
            if (c64_slope_kind(V.epi1) <= 1 && !(kdbg & (1 << 20))) twrite(std::integral_constant<int, 1>{});      // (none: slope 1, max(z, z) = z)
            else twrite(std::integral_constant<int, 2>{});
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();           // T complete
        stamp(blk, 4);
        // (the first weight fragments of conv2 were prefetched into afr[0] during step 17: 18 & 1 == 0)
        // ================= phase 3: conv2 from T ==================================================
        int prow[PT2], pcol[PT2];
        int boff2[PT2][3];
        // opaque zero: keeps hipcc from hoisting this phase's address arithmetic above conv1's K loop, where it would
        // be spilled to scratch (the loop runs at the register limit)
        int opaque0;
        asm volatile("v_mov_b32 %0, 0" : "=v"(opaque0));
#pragma unroll
        for (int k = 0; k < PT2; ++k) {
            const int t = min(wave * PT2 + k, NPT2 - 1);
            prow[k] = t / TWT;
            pcol[k] = (t - prow[k] * TWT) * 16;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int col = pcol[k] + pl + opaque0 + dx;
                const int P = prow[k] * ROWP + col;
                boff2[k][dx] = P * 128 + ((q ^ swz(P, col)) << 4);
            }
            if (wave * PT2 + k >= NPT2) prow[k] = 1 << 20;     // padding tile: the epilogue's bounds check drops it
        }
        u32x4 rres[CHAIN ? PT2 : 1][2];     // chain: residual prefetched under the last K-steps of conv2
        f32x4 acc2[PT2][4];
#pragma unroll
        for (int k = 0; k < PT2; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc2[k][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 bfr2[2][PT2];
#pragma unroll
        for (int k = 0; k < PT2; ++k) bfr2[0][k] = *(const bf16x8*)(lds + boff2[k][0]);
        if (!(kdbg & 8))
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            const int g = 18 + j;
            if (g + 1 < 36) {
                // chain: the residual (this workgroup's own output of the previous block, or the chain input) is
                // fetched under the last K-steps, right after the last weight step went out - always 2*PT2 loads per
                // wave (clamped addresses), because the hand-offs below count them
                handoff(g);
                if constexpr (CHAIN) {
                    if ((g + 1) % KS == 0 && (g + 1) / KS == S_LAST_ISSUE) {
#pragma unroll
                        for (int k = 0; k < PT2; ++k) {
                            const int oy = min(y0 + min(prow[k], TH - 1), p.H - 1), ox = min(x0 + pcol[k] + pl, p.W - 1);
                            const bf16_t* rp = V.x + (((size_t)n * p.H + oy) * p.W) * 64 + (unsigned)(ox * 64 + 8 * q);
                            rres[k][0] = *(const u32x4*)rp;
                            rres[k][1] = *(const u32x4*)(rp + 32);
                        }
                    }
                }
                const unsigned char* wslot = frag(g + 1);
#pragma unroll
                for (int t = 0; t < 4; ++t) afr[(j + 1) & 1][t] = *(const bf16x8*)(wslot + t * 1024);
                const int tapn = (j + 1) >> 1, halfn = (j + 1) & 1;
                const int dyn = tapn / 3, dxn = tapn - 3 * dyn;
#pragma unroll
                for (int k = 0; k < PT2; ++k)
                    bfr2[(j + 1) & 1][k] =
                        *(const bf16x8*)(lds + ((boff2[k][dxn] ^ (halfn << 6)) + dyn * (ROWP * 128)));
            }
#pragma unroll
            for (int k = 0; k < PT2; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc2[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[j & 1][t], bfr2[j & 1][k], acc2[k][t], 0, 0, 0);
            if (j + 1 < 18) sched_interleave<0, 4 + PT2, 4 * PT2>();
        }
        if constexpr (CHAIN) {
            const C64Epi& o = V.epi2;
            __builtin_amdgcn_s_barrier();       // every wave is done reading T: the next block's x tile goes there
            stamp(blk, 5);
            f32x2 sc2[8], sh2[8], sl2[8];
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
                        sc2[4 * hi + 2 * v4 + e] = (f32x2){a[2 * e], a[2 * e + 1]};
                        sh2[4 * hi + 2 * v4 + e] = (f32x2){bsh[2 * e], bsh[2 * e + 1]};
                        sl2[4 * hi + 2 * v4 + e] = (f32x2){csl[2 * e], csl[2 * e + 1]};
                    }
                }
            auto store_tiles = [&](auto SLOPE) {
            int opq;
            asm volatile("v_mov_b32 %0, 0" : "=v"(opq));
            // pass 0: the tiles of the two top and two bottom rows - what the vertical neighbours' halos need; then the flag goes
            // out; pass 1: the centre rows, whose stores drain under the next block
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int k = 0; k < PT2; ++k) {
                if (prow[k] >= TH) continue;                   // padding tile of the last wave (wave-uniform)
                // (with several tile columns the side neighbours need the end columns of the centre rows too: all in pass 0)
                if ((p.tiles_x > 1 || prow[k] < 2 || prow[k] >= TH - 2) != (pass == 0)) continue;
                const int oy = y0 + prow[k], ox = x0 + pcol[k] + pl + opq;
                const bool ok = oy < p.H && ox < p.W;
                u32x4 o0, o1;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int j = 2 * t + e;
                        const f32x2 a = {acc2[k][t][2 * e], acc2[k][t][2 * e + 1]};
                        f32x2 v;
                        if constexpr (decltype(SLOPE)::value) v = slope_act2(a, sc2[j], sh2[j], sl2[j]);
                        else v = __builtin_elementwise_fma(a, sc2[j], sh2[j]);
                        const unsigned rw = rres[k][j >> 2][j & 3];
                        v += (f32x2){__builtin_bit_cast(float, rw << 16), __builtin_bit_cast(float, rw & 0xffff0000u)};
                        const unsigned w = ok ? pack_bf2(v[0], v[1]) : 0u;      // outside the image: zeros in the next x tile
                        if (t < 2) o0[j & 3] = w; else o1[j & 3] = w;
                    }
                if (ok && !(p.dbg & 32768)) {      // agent-scope stores: the neighbours (other CUs, maybe other XCDs) read them in this launch
                    bf16_t* op = o.y + (((size_t)n * p.H + oy) * p.W) * 64 + (unsigned)(ox * 64 + 8 * q);
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(op), "v"(o0) : "memory");
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(op + 32), "v"(o1) : "memory");
                }
                const int col = pcol[k] + pl + opq + 2;
                const int P = (prow[k] + 2) * ROWP + col;
                const int a0 = P * 128 + ((q ^ swz(P, col)) << 4);
                *(u32x4*)(lds + a0) = o0;
                *(u32x4*)(lds + (a0 ^ 64)) = o1;
            }
            if (pass == 0 && blk + 1 < nblk) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's stores have completed at agent scope
                __builtin_amdgcn_s_barrier();                                 // ... and every wave's
                if (tid == 0)
                    __hip_atomic_store(p.flags + tile_id, p.target0 + (unsigned)blk + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                stamp(blk, 6);
            }
            }
            };
            if (o.act == SRMI_ACT_NONE) store_tiles(std::false_type{});
            else store_tiles(std::true_type{});
            if (blk + 1 < nblk) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();         // the next block's x tile centre is complete in LDS; `par` may be rewritten
                stamp(blk, 7);
            }
        } else {
        if (!(kdbg & 4)) {
            if constexpr (RES_LDS) {
                if (p.res_is_x) c64_epilogue<PT2, true>(acc2, par + 192, V.epi2, 0, n, y0, x0, prow, pcol, pl, q, rpre);
                else c64_epilogue<PT2>(acc2, par + 192, V.epi2, 0, n, y0, x0, prow, pcol, pl, q);
            } else {
                c64_epilogue<PT2>(acc2, par + 192, V.epi2, 0, n, y0, x0, prow, pcol, pl, q);
            }
        }
        }
    }
    }   // blocks
}

bool resblock_supported(const srmi_conv_desc& d1, const srmi_conv_desc& d2) {
    if (!c64_supported(d1) || !c64_supported(d2)) return false;
    if (d1.Cout != 64 || d2.Cout != 64 || d1.shuffle != 1 || d2.shuffle != 1) return false;
    if (d1.N != d2.N || d1.H != d2.H || d1.W != d2.W) return false;
    if (d1.out_cstride && d1.out_cstride != 64) return false;   // T must be a plain 64-channel tensor
    if (d1.out_coffset || (d2.in_cstride && d2.in_cstride != 64)) return false;
    // the staging addresses one image through a 32-bit buffer descriptor with two out-of-range markers
    if ((unsigned long long)d1.H * d1.W * (d1.in_cstride ? d1.in_cstride : 64) * 2 > (1ull << 30)) return false;
    return true;
}

int resblock_launch(const srmi_conv_args& a1, const srmi_conv_args& a2, hipStream_t s) {
    const srmi_conv_desc& d1 = a1.d;
    const srmi_conv_desc& d2 = a2.d;
    if (((uintptr_t)a1.x | (uintptr_t)a1.w | (uintptr_t)a2.w | (uintptr_t)a2.y | (uintptr_t)a2.residual) & 15) {
        set_error("resblock_bf16: buffers must be 16-byte aligned");
        return SRMI_EINVAL;
    }
    if (a1.residual) { set_error("resblock_bf16: conv1 cannot take a residual"); return SRMI_EUNSUPPORTED; }
    if (a2.residual && d2.res_dtype != SRMI_BF16) { set_error("resblock_bf16: residual must be bf16"); return SRMI_EUNSUPPORTED; }
    if (a2.y == a1.x) {
        set_error("resblock_bf16: output must not alias the input (neighbouring tiles read its halo)");
        return SRMI_EINVAL;
    }
    ResBlockParams p;
    p.v.x = (const bf16_t*)a1.x; p.v.w1 = (const bf16_t*)a1.w; p.v.w2 = (const bf16_t*)a2.w;
    p.v.epi1.scale = a1.scale; p.v.epi1.shift = a1.shift; p.v.epi1.alpha = a1.alpha; p.v.epi1.res = nullptr; p.v.epi1.y = nullptr;
    p.v.epi1.H = d1.H; p.v.epi1.W = d1.W; p.v.epi1.Cout = 64; p.v.epi1.out_cs = 64; p.v.epi1.out_co = 0; p.v.epi1.mode = C64_PLAIN;
    p.v.epi1.act = d1.act; p.v.epi1.act_param = d1.act_param; p.v.epi1.alpha_pc = d1.alpha_per_channel;
    p.v.epi2.scale = a2.scale; p.v.epi2.shift = a2.shift; p.v.epi2.alpha = a2.alpha;
    p.v.epi2.res = (const bf16_t*)a2.residual; p.v.epi2.y = (bf16_t*)a2.y;
    p.v.epi2.H = d2.H; p.v.epi2.W = d2.W; p.v.epi2.Cout = 64;
    p.v.epi2.out_cs = d2.out_cstride ? d2.out_cstride : 64; p.v.epi2.out_co = d2.out_coffset; p.v.epi2.mode = C64_PLAIN;
    p.v.epi2.act = d2.act; p.v.epi2.act_param = d2.act_param; p.v.epi2.alpha_pc = d2.alpha_per_channel;
    p.N = d1.N; p.H = d1.H; p.W = d1.W; p.in_cs = d1.in_cstride ? d1.in_cstride : 64;
    p.dbg = dbg_flags();
    if (p.dbg & 512) p.v.epi2.res = nullptr;     // ablation: no residual read
    p.res_is_x = p.v.epi2.res != nullptr && (const void*)p.v.epi2.res == (const void*)p.v.x && p.in_cs == 64 && !(p.dbg & 2048);
    if (p.dbg & 1024) p.v.epi2.H = 0;            // ablation: epilogue arithmetic only (every pixel fails the bounds check: no loads, no stores)
    if (d1.N == 0 || d1.H == 0 || d1.W == 0) return SRMI_OK;
    const bool big = d1.W > 64 && d1.H >= 6;
    // Tile shape of the big variant: 6 x 96 (conv1 region 8 x 98 = 49 pixel tiles -> 56 slots on 8 waves, halo tile 10 x 100)
    // or 12 x 48 (region 14 x 50 = 44 tiles -> 48 slots, halo tile 16 x 52: 14 % fewer conv-1 MFMAs and 17 % less staging
    // for the same 576 output pixels).  Cost = rounds of 256 workgroups x MFMA slots per workgroup (conv1 + conv2).
    const long long nb96 = (long long)d1.N * ((d1.W + 95) / 96) * ((d1.H + 5) / 6);
    const long long nb48 = (long long)d1.N * ((d1.W + 47) / 48) * ((d1.H + 11) / 12);
    const bool t12 = big && d1.H >= 12 && !(p.dbg & (1 << 23)) &&
                     ((nb48 + 255) / 256) * (48 + 40) <= ((nb96 + 255) / 256) * (56 + 40);
    if (big && (p.dbg & 4096)) {        // 6 x 48 tiles, two workgroups per CU
        p.tiles_x = (d1.W + 47) / 48; p.tiles_y = (d1.H + 5) / 6;
        const long long nb = (long long)d1.N * p.tiles_x * p.tiles_y;
        if (nb > 0x7fffffffLL) { set_error("resblock_bf16: grid too large"); return SRMI_EINVAL; }
        hipLaunchKernelGGL((resblock_c64_kernel<6, 3, 4, 3, 2, false, 1, 2>), dim3((unsigned)nb), dim3(256), 0, s, p);
    } else if (t12) {
        p.tiles_x = (d1.W + 47) / 48; p.tiles_y = (d1.H + 11) / 12;
        if (nb48 > 0x7fffffffLL) { set_error("resblock_bf16: grid too large"); return SRMI_EINVAL; }
        hipLaunchKernelGGL((resblock_c64_kernel<12, 3, 8, 4, 1, false, 1>), dim3((unsigned)nb48), dim3(512), 0, s, p);
    } else if (big) {
        p.tiles_x = (d1.W + 95) / 96; p.tiles_y = (d1.H + 5) / 6;
        const long long nb = (long long)d1.N * p.tiles_x * p.tiles_y;
        if (nb > 0x7fffffffLL) { set_error("resblock_bf16: grid too large"); return SRMI_EINVAL; }
        if (p.dbg & 128) hipLaunchKernelGGL((resblock_c64_kernel<6, 6, 4, 4, 1, false>), dim3((unsigned)nb), dim3(256), 0, s, p);
        // (measured and not adopted, same box: tap-sized ring slots = half the hand-off barriers need the unpadded pitch to
        //  fit LDS: KS=2/PITCH=2 31.1 us, KS=2/PITCH=8/3 slots 31.0, KS=1/PITCH=8 31.2, KS=1/PITCH=2 31.4 vs 30.6 for this one)
        else hipLaunchKernelGGL((resblock_c64_kernel<6, 6, 8, 4, 1, false, 1>), dim3((unsigned)nb), dim3(512), 0, s, p);
    } else {
        p.tiles_x = (d1.W + 31) / 32; p.tiles_y = (d1.H + 3) / 4;
        const long long nb = (long long)d1.N * p.tiles_x * p.tiles_y;
        if (nb > 0x7fffffffLL) { set_error("resblock_bf16: grid too large"); return SRMI_EINVAL; }
        hipLaunchKernelGGL((resblock_c64_kernel<4, 2, 4, 4, 1, false>), dim3((unsigned)nb), dim3(256), 0, s, p);
    }
    return hip_check(hipGetLastError(), "resblock_c64_kernel");
}

// ---- persistent chain -----------------------------------------------------------------------------------------------
// Flag words live in a file-scope device array: RB_FLAG_SETS sets, used round-robin by successive launches, values grow
// monotonically (epoch * 64 + blocks published), so nothing has to be cleared between launches.
constexpr int RB_FLAG_SETS = 8, RB_FLAG_TILES = 1024;
static __device__ unsigned g_chain_flags[RB_FLAG_SETS * RB_FLAG_TILES];
static __device__ unsigned g_chain_timeouts;
static __device__ unsigned long long g_chain_trace[RB_FLAG_TILES * 8];

namespace {
std::atomic<int> g_chain_enabled{-1};     // -1: undecided = the SRMI_CHAIN environment variable, else off; see srmi_chain_enable
int chain_enabled() {
    int v = g_chain_enabled.load();
    if (v < 0) {
        const char* e = getenv("SRMI_CHAIN");
        v = (e && atoi(e) != 0) ? 1 : 0;
        g_chain_enabled.store(v);
    }
    return v;
}
struct ChainDev { unsigned* flags = nullptr; unsigned* timeouts = nullptr; unsigned long long* trace = nullptr; int cus = 0; unsigned epoch = 0; bool tried = false; };
ChainDev g_chain_dev[16];
std::mutex g_chain_mutex;

ChainDev* chain_dev() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    ChainDev& c = g_chain_dev[dev];
    if (!c.tried) {
        c.tried = true;
        void* f = nullptr; void* t = nullptr;
        hipDeviceProp_t prop;
        if (hipGetSymbolAddress(&f, HIP_SYMBOL(g_chain_flags)) == hipSuccess &&
            hipGetSymbolAddress(&t, HIP_SYMBOL(g_chain_timeouts)) == hipSuccess &&
            hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            c.flags = (unsigned*)f; c.timeouts = (unsigned*)t; c.cus = prop.multiProcessorCount;
            void* tr = nullptr;
            if (hipGetSymbolAddress(&tr, HIP_SYMBOL(g_chain_trace)) == hipSuccess) c.trace = (unsigned long long*)tr;
        }
    }
    return c.flags ? &c : nullptr;
}
}  // namespace

// How many of the `npairs` fusable (conv1, conv2) pairs starting at a[0] can run as ONE persistent launch (0: use the
// per-block launches).  Conditions: same shape, every block's residual is its own input, block i+1 reads what block i
// wrote, dense 64-channel tensors, the big tile, and every tile resident at once.
bool resblock_chain_on() { return chain_enabled() != 0; }

int resblock_chain_length(const srmi_conv_args* a, int npairs) {
    if (npairs < 2 || !chain_enabled()) return 0;
    const srmi_conv_desc& d = a[0].d;
    if (!(d.W > 64 && d.H >= 6)) return 0;
    std::lock_guard<std::mutex> lock(g_chain_mutex);
    ChainDev* c = chain_dev();
    if (!c) return 0;
    const long long nb = (long long)d.N * ((d.W + 95) / 96) * ((d.H + 5) / 6);
    if (nb > c->cus || nb > RB_FLAG_TILES) return 0;
    int m = 0;
    for (; m < npairs && m < RB_CHAIN_MAX; ++m) {
        const srmi_conv_args& a1 = a[2 * m];
        const srmi_conv_args& a2 = a[2 * m + 1];
        if (a1.d.N != d.N || a1.d.H != d.H || a1.d.W != d.W) break;
        if ((a1.d.in_cstride && a1.d.in_cstride != 64) || (a2.d.out_cstride && a2.d.out_cstride != 64) || a2.d.out_coffset) break;
        if (!a2.residual || a2.residual != a1.x || a2.d.res_dtype != SRMI_BF16) break;
        if (m > 0 && a1.x != a[2 * m - 1].y) break;
        if (a2.y == a1.x) break;
    }
    return m >= 2 ? m : 0;
}

int resblock_chain_launch(const srmi_conv_args* a, int nblocks, hipStream_t s) {
    ResChainParams p;
    const srmi_conv_desc& d = a[0].d;
    for (int i = 0; i < nblocks; ++i) {
        const srmi_conv_args& a1 = a[2 * i];
        const srmi_conv_args& a2 = a[2 * i + 1];
        if (((uintptr_t)a1.x | (uintptr_t)a1.w | (uintptr_t)a2.w | (uintptr_t)a2.y) & 15) {
            set_error("resblock chain: buffers must be 16-byte aligned");
            return SRMI_EINVAL;
        }
        ResBlockVar& v = p.blk[i];
        v.x = (const bf16_t*)a1.x; v.w1 = (const bf16_t*)a1.w; v.w2 = (const bf16_t*)a2.w;
        v.epi1.scale = a1.scale; v.epi1.shift = a1.shift; v.epi1.alpha = a1.alpha; v.epi1.res = nullptr; v.epi1.y = nullptr;
        v.epi1.H = d.H; v.epi1.W = d.W; v.epi1.Cout = 64; v.epi1.out_cs = 64; v.epi1.out_co = 0; v.epi1.mode = C64_PLAIN;
        v.epi1.act = a1.d.act; v.epi1.act_param = a1.d.act_param; v.epi1.alpha_pc = a1.d.alpha_per_channel;
        v.epi2.scale = a2.scale; v.epi2.shift = a2.shift; v.epi2.alpha = a2.alpha;
        v.epi2.res = (const bf16_t*)a2.residual; v.epi2.y = (bf16_t*)a2.y;
        v.epi2.H = d.H; v.epi2.W = d.W; v.epi2.Cout = 64; v.epi2.out_cs = 64; v.epi2.out_co = 0; v.epi2.mode = C64_PLAIN;
        v.epi2.act = a2.d.act; v.epi2.act_param = a2.d.act_param; v.epi2.alpha_pc = a2.d.alpha_per_channel;
    }
    p.nblocks = nblocks;
    p.N = d.N; p.H = d.H; p.W = d.W; p.in_cs = 64;
    p.tiles_x = (d.W + 95) / 96; p.tiles_y = (d.H + 5) / 6;
    p.res_is_x = 1;
    p.dbg = dbg_flags() & (32 | 32768 | 65536);      // tile-order switch; timing experiments: 32768 no output stores, 65536 no halo loads
    const unsigned nb = (unsigned)d.N * p.tiles_x * p.tiles_y;
    {
        std::lock_guard<std::mutex> lock(g_chain_mutex);
        ChainDev* c = chain_dev();
        if (!c) { set_error("resblock chain: no flag memory on this device"); return SRMI_EHIP; }
        c->epoch += 1;
        p.flags = c->flags + (c->epoch % RB_FLAG_SETS) * RB_FLAG_TILES;
        p.timeouts = c->timeouts;
        p.trace = (dbg_flags() & 16384) ? c->trace : nullptr;
        p.target0 = c->epoch * 64u;
    }
    hipLaunchKernelGGL((resblock_c64_kernel<6, 6, 8, 4, 1, true, 1>), dim3(nb), dim3(512), 0, s, p);
    return hip_check(hipGetLastError(), "resblock_c64_kernel (chain)");
}

int resblock_chain_trace(unsigned long long* out, int n) {
    std::lock_guard<std::mutex> lock(g_chain_mutex);
    ChainDev* c = chain_dev();
    if (!c || !c->trace || n > RB_FLAG_TILES * 8) return SRMI_EINVAL;
    return hip_check(hipMemcpy(out, c->trace, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost), "chain trace");
}

int resblock_chain_enable(int on) {
    const int prev = chain_enabled();
    g_chain_enabled.store(on ? 1 : 0);
    return prev;
}

// number of neighbour waits that gave up since the library was loaded (synchronises the device; tests)
int resblock_chain_timeouts(unsigned* count) {
    std::lock_guard<std::mutex> lock(g_chain_mutex);
    ChainDev* c = chain_dev();
    *count = 0;
    if (!c) return SRMI_OK;
    return hip_check(hipMemcpy(count, c->timeouts, sizeof(unsigned), hipMemcpyDeviceToHost), "chain timeouts");
}

}  // namespace srmi
