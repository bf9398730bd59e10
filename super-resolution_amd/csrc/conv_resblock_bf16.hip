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
    int dbg;  // ablation (SRMI_DBG env, ablation build only): 1 skip staging, 2 skip conv1 K loop, 8 skip conv2 K loop, 4 skip epilogue, 16 skip T write
    int exp;  // scheduling experiments (SRMI_EXP env, ablation build only; results stay correct)
#ifdef SRMI_ABLATE
    unsigned long long* stamps;   // per launch slot: [workgroup][8] s_memrealtime stamps (100 MHz), or nullptr
#endif
};

#ifdef SRMI_ABLATE
#define RB_STAMP(i) do { if (p.stamps) { st[i] = __builtin_amdgcn_s_memrealtime(); ck[i] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define RB_STAMP(i) do { } while (0)
#endif

constexpr int RB_STEP_BYTES = 4096;   // one K-step: 4 channel tiles x 64 lanes x 16 B

// NW = waves per workgroup: 4 (one per SIMD, 512 registers each) or 8 (two per SIMD, 256 each: the second wave
// of a SIMD fills the first one's LDS / barrier / VALU-epilogue bubbles; same tile, same LDS, half the pixel
// tiles per wave).
//
// SLOTS = depth of the weight ring (K-steps), WGPC = workgroups per CU the register budget is cut for.  The 6 x 48 tile
// with 4 waves and a 3-slot ring needs 78.5 KiB of LDS and 256 registers: TWO workgroups share a CU, and while one
// sits in a phase that leaves the matrix cores idle (launch, staging, T write-back, epilogue, store drain) the other
// one computes - provided they are out of phase, which two streams (or several rounds) give.
template <int TH, int TWT, int NW, int SLOTS, int WGPC, int KS = 1, int PITCH = 16>
__global__ __launch_bounds__(NW * 64, WGPC) void resblock_c64_kernel(ResBlockParams p) {
    constexpr int NTH = NW * 64;
    constexpr int TW = TWT * 16;
    constexpr int XR = TH + 4, XC = TW + 4;          // x halo tile
    constexpr int TR = TH + 2, TC = TW + 2;          // T region
    constexpr int ROWP = (XC + PITCH - 1) / PITCH * PITCH;   // PITCH: 16 (padded), 8, or 2 (= XC, unpadded)   // LDS row pitch in pixels: even, so a row starts on a 256-B bank row and
                                                     // the swizzle key (col >> 1) & 7 is the same for every row (tap dy = +pitch)
    static_assert(ROWP % 2 == 0, "pixel pairs must not straddle rows");
    // swizzle key of LDS pixel P = row * ROWP + col (the P form keeps fewer values alive when the pitch allows it)
    auto swz = [](int P, int col) { return ROWP % 8 == 0 ? c64_key(P) : c64_key(col); };
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
#ifdef SRMI_ABLATE
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ck[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    RB_STAMP(0);

    // XCD-aware tile order: consecutive workgroup ids go round-robin to the 8 XCDs (one L2 each), so give each
    // XCD a CONTIGUOUS run of tiles - vertically adjacent tiles then share their halo rows through one L2.
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0 && !(p.dbg & 32)) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int kdbg = p.dbg;                   // phase-ablation switches (SRMI_DBG, investigation only)
    {   // SRMI_EXP (ablation build): de-phase the two workgroups that share a CU - one gets instruction priority, the other
        // may start its loads late
        const int ex = p.exp;
        if (ex & 6) {
            const bool low = (ex & 2) ? blockIdx.x >= (gridDim.x >> 1) : ((blockIdx.x >> 3) & 1) != 0;
            if (low) {
                __builtin_amdgcn_s_setprio(0);
                if (ex & 8) __builtin_amdgcn_s_sleep(40);
                if (ex & 16) __builtin_amdgcn_s_sleep(80);
            } else {
                __builtin_amdgcn_s_setprio(3);
            }
        }
    }
    const ResBlockVar& V = p.v;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int pl = lane & 15, q = lane >> 4;

    // (pinned in scalar registers: the pointers are not re-loaded from the kernel arguments inside the K loops)
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
        const unsigned row_bytes = (unsigned)p.W * p.in_cs * 2;
        const int ch = tid & 7, c0 = tid >> 3;
        u32x4 va[NA > 0 ? NA * XR : 1], vb[NB > 0 ? NB : 1];
        unsigned la[NA > 0 ? NA : 1], lb[NB > 0 ? NB : 1];
#pragma unroll
        for (int g = 0; g < NA; ++g) {
            const int col = g * CPT + c0, gx = x0 - 2 + col;
            const unsigned coff = (gx >= 0 && gx < p.W) ? (unsigned)gx * p.in_cs * 2 + ch * 16 : OOB_COL;
            la[g] = col * 128 + ((ch ^ c64_key(col)) << 4);
#pragma unroll
            for (int row = 0; row < XR; ++row) {
                const int gy = y0 - 2 + row;                               // wave-uniform
                const unsigned roff = (gy >= 0 && gy < p.H) ? (unsigned)gy * row_bytes : OOB_ROW;
                va[g * XR + row] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, coff + roff, 0, 0));
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int sidx = tid + i * NTH;
            const int row = sidx / (RB * 8), rem = sidx - row * (RB * 8);
            const int col = NA * CPT + (rem >> 3), chb = rem & 7;
            const int gy = y0 - 2 + row, gx = x0 - 2 + col;
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
    RB_STAMP(1);
    c64_params_store(par, pregs1, 1, tid);
    c64_params_store(par + 192, pregs2, 1, tid);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // also retires weight steps 0..2
    RB_STAMP(2);
    __builtin_amdgcn_s_barrier();
    RB_STAMP(3);

    // one K-step hand-off: step g+1 landed (own copy: counted vmcnt, others': barrier); slot of step g-1
    // is free (its fragments were consumed by the MFMAs of step g-1, which every wave has issued).
    // In slot units: called at the top of step g when step g+1 opens slot S = (g+1)/KS.  Slots up to S+SLOTS-3 are in
    // flight; all but S may stay so.  Then slot S+SLOTS-2 is issued into the ring position of slot S-2, whose fragments
    // were consumed by MFMAs every wave has issued.
    auto handoff = [&](int g) {
        if ((g + 1) % KS != 0) return;
        const int S = (g + 1) / KS;
        // slots S+1 .. S+SLOTS-3 (those that exist) may stay in flight
        const int ahead = (SLOTS - 3) < (NS - 1 - S) ? (SLOTS - 3) : (NS - 1 - S);
        if (!(kdbg & (1 << 17))) {       // (timing experiment: no wait for the weights)
            if (SLOTS >= 5 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * 2) : "memory");
            else if (SLOTS >= 4 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!(kdbg & (1 << 18))) __builtin_amdgcn_s_barrier();      // (timing experiment: no hand-off barrier)
        if (S + SLOTS - 2 < NS && !(kdbg & (1 << 19))) issue_slot(S + SLOTS - 2);     // (timing experiment: no weight stream)
    };
    // LDS address of the fragments of K-step g
    auto frag = [&](int g) { return wring + ((g / KS) % SLOTS) * SLOT_BYTES + (g % KS) * RB_STEP_BYTES + lane * 16; };

    // Tiles per wave.  With two waves per SIMD (waves w and w + NW/2 share one) the SIMD's MFMA time per K-step is the SUM of
    // its two waves' tile counts, so the two halves of the workgroup may take different counts: 44 conv-1 tiles = 4 x (6 + 5)
    // and 36 conv-2 tiles = 4 x (5 + 4) fill every slot, where 8 x 6 and 8 x 5 left 4 + 4 of 88 slots recomputing a
    // neighbour's tile (9 % of the MFMAs - and at the 1.69 GHz the matrix cores are power-limited to, wasted MFMAs also
    // cost clock).  Two instantiations of the phases, chosen by a wave-uniform branch; both execute the same barriers.
    constexpr bool SPLIT1 = NW == 8 && NT1 <= (NW / 2) * (2 * PT1 - 1);
    constexpr bool SPLIT2 = NW == 8 && NPT2 <= (NW / 2) * (2 * PT2 - 1);
    constexpr int PT1B = SPLIT1 ? PT1 - 1 : PT1, PT2B = SPLIT2 ? PT2 - 1 : PT2;
    auto body = [&](auto P1, auto P2, const int t1base, const int t2base) {
    constexpr int PT1c = decltype(P1)::value, PT2c = decltype(P2)::value;
    // ================= phase 1: conv1 over the flattened T region =================================
    int boff1[PT1c][3];
    unsigned inside = 0;     // bit k: this lane's pixel of tile k lies inside the image (else T = 0: conv2's zero padding)
#pragma unroll
    for (int k = 0; k < PT1c; ++k) {
        const int pix = min((t1base + k) * 16 + pl, TR * TC - 1);   // padding tiles recompute the last pixel
        const int ti = pix / TC, tj = pix - ti * TC;
        const int gy = y0 - 1 + ti, gx = x0 - 1 + tj;
        inside |= (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? 1u << k : 0u;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int P = ti * ROWP + tj + dx;
            boff1[k][dx] = P * 128 + ((q ^ swz(P, tj + dx)) << 4);
        }
    }
    asm volatile("" : "+v"(inside));      // materialise the mask HERE (hipcc would sink it below the K loop and keep its 2*PT1c inputs alive)
    {
        f32x4 acc[PT1c][4];
#pragma unroll
        for (int k = 0; k < PT1c; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[k][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 afr[2][4], bfr[2][PT1c];
#pragma unroll
        for (int t = 0; t < 4; ++t) afr[0][t] = *(const bf16x8*)(wring + lane * 16 + t * 1024);
#pragma unroll
        for (int k = 0; k < PT1c; ++k) bfr[0][k] = *(const bf16x8*)(lds + boff1[k][0]);
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

        RB_STAMP(4);
        // ---- phase 2: T = act(affine1(acc)) as bf16 over the x tile (zero outside the image) ----
        // The residual IS the centre of the x tile: take it from LDS now (the operand fragments are dead, T is about
        // to overwrite the tile) instead of re-reading it from L2 in the epilogue.  PT2c * 8 registers, kept across
        // conv2; worth it while they fit (8-wave variant, small tile) - at 9 tiles per wave hipcc spills them.
        constexpr bool RES_LDS = PT2c * 8 <= 40;
        u32x4 rpre[RES_LDS ? PT2c : 1][2];
        if constexpr (RES_LDS) {
            if (p.res_is_x) {
                int opr;      // (opaque zero in the address: keeps these loads where they are written)
                asm volatile("v_mov_b32 %0, 0" : "=v"(opr));
#pragma unroll
                for (int k = 0; k < PT2c; ++k) {
                    const int t = min(t2base + k, NPT2 - 1);
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
        RB_STAMP(5);
        // (the first weight fragments of conv2 were prefetched into afr[0] during step 17: 18 & 1 == 0)
        // ================= phase 3: conv2 from T ==================================================
        int prow[PT2c], pcol[PT2c];
        int boff2[PT2c][3];
        // opaque zero: keeps hipcc from hoisting this phase's address arithmetic above conv1's K loop, where it would
        // be spilled to scratch (the loop runs at the register limit)
        int opaque0;
        asm volatile("v_mov_b32 %0, 0" : "=v"(opaque0));
#pragma unroll
        for (int k = 0; k < PT2c; ++k) {
            const int t = min(t2base + k, NPT2 - 1);
            prow[k] = t / TWT;
            pcol[k] = (t - prow[k] * TWT) * 16;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int col = pcol[k] + pl + opaque0 + dx;
                const int P = prow[k] * ROWP + col;
                boff2[k][dx] = P * 128 + ((q ^ swz(P, col)) << 4);
            }
            if (t2base + k >= NPT2) prow[k] = 1 << 20;     // padding tile: the epilogue's bounds check drops it
        }
        f32x4 acc2[PT2c][4];
#pragma unroll
        for (int k = 0; k < PT2c; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc2[k][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 bfr2[2][PT2c];
#pragma unroll
        for (int k = 0; k < PT2c; ++k) bfr2[0][k] = *(const bf16x8*)(lds + boff2[k][0]);
        if (!(kdbg & 8))
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            const int g = 18 + j;
            if (g + 1 < 36) {
                handoff(g);
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
        RB_STAMP(6);
        if (!(kdbg & 4)) {
            if constexpr (RES_LDS) {
                if (p.res_is_x) c64_epilogue<PT2c, true>(acc2, par + 192, V.epi2, 0, n, y0, x0, prow, pcol, pl, q, rpre);
                else c64_epilogue<PT2c>(acc2, par + 192, V.epi2, 0, n, y0, x0, prow, pcol, pl, q);
            } else {
                c64_epilogue<PT2c>(acc2, par + 192, V.epi2, 0, n, y0, x0, prow, pcol, pl, q);
            }
        }
#ifdef SRMI_ABLATE
        if (p.stamps) {
            st[7] = __builtin_amdgcn_s_memrealtime(); ck[7] = __builtin_amdgcn_s_memtime();
            if (lane == 0 && wave == 0) {      // row 0: real time (100 MHz), row 1: shader clock counter
                unsigned long long* d = p.stamps + (size_t)blockIdx.x * 16;
#pragma unroll
                for (int i = 0; i < 8; ++i) { d[i] = st[i]; d[8 + i] = ck[i]; }
            }
        }
#endif
    }
    };
    if constexpr (SPLIT1 || SPLIT2) {
        if (wave < NW / 2) body(std::integral_constant<int, PT1>{}, std::integral_constant<int, PT2>{}, wave * PT1, wave * PT2);
        else body(std::integral_constant<int, PT1B>{}, std::integral_constant<int, PT2B>{},
                  (NW / 2) * PT1 + (wave - NW / 2) * PT1B, (NW / 2) * PT2 + (wave - NW / 2) * PT2B);
    } else {
        body(std::integral_constant<int, PT1>{}, std::integral_constant<int, PT2>{}, wave * PT1, wave * PT2);
    }
}

#ifdef SRMI_ABLATE
// stamp buffer of the ablation build: SRMI_STAMPS=<launches> allocates [launches][1024 workgroups][2 waves][8] u64 at the first
// launch; srmi_dbg_resblock_stamps copies it to the host.
static unsigned long long* g_stamps = nullptr;
static int g_stamp_slots = 0, g_stamp_next = 0;
static unsigned long long* next_stamp_slot(unsigned nb) {
    static const int want = [] { const char* e = getenv("SRMI_STAMPS"); return e ? atoi(e) : 0; }();
    if (!want || nb > 1024) return nullptr;
    if (!g_stamps) {
        if (hipMalloc((void**)&g_stamps, (size_t)want * 1024 * 16 * 8) != hipSuccess) return nullptr;
        (void)hipMemset(g_stamps, 0, (size_t)want * 1024 * 16 * 8);
        g_stamp_slots = want;
    }
    if (g_stamp_next >= g_stamp_slots) return nullptr;
    return g_stamps + (size_t)(g_stamp_next++) * 1024 * 16;
}
extern "C" int srmi_dbg_resblock_stamps(void* host_dst, size_t bytes, int* launches, int reset) {
    if (launches) *launches = g_stamp_next;
    if (g_stamps && host_dst) (void)hipMemcpy(host_dst, g_stamps, bytes, hipMemcpyDeviceToHost);
    if (reset) g_stamp_next = 0;
    return 0;
}
#endif

bool resblock_supported(const srmi_conv_desc& d1, const srmi_conv_desc& d2) {
    if (!c64_supported(d1) || !c64_supported(d2)) return false;
    if (d1.Cout != 64 || d2.Cout != 64 || d1.shuffle != 1 || d2.shuffle != 1) return false;
    if (d1.N != d2.N || d1.H != d2.H || d1.W != d2.W) return false;
    if (d1.out_cstride && d1.out_cstride != 64) return false;   // T must be a plain 64-channel tensor
    if (d1.out_coffset || (d2.in_cstride && d2.in_cstride != 64)) return false;
    if (d2.res_cstride && d2.res_cstride != 64) return false;   // dense residual only
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
    p.exp = exp_flags();
#ifdef SRMI_ABLATE
    p.stamps = nullptr;
#endif
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
    if (big && ((p.dbg & 4096) || (p.exp & 1))) {        // 6 x 48 tiles, two workgroups per CU
        p.tiles_x = (d1.W + 47) / 48; p.tiles_y = (d1.H + 5) / 6;
        const long long nb = (long long)d1.N * p.tiles_x * p.tiles_y;
        if (nb > 0x7fffffffLL) { set_error("resblock_bf16: grid too large"); return SRMI_EINVAL; }
        hipLaunchKernelGGL((resblock_c64_kernel<6, 3, 4, 3, 2, 1, 2>), dim3((unsigned)nb), dim3(256), 0, s, p);
    } else if (t12) {
        p.tiles_x = (d1.W + 47) / 48; p.tiles_y = (d1.H + 11) / 12;
        if (nb48 > 0x7fffffffLL) { set_error("resblock_bf16: grid too large"); return SRMI_EINVAL; }
        // weight ring: three tap-sized slots (half the hand-off barriers of four K-step slots; the padded pitch still fits
        // at this tile: 157 KB) - 29.4 vs 29.75 us per launch in three alternating runs; SRMI_DBG bit 22: the K-step ring
        // (four tap slots with an 8-pixel pitch, 149 KB: 29.0 vs 28.2 us - the narrower pitch costs more than the deeper ring gains)
        if (p.dbg & (1 << 22)) hipLaunchKernelGGL((resblock_c64_kernel<12, 3, 8, 4, 1, 1>), dim3((unsigned)nb48), dim3(512), 0, s, p);
#ifdef SRMI_ABLATE
        // round 3, same call: 56-pixel pitch with 4 / 5 / 3 tap slots (149-157 KB): 24.8-25.3 us each against 25.2-25.3 for
        // the product form - with the conflict-free swizzle the pitch costs nothing, and the ring depth buys nothing
        else if (p.exp & 32) hipLaunchKernelGGL((resblock_c64_kernel<12, 3, 8, 4, 1, 2, 8>), dim3((unsigned)nb48), dim3(512), 0, s, p);
        else if (p.exp & 64) hipLaunchKernelGGL((resblock_c64_kernel<12, 3, 8, 5, 1, 2, 8>), dim3((unsigned)nb48), dim3(512), 0, s, p);
        else if (p.exp & 128) hipLaunchKernelGGL((resblock_c64_kernel<12, 3, 8, 3, 1, 2, 8>), dim3((unsigned)nb48), dim3(512), 0, s, p);
#endif
        else {
#ifdef SRMI_ABLATE
            p.stamps = next_stamp_slot((unsigned)nb48);
#endif
            hipLaunchKernelGGL((resblock_c64_kernel<12, 3, 8, 3, 1, 2>), dim3((unsigned)nb48), dim3(512), 0, s, p);
        }
    } else if (big) {
        p.tiles_x = (d1.W + 95) / 96; p.tiles_y = (d1.H + 5) / 6;
        const long long nb = (long long)d1.N * p.tiles_x * p.tiles_y;
        if (nb > 0x7fffffffLL) { set_error("resblock_bf16: grid too large"); return SRMI_EINVAL; }
        if (p.dbg & 128) hipLaunchKernelGGL((resblock_c64_kernel<6, 6, 4, 4, 1>), dim3((unsigned)nb), dim3(256), 0, s, p);
        // (measured and not adopted, same box: tap-sized ring slots = half the hand-off barriers need the unpadded pitch to
        //  fit LDS: KS=2/PITCH=2 31.1 us, KS=2/PITCH=8/3 slots 31.0, KS=1/PITCH=8 31.2, KS=1/PITCH=2 31.4 vs 30.6 for this one)
        else hipLaunchKernelGGL((resblock_c64_kernel<6, 6, 8, 4, 1, 1>), dim3((unsigned)nb), dim3(512), 0, s, p);
    } else {
        p.tiles_x = (d1.W + 31) / 32; p.tiles_y = (d1.H + 3) / 4;
        const long long nb = (long long)d1.N * p.tiles_x * p.tiles_y;
        if (nb > 0x7fffffffLL) { set_error("resblock_bf16: grid too large"); return SRMI_EINVAL; }
        hipLaunchKernelGGL((resblock_c64_kernel<4, 2, 4, 4, 1>), dim3((unsigned)nb), dim3(256), 0, s, p);
    }
    return hip_check(hipGetLastError(), "resblock_c64_kernel");
}

}  // namespace srmi
