// f32 convolution on the bf16 matrix cores by operand splitting, gfx950.
//
// An f32 value is the exact sum of three bf16 values up to 2^-27 of its magnitude: hi = bf16(x), mid = bf16(x - hi),
// lo = bf16(x - hi - mid) (both differences are exact in f32).  A product of two such sums, with the terms below
// 2^-24 of it dropped, needs SIX bf16 products accumulated in f32:
//     x * w  ~  hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid        (dropped: mid*lo, lo*mid, lo*lo  <= 3 * 2^-25 |x w|)
// i.e. the rounding of an f32 FMA chain, not of a bf16 conv.  v_mfma_f32_32x32x16_bf16 retires 16 k per 32 cycles against
// 2 k per 64 cycles of v_mfma_f32_32x32x2_f32, so six of them per k are still 2.7x the exact-f32 MFMA rate - the only way
// past the 157 TFLOP/s f32 peak that bounds the SRGAN training step (VGG19 and discriminator convs).
//
// Structure = conv_f32_tiled.hip's lean variant (same GenericParams, prologue, split-K and epilogue):
//   activations  NHWC f32, 16-byte loads, split into the three planes when they are written to LDS;
//   weights      pre-split at pack time: [plane][tap][n = produced channel][k = contracted channel] bf16, k contiguous
//                (b3_pack_kernel), so a weight tile goes global -> LDS without touching the vector ALU;
//   LDS          one buffer of [plane][row][32 k] bf16 with an 80-byte row pitch (ds_read_b128 of 8 k per lane is
//                conflict-free), registers prefetch the next chunk under the MFMAs;
//   a wave       TM x TN tiles of 32 x 32, per 16-k slab 3 (TM + TN) ds_read_b128 and 6 TM TN MFMAs.
// Selected from the descriptor alone (b3_fwd_ok / b3_dgrad_ok: Cin and Cout multiples of 32), because the packed operand
// has a different format; SRMI_F32_EXACT=1 keeps every conv on the exact-f32 MFMA kernel.
#include <stdlib.h>

#include <type_traits>

#include "generic_common.h"

namespace srmi {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// contracted channels per chunk: BK = 32, or 64 for small layers that run one workgroup per CU without a K split (half the
// chunk hand-offs, each of which costs a global-load latency there).  LDS rows are BK + 8 bf16 (80 / 144 bytes: the 16-byte
// operand reads of 16 consecutive rows fall into 16 distinct bank groups either way).

__device__ __forceinline__ int b3_fastdiv(int n, unsigned mul, int sh) {
    return (int)(((unsigned long long)(unsigned)n * mul) >> sh);
}
__device__ __forceinline__ unsigned b3_bit_range(int lo, int hi) { return hi >= lo ? (2u << hi) - (1u << lo) : 0u; }

// x -> (hi, mid, lo) bf16 bits, round to nearest even at every step
__device__ __forceinline__ void b3_split(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    const __bf16 l = (__bf16)r2;
    hi = __builtin_bit_cast(unsigned short, h);
    mid = __builtin_bit_cast(unsigned short, m);
    lo = __builtin_bit_cast(unsigned short, l);
}

// two values at a time: v_cvt_pk_bf16_f32 rounds both and leaves them packed as the LDS word wants them
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void b3_split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{x0, x1}, bf16x2_t));
    const float r0 = x0 - __builtin_bit_cast(float, hi << 16), r1 = x1 - __builtin_bit_cast(float, hi & 0xffff0000u);
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, bf16x2_t));
    const float q0 = r0 - __builtin_bit_cast(float, mid << 16), q1 = r1 - __builtin_bit_cast(float, mid & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{q0, q1}, bf16x2_t));
}

template <int WM, int WN, int TM, int TN, int B3_BK>
__global__ __launch_bounds__(256) void conv_b3_kernel(GenericParams p) {
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int B3_PITCH = B3_BK + 8;
    constexpr int NPIECE = B3_BK / 4;                     // f32x4 pieces per pixel and chunk
    constexpr int A_LOADS = BM * NPIECE / 256;            // ... per thread
    constexpr int BPR = B3_BK / 8;                        // 16-byte pieces per weight row and plane
    constexpr int B_TOTAL = 3 * BN * BPR;                 // ... of the weight tile
    constexpr int B_LOADS = (B_TOTAL + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned short As[3][BM][B3_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[3][BN][B3_PITCH];

    const int tid = threadIdx.x;
#ifdef SRMI_B3_STAMPS
    long long stamp[5];
    stamp[0] = wall_clock64();
#define B3_STAMP(k) stamp[k] = wall_clock64()
#else
#define B3_STAMP(k)
#endif
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM;
    const int co0 = blockIdx.y * BN;

    const int cpt = p.Cin / B3_BK;                        // chunks per tap (Cin % 32 == 0)
    const int KC = p.kh * p.kw * cpt;
    int kc0 = 0, kc1 = KC;
    if (p.ksplit > 1) {
        const int per = (KC + p.ksplit - 1) / p.ksplit;
        kc0 = min(KC, (int)blockIdx.z * per);
        kc1 = min(KC, kc0 + per);
    }

    // this thread's activation pieces: pixel idx / 8, channels 4 * (idx % 8) .. + 3 of the chunk (see conv_f32_tiled.hip)
    int a_img[A_LOADS], a_base[A_LOADS];
    unsigned a_rm[A_LOADS], a_cm[A_LOADS];
#pragma unroll
    for (int l = 0; l < A_LOADS; ++l) {
        const int idx = tid + 256 * l;
        const int m = m0 + idx / NPIECE;
        a_img[l] = 0; a_base[l] = 0; a_rm[l] = 0u; a_cm[l] = 0u;
        if (m < p.M) {
            const int tq = b3_fastdiv(m, p.wo_mul, p.wo_sh);
            const int ox = m - tq * p.Wo;
            const int n = b3_fastdiv(tq, p.ho_mul, p.ho_sh);
            const int oy = tq - n * p.Ho;
            a_img[l] = n * p.H * p.W;
            const int by = p.transposed ? oy + p.pad : oy * p.stride - p.pad;
            const int bx = p.transposed ? ox + p.pad : ox * p.stride - p.pad;
            a_base[l] = by * p.W + bx;
            if (!p.transposed) {
                a_rm[l] = b3_bit_range(max(0, -by), min(p.kh - 1, p.H - 1 - by));
                a_cm[l] = b3_bit_range(max(0, -bx), min(p.kw - 1, p.W - 1 - bx));
            } else {                                      // stride 1 (checked on the host): 0 <= b - k < extent
                a_rm[l] = b3_bit_range(max(0, by - p.H + 1), min(p.kh - 1, by));
                a_cm[l] = b3_bit_range(max(0, bx - p.W + 1), min(p.kw - 1, bx));
            }
        }
    }
    const float* xf = (const float*)p.x;
    const unsigned short* wq = (const unsigned short*)p.w;
    const int in_cs = p.in_cs, Wi = p.W, Cin = p.Cin;
    const size_t w_plane = (size_t)p.kh * p.kw * p.Cout * Cin;        // elements per weight plane
    f32x4 ra[A_LOADS];
    u32x4 rb[B_LOADS];

    int nx_ky, nx_kx, nx_c0;
    {
        const int tap0 = kc0 / cpt;
        nx_c0 = (kc0 - tap0 * cpt) * B3_BK;
        nx_ky = tap0 / p.kw;
        nx_kx = tap0 - nx_ky * p.kw;
    }
    auto load_chunk = [&]() {
        const int ky = nx_ky, kx = nx_kx, c0 = nx_c0;
        const int tapoff = ky * Wi + kx;
#pragma unroll
        for (int l = 0; l < A_LOADS; ++l) {
            const int idx = tid + 256 * l;
            const bool v = ((a_rm[l] >> ky) & (a_cm[l] >> kx) & 1u) != 0u;
            f32x4 q = {0.f, 0.f, 0.f, 0.f};
            if (v) {
                const int pix = p.transposed ? a_base[l] - tapoff : a_base[l] + tapoff;
                q = *(const f32x4*)(xf + (size_t)(a_img[l] + pix) * in_cs + c0 + 4 * (idx % NPIECE));
            }
            ra[l] = q;
        }
        const unsigned short* wt = wq + ((size_t)(ky * p.kw + kx) * p.Cout) * Cin + c0;
#pragma unroll
        for (int l = 0; l < B_LOADS; ++l) {
            const int idx = tid + 256 * l;
            const int plane = idx / (BN * BPR), rem = idx % (BN * BPR);
            const int row = rem / BPR, q4 = rem % BPR;
            u32x4 q = {0u, 0u, 0u, 0u};
            if ((B_TOTAL % 256 == 0 || idx < B_TOTAL) && co0 + row < p.Cout) q = *(const u32x4*)(wt + plane * w_plane + (size_t)(co0 + row) * Cin + 8 * q4);
            rb[l] = q;
        }
        nx_c0 += B3_BK;
        if (nx_c0 >= Cin) {
            nx_c0 = 0;
            if (++nx_kx == p.kw) { nx_kx = 0; ++nx_ky; }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int l = 0; l < A_LOADS; ++l) {
            const int idx = tid + 256 * l;
            const int row = idx / NPIECE, k = 4 * (idx % NPIECE);
            unsigned hi[2], mid[2], lo[2];
            b3_split2(ra[l][0], ra[l][1], hi[0], mid[0], lo[0]);
            b3_split2(ra[l][2], ra[l][3], hi[1], mid[1], lo[1]);
            *(u32x2*)&As[0][row][k] = u32x2{hi[0], hi[1]};
            *(u32x2*)&As[1][row][k] = u32x2{mid[0], mid[1]};
            *(u32x2*)&As[2][row][k] = u32x2{lo[0], lo[1]};
        }
#pragma unroll
        for (int l = 0; l < B_LOADS; ++l) {
            const int idx = tid + 256 * l;
            const int plane = idx / (BN * BPR), rem = idx % (BN * BPR);
            if (B_TOTAL % 256 == 0 || idx < B_TOTAL) *(u32x4*)&Bs[plane][rem / BPR][8 * (rem % BPR)] = rb[l];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    B3_STAMP(1);
    if (kc0 < kc1) {
        load_chunk();
        store_chunk();
    }
    __syncthreads();
    B3_STAMP(2);
    for (int c = kc0; c < kc1; ++c) {
        const bool more = c + 1 < kc1;
        if (more) load_chunk();                           // in flight under the MFMAs below
#pragma unroll
        for (int s = 0; s < B3_BK / 16; ++s) {
            const int k0 = 16 * s + 8 * h;
            bf16x8_t av[3][TM], bv[3][TN];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int a = 0; a < TM; ++a) av[pl][a] = *(const bf16x8_t*)&As[pl][(wm * TM + a) * 32 + i][k0];
#pragma unroll
                for (int b = 0; b < TN; ++b) bv[pl][b] = *(const bf16x8_t*)&Bs[pl][(wn * TN + b) * 32 + i][k0];
            }
            // the six products, smallest first; consecutive MFMAs go to different accumulators
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[t]][a], bv[PB[t]][b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();                                  // every wave has read the tile
        if (more) store_chunk();
        __syncthreads();
    }

    B3_STAMP(3);
    auto epi = [&](auto A) {
        constexpr int a = decltype(A)::value;
        const int mt = m0 + (wm * TM + a) * 32;
        int n = 0, oy = 0, ox = 0;
        if (p.r != 1 || p.out_nchw) {
            const int m = max(0, min(mt + i, p.M - 1));
            const int tq = b3_fastdiv(m, p.wo_mul, p.wo_sh);
            ox = m - tq * p.Wo;
            n = b3_fastdiv(tq, p.ho_mul, p.ho_sh);
            oy = tq - n * p.Ho;
        }
        generic_epilogue<TN>(p, acc[a], mt, co0 + wn * TN * 32, n, oy, ox, lane, blockIdx.z);
    };
    epi(std::integral_constant<int, 0>{});
    if constexpr (TM > 1) epi(std::integral_constant<int, 1>{});
    static_assert(TM <= 2, "epilogue is written out for TM <= 2");
#ifdef SRMI_B3_STAMPS      // investigation build: phase time stamps of one workgroup over its own first output row (results wrong)
    if (blockIdx.x == (gridDim.x >> 1) && blockIdx.y == 0 && blockIdx.z == 0) {
        __syncthreads();
        if (tid == 0) {
            B3_STAMP(4);
            for (int k = 0; k < 5; ++k) ((long long*)((float*)p.y + (size_t)m0 * p.out_cs))[k] = stamp[k];
        }
    }
#endif
}

// ---- operand pack: OIHW f32 -> [plane][tap][n][k] bf16 --------------------------------------------------------------
// forward: n = co, k = ci;  dgrad (transposed = 1): n = ci, k = co (same tap index: the kernel subtracts the tap offset)
__global__ __launch_bounds__(256) void b3_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cin,
                                                      int Cout, int taps, int transposed, size_t plane) {
    const int Nn = transposed ? Cin : Cout, Kk = transposed ? Cout : Cin;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < plane; idx += (size_t)gridDim.x * 256) {
        const int k = (int)(idx % Kk);
        const size_t t1 = idx / Kk;
        const int n = (int)(t1 % Nn);
        const int tap = (int)(t1 / Nn);
        const int co = transposed ? k : n, ci = transposed ? n : k;
        unsigned hi, mid, lo;
        b3_split(w[((size_t)co * Cin + ci) * taps + tap], hi, mid, lo);
        out[idx] = (unsigned short)hi;
        out[plane + idx] = (unsigned short)mid;
        out[2 * plane + idx] = (unsigned short)lo;
    }
}

static bool b3_enabled() {
    static const bool v = [] { const char* e = getenv("SRMI_F32_EXACT"); return !(e && atoi(e)); }();
    return v;
}

// Decided from the WEIGHT's shape alone (the host caches packed operands per weight, not per input size).
bool b3_fwd_ok(const srmi_conv_desc& d) {
    return b3_enabled() && d.compute == SRMI_COMPUTE_F32 && d.in_layout == SRMI_NHWC && d.in_dtype == SRMI_F32 &&
           d.Cin % 32 == 0 && d.Cout % 32 == 0 && d.kh * d.kw <= 25 && (long long)d.kh * d.kw * d.Cin >= 256;
}
bool b3_dgrad_ok(const srmi_conv_desc& d) {
    return b3_enabled() && d.stride == 1 && d.Cin % 32 == 0 && d.Cout % 32 == 0 && d.kh * d.kw <= 25 &&
           (long long)d.kh * d.kw * d.Cout >= 256;
}

// Operand of the stride-2 dgrad in its sub-pixel form (conv_generic.hip: dgrad_subpixel - a forward 2x2 conv of dz to
// 4 Cin channels): [plane][tap (a, b)][n = (py, px, ci)][k = co]; taps that do not exist for a parity are zero.
__global__ __launch_bounds__(256) void b3_pack_sub_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cin,
                                                          int Cout, size_t plane) {
    const int Nn = 4 * Cin;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < plane; idx += (size_t)gridDim.x * 256) {
        const int k = (int)(idx % Cout);
        const size_t t1 = idx / Cout;
        const int nn = (int)(t1 % Nn);
        const int tap = (int)(t1 / Nn);
        const int sub = nn / Cin, ci = nn - sub * Cin;
        const int py = sub >> 1, px = sub & 1, a = tap >> 1, b = tap & 1;
        const int ky = py + 1 - 2 * a, kx = px + 1 - 2 * b;
        float v = 0.f;
        if (ky >= 0 && ky < 3 && kx >= 0 && kx < 3) v = w[(((size_t)k * Cin + ci) * 3 + ky) * 3 + kx];
        unsigned hi, mid, lo;
        b3_split(v, hi, mid, lo);
        out[idx] = (unsigned short)hi;
        out[plane + idx] = (unsigned short)mid;
        out[2 * plane + idx] = (unsigned short)lo;
    }
}

bool b3_dgrad_sub_ok(const srmi_conv_desc& d) { return b3_enabled() && d.Cin % 32 == 0 && d.Cout % 32 == 0 && !(dbg_flags() & (1 << 8)); }

size_t b3_sub_packed_bytes(const srmi_conv_desc& d) { return (size_t)3 * 4 * 4 * d.Cin * d.Cout * sizeof(unsigned short); }

int b3_pack_sub(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s) {
    const size_t plane = (size_t)4 * 4 * d.Cin * d.Cout;
    if (plane == 0) return SRMI_OK;
    size_t blocks = (plane + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(b3_pack_sub_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w_oihw, (unsigned short*)w_packed, d.Cin, d.Cout,
                       plane);
    return hip_check(hipGetLastError(), "b3_pack_sub_kernel");
}

size_t b3_packed_bytes(const srmi_conv_desc& d) { return (size_t)3 * d.kh * d.kw * d.Cin * d.Cout * sizeof(unsigned short); }

int b3_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, int transposed, hipStream_t s) {
    const size_t plane = (size_t)d.kh * d.kw * d.Cin * d.Cout;
    if (plane == 0) return SRMI_OK;
    size_t blocks = (plane + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(b3_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w_oihw, (unsigned short*)w_packed, d.Cin, d.Cout,
                       d.kh * d.kw, transposed, plane);
    return hip_check(hipGetLastError(), "b3_pack_kernel");
}

void f32t_magic(int d, unsigned* mul, int* sh);
void f32t_plan(const GenericParams& p, bool allow_split, int* cfg, int* ks);

// Tile, K split and chunk size.  conv_f32_tiled.hip's plan, except that a small layer it would split over K (partial tiles
// through the workspace + a reduce launch: ~10 us of fixed cost against ~2 us of MFMA work here) runs UNSPLIT on 64 x 64
// tiles with 64-channel chunks when that still gives half the CUs a workgroup and at most 24 chunks each.
static const int kB3BM[5] = {128, 128, 128, 64, 128}, kB3BN[5] = {128, 64, 32, 64, 32};

void b3_plan(const GenericParams& p, bool allow_split, int* cfg, int* ks, int* bk) {
    *bk = 32;
    f32t_plan(p, true, cfg, ks);                           // "would split" is a property of the layer, with or without workspace
    const long long wg3 = (long long)((p.M + 63) / 64) * ((p.Cout + 63) / 64);
    if (*ks > 1 && p.Cin % 64 == 0 && wg3 >= 128 && (long long)p.kh * p.kw * (p.Cin / 64) <= 24) {
        *cfg = 3; *ks = 1; *bk = 64;
        return;
    }
    f32t_plan(p, allow_split, cfg, ks);
    if (*cfg == 2) *cfg = 4;                               // (a 256-pixel tile's three planes do not fit the static LDS)
}

int b3_ksplit(const GenericParams& p) {
    int cfg, ks, bk;
    b3_plan(p, true, &cfg, &ks, &bk);
    return ks;
}

// p.b3 is set: the operand is in this file's format.  The launch-time requirements (16-byte activation rows) hold for
// every dense NHWC tensor; anything else is refused loudly - there is no other kernel that could read this operand.
int b3_launch(const GenericParams& p0) {
    GenericParams p = p0;
    if (!p.vec_ok || p.in_cs % 4 != 0 || p.in_bf16 || p.in_nchw || (p.transposed && p.stride > 1)) {
        set_error("conv (split-bf16 f32 kernel): input must be NHWC f32 with 16-byte aligned rows");
        return SRMI_EUNSUPPORTED;
    }
    int c, ks, bk;
    b3_plan(p, p.ksplit > 1, &c, &ks, &bk);               // p.ksplit was fixed by generic_run (1 when there is no workspace)
    f32t_magic(p.Wo > 0 ? p.Wo : 1, &p.wo_mul, &p.wo_sh);
    f32t_magic(p.Ho > 0 ? p.Ho : 1, &p.ho_mul, &p.ho_sh);
    const int BM = kB3BM[c], BN = kB3BN[c];
    dim3 grid((unsigned)((p.M + BM - 1) / BM), (unsigned)((p.Cout + BN - 1) / BN), (unsigned)p.ksplit);
    hipStream_t s = p.stream;
    if (bk == 64) {
        hipLaunchKernelGGL((conv_b3_kernel<2, 2, 1, 1, 64>), grid, dim3(256), 0, s, p);
    } else {
        switch (c) {
            case 0: hipLaunchKernelGGL((conv_b3_kernel<2, 2, 2, 2, 32>), grid, dim3(256), 0, s, p); break;
            case 1: hipLaunchKernelGGL((conv_b3_kernel<2, 2, 2, 1, 32>), grid, dim3(256), 0, s, p); break;
            case 3: hipLaunchKernelGGL((conv_b3_kernel<2, 2, 1, 1, 32>), grid, dim3(256), 0, s, p); break;
            default: hipLaunchKernelGGL((conv_b3_kernel<4, 1, 1, 1, 32>), grid, dim3(256), 0, s, p); break;
        }
    }
    return hip_check(hipGetLastError(), "conv_b3_kernel");
}

}  // namespace srmi
