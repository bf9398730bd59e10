// LDS-tiled exact-f32 conv2d (forward and, in transposed-gather mode, dgrad) on v_mfma_f32_32x32x2_f32, gfx950.
//
// The roofline kernel of the exact path: the <= 1e-4 inference mode and every forward / input-gradient conv of the
// SRGAN training step (P/mode.py:101, :149-179: G, D and the VGG19 stack).  Same arithmetic contract as
// conv_generic.hip (the f32 MFMA is bit-for-bit an fmaf chain; same packed weights, same fused epilogue, same
// deterministic split-K through a workspace), different data movement:
//
//   * implicit GEMM, M = N*Ho*Wo flattened output pixels, N = Cout, K = (tap, input channel).  One workgroup
//     (4 waves) owns a BM x BN output tile and walks K in chunks of BK channels of one tap;
//   * per chunk the BM x BK activation slab (gathered: each pixel's BK channels are 4*BK contiguous bytes of the NHWC
//     tensor, 16-B loads, zero fill for padding / stride holes) and the BK x BN weight slab (rows of the packed
//     [tap][Cin][Cout] array, fully coalesced) go global -> registers -> LDS, DOUBLE-BUFFERED: the loads of chunk c+1
//     are in flight while the MFMAs of chunk c run, one barrier per chunk;
//   * LDS images are K-major so that an MFMA operand read is one conflict-free ds_read_b32 per 32 lanes
//     (A: [k][pixel], pixel column XOR-ed with 8 * (k / 4) so the transposing stores are conflict-free too; B: [k][cout]);
//   * each wave keeps TM x TN accumulator tiles of 32 x 32 (register blocking: TM + TN operand reads feed TM * TN
//     MFMAs of 64 cycles each, so LDS bandwidth is a few per cent of what the MFMAs need).
//
// Algorithmic FLOPs = 2 * M * Cout * kh * kw * Cin; roofline = 157 TFLOP/s (f32 MFMA = the f32 vector rate).
#include <type_traits>

#include "generic_common.h"

namespace srmi {

// BK: channels per chunk (16; 8 for Cin <= 8).  WM x WN waves (= 4), each TM x TN tiles of 32 x 32.
//
// FAST (chosen on the host, f32t_fast): 16-byte activation loads, Cin a multiple of BK, no strided transposed taps, no
// ablation switches.  The chunk loop then has no per-element channel checks, no scalar-load fallback and no run-time
// divisions (tap and channel offset advance incrementally), which keeps its scalar state in registers: the general
// variant re-reads its kernel arguments from memory several times per chunk.
__device__ __forceinline__ int fastdiv(int n, unsigned mul, int sh) {      // n / d for 0 <= n < 2^31 (f32t_magic)
    return (int)(((unsigned long long)(unsigned)n * mul) >> sh);
}

// bits lo..hi of a 32-bit mask (empty when hi < lo; hi <= 31)
__device__ __forceinline__ unsigned bit_range(int lo, int hi) { return hi >= lo ? (2u << hi) - (1u << lo) : 0u; }

template <int BK, int WM, int WN, int TM, int TN, bool FAST>
__global__ __launch_bounds__(256) void conv_f32t_kernel(GenericParams p) {
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int NPIECE = BK / 4;                      // 16-B pieces per pixel and chunk
    constexpr int A_TOTAL = BM * NPIECE, B_TOTAL = BK * BN / 4;
    constexpr int A_LOADS = (A_TOTAL + 255) / 256, B_LOADS = (B_TOTAL + 255) / 256;
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM;
    const int co0 = blockIdx.y * BN;

    // ---- K range of this workgroup (split-K slice) in chunks of BK channels of one tap ----
    const int cpt = (p.Cin_pad + BK - 1) / BK;          // chunks per tap
    const int KC = p.kh * p.kw * cpt;
    int kc0 = 0, kc1 = KC;
    if (p.ksplit > 1) {
        const int per = (KC + p.ksplit - 1) / p.ksplit;
        kc0 = min(KC, (int)blockIdx.z * per);
        kc1 = min(KC, kc0 + per);
    }

    // ---- this thread's share of the activation slab: pixel (idx / NPIECE), 16-B piece (idx % NPIECE) ----
    // Everything that does not change from chunk to chunk is computed ONCE: the pixel index the tap offset is added to
    // (forward: (oy*s - pad) * W + (ox*s - pad); dgrad: ((oy + pad) * W + (ox + pad)), from which the tap offset is
    // subtracted and the difference divided by the stride) and one validity bit per tap (padding, stride holes).  A chunk
    // then costs a handful of vector instructions per load - on the small, latency-bound layers of the training step
    // the address arithmetic used to be a third of the time.
    int a_img[A_LOADS], a_base[A_LOADS];                 // image's first pixel (n * H * W), base pixel index inside it
    // validity of a tap = (its kernel row reads a real input row) & (its kernel column reads a real input column):
    // bit ky of a_rm, bit kx of a_cm (kh, kw <= 32, checked on the host).  Plain scalars, never indexed by a run-time
    // value (a dynamically indexed register array goes to scratch memory).
    unsigned a_rm[A_LOADS], a_cm[A_LOADS];
#pragma unroll
    for (int l = 0; l < A_LOADS; ++l) {
        const int idx = tid + 256 * l;
        const int m = m0 + idx / NPIECE;
        a_img[l] = 0; a_base[l] = 0; a_rm[l] = 0u; a_cm[l] = 0u;
        if (idx < A_TOTAL && m < p.M) {
            // (a run-time integer division costs ~40 instructions per lane on this ISA: multiply-shift with host-made
            // constants instead)
            const int tq = fastdiv(m, p.wo_mul, p.wo_sh);
            const int ox = m - tq * p.Wo;
            const int n = fastdiv(tq, p.ho_mul, p.ho_sh);
            const int oy = tq - n * p.Ho;
            a_img[l] = n * p.H * p.W;
            const int by = p.transposed ? oy + p.pad : oy * p.stride - p.pad;
            const int bx = p.transposed ? ox + p.pad : ox * p.stride - p.pad;
            a_base[l] = by * p.W + bx;
            auto axis_mask = [&](int b, int taps, int extent) {
                if (!p.transposed) return bit_range(max(0, -b), min(taps - 1, extent - 1 - b));     // 0 <= b + k < extent
                if (FAST || p.stride == 1) return bit_range(max(0, b - extent + 1), min(taps - 1, b));   // 0 <= b - k < extent
                unsigned mk = 0u;
                for (int k = 0; k < taps; ++k) {             // strided transposed taps: (b - k) divisible by the stride
                    const int tcoord = b - k;
                    bool v = tcoord >= 0;
                    int ic;
                    if (p.stride == 2) { v = v && !(tcoord & 1); ic = tcoord >> 1; }
                    else { v = v && tcoord % p.stride == 0; ic = tcoord / p.stride; }
                    if (v && ic < extent) mk |= 1u << k;
                }
                return mk;
            };
            a_rm[l] = axis_mask(by, p.kh, p.H);
            a_cm[l] = axis_mask(bx, p.kw, p.W);
        }
    }
    const float* xf = (const float*)p.x;
    f32x4 ra[A_LOADS], rb[B_LOADS];

    // FAST: (tap row, tap column, first channel) of the NEXT chunk to load; chunks are loaded in order, so the state
    // advances by BK channels per call instead of dividing the chunk index
    int nx_ky = 0, nx_kx = 0, nx_c0 = 0;
    if constexpr (FAST) {
        const int tap0 = kc0 / cpt;
        nx_c0 = (kc0 - tap0 * cpt) * BK;
        nx_ky = tap0 / p.kw;
        nx_kx = tap0 - nx_ky * p.kw;
    }
    const int in_cs = p.in_cs, Wi = p.W;
    const size_t w_tap = (size_t)p.Cin_pad * p.Cout_pad;
    const int b_row = tid / (BN / 4), b_c4 = tid % (BN / 4);         // weight slab piece of load 0 (256 / (BN/4) rows per load)
    const bool b_ok = co0 + 4 * b_c4 < p.Cout_pad;

    auto load_chunk = [&](int c) {
        if constexpr (FAST) {
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
            const float* wsrc = p.w + (size_t)(ky * p.kw + kx) * w_tap + (size_t)c0 * p.Cout_pad + co0 + 4 * b_c4;
#pragma unroll
            for (int l = 0; l < B_LOADS; ++l) {
                const int row = b_row + l * (256 / (BN / 4));
                f32x4 q = {0.f, 0.f, 0.f, 0.f};
                if (b_ok && (B_TOTAL % 256 == 0 || row < BK)) q = *(const f32x4*)(wsrc + (size_t)row * p.Cout_pad);
                rb[l] = q;
            }
            nx_c0 += BK;
            if (nx_c0 >= p.Cin_pad) {
                nx_c0 = 0;
                if (++nx_kx == p.kw) { nx_kx = 0; ++nx_ky; }
            }
            return;
        }
        const int tap = c / cpt;                         // scalar
        const int c0 = (c - tap * cpt) * BK;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
        const int tapoff = ky * p.W + kx;
#pragma unroll
        for (int l = 0; l < A_LOADS; ++l) {
            const int idx = tid + 256 * l;
            const int cb = c0 + 4 * (idx % NPIECE);
            const bool v = ((a_rm[l] >> ky) & (a_cm[l] >> kx) & 1u) && cb < p.Cin;
            f32x4 q = {0.f, 0.f, 0.f, 0.f};
            if (v) {
                int pix = p.transposed ? a_base[l] - tapoff : a_base[l] + tapoff;
                if (p.transposed && p.stride > 1) pix = p.stride == 2 ? pix >> 1 : pix / p.stride;   // exact: the tap is valid
                const float* src = xf + (size_t)(a_img[l] + pix) * p.in_cs + cb;
                if (p.vec_ok) {
                    q = *(const f32x4*)src;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (cb + j < p.Cin) q[j] = src[j];
                }
            }
            ra[l] = q;
        }
#pragma unroll
        for (int l = 0; l < B_LOADS; ++l) {
            const int idx = tid + 256 * l;
            const int row = idx / (BN / 4), c4 = idx % (BN / 4);
            f32x4 q = {0.f, 0.f, 0.f, 0.f};
            if (idx < B_TOTAL && c0 + row < p.Cin_pad && co0 + 4 * c4 < p.Cout_pad)
                q = *(const f32x4*)(p.w + ((size_t)tap * p.Cin_pad + c0 + row) * p.Cout_pad + co0 + 4 * c4);
            rb[l] = q;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int l = 0; l < A_LOADS; ++l) {
            const int idx = tid + 256 * l;
            if (idx < A_TOTAL) {
                const int piece = idx % NPIECE;
                const int col = (idx / NPIECE) ^ ((piece * 8) & 31);
#pragma unroll
                for (int j = 0; j < 4; ++j) As[buf][4 * piece + j][col] = ra[l][j];
            }
        }
#pragma unroll
        for (int l = 0; l < B_LOADS; ++l) {
            const int idx = tid + 256 * l;
            if (idx < B_TOTAL) *(f32x4*)&Bs[buf][idx / (BN / 4)][4 * (idx % (BN / 4))] = rb[l];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    if (kc0 < kc1) {
        load_chunk(kc0);
        store_chunk(0);
    }
    __syncthreads();
    int buf = 0;
    for (int c = kc0; c < kc1; ++c) {
        const bool more = c + 1 < kc1;
        if (more && (FAST || !(p.dbg & (1 << 26)))) load_chunk(c + 1);          // in flight under the MFMAs below
        if (FAST || !(p.dbg & (1 << 28)))
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int k = 2 * kk + h;
            const int sw = ((k >> 2) * 8) & 31;
            float av[TM], bv[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) av[a] = As[buf][k][((wm * TM + a) * 32 + i) ^ sw];
#pragma unroll
            for (int b = 0; b < TN; ++b) bv[b] = Bs[buf][k][(wn * TN + b) * 32 + i];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        if (more && (FAST || !(p.dbg & (1 << 27)))) store_chunk(buf ^ 1);
        if (FAST || !(p.dbg & (1 << 29))) __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue: one call per 32-pixel row of tiles (shared with conv_generic_kernel) ----
    // (no `continue` / early exit inside: the loop must unroll completely, or acc[a] becomes a run-time index into the
    // accumulator registers and the whole array moves to scratch memory)
    auto epi = [&](auto A) {
        constexpr int a = decltype(A)::value;
        const int mt = m0 + (wm * TM + a) * 32;          // rows past M are dropped by the epilogue's own bounds check
        int n = 0, oy = 0, ox = 0;
        if (p.r != 1 || p.out_nchw) {                    // (the dense NHWC store needs no pixel coordinates)
            const int m = max(0, min(mt + i, p.M - 1));
            const int tq = fastdiv(m, p.wo_mul, p.wo_sh);
            ox = m - tq * p.Wo;
            n = fastdiv(tq, p.ho_mul, p.ho_sh);
            oy = tq - n * p.Ho;
        }
        if (FAST || !(p.dbg & (1 << 30))) generic_epilogue<TN>(p, acc[a], mt, co0 + wn * TN * 32, n, oy, ox, lane, blockIdx.z);
    };
    epi(std::integral_constant<int, 0>{});
    if constexpr (TM > 1) epi(std::integral_constant<int, 1>{});
    static_assert(TM <= 2, "epilogue is written out for TM <= 2");
}

// Is the tiled kernel applicable?  NHWC f32 input; everything else (any k / stride / pad / channels, output layout and
// dtype, sub-pixel stores, channel slices, residual) is the shared epilogue's business.
bool f32t_supported(const GenericParams& p) {
    return !p.in_nchw && !p.in_bf16 && p.kh <= 32 && p.kw <= 32 && !(dbg_flags() & (1 << 24));
}

// Workgroup tile and K split.  The f32 MFMA is slow (64 cycles per 32 x 32 x 2 step), so a small layer is bound by the
// serial accumulation chain of ONE wave and by the fixed cost of a chunk (global-load latency, LDS hand-off, barrier),
// which only other resident workgroups hide: the plan wants >= 2 workgroups per CU.
//   0: 128 x 128, 4 tiles per wave (most operand reuse)    1: 128 x 64, 2 per wave    3: 64 x 64, 1 per wave
//   2: 256 x 32, 2 per wave (Cout <= 32)                   4: 128 x 32, 1 per wave (Cout <= 32)
// Largest tile that still gives >= 512 workgroups; else, when the epilogue is linear (scale NULL, no activation /
// residual / shuffle, dense f32 output: the training path) and the caller brought a workspace, the 128 x 64 tile with
// K split over <= 16 slices (each slice costs one more pass over the output: stored, then added in slice order);
// else the smallest tile.
static const int kBM[5] = {128, 128, 256, 64, 128}, kBN[5] = {128, 64, 32, 64, 32};

static int f32t_bk(const GenericParams& p) {
    return p.Cin_pad <= 8 ? 8 : (p.Cin_pad >= 32 ? 32 : 16);
}

static long long f32t_wgs(const GenericParams& p, int c) {
    return (long long)((p.M + kBM[c] - 1) / kBM[c]) * ((p.Cout + kBN[c] - 1) / kBN[c]);
}

void f32t_plan(const GenericParams& p, bool allow_split, int* cfg, int* ks) {
    if (dbg_flags() & (1 << 11)) allow_split = false;     // experiment: never split K
    const int BK = f32t_bk(p);
    const int KC = p.kh * p.kw * ((p.Cin_pad + BK - 1) / BK);
    const bool linear = (!(p.scale || p.act != SRMI_ACT_NONE || p.res || p.out_bf16 || p.r != 1) &&
                         (p.out_nchw || p.out_cs == p.Cf)) || generic_subpix2_shape(p);
    *ks = 1;
    if (p.Cout <= 32) {
        *cfg = f32t_wgs(p, 2) >= 512 ? 2 : 4;
    } else if (p.Cout >= 96 && f32t_wgs(p, 0) >= 512) {
        *cfg = 0;
    } else if (f32t_wgs(p, 1) >= 512) {
        *cfg = 1;
    } else if (f32t_wgs(p, 3) >= 384 || !(allow_split && linear) || KC < 8) {
        *cfg = 3;
    } else {
        *cfg = f32t_wgs(p, 1) >= 32 ? 1 : 3;
    }
    if (allow_split && linear && KC >= 8) {
        const long long wgs = f32t_wgs(p, *cfg);
        if (wgs < 384) {
            long long k = (640 + wgs - 1) / wgs;
            if (k > KC / 4) k = KC / 4;                   // at least four chunks per slice
            if (k > 16) k = 16;
            if (k > 1) *ks = (int)k;
        }
    }
}

int f32t_ksplit(const GenericParams& p) {
    int cfg, ks;
    f32t_plan(p, true, &cfg, &ks);
    return ks;
}

// multiply-shift constants of n / d for 0 <= n < 2^31, d >= 1: s = ceil(log2 d), mul = floor(2^(31+s) / d) + 1 (< 2^32),
// n / d = (n * mul) >> (31 + s)  (error term n * e / (d * 2^(31+s)) < 1 / d for e <= d <= 2^s)
void f32t_magic(int d, unsigned* mul, int* sh) {
    int s = 0;
    while ((1LL << s) < d) ++s;
    *mul = (unsigned)(((1ULL << (31 + s)) / (unsigned long long)d) + 1ULL);
    *sh = 31 + s;
}

// the lean chunk loop's preconditions (see the kernel)
static bool f32t_fast(const GenericParams& p, int BK) {
    return p.vec_ok && p.Cin % BK == 0 && p.Cin_pad == p.Cin && !(p.transposed && p.stride > 1) && (p.dbg & (31 << 26)) == 0 &&
           p.in_cs % 4 == 0;
}

int f32t_launch(const GenericParams& p0) {
    GenericParams p = p0;
    int c, ks;
    p.dbg = dbg_flags();
    f32t_plan(p, p.ksplit > 1, &c, &ks);                 // p.ksplit was fixed by generic_run (1 when there is no workspace)
    f32t_magic(p.Wo > 0 ? p.Wo : 1, &p.wo_mul, &p.wo_sh);
    f32t_magic(p.Ho > 0 ? p.Ho : 1, &p.ho_mul, &p.ho_sh);
    const int BM = kBM[c], BN = kBN[c];
    const int BK = f32t_bk(p);
    const bool fast = f32t_fast(p, BK);
    dim3 grid((unsigned)((p.M + BM - 1) / BM), (unsigned)((p.Cout + BN - 1) / BN), (unsigned)p.ksplit);
    hipStream_t s = p.stream;
#define SRMI_F32T(BKv, a, b, c_, d_)                                                                      \
    do {                                                                                                  \
        if (fast) hipLaunchKernelGGL((conv_f32t_kernel<BKv, a, b, c_, d_, true>), grid, dim3(256), 0, s, p);   \
        else hipLaunchKernelGGL((conv_f32t_kernel<BKv, a, b, c_, d_, false>), grid, dim3(256), 0, s, p);       \
    } while (0)
#define SRMI_F32T_CFG(BKv)                              \
    switch (c) {                                        \
        case 0: SRMI_F32T(BKv, 2, 2, 2, 2); break;      \
        case 1: SRMI_F32T(BKv, 2, 2, 2, 1); break;      \
        case 2: SRMI_F32T(BKv, 4, 1, 2, 1); break;      \
        case 3: SRMI_F32T(BKv, 2, 2, 1, 1); break;      \
        default: SRMI_F32T(BKv, 4, 1, 1, 1); break;     \
    }
    if (BK == 8) { SRMI_F32T_CFG(8) } else if (BK == 16) { SRMI_F32T_CFG(16) } else { SRMI_F32T_CFG(32) }
#undef SRMI_F32T_CFG
#undef SRMI_F32T
    return hip_check(hipGetLastError(), "conv_f32t_kernel");
}

}  // namespace srmi
