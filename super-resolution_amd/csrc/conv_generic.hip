// Generic fused conv2d on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// Serves every shape the fast kernels do not: any kh x kw, stride, pad, Cin, Cout, NHWC/NCHW,
// f32/bf16 storage, sub-pixel stores of any factor.  It is the <=1e-4 parity path
// (SRMI_COMPUTE_F32): the f32 MFMA is bit-for-bit an fmaf chain, so results differ from the
// reference's fp32 conv only by summation order.
//
// Mapping: implicit GEMM, M = N*Ho*Wo output pixels (flattened, so stride-2 and ragged shapes
// need no special cases), N = Cout, K = kh*kw*Cin.  One wave owns a 32(M) x 32*NT(N) tile and
// walks K straight from global memory (the activations of one tile are re-read from L1/L2 by
// the taps; no LDS: this kernel is the fallback, not the roofline kernel).
// K order inside a tap is permuted so each lane reads 4 consecutive channels (one 16-B load):
// k-pair (c0+j, c0+4+j) is fed to MFMA j; A and B use the same permutation so the sum is the
// same set of products.
#include "generic_common.h"

namespace srmi {

template <int NT>
__global__ __launch_bounds__(256) void conv_generic_kernel(GenericParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int m0 = (blockIdx.x * 4 + wave) * 32;
    if (m0 >= p.M) return;  // wave-uniform; the kernel has no barriers
    const int co0 = blockIdx.y * (32 * NT);

    const int m = m0 + i;
    const bool mvalid = m < p.M;
    const int mm = mvalid ? m : 0;
    const int ox = mm % p.Wo;
    const int tq = mm / p.Wo;
    const int oy = tq % p.Ho;
    const int n = tq / p.Ho;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const float* xf = (const float*)p.x;
    const bf16_t* xb = (const bf16_t*)p.x;
    const size_t HW = (size_t)p.H * p.W;

    // K space = (tap, 8-channel chunk); a split-K launch (blockIdx.z) walks only its slice of it
    const int cpt = p.Cin_pad >> 3;                 // chunks per tap
    const int KC = p.kh * p.kw * cpt;
    int kc = 0, kc1 = KC;
    if (p.ksplit > 1) {
        const int per = (KC + p.ksplit - 1) / p.ksplit;
        kc = blockIdx.z * per;
        kc1 = min(KC, kc + per);
    }
    while (kc < kc1) {
        const int tap = kc / cpt;
        const int cbeg = (kc - tap * cpt) << 3;
        const int cend = min(p.Cin_pad, cbeg + ((kc1 - kc) << 3));
        kc += (cend - cbeg) >> 3;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
        int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
        bool vt = true;
        if (p.transposed) {
            const int ty = oy + p.pad - ky, tx = ox + p.pad - kx;
            vt = ty >= 0 && tx >= 0 && ty % p.stride == 0 && tx % p.stride == 0;
            iy = ty / p.stride;
            ix = tx / p.stride;
        }
        const bool v = mvalid && vt && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const float* wt = p.w + (size_t)tap * p.Cin_pad * p.Cout_pad + co0 + i;
        const size_t pix = v ? (((size_t)n * p.H + iy) * p.W + ix) : 0;
        for (int c0 = cbeg; c0 < cend; c0 += 8) {
            const int cb = c0 + 4 * h;
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            if (v) {
                if (p.vec_ok) {  // NHWC, Cin % 8 == 0, 16-B aligned rows
                    if (p.in_bf16) {
                        const u16x4 q = *(const u16x4*)(xb + pix * p.in_cs + cb);
                        a[0] = bf2f(q[0]); a[1] = bf2f(q[1]); a[2] = bf2f(q[2]); a[3] = bf2f(q[3]);
                    } else {
                        const f32x4 q = *(const f32x4*)(xf + pix * p.in_cs + cb);
                        a[0] = q[0]; a[1] = q[1]; a[2] = q[2]; a[3] = q[3];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = cb + j;
                        if (c < p.Cin) {
                            const size_t idx = p.in_nchw
                                ? ((size_t)n * p.Cin + c) * HW + (size_t)iy * p.W + ix
                                : pix * p.in_cs + c;
                            a[j] = p.in_bf16 ? bf2f(xb[idx]) : xf[idx];
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* wr = wt + (size_t)(cb + j) * p.Cout_pad;
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], wr[32 * t], acc[t], 0, 0, 0);
            }
        }
    }

    generic_epilogue<NT>(p, acc, m0, co0, n, oy, ox, lane, blockIdx.z);
}

// split-bf16 f32 kernel (conv_b3_f32.hip): chosen from the descriptor, with its own operand format
bool b3_fwd_ok(const srmi_conv_desc& d);
bool b3_dgrad_ok(const srmi_conv_desc& d);
size_t b3_packed_bytes(const srmi_conv_desc& d);
int b3_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, int transposed, hipStream_t s);
int b3_launch(const GenericParams& p);
int b3_ksplit(const GenericParams& p);
bool b3_dgrad_sub_ok(const srmi_conv_desc& d);
size_t b3_sub_packed_bytes(const srmi_conv_desc& d);
int b3_pack_sub(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s);
// its 3x3 / 64-channel halo-tile sibling (conv_b3h_f32.hip; p.b3 == 2)
bool b3h_fwd_ok(const srmi_conv_desc& d);
bool b3h_dgrad_ok(const srmi_conv_desc& d);
size_t b3h_packed_bytes(const srmi_conv_desc& d);
int b3h_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, int transposed, hipStream_t s);
int b3h_launch(const GenericParams& p);

// OIHW f32 -> [tap][Cin_pad][Cout_pad] f32, zero padded
__global__ void generic_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout,
                                    int kh, int kw, int Cin_pad, int Cout_pad, size_t total) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(idx % Cout_pad);
        const size_t t1 = idx / Cout_pad;
        const int c = (int)(t1 % Cin_pad);
        const int tap = (int)(t1 / Cin_pad);
        float v = 0.f;
        if (co < Cout && c < Cin) {
            const int ky = tap / kw, kx = tap % kw;
            v = w[(((size_t)co * Cin + c) * kh + ky) * kw + kx];
        }
        out[idx] = v;
    }
}

// dgrad operand: OIHW f32 -> [tap][Cout_pad8 (contracted)][Cin_pad32 (produced)] f32, same tap index as forward
__global__ void generic_pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout,
                                          int kh, int kw, int K_pad, int N_pad, size_t total) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(idx % N_pad);
        const size_t t1 = idx / N_pad;
        const int co = (int)(t1 % K_pad);
        const int tap = (int)(t1 / K_pad);
        float v = 0.f;
        if (co < Cout && ci < Cin) {
            const int ky = tap / kw, kx = tap % kw;
            v = w[(((size_t)co * Cin + ci) * kh + ky) * kw + kx];
        }
        out[idx] = v;
    }
}

// Stride-2 dgrad as a SUB-PIXEL convolution.  For a 3x3 / stride 2 / pad 1 conv on an even-sized input the input gradient
// is  dx[2 y2 + py, 2 x2 + px] = sum over a, b in {0, 1} of dz[y2 + a, x2 + b] . w[.., ky, kx]  with ky = py + 1 - 2 a,
// kx = px + 1 - 2 b (taps outside 0..2 do not exist): a stride-1 conv of dz with a 2x2 kernel (reads one row / column
// past the end: zero) to 4 Cin channels, stored through depth_to_space(2).  9 of its 16 tap slots are real, against 9 of
// 36 when every dx pixel gathers all nine taps and masks the ones whose parity does not fit (the "transposed" mode of
// the kernels): 2.25x fewer MFMAs, on the forward kernel, through the sub-pixel store the epilogue already has.
// SRMI_DBG bit 14: the gather form.
bool dgrad_subpixel(const srmi_conv_desc& d) {
    if (dbg_flags() & (1 << 14)) return false;
    return d.stride == 2 && d.kh == 3 && d.kw == 3 && d.pad == 1 && d.H % 2 == 0 && d.W % 2 == 0 && d.H > 0 && d.W > 0 &&
           d.Cin % 32 == 0;      // (lean sub-pixel epilogue: whole 32-channel tiles).  Shape of the WEIGHT and stride / pad / H, W
                                 // parity only - never N: the packed operand format follows this predicate and is cached per weight
}

// its operand: OIHW f32 -> [tap (a, b)][Cout_pad8 (contracted)][(py, px, ci) padded to 32 (produced)]
__global__ void generic_pack_dgrad_sub_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout,
                                              int K_pad, int N_pad, size_t total) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int nn = (int)(idx % N_pad);
        const size_t t1 = idx / N_pad;
        const int co = (int)(t1 % K_pad);
        const int tap = (int)(t1 / K_pad);
        float v = 0.f;
        if (co < Cout && nn < 4 * Cin) {
            const int sub = nn / Cin, ci = nn - sub * Cin;
            const int py = sub >> 1, px = sub & 1, a = tap >> 1, b = tap & 1;
            const int ky = py + 1 - 2 * a, kx = px + 1 - 2 * b;
            if (ky >= 0 && ky < 3 && kx >= 0 && kx < 3) v = w[(((size_t)co * Cin + ci) * 3 + ky) * 3 + kx];
        }
        out[idx] = v;
    }
}

// ---- the same packs for operands without padding, through LDS (both sides of the transposition coalesced) -----------
// The gather kernels above read 4 bytes per lane at a stride of Cin*kh*kw floats; at the discriminator's first Linear
// (75 MB of weights, re-packed after every optimizer step in both forms) that was 0.5 ms per training step.
//
// forward operand, Cin*taps % 32 == 0 and Cout % 32 == 0: a 32(co) x 32(j = c*taps + tap) tile of the OIHW matrix is read
// along j and written along co, row j going to row (tap, c) of the packed operand.
__global__ __launch_bounds__(256) void generic_pack_tiled_kernel(const float* __restrict__ w, float* __restrict__ out,
                                                                 int Cin, int Cout, int taps) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const size_t J = (size_t)Cin * taps;
    const size_t j0 = (size_t)blockIdx.x * 32;
    const int co0 = blockIdx.y * 32;
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[ty + 8 * r][tx] = w[(size_t)(co0 + ty + 8 * r) * J + j0 + tx];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const size_t j = j0 + ty + 8 * r;
        const int c = (int)(j / taps), tap = (int)(j - (size_t)c * taps);
        out[((size_t)tap * Cin + c) * Cout + co0 + tx] = tile[tx][ty + 8 * r];
    }
}

// dgrad operands, Cin % 32 == 0, Cout % 8 == 0, taps <= 128: workgroup (ci block of 32, co) reads the contiguous
// [32 ci][taps] run of its output channel and writes one 128-byte row of 32 ci per tap, at
//   general form   [tap][co][ci]          full-map form   [co][tap][ci]   (dgrad_fullmap below)
__global__ __launch_bounds__(256) void generic_pack_dgrad_rows_kernel(const float* __restrict__ w, float* __restrict__ out,
                                                                      int Cin, int Cout, int taps, int fullmap) {
    __shared__ float run[32 * 128 + 32];
    const int ci0 = blockIdx.x * 32, co = blockIdx.y;
    const float* src = w + ((size_t)co * Cin + ci0) * taps;
    const int n = 32 * taps;
    for (int idx = threadIdx.x; idx < n; idx += 256) run[idx + idx / taps] = src[idx];      // (+1 per ci: bank skew)
    __syncthreads();
    for (int idx = threadIdx.x; idx < n; idx += 256) {
        const int tap = idx >> 5, ci = idx & 31;
        const size_t row = fullmap ? (size_t)co * taps + tap : (size_t)tap * Cout + co;
        out[row * Cin + ci0 + ci] = run[ci * taps + tap + ci];
    }
}

// A conv whose kernel covers the whole input map without padding (nn.Linear on a flattened feature map): one output
// pixel per image.  Its input gradient is a plain GEMM  dx[n][(tap, ci)] = sum over co of dz[n][co] * w[co][(tap, ci)] -
// a FORWARD 1x1 conv of dz [N,1,1,Cout] to kh*kw*Cin channels, whose NHWC output is dx itself.  As a gathering
// transposed conv every dx pixel walked all kh*kw taps to find the one that exists (1 of 36 at the discriminator's
// Linear: 275 us for 75 MB of weights).
bool dgrad_fullmap(const srmi_conv_desc& d) {
    if (dbg_flags() & (1 << 12)) return false;
    return d.pad == 0 && d.kh == d.H && d.kw == d.W && d.kh * d.kw > 1 && d.kh * d.kw <= 128 && d.Cin % 32 == 0 &&
           d.Cout % 8 == 0 && d.Cout <= 65535 && (long long)d.kh * d.kw * d.Cin < 0x7fffffffLL;
}

// the gathering (transposed) dgrad form on the split-bf16 kernel; the sub-pixel and full-map forms keep their own operands
static bool dgrad_b3(const srmi_conv_desc& d) { return !dgrad_subpixel(d) && !dgrad_fullmap(d) && b3_dgrad_ok(d); }
static bool dgrad_b3h(const srmi_conv_desc& d) { return !dgrad_subpixel(d) && !dgrad_fullmap(d) && b3h_dgrad_ok(d); }

size_t generic_dgrad_packed_bytes(const srmi_conv_desc& d) {
    if (dgrad_b3h(d)) return b3h_packed_bytes(d);
    if (dgrad_b3(d)) return b3_packed_bytes(d);
    if (dgrad_subpixel(d) && b3_dgrad_sub_ok(d)) return b3_sub_packed_bytes(d);
    if (dgrad_subpixel(d)) return (size_t)4 * round_up(d.Cout, 8) * round_up(4 * d.Cin, 32) * sizeof(float);
    return (size_t)d.kh * d.kw * round_up(d.Cout, 8) * round_up(d.Cin, 32) * sizeof(float);
}

int generic_dgrad_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s) {
    if (dgrad_b3h(d)) return b3h_pack(d, w_oihw, w_packed, 1, s);
    if (dgrad_b3(d)) return b3_pack(d, w_oihw, w_packed, 1, s);
    if (dgrad_subpixel(d) && b3_dgrad_sub_ok(d)) return b3_pack_sub(d, w_oihw, w_packed, s);
    if (dgrad_subpixel(d)) {
        const int K_pad = round_up(d.Cout, 8), N_pad = round_up(4 * d.Cin, 32);
        const size_t total = (size_t)4 * K_pad * N_pad;
        size_t blocks = (total + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(generic_pack_dgrad_sub_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w_oihw, (float*)w_packed,
                           d.Cin, d.Cout, K_pad, N_pad, total);
        return hip_check(hipGetLastError(), "generic_pack_dgrad_sub_kernel");
    }
    const int K_pad = round_up(d.Cout, 8), N_pad = round_up(d.Cin, 32);
    const size_t total = (size_t)d.kh * d.kw * K_pad * N_pad;
    if (total == 0) return SRMI_OK;
    const bool fullmap = dgrad_fullmap(d);
    if (fullmap || (d.Cin % 32 == 0 && d.Cout % 8 == 0 && d.kh * d.kw <= 128 && d.Cout <= 65535 && total >= (1u << 20))) {
        hipLaunchKernelGGL(generic_pack_dgrad_rows_kernel, dim3((unsigned)(d.Cin / 32), (unsigned)d.Cout), dim3(256), 0, s,
                           w_oihw, (float*)w_packed, d.Cin, d.Cout, d.kh * d.kw, fullmap ? 1 : 0);
        return hip_check(hipGetLastError(), "generic_pack_dgrad_rows_kernel");
    }
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(generic_pack_dgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w_oihw, (float*)w_packed,
                       d.Cin, d.Cout, d.kh, d.kw, K_pad, N_pad, total);
    return hip_check(hipGetLastError(), "generic_pack_dgrad_kernel");
}

size_t generic_packed_bytes(const srmi_conv_desc& d) {
    if (b3h_fwd_ok(d)) return b3h_packed_bytes(d);
    if (b3_fwd_ok(d)) return b3_packed_bytes(d);
    return (size_t)d.kh * d.kw * round_up(d.Cin, 8) * round_up(d.Cout, 32) * sizeof(float);
}

int generic_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s) {
    if (b3h_fwd_ok(d)) return b3h_pack(d, w_oihw, w_packed, 0, s);
    if (b3_fwd_ok(d)) return b3_pack(d, w_oihw, w_packed, 0, s);
    const int Cin_pad = round_up(d.Cin, 8), Cout_pad = round_up(d.Cout, 32);
    const size_t total = (size_t)d.kh * d.kw * Cin_pad * Cout_pad;
    if (total == 0) return SRMI_OK;
    if (d.Cin % 8 == 0 && d.Cout % 32 == 0 && ((size_t)d.Cin * d.kh * d.kw) % 32 == 0 && d.Cout / 32 <= 65535 &&
        total >= (1u << 20)) {
        hipLaunchKernelGGL(generic_pack_tiled_kernel, dim3((unsigned)((size_t)d.Cin * d.kh * d.kw / 32), (unsigned)(d.Cout / 32)),
                           dim3(256), 0, s, w_oihw, (float*)w_packed, d.Cin, d.Cout, d.kh * d.kw);
        return hip_check(hipGetLastError(), "generic_pack_tiled_kernel");
    }
    const int block = 256;
    size_t blocks = (total + block - 1) / block;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(generic_pack_kernel, dim3((unsigned)blocks), dim3(block), 0, s, w_oihw, (float*)w_packed,
                       d.Cin, d.Cout, d.kh, d.kw, Cin_pad, Cout_pad, total);
    return hip_check(hipGetLastError(), "generic_pack_kernel");
}

static int generic_run(const struct GenericParams& p);


// vector-ALU kernels for convs with <= 4 channels on one side (conv_thin_f32.hip): 0 = not such a conv
int thinf_kind(const GenericParams& p);
int thinf_launch(const GenericParams& p);

// LDS-tiled kernel for NHWC f32 inputs (conv_f32_tiled.hip); this file's kernel serves the other input layouts / dtypes
bool f32t_supported(const GenericParams& p);
int f32t_ksplit(const GenericParams& p);
int f32t_launch(const GenericParams& p);

// y[i] = partial[0][i] + partial[1][i] + ... in slice order (slice 0 carries the shift)
__global__ void generic_splitk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ y, size_t n,
                                             int ksplit) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = partial[i];
        for (int z = 1; z < ksplit; ++z) v += partial[(size_t)z * n + i];
        y[i] = v;
    }
}

// the same for n % 4 == 0 and 16-byte aligned buffers: 16 bytes per lane and slice (same order of additions per element)
__global__ void generic_splitk_reduce_v4_kernel(const f32x4* __restrict__ partial, f32x4* __restrict__ y, size_t n4,
                                                int ksplit) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 v = partial[i];
#pragma unroll 4
        for (int z = 1; z < ksplit; ++z) v += partial[(size_t)z * n4 + i];
        y[i] = v;
    }
}

// number of K slices generic_run will use for these parameters (1 = no split)
static int generic_ksplit(const GenericParams& p) {
    const unsigned gy = p.Cout > 32 ? (unsigned)((p.Cout + 63) / 64) : 1u;
    const long long waves = (long long)((p.M + 31) / 32) * gy;
    const int KC = p.kh * p.kw * (p.Cin_pad / 8);
    if (p.scale || p.act != SRMI_ACT_NONE || p.res || p.out_bf16 || p.r != 1 || waves >= 2048 || KC < 16) return 1;
    if (!p.out_nchw && p.out_cs != p.Cf) return 1;   // channel-slice outputs cannot be zero-filled / reduced densely
    long long ks = (4096 + waves - 1) / waves;
    if (ks > KC / 8) ks = KC / 8;                    // at least 8 chunks (64 channels of one tap) per slice
    if (ks > 256) ks = 256;
    return ks > 1 ? (int)ks : 1;
}

static void generic_fill(GenericParams& p, const srmi_conv_args& a, const ConvGeom& g);

static int pick_ksplit(const GenericParams& p) {
    if (thinf_kind(p)) return 1;
    if (p.b3 == 2) return 1;
    if (p.b3) return b3_ksplit(p);
    return f32t_supported(p) ? f32t_ksplit(p) : generic_ksplit(p);
}

// bytes of workspace that make the forward conv's split-K deterministic (0: it does not split)
size_t generic_workspace_bytes(const srmi_conv_desc& d, const ConvGeom& g) {
    srmi_conv_args a;
    memset(&a, 0, sizeof(a));
    a.d = d;
    GenericParams p;
    generic_fill(p, a, g);
    const long long M = (long long)d.N * g.Ho * g.Wo;
    if (M <= 0 || M > 0x7fffffffLL) return 0;
    p.M = (int)M;
    const int ks = pick_ksplit(p);
    return ks > 1 ? (size_t)ks * d.N * g.Cf * g.Hf * g.Wf * sizeof(float) : 0;
}

int generic_launch(const srmi_conv_args& a, const ConvGeom& g, hipStream_t s, void* ws, size_t ws_bytes, double* bn_stats) {
    const srmi_conv_desc& d = a.d;
    GenericParams p;
    generic_fill(p, a, g);
    if (bn_stats && p.b3 != 2) { set_error("conv: fused BN statistics are served by the halo-tile split kernel only"); return SRMI_EUNSUPPORTED; }
    p.bn_stats = bn_stats;
    p.stream = s;
    p.partial = (float*)ws;
    p.partial_stride = ws ? ws_bytes / sizeof(float) : 0;     // capacity in elements until generic_run fixes the stride
    const long long M = (long long)d.N * g.Ho * g.Wo;
    if (M > 0x7fffffffLL) { set_error("generic conv: N*Ho*Wo = %lld exceeds int32", M); return SRMI_EINVAL; }
    p.M = (int)M;
    if (p.M == 0) return SRMI_OK;
    return generic_run(p);
}

static void generic_fill(GenericParams& p, const srmi_conv_args& a, const ConvGeom& g) {
    const srmi_conv_desc& d = a.d;
    memset(&p, 0, sizeof(p));
    p.x = a.x; p.w = (const float*)a.w; p.scale = a.scale; p.shift = a.shift; p.alpha = a.alpha;
    p.res = a.residual; p.y = a.y;
    p.N = d.N; p.H = d.H; p.W = d.W; p.Cin = d.Cin; p.Cout = d.Cout; p.kh = d.kh; p.kw = d.kw;
    p.stride = d.stride; p.pad = d.pad;
    p.Ho = g.Ho; p.Wo = g.Wo; p.Cf = g.Cf; p.Hf = g.Hf; p.Wf = g.Wf;
    p.in_cs = g.in_cs; p.out_cs = g.out_cs; p.out_co = d.out_coffset; p.res_cs = g.res_cs;
    p.in_bf16 = d.in_dtype == SRMI_BF16; p.out_bf16 = d.out_dtype == SRMI_BF16;
    p.res_bf16 = d.res_dtype == SRMI_BF16;
    p.in_nchw = d.in_layout == SRMI_NCHW; p.out_nchw = d.out_layout == SRMI_NCHW;
    const size_t esz = p.in_bf16 ? 2 : 4;
    p.vec_ok = !p.in_nchw && d.Cin % 8 == 0 && (g.in_cs * esz) % 16 == 0 && ((uintptr_t)a.x % 16) == 0;
    p.act = d.act; p.act_param = d.act_param; p.alpha_pc = d.alpha_per_channel;
    p.r = d.shuffle; p.dcr = d.shuffle_order == SRMI_SHUFFLE_DCR;
    p.Cin_pad = round_up(d.Cin, 8); p.Cout_pad = round_up(d.Cout, 32);
    p.transposed = 0;
    p.ksplit = 1;
    p.b3 = b3h_fwd_ok(d) ? 2 : b3_fwd_ok(d) ? 1 : 0;
}

// The f32 kernel generic_run picks for the forward conv of `d` (16-byte aligned buffers assumed, as every caller's are).
const char* generic_kernel_label(const srmi_conv_desc& d, const ConvGeom& g) {
    srmi_conv_args a;
    memset(&a, 0, sizeof(a));
    a.d = d;
    GenericParams p;
    generic_fill(p, a, g);
    if (thinf_kind(p)) return "thin_f32";
    if (p.b3 == 2) return "b3h_f32";
    if (p.b3) return "b3_f32";
    return f32t_supported(p) ? "f32t" : "generic_f32";
}

static int generic_run(const GenericParams& p0) {
    GenericParams p = p0;
    hipStream_t s = p.stream;
    const unsigned gx = (unsigned)((p.M + 127) / 128);
    const unsigned gy = p.Cout > 32 ? (unsigned)((p.Cout + 63) / 64) : 1u;
    // Split K when the pixel x channel tiling alone cannot fill 256 CUs (deep, spatially small layers: the
    // discriminator tail, VGG conv4/conv5, Linear layers, their dgrads).  Needs a linear epilogue with unit
    // scale and a dense f32 output, which is what the training path asks for, and a workspace: the slices are
    // stored there and added in slice order, so the result is the same bits on every run.  There are NO float
    // atomics in this file: without a (large enough) workspace the conv simply does not split.
    if (thinf_kind(p)) {          // <= 4 channels on one side: HBM-bound, one pass on the vector ALU
        p.ksplit = 1;
        return thinf_launch(p);
    }
    const bool tiled = f32t_supported(p);
    p.ksplit = pick_ksplit(p);
    const size_t out_elems = (size_t)p.N * p.Cf * p.Hf * p.Wf;
    if (p.ksplit > 1) {
        if (p.partial && p.partial_stride >= out_elems * (size_t)p.ksplit) p.partial_stride = out_elems;
        else { p.ksplit = 1; p.partial = nullptr; }
    }
    dim3 grid(gx, gy, (unsigned)p.ksplit);
    int rc;
    if (p.b3 == 2) {
        rc = b3h_launch(p);
    } else if (p.b3) {
        rc = b3_launch(p);
    } else if (tiled) {
        rc = f32t_launch(p);
    } else {
        if (p.Cout > 32) {
            hipLaunchKernelGGL(conv_generic_kernel<2>, grid, dim3(256), 0, s, p);
        } else {
            hipLaunchKernelGGL(conv_generic_kernel<1>, grid, dim3(256), 0, s, p);
        }
        rc = hip_check(hipGetLastError(), "conv_generic_kernel");
    }
    if (rc || p.ksplit <= 1 || !p.partial) return rc;
    if (out_elems % 4 == 0 && ((uintptr_t)p.partial | (uintptr_t)p.y) % 16 == 0) {
        size_t blocks = (out_elems / 4 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(generic_splitk_reduce_v4_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const f32x4*)p.partial,
                           (f32x4*)p.y, out_elems / 4, p.ksplit);
        return hip_check(hipGetLastError(), "generic_splitk_reduce_kernel");
    }
    size_t blocks = (out_elems + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(generic_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p.partial, (float*)p.y,
                       out_elems, p.ksplit);
    return hip_check(hipGetLastError(), "generic_splitk_reduce_kernel");
}

static void generic_dgrad_fill(GenericParams& p, const srmi_conv_desc& d, const ConvGeom& g);

// bytes of workspace with which the dgrad of `d` splits K (0: it does not split)
size_t generic_dgrad_workspace_bytes(const srmi_conv_desc& d, const ConvGeom& g) {
    GenericParams p;
    generic_dgrad_fill(p, d, g);
    const long long M = (dgrad_subpixel(d) || dgrad_fullmap(d)) ? (long long)d.N * g.Ho * g.Wo : (long long)d.N * d.H * d.W;
    if (M <= 0 || M > 0x7fffffffLL) return 0;
    p.M = (int)M;
    const int ks = pick_ksplit(p);
    return ks > 1 ? (size_t)ks * d.N * d.H * d.W * d.Cin * sizeof(float) : 0;
}

// dx[N,H,W,Cin] = conv_transpose(dz[N,Ho,Wo,Cout], w): the forward descriptor `d` describes the conv whose
// input gradient is wanted; dz and dx are dense NHWC f32.
int generic_dgrad_launch(const srmi_conv_desc& d, const ConvGeom& g, const float* dz, const float* w_dgrad, float* dx,
                         hipStream_t s, void* ws, size_t ws_bytes) {
    GenericParams p;
    generic_dgrad_fill(p, d, g);
    p.stream = s;
    p.x = dz; p.w = w_dgrad; p.y = dx;
    p.vec_ok = d.Cout % 8 == 0 && ((uintptr_t)dz % 16) == 0;
    p.partial = (float*)ws;
    p.partial_stride = ws ? ws_bytes / sizeof(float) : 0;
    const long long M = (dgrad_subpixel(d) || dgrad_fullmap(d)) ? (long long)d.N * g.Ho * g.Wo : (long long)d.N * d.H * d.W;
    if (M > 0x7fffffffLL || (long long)d.N * d.H * d.W > 0x7fffffffLL) {
        set_error("dgrad: N*H*W = %lld exceeds int32", (long long)d.N * d.H * d.W);
        return SRMI_EINVAL;
    }
    p.M = (int)M;
    if (p.M == 0) return SRMI_OK;
    return generic_run(p);
}

static void generic_dgrad_fill(GenericParams& p, const srmi_conv_desc& d, const ConvGeom& g) {
    memset(&p, 0, sizeof(p));
    if (dgrad_subpixel(d)) {      // a forward 2x2 / stride 1 / pad 0 conv of dz to 4 Cin channels + depth_to_space(2)
        p.N = d.N; p.H = g.Ho; p.W = g.Wo; p.Cin = d.Cout; p.Cout = 4 * d.Cin; p.kh = 2; p.kw = 2;
        p.stride = 1; p.pad = 0;
        p.Ho = g.Ho; p.Wo = g.Wo; p.Cf = d.Cin; p.Hf = d.H; p.Wf = d.W;
        p.in_cs = d.Cout; p.out_cs = d.Cin; p.out_co = 0; p.res_cs = d.Cin;
        p.act = SRMI_ACT_NONE; p.r = 2; p.dcr = 1;
        p.Cin_pad = round_up(d.Cout, 8); p.Cout_pad = round_up(4 * d.Cin, 32);
        p.transposed = 0;
        p.ksplit = 1;
        p.b3 = b3_dgrad_sub_ok(d) ? 1 : 0;               // the split-bf16 kernel with the sub-pixel operand and store
        return;
    }
    if (dgrad_fullmap(d)) {       // a forward 1x1 conv of dz [N,1,1,Cout] to (tap, ci) channels: its NHWC output is dx
        const int Cq = d.kh * d.kw * d.Cin;
        p.N = d.N; p.H = 1; p.W = 1; p.Cin = d.Cout; p.Cout = Cq; p.kh = 1; p.kw = 1; p.stride = 1; p.pad = 0;
        p.Ho = 1; p.Wo = 1; p.Cf = Cq; p.Hf = 1; p.Wf = 1;
        p.in_cs = d.Cout; p.out_cs = Cq; p.out_co = 0; p.res_cs = Cq;
        p.act = SRMI_ACT_NONE; p.r = 1;
        p.Cin_pad = round_up(d.Cout, 8); p.Cout_pad = round_up(Cq, 32);
        p.transposed = 0;
        p.ksplit = 1;
        return;
    }
    p.N = d.N; p.H = g.Ho; p.W = g.Wo; p.Cin = d.Cout; p.Cout = d.Cin; p.kh = d.kh; p.kw = d.kw;
    p.stride = d.stride; p.pad = d.pad;
    p.Ho = d.H; p.Wo = d.W; p.Cf = d.Cin; p.Hf = d.H; p.Wf = d.W;
    p.in_cs = d.Cout; p.out_cs = d.Cin; p.out_co = 0; p.res_cs = d.Cin;
    p.act = SRMI_ACT_NONE; p.r = 1;
    p.Cin_pad = round_up(d.Cout, 8); p.Cout_pad = round_up(d.Cin, 32);
    p.transposed = 1;
    p.ksplit = 1;
    p.b3 = dgrad_b3h(d) ? 2 : dgrad_b3(d) ? 1 : 0;
}

}  // namespace srmi
