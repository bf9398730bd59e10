// Shared host/device helpers for libsrmi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/srmi.h"

namespace srmi {

typedef unsigned short bf16_t;  // raw bf16 bits
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf2f(bf16_t u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
// round-to-nearest-even, NaN stays NaN (plain cast -> v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
// two f32 -> packed bf16x2 (RNE) in ONE instruction; hipcc has no builtin for it and emits
// 2 cvt + shift + or for the scalar form (the epilogues are VALU-issue-bound at one wave per SIMD)
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

__device__ __forceinline__ float apply_act(float v, int act, float param, float alpha) {
    switch (act) {
        case SRMI_ACT_PRELU: return v >= 0.f ? v : v * alpha;
        case SRMI_ACT_LRELU: return v >= 0.f ? v : v * param;
        case SRMI_ACT_RELU: return v > 0.f ? v : 0.f;
        case SRMI_ACT_TANH: return tanhf(v);
        case SRMI_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

// ---- host side ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_check(hipError_t e, const char* what);

struct ConvGeom {
    int Ho, Wo;        // conv output spatial size
    int Cf;            // final channels = Cout / r^2
    int Hf, Wf;        // final spatial size = Ho*r, Wo*r
    int in_cs, out_cs; // effective channel strides
    int res_cs;        // residual channels per pixel
};
int conv_geom(const srmi_conv_desc& d, ConvGeom* g);  // validates, fills g

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// Spatial tile choice of the LDS-tiled 3x3 kernels: 0 = 6 x 96, 1 = 8 x 64, 2 = 4 x 32.  Cost = rounds of 256
// workgroups x tile area (ragged edges and a partly filled last round are both paid for); the small tile keeps
// one wave per SIMD busy only a quarter as well, so it is chosen only when it is >1.5x cheaper.
static inline int pick_tile3x3(int N, int H, int W) {
    static const int th[3] = {6, 8, 4}, tw[3] = {96, 64, 32};
    double best = 0; int bi = 0;
    for (int i = 0; i < 3; ++i) {
        const long long wg = (long long)N * ((H + th[i] - 1) / th[i]) * ((W + tw[i] - 1) / tw[i]);
        double cost = (double)((wg + 255) / 256) * th[i] * tw[i] * (i == 2 ? 1.5 : 1.0);
        if (i == 0 || cost < best) { best = cost; bi = i; }
    }
    return bi;
}

// Investigation knobs (ablation switches of tools/ablate_*): they exist only in the ABLATION build of the library
// (`make ablate` -> lib/libsrmi_ablate.so, compiled with -DSRMI_ABLATE; the tools point SRMI_LIB at it), where they are
// read from the environment once per process: SRMI_DBG = bit mask that skips kernel phases (results wrong by
// construction), SRMI_EXP = scheduling experiments (results stay correct), SRMI_THIN_RB = rows per band of the thin
// kernel.  The product library ignores all three: the functions are constant 0 there.
#ifdef SRMI_ABLATE
int dbg_flags();
int exp_flags();
int dbg_thin_rows();
#else
static inline int dbg_flags() { return 0; }
static inline int exp_flags() { return 0; }
static inline int dbg_thin_rows() { return 0; }
#endif

// kernel families (dispatch.cpp decides, the .hip files implement)
enum Kernel { K_GENERIC_F32 = 0, K_C64_BF16 = 1, K_THIN_BF16 = 2, K_K9C3_BF16 = 3, K_CIN_BF16 = 4, K_PW_BF16 = 5, K_C1OUT_BF16 = 6, K_K9C1_BF16 = 7 };
const char* kernel_label(int k);
int select_kernel(const srmi_conv_desc& d);

// generic f32-MFMA kernel (conv_generic.hip)
size_t generic_packed_bytes(const srmi_conv_desc& d);
int generic_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s);
int generic_launch(const srmi_conv_args& a, const ConvGeom& g, hipStream_t s, void* ws = nullptr, size_t ws_bytes = 0,
                   double* bn_stats = nullptr);
long long b3h_tiles(const srmi_conv_desc& d);          // conv_b3h_f32.hip: its workgroups along x = statistics slots
bool b3h_fwd_ok(const srmi_conv_desc& d);
size_t generic_workspace_bytes(const srmi_conv_desc& d, const ConvGeom& g);
const char* generic_kernel_label(const srmi_conv_desc& d, const ConvGeom& g);   // which f32 kernel runs d: b3h_f32 | b3_f32 | f32t | generic_f32

// bf16 MFMA kernel for 3x3 / Cin=64 / Cout%64==0 (conv_c64_bf16.hip)
bool c64_supported(const srmi_conv_desc& d);
size_t c64_packed_bytes(const srmi_conv_desc& d);
int c64_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s);
int c64_launch(const srmi_conv_args& a, const ConvGeom& g, hipStream_t s);

// bf16 MFMA kernel for 3x3 / Cin=64 / Cout<=4 (conv_thin_bf16.hip): the HBM-bound last_conv
bool thin_supported(const srmi_conv_desc& d);
size_t thin_packed_bytes(const srmi_conv_desc& d);
int thin_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s);
int thin_launch(const srmi_conv_args& a, const ConvGeom& g, hipStream_t s);

// bf16 MFMA kernel for 9x9 / Cin<=3 / Cout=64 (conv_k9c3_bf16.hip): conv01
bool k9c3_supported(const srmi_conv_desc& d);
size_t k9c3_packed_bytes(const srmi_conv_desc& d);
int k9c3_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s);
int k9c3_launch(const srmi_conv_args& a, const ConvGeom& g, hipStream_t s);

// bf16 MFMA kernel for 3x3 / Cin % 32 == 0 / Cout % 32 == 0, K-chunked (conv_cin_bf16.hip): dense blocks, wide layers
bool cin_supported(const srmi_conv_desc& d);
size_t cin_packed_bytes(const srmi_conv_desc& d);
int cin_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s);
int cin_launch(const srmi_conv_args& a, const ConvGeom& g, hipStream_t s);

// 9x9 conv of a one-channel image, taps packed on the K axis (conv_k9c1_bf16.hip): SRCNN's first layer
bool k9c1_supported(const srmi_conv_desc& d);
size_t k9c1_packed_bytes(const srmi_conv_desc& d);
int k9c1_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s);
int k9c1_launch(const srmi_conv_args& a, const ConvGeom& g, hipStream_t s);

// HBM-bound SRCNN layer shapes (conv_srcnn_bf16.hip): 1x1 conv as a streaming bf16 GEMM; k x k conv to ONE channel
bool pw_supported(const srmi_conv_desc& d);
size_t pw_packed_bytes(const srmi_conv_desc& d);
int pw_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s);
int pw_launch(const srmi_conv_args& a, const ConvGeom& g, hipStream_t s);
bool c1_supported(const srmi_conv_desc& d);
size_t c1_packed_bytes(const srmi_conv_desc& d);
int c1_pack(const srmi_conv_desc& d, const float* w_oihw, void* w_packed, hipStream_t s);
int c1_launch(const srmi_conv_args& a, const ConvGeom& g, hipStream_t s);

// fused ResBlock (conv_resblock_bf16.hip)
bool resblock_supported(const srmi_conv_desc& d1, const srmi_conv_desc& d2);
int resblock_launch(const srmi_conv_args& a1, const srmi_conv_args& a2, hipStream_t s);
// chain of fused ResBlocks in one persistent launch (conv_chain_bf16.hip)
int chain_run_length(const srmi_conv_args* a, int n, bool (*fusable)(const srmi_conv_args*, const srmi_conv_args*),
                     int* npairs = nullptr, int* kinds = nullptr, int* head = nullptr);
int chain_launch(const srmi_conv_args* a, int head, int npairs, const int* kinds, hipStream_t s);
int chain_status();
// fused generator tail: up conv 64->256 + PixelShuffle(2) + PReLU + last conv 64->3 + tanh (conv_upfuse_bf16.hip)
bool upfuse_supported(const srmi_conv_desc& d1, const srmi_conv_desc& d2);
bool k9pw_supported(const srmi_conv_desc& d1, const srmi_conv_desc& d2);      // conv_k9c1_bf16.hip: 9x9 1 -> 64 + 1x1 64 -> 32
int k9pw_launch(const srmi_conv_args& a1, const srmi_conv_args& a2, hipStream_t s);
int upfuse_launch(const srmi_conv_args& a1, const srmi_conv_args& a2, hipStream_t s);

}  // namespace srmi
