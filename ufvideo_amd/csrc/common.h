// Shared device/host helpers for the gfx950 (MI355X / CDNA4) kernels.  HIP only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/ufv.h"      // the C ABI: every definition of an entry point sees its UFV_API declaration (-fvisibility=hidden)

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_PTR(p) ((void __attribute__((address_space(3)))*)(p))
#define GLB_PTR(p) ((const void __attribute__((address_space(1)))*)(p))

// ---- error plumbing (ufv_last_error) -------------------------------------------------
void ufv_set_error(const char* fmt, ...);
#define UFV_OK 0
#define UFV_EINVAL (-1)
#define UFV_EHIP (-2)
#define UFV_EUNSUPPORTED (-3)

#define UFV_CHECK_LAUNCH()                                                         \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            ufv_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return UFV_EHIP;                                                       \
        }                                                                          \
    } while (0)

#define UFV_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            ufv_set_error(__VA_ARGS__); \
            return UFV_EINVAL;          \
        }                               \
    } while (0)

// Runs BODY once per DEVICE and call site (hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device setting: a process-wide `static bool` left the
// kernels of a second device in the same process without it).  One bit per device ordinal in a 64-bit mask; two threads racing on a device's first launch
// both run BODY, which is idempotent.
#define UFV_ONCE_PER_DEVICE(...)                                          \
    do {                                                                  \
        static std::atomic<uint64_t> done_{0};                            \
        int dev_ = 0;                                                     \
        (void)hipGetDevice(&dev_);                                        \
        const uint64_t bit_ = 1ull << (dev_ & 63);                        \
        if (!(done_.load(std::memory_order_acquire) & bit_)) {            \
            __VA_ARGS__                                                   \
            done_.fetch_or(bit_, std::memory_order_release);              \
        }                                                                 \
    } while (0)

// ---- activations ----------------------------------------------------------------------
enum { ACT_NONE = 0, ACT_GELU_TANH = 1, ACT_GELU_ERF = 2, ACT_SILU = 3, ACT_RELU = 4, ACT_QUICK_GELU = 5, ACT_SIGMOID = 6 };

// GELU (tanh form), 0.5 x (1 + tanh u) with u = sqrt(2/pi) (x + 0.044715 x^3), written as x / (1 + exp(-2u)): the same function (0.5 (1 + tanh u) is the
// logistic of 2u) in one v_exp_f32 + one v_rcp_f32 and no branch.  libm's tanhf takes a polynomial or an exp path by magnitude, so a wave with both kinds of
// lanes ran both: the epilogue of the ViT's fc1 GEMM (80 M activations per call) cost 35-70 us by DATA on a 155 us GEMM (tools/lab/gelu_data_probe.py).
// Round 4: the exponent -2 log2(e) u = x (C0 + C1 x^2) with the constants folded (C0 = -2 log2(e) sqrt(2/pi), C1 = 0.044715 C0): mul, fma, mul instead of
// mul, mul, fma, mul, mul in front of the two transcendentals -- the fc1 epilogue (128 activations per lane and tile, two waves per SIMD) is VALU-bound on this
// function (per-tile timeline: epilogue 11.2 / 16.2 k ticks against 7.4 / 9.2 k for the same tile without it), and the three operations pair up as packed fp32.
__device__ __forceinline__ float gelu_tanh(float x) {
    const float a = x * fmaf(x * x, -0.10294324f, -2.3022082f);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a));
}

// logistic(x) = 1 / (1 + e^-x) as one v_exp_f32 + one v_rcp_f32 (<= 1 ulp from the quotient).  Written as `1 / (1 + __expf(-x))` hipcc emits the IEEE division
// sequence (v_div_scale x 2, v_rcp, 4 fma, v_div_fmas, v_div_fixup): 10 more vector instructions per element, which made the connector's "HBM-bound" LayerNorm + SiLU /
// depthwise + LN + SiLU / LN + add + SiLU kernels (66 M activations per call) and the SwiGLU epilogue of gate/up (64 pairs per lane and tile) VALU-bound (round 4).
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float silu_fast(float x) { return x * sigmoid_fast(x); }
__device__ __forceinline__ float swiglu_f(float g, float u) { return g * sigmoid_fast(g) * u; }      // SwiGLU: silu(gate) * up, every kernel's one formula

__device__ __forceinline__ float act_apply(float x, int act) {
    switch (act) {
        case ACT_GELU_TANH: return gelu_tanh(x);
        case ACT_GELU_ERF: return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
        case ACT_SILU: return silu_fast(x);
        case ACT_RELU: return x > 0.f ? x : 0.f;
        case ACT_QUICK_GELU: return x * sigmoid_fast(1.702f * x);
        case ACT_SIGMOID: return sigmoid_fast(x);
        default: return x;
    }
}

template <int ACT>
__device__ __forceinline__ float act_apply_t(float x) {
    if (ACT == ACT_GELU_TANH) return gelu_tanh(x);
    if (ACT == ACT_GELU_ERF) return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
    if (ACT == ACT_SILU) return silu_fast(x);
    if (ACT == ACT_RELU) return x > 0.f ? x : 0.f;
    if (ACT == ACT_QUICK_GELU) return x * sigmoid_fast(1.702f * x);
    if (ACT == ACT_SIGMOID) return sigmoid_fast(x);
    return x;
}

// run BODY with a compile-time activation id `ACT_` chosen from the runtime value (wave-uniform branch)
#define UFV_ACT_SWITCH(act_, ...)                                        \
    switch (act_) {                                                      \
        case ACT_GELU_TANH: { constexpr int ACT_ = ACT_GELU_TANH; __VA_ARGS__; } break;   \
        case ACT_GELU_ERF: { constexpr int ACT_ = ACT_GELU_ERF; __VA_ARGS__; } break;     \
        case ACT_SILU: { constexpr int ACT_ = ACT_SILU; __VA_ARGS__; } break;             \
        case ACT_RELU: { constexpr int ACT_ = ACT_RELU; __VA_ARGS__; } break;             \
        case ACT_QUICK_GELU: { constexpr int ACT_ = ACT_QUICK_GELU; __VA_ARGS__; } break; \
        case ACT_SIGMOID: { constexpr int ACT_ = ACT_SIGMOID; __VA_ARGS__; } break;       \
        default: { constexpr int ACT_ = ACT_NONE; __VA_ARGS__; } break;                   \
    }

// Row-norm arithmetic with a PINNED operation order, shared by the norm kernels of ops.hip (rmsnorm_k, layernorm_k, layernorm_pipe_k -- which must agree bit
// for bit -- and whatever computes the same norm elsewhere): which multiply-adds hipcc contracts into an fma is otherwise its own choice per kernel (round 5: the
// same source lines compiled as multiply-then-add in ops.hip and as fmas inside a GEMM epilogue experiment, one bf16 ulp apart in places).
// Sums of squares: products rounded, then added left to right; the LayerNorm output: ONE fma of the scaled deviation with weight and bias.
__device__ __forceinline__ float norm_sumsq4(f32x4 v) {
#pragma clang fp contract(off)
    return v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
}
__device__ __forceinline__ float norm_var_acc(float q, float x, float mean) {
#pragma clang fp contract(off)
    const float d = x - mean;
    return q + d * d;
}
__device__ __forceinline__ float norm_ln_out(float x, float mean, float rstd, float w, float b) {
    float t;
    {
#pragma clang fp contract(off)
        t = (x - mean) * rstd;
    }
    return __builtin_fmaf(t, w, b);
}

// MX block scale (round 5): the e8m0 byte of a block of e4m3 activations whose largest magnitude is amax: amax / 448 rounded UP to a power of two, so that no element of
// the block saturates (byte = biased exponent, + 1 when the mantissa is not zero; clamped to [1, 253]; an all-zero block takes 1).  scale = 2^(byte - 127),
// codes = rne_e4m3(x * 2^(127 - byte)).  One definition for the GEMM epilogues (gemm256_kernel.h) and ufv_quantize_mx (quant.hip); oracle.mx_quantize restates it.
__device__ __forceinline__ unsigned mx_scale_byte(float amax) {
    const unsigned b = __builtin_bit_cast(unsigned, amax * (1.0f / 448.0f));
    unsigned e = (b >> 23) + ((b & 0x7fffffu) ? 1u : 0u);
    e = e < 1u ? 1u : (e > 253u ? 253u : e);
    return e;
}

// Rotate-half RoPE of one (x1, x2) = (dim i, dim i + hd/2) pair, in the ONE operation order every forward kernel uses (the prefill table kernel, the
// on-the-fly forms, the decode steps and the QKV GEMM's fused epilogue): y1 = fma(x1, c, -(x2 s)), y2 = fma(x1, s, x2 c).  Written with explicit fmas so that
// the compiler's contraction choice cannot differ from kernel to kernel -- the fused and unfused prefill paths are bit-identical by construction
// (modeling_qwen2.py apply_rotary_pos_emb: q cos + rotate_half(q) sin).
__device__ __forceinline__ void rope_pair(float x1, float x2, float c, float sn, float& y1, float& y2) {
    y1 = __builtin_fmaf(x1, c, -(x2 * sn));
    y2 = __builtin_fmaf(x1, sn, x2 * c);
}

// ---- wave-level reductions (wave = 64 lanes) --------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x <= 1024, `red` = 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
