// HBM-bound kernels of the hot path (gfx950): norms, RoPE + KV-cache store, patchify, the
// projector's depthwise/SE/Conv3d-gather pieces, row gathers for the embedding splice, argmax,
// frame preprocessing.  All: 8-16 byte per-lane accesses, one wave (or block) per row, fp32 math.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "../../include/ufv.h"
#include "gemm_state.h"

// ---- error string ------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void ufv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* ufv_last_error(void) { return g_err; }
extern "C" int ufv_abi_version(void) { return UFV_ABI_VERSION; }

namespace {

// -------------------------------------------------------------------------------------------------
// row loaders: a wave owns a row of D (= 4*nv) elements, lane handles vec4 index lane + 64*i
// -------------------------------------------------------------------------------------------------
constexpr int MAXV = 16;   // D <= 4096

template <int DT>
__device__ __forceinline__ f32x4 load4(const void* p, int64_t idx) {
    if (DT == UFV_DT_F32) return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + idx);
    if (DT == UFV_DT_BF16) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p) + idx);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
    const h4 v = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(p) + idx);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <bool F32>
__device__ __forceinline__ void store4(void* p, int64_t idx, f32x4 v) {
    if (F32)
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p) + idx) = v;
    else
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p) + idx) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
}

// LayerNorm statistics of a row held in registers (two-pass, like torch)
__device__ __forceinline__ void ln_stats(const f32x4* x, int nv, int lane, int D, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nv) s += x[i][0] + x[i][1] + x[i][2] + x[i][3];
    mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) q = norm_var_acc(q, x[i][j], mean);
        }
    rstd = rsqrtf(wave_sum(q) / D + eps);
}

// NV = float4 chunks per lane (row length <= 256 * NV); ACT is a compile-time activation id or -1 = runtime `act`
template <int XDT, bool YF32, int NV, int ACT>
__global__ __launch_bounds__(256) void layernorm_k(const void* x, int ldx, void* y, int ldy, const float* w, const float* b,
                                                   int M, int D, float eps, int act) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int nv = D >> 2;
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < nv) v[i] = load4<XDT>(x, (int64_t)row * ldx + 4 * (lane + 64 * i));
    // statistics: two-pass in registers, like torch
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < nv) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < nv)
#pragma unroll
            for (int j = 0; j < 4; ++j) q = norm_var_acc(q, v[i][j], mean);
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < nv) {
            const int c = 4 * (lane + 64 * i);
            const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
            const f32x4 bb = b ? *reinterpret_cast<const f32x4*>(b + c) : f32x4{0, 0, 0, 0};
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = norm_ln_out(v[i][j], mean, rstd, ww[j], bb[j]);
                o[j] = ACT >= 0 ? act_apply_t<(ACT >= 0 ? ACT : 0)>(t) : act_apply(t, act);
            }
            store4<YF32>(y, (int64_t)row * ldy + c, o);
        }
}

// layernorm_k for bf16 rows of more than 1280 elements (the connector's LayerNorm + SiLU on 18 432 x 3584): the row stays PACKED in registers (2 per vec4 chunk instead of 4;
// the three passes convert again, one shift / and per element) and the affine parameters are staged once per block in LDS and read from there by the block's four rows.
// The element -> lane assignment, every operation and its order are layernorm_k's: bit-identical output (test).  96 -> ~60 registers: 8 waves per SIMD instead of 5.
template <bool YF32, int ACT, bool LDSW = true>
__global__ __launch_bounds__(256) void layernorm_packed_k(const bf16* x, int ldx, void* y, int ldy, const float* w, const float* b, int M, int D, float eps, int act) {
    extern __shared__ __attribute__((aligned(16))) float lnp[];                   // w [D] | b [D]
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nv = D >> 2;
    const bool rok = row < M;
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    u2 pv[MAXV];
    if (rok) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (lane + 64 * i < nv) pv[i] = *reinterpret_cast<const u2*>(x + (int64_t)row * ldx + 4 * (lane + 64 * i));
    }
    if constexpr (LDSW) {
        for (int i = threadIdx.x; i < nv; i += 256) {
            reinterpret_cast<f32x4*>(lnp)[i] = reinterpret_cast<const f32x4*>(w)[i];
            reinterpret_cast<f32x4*>(lnp + D)[i] = b ? reinterpret_cast<const f32x4*>(b)[i] : f32x4{0, 0, 0, 0};
        }
        __syncthreads();
    }
    if (!rok) return;
    auto unpack = [](u2 p) { return f32x4{__builtin_bit_cast(float, p[0] << 16), __builtin_bit_cast(float, p[0] & 0xffff0000u), __builtin_bit_cast(float, p[1] << 16),
                                            __builtin_bit_cast(float, p[1] & 0xffff0000u)}; };
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nv) {
            const f32x4 v = unpack(pv[i]);
            s += v[0] + v[1] + v[2] + v[3];
        }
    const float mean = wave_sum(s) / D;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) asm volatile("" : "+v"(pv[i]));
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nv) {
            const f32x4 v = unpack(pv[i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) q = norm_var_acc(q, v[j], mean);
        }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) asm volatile("" : "+v"(pv[i]));
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nv) {
            if constexpr (LDSW) __builtin_amdgcn_sched_barrier(0);
            const int c = 4 * (lane + 64 * i);
            const f32x4 v = unpack(pv[i]);
            f32x4 ww, bb;
            if constexpr (LDSW) { ww = *reinterpret_cast<const f32x4*>(lnp + c); bb = *reinterpret_cast<const f32x4*>(lnp + D + c); }
            else { ww = *reinterpret_cast<const f32x4*>(w + c); bb = b ? *reinterpret_cast<const f32x4*>(b + c) : f32x4{0, 0, 0, 0}; }
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = norm_ln_out(v[j], mean, rstd, ww[j], bb[j]);
                o[j] = ACT >= 0 ? act_apply_t<(ACT >= 0 ? ACT : 0)>(t) : act_apply(t, act);
            }
            store4<YF32>(y, (int64_t)row * ldy + c, o);
        }
}

// Pipelined form of layernorm_k for long inputs without activation: persistent blocks, a wave walks rows g, g + G, ... and asks for the NEXT row's
// chunks before it reduces and stores the current one (layernorm_k's waves all load, then all reduce, then all store: with ~2 rounds of blocks per CU the
// memory system idles through every reduce phase); weight and bias are staged once per block in LDS.  Same per-lane summation order as layernorm_k:
// bit-identical results.
template <int XDT, bool YF32, int NV>
__global__ __launch_bounds__(256) void layernorm_pipe_k(const void* x, int ldx, void* y, int ldy, const float* w, const float* b, int M, int D, float eps) {
    extern __shared__ __attribute__((aligned(16))) float ln_wb[];
    const int lane = threadIdx.x & 63, nv = D >> 2;
    for (int i = threadIdx.x; i < nv; i += 256) {
        reinterpret_cast<f32x4*>(ln_wb)[i] = reinterpret_cast<const f32x4*>(w)[i];
        reinterpret_cast<f32x4*>(ln_wb + D)[i] = b ? reinterpret_cast<const f32x4*>(b)[i] : f32x4{0, 0, 0, 0};
    }
    __syncthreads();
    const int G = gridDim.x * 4;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x4 v[NV], nx[NV];
    if (row < M) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < nv) v[i] = load4<XDT>(x, (int64_t)row * ldx + 4 * (lane + 64 * i));
    }
    while (row < M) {
        const int nrow = row + G;
        if (nrow < M) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (lane + 64 * i < nv) nx[i] = load4<XDT>(x, (int64_t)nrow * ldx + 4 * (lane + 64 * i));
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < nv) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        const float mean = wave_sum(s) / D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < nv)
#pragma unroll
                for (int j = 0; j < 4; ++j) q = norm_var_acc(q, v[i][j], mean);
        const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < nv) {
                const int c = 4 * (lane + 64 * i);
                const f32x4 ww = *reinterpret_cast<const f32x4*>(ln_wb + c);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(ln_wb + D + c);
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = norm_ln_out(v[i][j], mean, rstd, ww[j], bb[j]);
                store4<YF32>(y, (int64_t)row * ldy + c, o);
            }
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = nx[i];
        row = nrow;
    }
}

// rows stay in registers as bf16 (16-byte loads, 8 channels per chunk): half the registers of an fp32 copy, so more waves are
// resident to hide the three HBM streams (a, b in; out)
template <bool REPACK = false>
__device__ __forceinline__ void ln_stats8(bf16x8* x, int nc, int lane, int D, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (lane + 64 * i < nc)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)x[i][j];
    mean = wave_sum(s) / D;
    if (REPACK) {                       // the second pass converts again from the packed row instead of keeping 56 fp32 values alive across the reduction
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(x[i]));
    }
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (lane + 64 * i < nc)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = (float)x[i][j] - mean;
                q += d * d;
            }
    rstd = rsqrtf(wave_sum(q) / D + eps);
}

// REPACK: the two rows stay PACKED (bf16) in registers between the passes.  Left to itself hipcc keeps the fp32 conversions of both rows alive from the statistics to
// the output loop -- 178 registers, two waves per SIMD for a streaming kernel; an empty asm that "modifies" the packed registers makes every pass convert again (one
// shift / and per element): 116 registers, four waves per SIMD, 101 -> 86 us on 18 432 x 3584 (113 -> 89 with the shortcut's own norm), bit-identical.  (Staging the
// affine parameters in LDS on top of that was measured too: 90 us, and 141 us with four vectors -- 56 KB per block leaves two blocks per CU; not kept.)
template <bool REPACK>
__global__ __launch_bounds__(256) void ln_add_silu_k(const bf16* a, const float* wa, const float* ba, const bf16* b,
                                                     const float* wb, const float* bb, bf16* out, int M, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int nc = D >> 3;
    bf16x8 va[8], vb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (lane + 64 * i < nc) {
            va[i] = *reinterpret_cast<const bf16x8*>(a + (int64_t)row * D + 8 * (lane + 64 * i));
            vb[i] = *reinterpret_cast<const bf16x8*>(b + (int64_t)row * D + 8 * (lane + 64 * i));
        }
    float ma, ra, mb = 0.f, rb = 1.f;
    ln_stats8<REPACK>(va, nc, lane, D, eps, ma, ra);
    if (wb) ln_stats8<REPACK>(vb, nc, lane, D, eps, mb, rb);
    if constexpr (REPACK) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(va[i]), "+v"(vb[i]));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (lane + 64 * i < nc) {
            const int c = 8 * (lane + 64 * i);
            float w1[8], b1[8], w2[8], b2[8];
            *reinterpret_cast<f32x4*>(w1) = *reinterpret_cast<const f32x4*>(wa + c); *reinterpret_cast<f32x4*>(w1 + 4) = *reinterpret_cast<const f32x4*>(wa + c + 4);
            *reinterpret_cast<f32x4*>(b1) = *reinterpret_cast<const f32x4*>(ba + c); *reinterpret_cast<f32x4*>(b1 + 4) = *reinterpret_cast<const f32x4*>(ba + c + 4);
            if (wb) {
                *reinterpret_cast<f32x4*>(w2) = *reinterpret_cast<const f32x4*>(wb + c); *reinterpret_cast<f32x4*>(w2 + 4) = *reinterpret_cast<const f32x4*>(wb + c + 4);
                *reinterpret_cast<f32x4*>(b2) = *reinterpret_cast<const f32x4*>(bb + c); *reinterpret_cast<f32x4*>(b2 + 4) = *reinterpret_cast<const f32x4*>(bb + c + 4);
            }
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = ((float)va[i][j] - ma) * ra * w1[j] + b1[j];
                float s = (float)vb[i][j];
                if (wb) s = (s - mb) * rb * w2[j] + b2[j];
                o[j] = (bf16)act_apply_t<ACT_SILU>(t + s);
            }
            *reinterpret_cast<bf16x8*>(out + (int64_t)row * D + c) = o;
        }
}

template <bool YF32>
__global__ __launch_bounds__(256) void rmsnorm_k(const float* x, int ldx, void* y, int ldy, const float* w, int M, int D,
                                                 float eps) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // blockDim.x / 64 rows per block
    if (row >= M) return;
    const int nv = D >> 2;
    f32x4 v[MAXV];
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nv) {
            v[i] = load4<UFV_DT_F32>(x, (int64_t)row * ldx + 4 * (lane + 64 * i));
            q += norm_sumsq4(v[i]);
        }
    const float r = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nv) {
            const int c = 4 * (lane + 64 * i);
            const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
            store4<YF32>(y, (int64_t)row * ldy + c, f32x4{ww[0] * (v[i][0] * r), ww[1] * (v[i][1] * r), ww[2] * (v[i][2] * r),
                                                          ww[3] * (v[i][3] * r)});
        }
}

// -------------------------------------------------------------------------------------------------
// RoPE (rotate-half) on q in place; k roped + v copied into the KV cache
// one thread per (token, head, pair i < hd/2)
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_kv_k(bf16* qkv, int ldqkv, int S, int Hq, int Hkv, int hd, const float* inv_freq, int pos0,
                                                 bf16* kv, int ldkv) {
    // thread = (token s, head hh of q|k|v, chunk of 8 pairs): hd/2 must be a multiple of 8 (vector path) else scalar loop
    const int half = hd >> 1, cpr = half >> 3;               // chunks per row-half
    const int H = Hq + 2 * Hkv;
    const int64_t total = (int64_t)S * H * cpr;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int ch = id % cpr;
        const int hh = (id / cpr) % H;
        const int s = id / ((int64_t)cpr * H);
        bf16* row = qkv + (int64_t)s * ldqkv;
        const int i0 = ch * 8;
        if (hh < Hq + Hkv) {
            bf16* p = row + hh * hd;
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(p + i0), b2 = *reinterpret_cast<const bf16x8*>(p + half + i0);
            bf16x8 y1, y2;
            const float pos = (float)(pos0 + s);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float ang = pos * inv_freq[i0 + j];
                const float c = cosf(ang), sn = sinf(ang);
                const float x1 = (float)a[j], x2 = (float)b2[j];
                float r1, r2;
                rope_pair(x1, x2, c, sn, r1, r2);
                y1[j] = (bf16)r1;
                y2[j] = (bf16)r2;
            }
            bf16* d = (hh < Hq) ? p : kv + (int64_t)(pos0 + s) * ldkv + (hh - Hq) * hd;
            *reinterpret_cast<bf16x8*>(d + i0) = y1;
            *reinterpret_cast<bf16x8*>(d + half + i0) = y2;
        } else {
            const int hv = hh - Hq - Hkv;
            const bf16* p = row + (Hq + Hkv) * hd + hv * hd;
            bf16* d = kv + (int64_t)(pos0 + s) * ldkv + Hkv * hd + hv * hd;
            *reinterpret_cast<bf16x8*>(d + i0) = *reinterpret_cast<const bf16x8*>(p + i0);
            *reinterpret_cast<bf16x8*>(d + half + i0) = *reinterpret_cast<const bf16x8*>(p + half + i0);
        }
    }
}

// cos / sin of every (position, frequency) once per forward pass: tab[s][i] = cos((pos0+s) * inv_freq[i]), tab[s][half+i] = sin(..)
// (the same cosf / sinf as rope_kv_k, so the table form is bit-identical); every layer and every head re-uses it.
__global__ __launch_bounds__(256) void rope_table_k(const float* inv_freq, int pos0, int S, int half, float* tab) {
    const int64_t total = (int64_t)S * half;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int i = id % half, s = id / half;
        const float ang = (float)(pos0 + s) * inv_freq[i];
        tab[(int64_t)s * 2 * half + i] = cosf(ang);
        tab[(int64_t)s * 2 * half + half + i] = sinf(ang);
    }
}

// rope_kv_k with the angles' cos / sin read from the table (16-byte loads) instead of 16 transcendentals per thread
__global__ __launch_bounds__(256) void rope_kv_tab_k(bf16* qkv, int ldqkv, int S, int Hq, int Hkv, int hd, const float* __restrict__ tab, int pos0,
                                                     bf16* kv, int ldkv) {
    const int half = hd >> 1, cpr = half >> 3;
    const int H = Hq + 2 * Hkv;
    const int64_t total = (int64_t)S * H * cpr;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int ch = id % cpr;
        const int hh = (id / cpr) % H;
        const int s = id / ((int64_t)cpr * H);
        bf16* row = qkv + (int64_t)s * ldqkv;
        const int i0 = ch * 8;
        if (hh < Hq + Hkv) {
            bf16* p = row + hh * hd;
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(p + i0), b2 = *reinterpret_cast<const bf16x8*>(p + half + i0);
            const float* tr = tab + (int64_t)s * hd;
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(tr + i0), c1 = *reinterpret_cast<const f32x4*>(tr + i0 + 4);
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(tr + half + i0), s1 = *reinterpret_cast<const f32x4*>(tr + half + i0 + 4);
            bf16x8 y1, y2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float c = j < 4 ? c0[j & 3] : c1[j & 3], sn = j < 4 ? s0[j & 3] : s1[j & 3];
                const float x1 = (float)a[j], x2 = (float)b2[j];
                float r1, r2;
                rope_pair(x1, x2, c, sn, r1, r2);
                y1[j] = (bf16)r1;
                y2[j] = (bf16)r2;
            }
            bf16* d = (hh < Hq) ? p : kv + (int64_t)(pos0 + s) * ldkv + (hh - Hq) * hd;
            *reinterpret_cast<bf16x8*>(d + i0) = y1;
            *reinterpret_cast<bf16x8*>(d + half + i0) = y2;
        } else {
            const int hv = hh - Hq - Hkv;
            const bf16* p = row + (Hq + Hkv) * hd + hv * hd;
            bf16* d = kv + (int64_t)(pos0 + s) * ldkv + Hkv * hd + hv * hd;
            *reinterpret_cast<bf16x8*>(d + i0) = *reinterpret_cast<const bf16x8*>(p + i0);
            *reinterpret_cast<bf16x8*>(d + half + i0) = *reinterpret_cast<const bf16x8*>(p + half + i0);
        }
    }
}

// single new token (decode) with its position read from device memory: same arithmetic as rope_kv_scalar_k, launch arguments
// independent of the position (HIP-graph capture of the decode step)
__global__ void rope_kv1_dev_k(bf16* qkv, int Hq, int Hkv, int hd, const float* inv_freq, const int* __restrict__ pos_dev, bf16* kv, int ldkv) {
    const int half = hd >> 1;
    const int pos = *pos_dev;
    const int total = (Hq + 2 * Hkv) * half;
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < total; id += gridDim.x * blockDim.x) {
        const int i = id % half, hh = id / half;
        if (hh < Hq + Hkv) {
            bf16* p = qkv + hh * hd;
            const float ang = (float)pos * inv_freq[i];
            const float c = cosf(ang), sn = sinf(ang);
            const float x1 = (float)p[i], x2 = (float)p[half + i];
            bf16* d = (hh < Hq) ? p : kv + (int64_t)pos * ldkv + (hh - Hq) * hd;
            float r1, r2;
            rope_pair(x1, x2, c, sn, r1, r2);
            d[i] = (bf16)r1;
            d[half + i] = (bf16)r2;
        } else {
            const int hv = hh - Hq - Hkv;
            const bf16* p = qkv + (Hq + Hkv) * hd + hv * hd;
            bf16* d = kv + (int64_t)pos * ldkv + Hkv * hd + hv * hd;
            d[i] = p[i]; d[half + i] = p[half + i];
        }
    }
}

__global__ void add_int_k(int* p, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) *p += v; }

__global__ void rope_kv_scalar_k(bf16* qkv, int ldqkv, int S, int Hq, int Hkv, int hd, const float* inv_freq, int pos0, bf16* kv,
                                 int ldkv) {
    const int half = hd >> 1;
    const int64_t total = (int64_t)S * (Hq + 2 * Hkv) * half;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int i = id % half;
        const int hh = (id / half) % (Hq + 2 * Hkv);
        const int s = id / ((int64_t)half * (Hq + 2 * Hkv));
        bf16* row = qkv + (int64_t)s * ldqkv;
        if (hh < Hq + Hkv) {
            const float ang = (float)(pos0 + s) * inv_freq[i];
            const float c = cosf(ang), sn = sinf(ang);
            bf16* p = row + hh * hd;
            const float x1 = (float)p[i], x2 = (float)p[i + half];
            float y1, y2;
            rope_pair(x1, x2, c, sn, y1, y2);
            bf16* d = (hh < Hq) ? p : kv + (int64_t)(pos0 + s) * ldkv + (hh - Hq) * hd;
            d[i] = (bf16)y1;
            d[i + half] = (bf16)y2;
        } else {
            const int hv = hh - Hq - Hkv;
            const bf16* p = row + (Hq + Hkv) * hd + hv * hd;
            bf16* d = kv + (int64_t)(pos0 + s) * ldkv + Hkv * hd + hv * hd;
            d[i] = p[i];
            d[i + half] = p[i + half];
        }
    }
}

// -------------------------------------------------------------------------------------------------
// patchify: pixels [T,C,H,W] -> [T*gh*gw, Kpad], k = c*P*P + py*P + px.  One block per (patch row of
// the image, frame): reads are contiguous image rows (coalesced NCHW), writes contiguous k runs.
// -------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void patchify_k(const void* px, bf16* out, int T, int C, int H, int W, int P, int Kpad) {
    const int gw = W / P, gh = H / P;
    const int gy = blockIdx.x, t = blockIdx.y;
    const int K = C * P * P;
    // elements of this patch row: C * P (py) * W (x) ; x = gx*P + pxl
    const int n = C * P * W;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int x = e % W, py = (e / W) % P, c = e / (W * P);
        if (x >= gw * P) continue;                                    // W % P pixels on the right belong to no patch (a stride-P convolution without padding: 384 = 27 x 14 + 6)
        const int64_t src = (((int64_t)t * C + c) * H + (gy * P + py)) * W + x;
        float v;
        if (DT == UFV_DT_F32) v = reinterpret_cast<const float*>(px)[src];
        else if (DT == UFV_DT_BF16) v = (float)reinterpret_cast<const bf16*>(px)[src];
        else v = (float)reinterpret_cast<const _Float16*>(px)[src];
        const int gx = x / P, pxl = x % P;
        out[((int64_t)(t * gh + gy) * gw + gx) * Kpad + c * P * P + py * P + pxl] = (bf16)v;
    }
    // zero pad columns
    const int padn = Kpad - K;
    for (int e = threadIdx.x; e < gw * padn; e += blockDim.x) {
        const int gx = e / padn, k = K + e % padn;
        out[((int64_t)(t * gh + gy) * gw + gx) * Kpad + k] = (bf16)0.f;
    }
}

// The same rearrangement by RUNS for even P, W: one thread moves the P pixels of one (channel, patch row, patch) -- contiguous in the image row and in the patch's k
// range -- as P / 2 dwords (the scalar form above computes three divisions per 2-byte element: 42 us per 32-frame clip, 1 TB/s).  Same values, same rounding.
template <int DT>
__global__ __launch_bounds__(256) void patchify_runs_k(const void* px, bf16* out, int T, int C, int H, int W, int P, int Kpad) {
    const int gw = W / P, gh = H / P;
    const int gy = blockIdx.x, t = blockIdx.y;
    const int K = C * P * P, nrun = C * P * gw, hp = P >> 1;
    for (int r = threadIdx.x; r < nrun; r += blockDim.x) {
        const int py = r % P, c = (r / P) % C, gx = r / (P * C);                   // consecutive threads: consecutive k runs of ONE patch (contiguous stores; the image rows come from L2)
        const int64_t src = (((int64_t)t * C + c) * H + (gy * P + py)) * W + gx * P;
        unsigned* d = reinterpret_cast<unsigned*>(out + ((int64_t)(t * gh + gy) * gw + gx) * Kpad + c * P * P + py * P);
        if (DT == UFV_DT_BF16) {
            const unsigned* sp = reinterpret_cast<const unsigned*>(reinterpret_cast<const bf16*>(px) + src);
            for (int i = 0; i < hp; ++i) d[i] = sp[i];
        } else {
            const float2* sp = reinterpret_cast<const float2*>(reinterpret_cast<const float*>(px) + src);
            for (int i = 0; i < hp; ++i) {
                const float2 v = sp[i];
                const bf16x2 o = {(bf16)v.x, (bf16)v.y};
                d[i] = __builtin_bit_cast(unsigned, o);
            }
        }
    }
    const int padn = Kpad - K;
    for (int e = threadIdx.x; e < gw * padn; e += blockDim.x) {
        const int gx = e / padn, k = K + e % padn;
        out[((int64_t)(t * gh + gy) * gw + gx) * Kpad + k] = (bf16)0.f;
    }
}

// -------------------------------------------------------------------------------------------------
// depthwise 3x3 (pad 1) + LayerNorm(C) + SiLU, NHWC.  One block = 4 consecutive pixels of a row x all channels;
// wave w owns a quarter of the 8-channel chunks, so every weight chunk is loaded once per 4 pixels and every input
// column is loaded once for the 3 horizontal taps that use it; LayerNorm statistics are combined across the waves in LDS.
// -------------------------------------------------------------------------------------------------
// Block order: the hardware deals consecutive workgroups round-robin over the 8 XCDs, each with its own L2.  An output row reads three input rows and a segment two
// halo columns of its neighbours, so with REMAP block b works on logical index (b % 8) * (n / 8) + b / 8: every XCD walks its own contiguous range of
// (frame, row, segment) and finds the rows its previous blocks fetched in ITS L2 instead of fetching them again over the fabric.
template <int PX, bool REMAP>
__global__ __launch_bounds__(256) void dwconv_ln_silu_k(const bf16* __restrict__ x, bf16* __restrict__ y, const float* __restrict__ w9,
                                                        const float* __restrict__ lnw, const float* __restrict__ lnb, int F, int H, int W,
                                                        int C, float eps, int nblk) {
    constexpr int MAXI = 2;                             // C <= 4096 -> <= 128 chunks per wave -> <= 2 per lane
    __shared__ float red[2][4][PX];
    const int nseg = (W + PX - 1) / PX;
    int bid = blockIdx.x;
    if (REMAP) {
        const int per = (nblk + 7) >> 3;
        bid = (bid & 7) * per + (bid >> 3);
        if (bid >= nblk) return;
    }
    const int seg = bid % nseg, py = (bid / nseg) % H, f = bid / (nseg * H);
    const int x0 = seg * PX, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nc = C >> 3, cpw = (nc + 3) >> 2;
    int cidx[MAXI];
    bool cok[MAXI];
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int q = lane + 64 * i;
        cok[i] = q < cpw && wave * cpw + q < nc;
        cidx[i] = 8 * (wave * cpw + q);
    }
    float acc[MAXI][PX][8];
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
#pragma unroll
        for (int p = 0; p < PX; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][p][j] = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = py + dy;
        if (yy < 0 || yy >= H) continue;
        float wt[3][MAXI][8];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int i = 0; i < MAXI; ++i)
                if (cok[i]) {
                    const float* wk = w9 + ((dy + 1) * 3 + dx) * C + cidx[i];
                    const f32x4 a = *reinterpret_cast<const f32x4*>(wk), b = *reinterpret_cast<const f32x4*>(wk + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { wt[dx][i][j] = a[j]; wt[dx][i][4 + j] = b[j]; }
                }
        const bf16* row = x + ((int64_t)(f * H + yy) * W) * C;
#pragma unroll
        for (int col = -1; col <= PX; ++col) {
            const int xx = x0 + col;
            if (xx < 0 || xx >= W) continue;
#pragma unroll
            for (int i = 0; i < MAXI; ++i)
                if (cok[i]) {
                    const bf16x8 xv = *reinterpret_cast<const bf16x8*>(row + (int64_t)xx * C + cidx[i]);
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int p = col - (dx - 1);           // output pixel that sees this column through tap dx
                        if (p >= 0 && p < PX)
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[i][p][j] += (float)xv[j] * wt[dx][i][j];
                    }
                }
        }
    }
    // LayerNorm over C per pixel (mean, then centred variance), statistics combined across the 4 waves
    float mean[PX], rstd[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
            if (cok[i])
#pragma unroll
                for (int j = 0; j < 8; ++j) s += acc[i][p][j];
        s = wave_sum(s);
        if (lane == 0) red[0][wave][p] = s;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PX; ++p) mean[p] = (red[0][0][p] + red[0][1][p] + red[0][2][p] + red[0][3][p]) / C;
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
            if (cok[i])
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = acc[i][p][j] - mean[p];
                    q += d * d;
                }
        q = wave_sum(q);
        if (lane == 0) red[1][wave][p] = q;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PX; ++p) rstd[p] = rsqrtf((red[1][0][p] + red[1][1][p] + red[1][2][p] + red[1][3][p]) / C + eps);
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
        if (cok[i]) {
            const int c = cidx[i];
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(lnw + c), g1 = *reinterpret_cast<const f32x4*>(lnw + c + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(lnb + c), b1 = *reinterpret_cast<const f32x4*>(lnb + c + 4);
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                if (x0 + p >= W) continue;
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = (bf16)act_apply_t<ACT_SILU>((acc[i][p][j] - mean[p]) * rstd[p] * g0[j] + b0[j]);
                    o[4 + j] = (bf16)act_apply_t<ACT_SILU>((acc[i][p][4 + j] - mean[p]) * rstd[p] * g1[j] + b1[j]);
                }
                *reinterpret_cast<bf16x8*>(y + ((int64_t)(f * H + py) * W + x0 + p) * C + c) = o;
            }
        }
}

// out[f, c] = mean_p x[f*P + p, c].  Block = 8 row groups x 64 lanes, lane owns 8 channels (16-byte loads, 512 channels per
// block); each row group walks its rows four at a time so that several loads are in flight; groups combine through LDS.
__global__ __launch_bounds__(512) void colmean_k(const bf16* __restrict__ x, bf16* __restrict__ out, int F, int P, int C) {
    __shared__ float part[8][64][8];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 8, f = blockIdx.y;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    if (c < C) {
        const bf16* p = x + (int64_t)f * P * C + c;
        int r = g;
        for (; r + 24 < P; r += 32) {
            const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(p + (int64_t)r * C), v1 = *reinterpret_cast<const bf16x8*>(p + (int64_t)(r + 8) * C);
            const bf16x8 v2 = *reinterpret_cast<const bf16x8*>(p + (int64_t)(r + 16) * C), v3 = *reinterpret_cast<const bf16x8*>(p + (int64_t)(r + 24) * C);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += ((float)v0[j] + (float)v1[j]) + ((float)v2[j] + (float)v3[j]);
        }
        for (; r < P; r += 8) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(p + (int64_t)r * C);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[g][lane][j] = s[j];
    __syncthreads();
    if (g == 0 && c < C) {
        const float inv = 1.0f / P;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = s[j];
#pragma unroll
            for (int i = 1; i < 8; ++i) t += part[i][lane][j];
            o[j] = (bf16)(t * inv);
        }
        *reinterpret_cast<bf16x8*>(out + (int64_t)f * C + c) = o;
    }
}

__global__ __launch_bounds__(256) void scale_channels_k(bf16* x, const bf16* gate, int F, int P, int C) {
    const int64_t nv = (int64_t)F * P * C / 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i * 8;
        const int c = e % C;
        const int f = e / ((int64_t)P * C);
        bf16x8 v = *reinterpret_cast<bf16x8*>(x + e);
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(gate + (int64_t)f * C + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] * (float)g[j]);
        *reinterpret_cast<bf16x8*>(x + e) = v;
    }
}

// AvgPool3d(k = stride = (kt,kh,kw), no padding, floor) + SiLU over token-major x [T,H,W,C] (STP / spatial_pool sampler)
template <int ACT>
__global__ __launch_bounds__(256) void avgpool3d_silu_k(const bf16* x, bf16* out, int T, int H, int W, int C, int kt, int kh, int kw,
                                                        int To, int Ho, int Wo) {
    const int cv = C / 8;
    const int64_t total = (int64_t)To * Ho * Wo * cv;
    const float inv = 1.0f / (kt * kh * kw);
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(id % cv) * 8;
        const int64_t p = id / cv;
        const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), to = (int)(p / ((int64_t)Wo * Ho));
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int dt = 0; dt < kt; ++dt)
            for (int dh = 0; dh < kh; ++dh)
                for (int dw = 0; dw < kw; ++dw) {
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (((int64_t)(to * kt + dt) * H + ho * kh + dh) * W + wo * kw + dw) * C + c);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
                }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)act_apply_t<ACT>(acc[j] * inv);
        *reinterpret_cast<bf16x8*>(out + p * C + c) = o;
    }
}

// Conv3d gather: out row (to,ho,wo), col ((dt*kh+dh)*kw+dw)*C + c
__global__ __launch_bounds__(256) void conv3d_gather_k(const bf16* x, bf16* out, int T, int H, int W, int C, int kt, int kh,
                                                       int kw, int pad, int To, int Ho, int Wo) {
    const int cv = C / 8;
    const int64_t total = (int64_t)To * Ho * Wo * kt * kh * kw * cv;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = id % cv;
        int64_t r = id / cv;
        const int dw = r % kw; r /= kw;
        const int dh = r % kh; r /= kh;
        const int dt = r % kt; r /= kt;
        const int wo = r % Wo; r /= Wo;
        const int ho = r % Ho; r /= Ho;
        const int to = (int)r;
        const int t = to * kt + dt - pad, y = ho * kh + dh - pad, xx = wo * kw + dw - pad;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t >= 0 && t < T && y >= 0 && y < H && xx >= 0 && xx < W)
            v = *reinterpret_cast<const bf16x8*>(x + (((int64_t)t * H + y) * W + xx) * C + c8 * 8);
        *reinterpret_cast<bf16x8*>(out + id * 8) = v;
    }
}

// generic row gather/scatter with dtype conversion: one wave per row
template <int SDT, bool DF32>
__global__ __launch_bounds__(256) void gather_rows_k(const void* src, int64_t lds_, const int64_t* sidx, void* dst, int64_t ldd,
                                                     const int64_t* didx, int n, int D) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    const int64_t sr = sidx ? sidx[r] : r, dr = didx ? didx[r] : r;
    if (dr < 0) return;
    for (int c = lane; c < D; c += 64) {
        float v;
        if (sr < 0) v = 0.f;                      // padding row (window partition past the image edge)
        else if (SDT == UFV_DT_F32) v = reinterpret_cast<const float*>(src)[sr * lds_ + c];
        else if (SDT == UFV_DT_BF16) v = (float)reinterpret_cast<const bf16*>(src)[sr * lds_ + c];
        else v = (float)reinterpret_cast<const _Float16*>(src)[sr * lds_ + c];
        if (DF32) reinterpret_cast<float*>(dst)[dr * ldd + c] = v;
        else reinterpret_cast<bf16*>(dst)[dr * ldd + c] = (bf16)v;
    }
}

// the same copy with 8 elements per lane and access (rows 16-byte aligned, D % 8 == 0): RW = waves per row -- 1: four rows per block; 4: one row per block (the decode
// step's single embedding row was 56 one-element iterations of ONE wave: 25 us for 7 KB)
template <int SDT, bool DF32, int RW>
__global__ __launch_bounds__(256) void gather_rows8_k(const void* src, int64_t lds_, const int64_t* sidx, void* dst, int64_t ldd, const int64_t* didx, int n, int D) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = RW == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
    if (r >= n) return;
    const int64_t sr = sidx ? sidx[r] : r, dr = didx ? didx[r] : r;
    if (dr < 0) return;
    for (int c = (RW == 1 ? lane : threadIdx.x) * 8; c < D; c += 512 * RW) {
        float v[8];
        if (sr < 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
        } else if (SDT == UFV_DT_F32) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(src) + sr * lds_ + c), b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(src) + sr * lds_ + c + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
        } else if (SDT == UFV_DT_BF16) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(src) + sr * lds_ + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
        } else {
            typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
            const f16x8 a = *reinterpret_cast<const f16x8*>(reinterpret_cast<const _Float16*>(src) + sr * lds_ + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
        }
        if (DF32) {
            float* o = reinterpret_cast<float*>(dst) + dr * ldd + c;
            *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)v[j];
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(dst) + dr * ldd + c) = o;
        }
    }
}

// masked mean pooling: out[i, c] = sum_p feat[frame_of[i], p, c] * mask[i, p] / (sum_p mask[i,p] + 1e-8)
template <int DT>
__global__ __launch_bounds__(256) void mask_pool_k(const void* feat, const float* mask, const int32_t* frame_of, float* out,
                                                   int P, int C) {
    __shared__ float red[16];
    const int i = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    const float* m = mask + (int64_t)i * P;
    float cnt = 0.f;
    for (int p = threadIdx.x; p < P; p += blockDim.x) cnt += m[p];
    cnt = block_sum(cnt, red) + 1e-8f;
    if (c >= C) return;
    const int f = frame_of[i];
    float s = 0.f;
    for (int p = 0; p < P; ++p) {
        const float mv = m[p];
        if (mv != 0.f) {
            const int64_t idx = ((int64_t)f * P + p) * C + c;
            float v;
            if (DT == UFV_DT_F32) v = reinterpret_cast<const float*>(feat)[idx];
            else v = (float)reinterpret_cast<const bf16*>(feat)[idx];
            s += v * mv / cnt;      // same association as the reference: (x * mask / denorm).sum()
        }
    }
    out[(int64_t)i * C + c] = s;
}

// argmax with lowest-index tie-break (torch.argmax); single block
__global__ __launch_bounds__(1024) void argmax_k(const float* x, int N, int64_t* out) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    float best = -INFINITY;
    int idx = 0x7fffffff;
    auto take = [&](float v, int i) {
        if (v > best || (v == best && i < idx) || (v != v && !(best != best))) {   // NaN wins like torch
            best = v;
            idx = i;
        }
    };
    // 16-byte loads, four of them in flight per thread (one scalar load per iteration was a 148-deep latency chain: 79 us for
    // the 151 748 logits of the decode step); a thread visits its elements in increasing index order, so ties keep the lowest
    const int n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? (N >> 2) : 0;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    int c = threadIdx.x;
    for (; c + 3 * (int)blockDim.x < n4; c += 4 * blockDim.x) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = x4[c + u * blockDim.x];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) take(v[u][j], 4 * (c + u * blockDim.x) + j);
    }
    for (; c < n4; c += blockDim.x) {
        const f32x4 v = x4[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) take(v[j], 4 * c + j);
    }
    for (int i = 4 * n4 + threadIdx.x; i < N; i += blockDim.x) take(x[i], i);
    auto better = [](float v, int i, float bv_, int bi_) {
        const bool vn = v != v, bn = bv_ != bv_;
        if (vn != bn) return vn;
        if (vn && bn) return i < bi_;
        return v > bv_ || (v == bv_ && i < bi_);
    };
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (better(ov, oi, best, idx)) { best = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (better(bv[w], bi[w], best, idx)) { best = bv[w]; idx = bi[w]; }
        out[0] = idx;
    }
}

// The same arg-max over 64 blocks (one pass of 16-byte loads per block, all requested at once) for the decode step: block partials go to `ws` with agent-scope stores and
// the block that arrives last at the counter merges them (ws: 64 floats, 64 ints, one counter that is ZERO before the first call and returns to zero).  Same order
// relation as argmax_k (NaN beats everything, ties keep the lowest index): the result does not depend on the partition.  (One 1024-thread block walked the 600 KB of
// logits in 10 dependent round trips: 27 us per token.)
constexpr int ARGMAX_NB = 64;
__global__ __launch_bounds__(256) void argmax_blocks_k(const float* x, int N, int64_t* out, float* ws) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    __shared__ int last_flag;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    auto better = [](float v, int i, float bv_, int bi_) {
        const bool vn = v != v, bn = bv_ != bv_;
        if (vn != bn) return vn;
        if (vn && bn) return i < bi_;
        return v > bv_ || (v == bv_ && i < bi_);
    };
    const int per = ((N + ARGMAX_NB - 1) / ARGMAX_NB + 3) & ~3;           // a multiple of 4: block ranges start on 16-byte boundaries
    const int lo = blockIdx.x * per, hi = min(N, lo + per);
    constexpr int MAXV = 4;                                                // 16-byte loads per thread in flight (N <= 64 * 256 * 4 * MAXV = 262144 in one pass)
    f32x4 v[MAXV];
    const int n4 = hi > lo ? (hi - lo) >> 2 : 0;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x + lo);
    for (int c0 = threadIdx.x; c0 < n4; c0 += 256 * MAXV) {
#pragma unroll
        for (int u = 0; u < MAXV; ++u) v[u] = x4[min(c0 + u * 256, n4 - 1)];
#pragma unroll
        for (int u = 0; u < MAXV; ++u)
            if (c0 + u * 256 < n4)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (better(v[u][j], lo + 4 * (c0 + u * 256) + j, best, idx)) { best = v[u][j]; idx = lo + 4 * (c0 + u * 256) + j; }
    }
    for (int i = lo + 4 * n4 + threadIdx.x; i < hi; i += 256)
        if (better(x[i], i, best, idx)) { best = x[i]; idx = i; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (better(ov, oi, best, idx)) { best = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    int* wi = reinterpret_cast<int*>(ws + ARGMAX_NB);
    int* counter = wi + ARGMAX_NB;
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (better(bv[w], bi[w], best, idx)) { best = bv[w]; idx = bi[w]; }
        __hip_atomic_store(ws + blockIdx.x, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(wi + blockIdx.x, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // write-through stores acknowledged = visible device-wide
        last_flag = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ARGMAX_NB - 1;
    }
    __syncthreads();
    if (!last_flag || threadIdx.x >= 64) return;
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float pv;
    int pi;
    asm volatile("global_load_dword %0, %2, off sc0 sc1\n\tglobal_load_dword %1, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(pv), "=&v"(pi) : "v"(ws + threadIdx.x), "v"(wi + threadIdx.x) : "memory");
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(pv, o, 64);
        const int oi = __shfl_xor(pi, o, 64);
        if (better(ov, oi, pv, pi)) { pv = ov; pi = oi; }
    }
    if (threadIdx.x == 0) out[0] = pi;
}

__global__ __launch_bounds__(256) void preprocess_u8_k(const uint8_t* fr, bf16* out, int T, int H, int W, float m0, float m1,
                                                       float m2, float s0, float s1, float s2) {
    const int64_t total = (int64_t)T * 3 * H * W;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int x = id % W, y = (id / W) % H, c = (id / ((int64_t)W * H)) % 3;
        const int t = id / ((int64_t)3 * W * H);
        const float v = (float)fr[(((int64_t)t * H + y) * W + x) * 3 + c] * (1.0f / 255.0f);
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        out[id] = (bf16)((v - mean) / sd);
    }
}

template <int SDT, int DDT>
__global__ __launch_bounds__(256) void convert_k(const void* s, void* d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v;
        if (SDT == UFV_DT_F32) v = reinterpret_cast<const float*>(s)[i];
        else if (SDT == UFV_DT_BF16) v = (float)reinterpret_cast<const bf16*>(s)[i];
        else v = (float)reinterpret_cast<const _Float16*>(s)[i];
        if (DDT == UFV_DT_F32) reinterpret_cast<float*>(d)[i] = v;
        else if (DDT == UFV_DT_BF16) reinterpret_cast<bf16*>(d)[i] = (bf16)v;
        else reinterpret_cast<_Float16*>(d)[i] = (_Float16)v;
    }
}

// general im2col for Conv2d(k, stride, pad): pixels [B,C,H,W] -> bf16 [B*Ho*Wo, Kpad], k = c*ks*ks + ky*ks + kx
template <int DT>
__global__ __launch_bounds__(256) void im2col_k(const void* px, bf16* out, int B, int C, int H, int W, int ks, int stride, int pad,
                                                int Ho, int Wo, int Kpad) {
    const int K = C * ks * ks;
    const int64_t total = (int64_t)B * Ho * Wo * Kpad;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int k = id % Kpad;
        const int64_t row = id / Kpad;
        float v = 0.f;
        if (k < K) {
            const int kx = k % ks, ky = (k / ks) % ks, c = k / (ks * ks);
            const int ox = row % Wo, oy = (row / Wo) % Ho, b = row / ((int64_t)Wo * Ho);
            const int y = oy * stride + ky - pad, x = ox * stride + kx - pad;
            if (y >= 0 && y < H && x >= 0 && x < W) {
                const int64_t src = (((int64_t)b * C + c) * H + y) * W + x;
                if (DT == UFV_DT_F32) v = reinterpret_cast<const float*>(px)[src];
                else if (DT == UFV_DT_BF16) v = (float)reinterpret_cast<const bf16*>(px)[src];
                else v = (float)reinterpret_cast<const _Float16*>(px)[src];
            }
        }
        out[id] = (bf16)v;
    }
}

// 2x2/s2 max pool over a token grid: x rows [Bw*H*W] (pitch ldx) x C -> out rows [Bw*(H/2)*(W/2)] (pitch ldo)
template <int DT>
__global__ __launch_bounds__(256) void maxpool2x2_k(const void* x, int64_t ldx, void* out, int64_t ldo, int Bw, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)Bw * Ho * Wo * C;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int c = id % C;
        const int64_t r = id / C;
        const int ox = r % Wo, oy = (r / Wo) % Ho, b = r / ((int64_t)Wo * Ho);
        const int64_t r00 = ((int64_t)b * H + 2 * oy) * W + 2 * ox;
        float m;
        if (DT == UFV_DT_F32) {
            const float* p = reinterpret_cast<const float*>(x);
            m = fmaxf(fmaxf(p[r00 * ldx + c], p[(r00 + 1) * ldx + c]), fmaxf(p[(r00 + W) * ldx + c], p[(r00 + W + 1) * ldx + c]));
            reinterpret_cast<float*>(out)[r * ldo + c] = m;
        } else {
            const bf16* p = reinterpret_cast<const bf16*>(x);
            m = fmaxf(fmaxf((float)p[r00 * ldx + c], (float)p[(r00 + 1) * ldx + c]),
                      fmaxf((float)p[(r00 + W) * ldx + c], (float)p[(r00 + W + 1) * ldx + c]));
            reinterpret_cast<bf16*>(out)[r * ldo + c] = (bf16)m;
        }
    }
}

// dst[didx[i]][:] += src[i][:]  (fp32 dst; rows with didx < 0 are padding and skipped); one wave per row
template <int SDT>
__global__ __launch_bounds__(256) void add_rows_k(const void* src, int64_t lds_, float* dst, int64_t ldd, const int64_t* didx, int n, int D) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    const int64_t dr = didx ? didx[r] : r;
    if (dr < 0) return;
    for (int c = lane; c < D; c += 64) {
        const float v = (SDT == UFV_DT_F32) ? reinterpret_cast<const float*>(src)[(int64_t)r * lds_ + c]
                                            : (float)reinterpret_cast<const bf16*>(src)[(int64_t)r * lds_ + c];
        dst[dr * ldd + c] += v;
    }
}

// FPN top-down: x[b, y, x, :] += prev[b, y/2, x/2, :]   (nearest x2 upsample, fp32 NHWC)
__global__ __launch_bounds__(256) void upsample2x_add_k(float* x, const float* prev, int B, int H, int W, int C) {
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int c = id % C;
        const int64_t r = id / C;
        const int xx = r % W, yy = (r / W) % H, b = r / ((int64_t)W * H);
        x[id] += prev[(((int64_t)b * (H / 2) + yy / 2) * (W / 2) + xx / 2) * C + c];
    }
}

inline int grid_for(int64_t n, int per_block = 256) {
    const int64_t g = (n + per_block - 1) / per_block;
    return (int)(g < 8192 ? g : 8192);
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int ufv_layernorm(const void* x, int x_dtype, int ldx, void* y, int y_f32, int ldy, const float* w, const float* b,
                             int M, int D, float eps, int act, void* stream) {
    UFV_REQUIRE(x && y && w && M > 0 && D > 0, "ufv_layernorm: bad arguments");
    UFV_REQUIRE(D % 4 == 0 && D <= 4 * 64 * MAXV && ldx % 4 == 0 && ldy % 4 == 0, "ufv_layernorm: D=%d must be a multiple of 4 and <= %d", D, 4 * 64 * MAXV);
    dim3 g(cdiv(M, 4)), blk(256);
    // long inputs without activation (the ViT's 18 432 x 1152 stream): the pipelined persistent form, >= 4 rows per wave, <= 4 blocks per CU
    // (measured on 18 432 x 1152 fp32 -> bf16: 30.3 -> 22.9 us from cache, 38.5 -> 27.4 us from HBM; 2 / 3 / 4 blocks per CU within 2 %)
    if (act == ACT_NONE && D <= 1280 && !y_f32 && M >= 4096 && (x_dtype == UFV_DT_F32 || x_dtype == UFV_DT_BF16)) {
        const int blocks = M / 16 < 1024 ? M / 16 : 1024;
        if (x_dtype == UFV_DT_F32)
            hipLaunchKernelGGL((layernorm_pipe_k<UFV_DT_F32, false, 5>), dim3(blocks), blk, 2 * D * sizeof(float), ST(stream), x, ldx, y, ldy, w, b, M, D, eps);
        else
            hipLaunchKernelGGL((layernorm_pipe_k<UFV_DT_BF16, false, 5>), dim3(blocks), blk, 2 * D * sizeof(float), ST(stream), x, ldx, y, ldy, w, b, M, D, eps);
        UFV_CHECK_LAUNCH();
        return UFV_OK;
    }
    static const bool no_packed = getenv("UFV_LN_NO_PACKED") != nullptr;            // same-box A/B switch
    if (!no_packed && x_dtype == UFV_DT_BF16 && D > 1280 && (uintptr_t)w % 16 == 0 && (!b || (uintptr_t)b % 16 == 0)) {
        const size_t par = 2 * (size_t)D * sizeof(float);
        // (the same kernel with the parameters as global loads in the output loop, LDSW = false: 416 us instead of 62 -- hipcc serialises them behind the packed-row fences)
#define LNP_LAUNCH(YF, ACT_) hipLaunchKernelGGL((layernorm_packed_k<YF, ACT_, true>), g, blk, par, ST(stream), (const bf16*)x, ldx, y, ldy, w, b, M, D, eps, act)
        if (y_f32) { if (act == ACT_NONE) LNP_LAUNCH(true, ACT_NONE); else if (act == ACT_SILU) LNP_LAUNCH(true, ACT_SILU); else LNP_LAUNCH(true, -1); }
        else { if (act == ACT_NONE) LNP_LAUNCH(false, ACT_NONE); else if (act == ACT_SILU) LNP_LAUNCH(false, ACT_SILU); else LNP_LAUNCH(false, -1); }
#undef LNP_LAUNCH
        UFV_CHECK_LAUNCH();
        return UFV_OK;
    }
#define LN_LAUNCH2(XD, YF, NV_, ACT_) hipLaunchKernelGGL((layernorm_k<XD, YF, NV_, ACT_>), g, blk, 0, ST(stream), x, ldx, y, ldy, w, b, M, D, eps, act)
#define LN_LAUNCH(XD, YF)                                                         \
    do {                                                                          \
        if (act == ACT_NONE && D <= 1280) LN_LAUNCH2(XD, YF, 5, ACT_NONE);        \
        else if (act == ACT_NONE) LN_LAUNCH2(XD, YF, MAXV, ACT_NONE);             \
        else if (act == ACT_SILU && D <= 1280) LN_LAUNCH2(XD, YF, 5, ACT_SILU);   \
        else if (act == ACT_SILU) LN_LAUNCH2(XD, YF, MAXV, ACT_SILU);             \
        else LN_LAUNCH2(XD, YF, MAXV, -1);                                        \
    } while (0)
    if (x_dtype == UFV_DT_F32) { if (y_f32) LN_LAUNCH(UFV_DT_F32, true); else LN_LAUNCH(UFV_DT_F32, false); }
    else if (x_dtype == UFV_DT_BF16) { if (y_f32) LN_LAUNCH(UFV_DT_BF16, true); else LN_LAUNCH(UFV_DT_BF16, false); }
    else { ufv_set_error("ufv_layernorm: unsupported input dtype %d", x_dtype); return UFV_EINVAL; }
#undef LN_LAUNCH2
#undef LN_LAUNCH
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_ln_add_silu(const void* a, const float* wa, const float* ba, const void* b, const float* wb, const float* bb,
                               void* out, int M, int D, float eps, void* stream) {
    UFV_REQUIRE(a && wa && ba && b && out && M > 0, "ufv_ln_add_silu: bad arguments");
    UFV_REQUIRE(D % 8 == 0 && D <= 4096, "ufv_ln_add_silu: D=%d must be a multiple of 8 and <= 4096", D);
    static const bool unpacked = getenv("UFV_LN_ADD_UNPACKED") != nullptr;          // same-box A/B switch: the round-4 form
    if (unpacked)
        hipLaunchKernelGGL((ln_add_silu_k<false>), dim3(cdiv(M, 4)), dim3(256), 0, ST(stream), (const bf16*)a, wa, ba, (const bf16*)b, wb, bb, (bf16*)out, M, D, eps);
    else
        hipLaunchKernelGGL((ln_add_silu_k<true>), dim3(cdiv(M, 4)), dim3(256), 0, ST(stream), (const bf16*)a, wa, ba, (const bf16*)b, wb, bb, (bf16*)out, M, D, eps);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_rmsnorm(const float* x, int ldx, void* y, int y_f32, int ldy, const float* w, int M, int D, float eps,
                           void* stream) {
    UFV_REQUIRE(x && y && w && M > 0, "ufv_rmsnorm: bad arguments");
    UFV_REQUIRE(D % 4 == 0 && D <= 4 * 64 * MAXV && ldx % 4 == 0 && ldy % 4 == 0, "ufv_rmsnorm: D=%d unsupported", D);
    // one wave per row.  Few rows (the decoder's S = 2399: 9.4 rows per CU): one-wave blocks, so that the CUs' row counts differ by one row, not by a block
    // of four (600 four-row blocks gave CUs 8 or 12 rows)
    const int rpb = M < 8192 ? 1 : 4;
    if (y_f32) hipLaunchKernelGGL((rmsnorm_k<true>), dim3(cdiv(M, rpb)), dim3(64 * rpb), 0, ST(stream), x, ldx, y, ldy, w, M, D, eps);
    else hipLaunchKernelGGL((rmsnorm_k<false>), dim3(cdiv(M, rpb)), dim3(64 * rpb), 0, ST(stream), x, ldx, y, ldy, w, M, D, eps);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_rope_kv(void* qkv, int ldqkv, int S, int Hq, int Hkv, int hd, const float* inv_freq, int pos0, void* kv_cache,
                           int ldkv, void* stream) {
    UFV_REQUIRE(qkv && inv_freq && kv_cache && S > 0 && hd % 2 == 0, "ufv_rope_kv: bad arguments");
    const bool vec = (hd % 16 == 0) && (ldqkv % 8 == 0) && (ldkv % 8 == 0) && ((uintptr_t)qkv % 16 == 0) && ((uintptr_t)kv_cache % 16 == 0);
    if (vec) {
        const int64_t total = (int64_t)S * (Hq + 2 * Hkv) * (hd / 16);
        hipLaunchKernelGGL(rope_kv_k, dim3(grid_for(total)), dim3(256), 0, ST(stream), (bf16*)qkv, ldqkv, S, Hq, Hkv, hd, inv_freq, pos0,
                           (bf16*)kv_cache, ldkv);
    } else {
        const int64_t total = (int64_t)S * (Hq + 2 * Hkv) * (hd / 2);
        hipLaunchKernelGGL(rope_kv_scalar_k, dim3(grid_for(total)), dim3(256), 0, ST(stream), (bf16*)qkv, ldqkv, S, Hq, Hkv, hd, inv_freq,
                           pos0, (bf16*)kv_cache, ldkv);
    }
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_rope_kv1_dev(void* qkv, int Hq, int Hkv, int hd, const float* inv_freq, const int* pos_dev, void* kv_cache, int ldkv,
                                void* stream) {
    UFV_REQUIRE(qkv && inv_freq && pos_dev && kv_cache && hd % 2 == 0, "ufv_rope_kv1_dev: bad arguments");
    const int total = (Hq + 2 * Hkv) * (hd / 2);
    hipLaunchKernelGGL(rope_kv1_dev_k, dim3(cdiv(total, 256)), dim3(256), 0, ST(stream), (bf16*)qkv, Hq, Hkv, hd, inv_freq, pos_dev,
                       (bf16*)kv_cache, ldkv);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_add_int(int* p, int v, void* stream) {
    UFV_REQUIRE(p, "ufv_add_int: null pointer");
    hipLaunchKernelGGL(add_int_k, dim3(1), dim3(64), 0, ST(stream), p, v);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_rope_table(const float* inv_freq, int pos0, int S, int hd, float* table, void* stream) {
    UFV_REQUIRE(inv_freq && table && S > 0 && hd % 2 == 0, "ufv_rope_table: bad arguments");
    hipLaunchKernelGGL(rope_table_k, dim3(grid_for((int64_t)S * (hd / 2))), dim3(256), 0, ST(stream), inv_freq, pos0, S, hd / 2, table);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_rope_kv_table(void* qkv, int ldqkv, int S, int Hq, int Hkv, int hd, const float* table, int pos0, void* kv_cache,
                                 int ldkv, void* stream) {
    UFV_REQUIRE(qkv && table && kv_cache && S > 0, "ufv_rope_kv_table: bad arguments");
    UFV_REQUIRE((hd % 16 == 0) && (ldqkv % 8 == 0) && (ldkv % 8 == 0) && ((uintptr_t)qkv % 16 == 0) && ((uintptr_t)kv_cache % 16 == 0) &&
                ((uintptr_t)table % 16 == 0), "ufv_rope_kv_table: needs hd %% 16 == 0 and 16-byte aligned rows (hd=%d)", hd);
    const int64_t total = (int64_t)S * (Hq + 2 * Hkv) * (hd / 16);
    hipLaunchKernelGGL(rope_kv_tab_k, dim3(grid_for(total)), dim3(256), 0, ST(stream), (bf16*)qkv, ldqkv, S, Hq, Hkv, hd, table, pos0,
                       (bf16*)kv_cache, ldkv);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_patchify(const void* pixels, int dtype, void* out, int T, int C, int H, int W, int P, int Kpad, void* stream) {
    // H, W need not be multiples of P: nn.Conv2d(kernel = stride = P, padding 'valid') drops the H % P bottom rows and W % P right columns (SigLIP so400m at 384 px: 27 x 27 patches)
    UFV_REQUIRE(pixels && out && T > 0 && P > 0 && H >= P && W >= P && Kpad >= C * P * P, "ufv_patchify: bad arguments");
    dim3 g(H / P, T), blk(256);
    // runs of P pixels as dwords when rows and runs are dword-aligned (P, W even: 14 / 336) and 8-byte aligned for fp32 pixels
    const bool runs = P % 2 == 0 && W % 2 == 0 && Kpad % 2 == 0 && (uintptr_t)pixels % 8 == 0 && (uintptr_t)out % 4 == 0 && (dtype == UFV_DT_F32 || dtype == UFV_DT_BF16);
    if (runs && dtype == UFV_DT_F32) hipLaunchKernelGGL((patchify_runs_k<UFV_DT_F32>), g, blk, 0, ST(stream), pixels, (bf16*)out, T, C, H, W, P, Kpad);
    else if (runs) hipLaunchKernelGGL((patchify_runs_k<UFV_DT_BF16>), g, blk, 0, ST(stream), pixels, (bf16*)out, T, C, H, W, P, Kpad);
    else if (dtype == UFV_DT_F32) hipLaunchKernelGGL((patchify_k<UFV_DT_F32>), g, blk, 0, ST(stream), pixels, (bf16*)out, T, C, H, W, P, Kpad);
    else if (dtype == UFV_DT_BF16) hipLaunchKernelGGL((patchify_k<UFV_DT_BF16>), g, blk, 0, ST(stream), pixels, (bf16*)out, T, C, H, W, P, Kpad);
    else if (dtype == UFV_DT_F16) hipLaunchKernelGGL((patchify_k<UFV_DT_F16>), g, blk, 0, ST(stream), pixels, (bf16*)out, T, C, H, W, P, Kpad);
    else { ufv_set_error("ufv_patchify: unsupported dtype %d", dtype); return UFV_EINVAL; }
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_dwconv3x3_ln_silu(const void* x, void* y, const float* w9, const float* lnw, const float* lnb, int F, int H,
                                     int W, int C, float eps, void* stream) {
    UFV_REQUIRE(x && y && w9 && lnw && lnb && F > 0, "ufv_dwconv3x3_ln_silu: bad arguments");
    UFV_REQUIRE(C % 8 == 0 && C <= 4096, "ufv_dwconv3x3_ln_silu: C=%d must be a multiple of 8 and <= 4096", C);
    // round 5 lab: 8 pixels per block (halves the weight re-reads) lost to the occupancy it costs (192 registers: 151 -> 172 us); the XCD-aware order alone: 151 -> 132 us
    static const bool no_remap = getenv("UFV_DWCONV_NO_REMAP") != nullptr;          // same-box A/B switch
    const int nblk = F * H * cdiv(W, 4);
    if (no_remap)
        hipLaunchKernelGGL((dwconv_ln_silu_k<4, false>), dim3(nblk), dim3(256), 0, ST(stream), (const bf16*)x, (bf16*)y, w9, lnw, lnb, F, H, W, C, eps, nblk);
    else
        hipLaunchKernelGGL((dwconv_ln_silu_k<4, true>), dim3(cdiv(nblk, 8) * 8), dim3(256), 0, ST(stream), (const bf16*)x, (bf16*)y, w9, lnw, lnb, F, H, W, C, eps, nblk);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_colmean(const void* x, void* out, int F, int P, int C, void* stream) {
    UFV_REQUIRE(x && out && F > 0 && P > 0 && C % 8 == 0, "ufv_colmean: C must be a multiple of 8");
    hipLaunchKernelGGL(colmean_k, dim3(cdiv(C / 8, 64), F), dim3(512), 0, ST(stream), (const bf16*)x, (bf16*)out, F, P, C);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_scale_channels(void* x, const void* gate, int F, int P, int C, void* stream) {
    UFV_REQUIRE(x && gate && C % 8 == 0, "ufv_scale_channels: C must be a multiple of 8");
    hipLaunchKernelGGL(scale_channels_k, dim3(grid_for((int64_t)F * P * C / 8)), dim3(256), 0, ST(stream), (bf16*)x, (const bf16*)gate,
                       F, P, C);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_conv3d_gather(const void* x, void* out, int T, int H, int W, int C, int kt, int kh, int kw, int pad,
                                 void* stream) {
    UFV_REQUIRE(x && out && C % 8 == 0, "ufv_conv3d_gather: C must be a multiple of 8");
    const int To = (T + 2 * pad - kt) / kt + 1, Ho = (H + 2 * pad - kh) / kh + 1, Wo = (W + 2 * pad - kw) / kw + 1;
    UFV_REQUIRE(To > 0 && Ho > 0 && Wo > 0, "ufv_conv3d_gather: empty output");
    const int64_t total = (int64_t)To * Ho * Wo * kt * kh * kw * (C / 8);
    hipLaunchKernelGGL(conv3d_gather_k, dim3(grid_for(total)), dim3(256), 0, ST(stream), (const bf16*)x, (bf16*)out, T, H, W, C, kt,
                       kh, kw, pad, To, Ho, Wo);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_avgpool3d_silu(const void* x, void* out, int T, int H, int W, int C, int kt, int kh, int kw, void* stream) {
    UFV_REQUIRE(x && out && C % 8 == 0 && kt > 0 && kh > 0 && kw > 0, "ufv_avgpool3d_silu: C must be a multiple of 8");
    const int To = T / kt, Ho = H / kh, Wo = W / kw;
    UFV_REQUIRE(To > 0 && Ho > 0 && Wo > 0, "ufv_avgpool3d_silu: empty output");
    hipLaunchKernelGGL(avgpool3d_silu_k<ACT_SILU>, dim3(grid_for((int64_t)To * Ho * Wo * (C / 8))), dim3(256), 0, ST(stream), (const bf16*)x,
                       (bf16*)out, T, H, W, C, kt, kh, kw, To, Ho, Wo);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

// the same pooling without the activation (training keeps the pre-activation: train_projector.py)
extern "C" int ufv_avgpool3d(const void* x, void* out, int T, int H, int W, int C, int kt, int kh, int kw, void* stream) {
    UFV_REQUIRE(x && out && C % 8 == 0 && kt > 0 && kh > 0 && kw > 0, "ufv_avgpool3d: C must be a multiple of 8");
    const int To = T / kt, Ho = H / kh, Wo = W / kw;
    UFV_REQUIRE(To > 0 && Ho > 0 && Wo > 0, "ufv_avgpool3d: empty output");
    hipLaunchKernelGGL(avgpool3d_silu_k<ACT_NONE>, dim3(grid_for((int64_t)To * Ho * Wo * (C / 8))), dim3(256), 0, ST(stream), (const bf16*)x,
                       (bf16*)out, T, H, W, C, kt, kh, kw, To, Ho, Wo);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_gather_rows(const void* src, int src_dtype, int64_t ld_src, const int64_t* src_idx, void* dst, int dst_dtype,
                               int64_t ld_dst, const int64_t* dst_idx, int n, int D, void* stream) {
    if (n == 0) return UFV_OK;
    UFV_REQUIRE(src && dst && n > 0 && D > 0, "ufv_gather_rows: bad arguments");
    dim3 g(cdiv(n, 4)), blk(256);
    const int ses = src_dtype == UFV_DT_F32 ? 4 : 2, des = dst_dtype == UFV_DT_F32 ? 4 : 2;
    const bool vec = D % 8 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0) && (ld_src * ses) % 16 == 0 && (ld_dst * des) % 16 == 0;
    const bool few = n <= 64;                    // few rows: one block per row
#define GR(SD, DF) do { if (vec && few) hipLaunchKernelGGL((gather_rows8_k<SD, DF, 4>), dim3(n), blk, 0, ST(stream), src, ld_src, src_idx, dst, ld_dst, dst_idx, n, D); \
                        else if (vec) hipLaunchKernelGGL((gather_rows8_k<SD, DF, 1>), g, blk, 0, ST(stream), src, ld_src, src_idx, dst, ld_dst, dst_idx, n, D); \
                        else hipLaunchKernelGGL((gather_rows_k<SD, DF>), g, blk, 0, ST(stream), src, ld_src, src_idx, dst, ld_dst, dst_idx, n, D); } while (0)
    const bool df32 = dst_dtype == UFV_DT_F32;
    UFV_REQUIRE(dst_dtype == UFV_DT_F32 || dst_dtype == UFV_DT_BF16, "ufv_gather_rows: dst must be f32 or bf16");
    if (src_dtype == UFV_DT_F32) { if (df32) GR(UFV_DT_F32, true); else GR(UFV_DT_F32, false); }
    else if (src_dtype == UFV_DT_BF16) { if (df32) GR(UFV_DT_BF16, true); else GR(UFV_DT_BF16, false); }
    else if (src_dtype == UFV_DT_F16) { if (df32) GR(UFV_DT_F16, true); else GR(UFV_DT_F16, false); }
    else { ufv_set_error("ufv_gather_rows: unsupported src dtype"); return UFV_EINVAL; }
#undef GR
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_mask_pool(const void* feat, int feat_dtype, const float* mask, const int32_t* frame_of, float* out, int q, int P,
                             int C, void* stream) {
    if (q == 0) return UFV_OK;
    UFV_REQUIRE(feat && mask && frame_of && out, "ufv_mask_pool: bad arguments");
    dim3 g(cdiv(C, 256), q), blk(256);
    if (feat_dtype == UFV_DT_F32) hipLaunchKernelGGL((mask_pool_k<UFV_DT_F32>), g, blk, 0, ST(stream), feat, mask, frame_of, out, P, C);
    else if (feat_dtype == UFV_DT_BF16) hipLaunchKernelGGL((mask_pool_k<UFV_DT_BF16>), g, blk, 0, ST(stream), feat, mask, frame_of, out, P, C);
    else { ufv_set_error("ufv_mask_pool: unsupported dtype"); return UFV_EINVAL; }
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_argmax(const float* logits, int N, int64_t* out, void* stream) {
    UFV_REQUIRE(logits && out && N > 0, "ufv_argmax: bad arguments");
    hipLaunchKernelGGL(argmax_k, dim3(1), dim3(1024), 0, ST(stream), logits, N, out);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int64_t ufv_argmax_ws_bytes(void) { return (int64_t)(sizeof(float) * ARGMAX_NB + sizeof(int) * (ARGMAX_NB + 1)); }

// ws: ufv_argmax_ws_bytes() bytes, 16-byte aligned, whose last int is ZERO before the first call (it returns to zero); logits 16-byte aligned
extern "C" int ufv_argmax_ws(const float* logits, int N, int64_t* out, void* ws, void* stream) {
    UFV_REQUIRE(logits && out && ws && N > 0 && ((uintptr_t)logits % 16 == 0) && ((uintptr_t)ws % 4 == 0), "ufv_argmax_ws: bad arguments");
    hipLaunchKernelGGL(argmax_blocks_k, dim3(ARGMAX_NB), dim3(256), 0, ST(stream), logits, N, out, (float*)ws);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_preprocess_u8(const uint8_t* frames, void* out, int T, int H, int W, const float* mean3, const float* std3,
                                 void* stream) {
    UFV_REQUIRE(frames && out && mean3 && std3 && T > 0, "ufv_preprocess_u8: bad arguments");
    hipLaunchKernelGGL(preprocess_u8_k, dim3(grid_for((int64_t)T * 3 * H * W)), dim3(256), 0, ST(stream), frames, (bf16*)out, T, H, W,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_convert(const void* src, int sd, void* dst, int dd, int64_t n, void* stream) {
    if (n == 0) return UFV_OK;
    UFV_REQUIRE(src && dst && sd >= 0 && sd <= 2 && dd >= 0 && dd <= 2, "ufv_convert: bad arguments");
    dim3 g(grid_for(n)), blk(256);
#define CV(S, D) hipLaunchKernelGGL((convert_k<S, D>), g, blk, 0, ST(stream), src, dst, n)
    switch (sd * 3 + dd) {
        case 0: CV(0, 0); break; case 1: CV(0, 1); break; case 2: CV(0, 2); break;
        case 3: CV(1, 0); break; case 4: CV(1, 1); break; case 5: CV(1, 2); break;
        case 6: CV(2, 0); break; case 7: CV(2, 1); break; case 8: CV(2, 2); break;
    }
#undef CV
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_im2col(const void* pixels, int dtype, void* out, int B, int C, int H, int W, int ks, int stride, int pad, int Kpad,
                          void* stream) {
    UFV_REQUIRE(pixels && out && B > 0 && ks > 0 && stride > 0 && Kpad >= C * ks * ks, "ufv_im2col: bad arguments");
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    UFV_REQUIRE(Ho > 0 && Wo > 0, "ufv_im2col: empty output");
    dim3 g(grid_for((int64_t)B * Ho * Wo * Kpad)), blk(256);
    if (dtype == UFV_DT_F32) hipLaunchKernelGGL((im2col_k<UFV_DT_F32>), g, blk, 0, ST(stream), pixels, (bf16*)out, B, C, H, W, ks, stride, pad, Ho, Wo, Kpad);
    else if (dtype == UFV_DT_BF16) hipLaunchKernelGGL((im2col_k<UFV_DT_BF16>), g, blk, 0, ST(stream), pixels, (bf16*)out, B, C, H, W, ks, stride, pad, Ho, Wo, Kpad);
    else if (dtype == UFV_DT_F16) hipLaunchKernelGGL((im2col_k<UFV_DT_F16>), g, blk, 0, ST(stream), pixels, (bf16*)out, B, C, H, W, ks, stride, pad, Ho, Wo, Kpad);
    else { ufv_set_error("ufv_im2col: unsupported dtype %d", dtype); return UFV_EINVAL; }
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_maxpool2x2(const void* x, int dtype, int64_t ldx, void* out, int64_t ldo, int Bw, int H, int W, int C, void* stream) {
    UFV_REQUIRE(x && out && Bw > 0 && H >= 2 && W >= 2 && C > 0, "ufv_maxpool2x2: bad arguments");
    dim3 g(grid_for((int64_t)Bw * (H / 2) * (W / 2) * C)), blk(256);
    if (dtype == UFV_DT_F32) hipLaunchKernelGGL((maxpool2x2_k<UFV_DT_F32>), g, blk, 0, ST(stream), x, ldx, out, ldo, Bw, H, W, C);
    else if (dtype == UFV_DT_BF16) hipLaunchKernelGGL((maxpool2x2_k<UFV_DT_BF16>), g, blk, 0, ST(stream), x, ldx, out, ldo, Bw, H, W, C);
    else { ufv_set_error("ufv_maxpool2x2: unsupported dtype %d", dtype); return UFV_EINVAL; }
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_add_rows(const void* src, int src_dtype, int64_t ld_src, float* dst, int64_t ld_dst, const int64_t* dst_idx, int n, int D,
                            void* stream) {
    if (n == 0) return UFV_OK;
    UFV_REQUIRE(src && dst && n > 0 && D > 0, "ufv_add_rows: bad arguments");
    dim3 g(cdiv(n, 4)), blk(256);
    if (src_dtype == UFV_DT_F32) hipLaunchKernelGGL((add_rows_k<UFV_DT_F32>), g, blk, 0, ST(stream), src, ld_src, dst, ld_dst, dst_idx, n, D);
    else if (src_dtype == UFV_DT_BF16) hipLaunchKernelGGL((add_rows_k<UFV_DT_BF16>), g, blk, 0, ST(stream), src, ld_src, dst, ld_dst, dst_idx, n, D);
    else { ufv_set_error("ufv_add_rows: unsupported dtype %d", src_dtype); return UFV_EINVAL; }
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_upsample2x_add(float* x, const float* prev, int B, int H, int W, int C, void* stream) {
    UFV_REQUIRE(x && prev && B > 0 && H % 2 == 0 && W % 2 == 0, "ufv_upsample2x_add: bad arguments");
    hipLaunchKernelGGL(upsample2x_add_k, dim3(grid_for((int64_t)B * H * W * C)), dim3(256), 0, ST(stream), x, prev, B, H, W, C);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}
