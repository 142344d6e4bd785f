// Shared GEMM epilogue (bias -> activation -> fp32 residual -> bf16|fp32 store), 4 consecutive columns per lane.
#pragma once
#include "common.h"

struct Epi {
    const float* bias;     // [N] or null
    const float* resid;    // fp32 [M, ldr] or null (added after activation)
    void* out;             // bf16 or fp32
    int ldr, ldc, act;
    int resid_rows;        // >0: residual row = m % resid_rows (broadcast table, e.g. position embeddings)
    int resid_bf16 = 0;    // the residual is bf16 [M, ldr] (bf16 outputs only): the in-place update of a bf16 residual stream, out = bf16(acc + bias + float(resid))
    const float* scale_m;  // fp8 operands only: per-row scale of A [M] and per-row scale of W [N]; acc *= scale_m[m] * scale_n[n]
    const float* scale_n;
    int dump_f32;          // stream-K only: store the raw accumulators as fp32 whatever the output type (partial tile dump)
    int ksplit;            // > 1: split-K launch (gridDim.y = ksplit), partial tiles are added to the fp32 output atomically
    // fused QKV projection + RoPE + KV-cache append (gemm_nt_256<..., ROPE = true>, epilogue256_rope): `out` / `ldc` is the q buffer [M, Hq * 128]
    const float* rope_tab = nullptr; // f32 [M, 128]: cos (64) | sin (64) of every row's position (ufv_rope_table)
    void* out_kv = nullptr;          // bf16 KV-cache row of the call's first position: row m = [Hkv * 128 k | Hkv * 128 v] at out_kv + m * ldkv
    int ldkv = 0, rope_hq = 0, rope_hkv = 0;
    // MX-style block scales of e4m3 activations (gemm_nt_256<..., MX>): one e8m0 byte per (row, 32 consecutive K / N elements), value 2^(byte - 127), stored
    // as [ceil(M / 64)][ceil(K / 512)][64 rows][16]: per (row, group of four K-tiles) 16 bytes, byte 4 (b & 3) + (b >> 2) = block b of the group (b & 3 = the 32-element
    // block inside its K-tile); the groups of a 64-row block follow each other (include/ufv.h ufv_quantize_mx)
    const unsigned char* a_bscale = nullptr;   // MX & 1: scales of the A operand; ld_abs = bytes per 64-row block (= 1024 ceil(K / 512)); scale_m is not used then
    unsigned char* out_bscale = nullptr;       // MX & 2: the epilogue writes e4m3 codes to `out` and their block scales here, ld_obs bytes per 64-row block
    int ld_abs = 0, ld_obs = 0;
};

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// four / one residual value(s) at element index idx of the residual buffer, whatever its type
__device__ __forceinline__ f32x4 resid_load4(const Epi& e, size_t idx) {
    if (e.resid_bf16) {
        const u32x2 c = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16*>(e.resid) + idx);
        return f32x4{__builtin_bit_cast(float, c[0] << 16), __builtin_bit_cast(float, c[0] & 0xffff0000u), __builtin_bit_cast(float, c[1] << 16),
                     __builtin_bit_cast(float, c[1] & 0xffff0000u)};
    }
    return *reinterpret_cast<const f32x4*>(e.resid + idx);
}
__device__ __forceinline__ float resid_load1(const Epi& e, size_t idx) {
    return e.resid_bf16 ? (float)reinterpret_cast<const bf16*>(e.resid)[idx] : e.resid[idx];
}
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <bool OUT_F32, int ACT>
__device__ __forceinline__ void epi_store4(const Epi& e, int m, int n, float v0, float v1, float v2, float v3) {
    // n is a multiple of 4; the four values are columns n..n+3 of row m
    float v[4] = {v0, v1, v2, v3};
    if (e.bias) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(e.bias + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += b[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = act_apply_t<ACT>(v[j]);
    if (e.resid) {
        const int mr = e.resid_rows > 0 ? m % e.resid_rows : m;
        const f32x4 r = resid_load4(e, (size_t)mr * e.ldr + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += r[j];
    }
    if (OUT_F32) {
        f32x4 o = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(e.out) + (size_t)m * e.ldc + n) = o;
    } else {
        bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(e.out) + (size_t)m * e.ldc + n) = o;
    }
}


// same as epi_store4 with the bias already in registers (ignored when e.bias is null).  DUMP (stream-K kernels): when
// e.dump_f32 is set the four values are stored untouched as fp32 -- the partial-tile dump reuses the epilogue's store code so
// that the accumulators have a single consumer in the control flow (a second consumer block makes hipcc copy / spill them).
template <bool OUT_F32, int ACT, bool DUMP = false>
__device__ __forceinline__ void epi_store4b(const Epi& e, int m, int n, float v0, float v1, float v2, float v3, f32x4 b) {
    if (DUMP && e.dump_f32) {
        // system-scope (write-through) store: the partial tile is read by a block on another XCD, whose L2 is not coherent with
        // ours; sc0 sc1 makes the line visible in memory when the store is acknowledged -- no L2 write-back fence needed.
        f32x4 o = {v0, v1, v2, v3};
        float* p = reinterpret_cast<float*>(e.out) + (size_t)m * e.ldc + n;
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" ::"v"(p), "v"(o) : "memory");       // s_nop: store-data hazard the compiler cannot see inside asm
        return;
    }
    float v[4] = {v0, v1, v2, v3};      // b = bias already in registers (zeros when there is none)
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = act_apply_t<ACT>(v[j] + b[j]);
    if (e.resid) {
        const int mr = e.resid_rows > 0 ? m % e.resid_rows : m;
        const f32x4 r = resid_load4(e, (size_t)mr * e.ldr + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += r[j];
    }
    if (OUT_F32) {
        f32x4 o = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(e.out) + (size_t)m * e.ldc + n) = o;
    } else {
        bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(e.out) + (size_t)m * e.ldc + n) = o;
    }
}

int ufv_launch_gemm256(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, bool out_f32,
                       bool swiglu, bool fp8, bool streamk, int shape, hipStream_t st);
// gemm256_m.hip / gemm256_m2.hip: the MX instantiations (mx & 1: block-scaled A operand at every named shape; mx & 2: 256 x 256 tiles whose epilogue emits MX)
int ufv_launch_pp_mx(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, bool out_f32, bool swiglu, int shape, int mx, hipStream_t st);
// gemm256_r.hip: the fused QKV + RoPE + KV-append instantiations of the ping-pong kernel (shape 1332)
int ufv_launch_pp_rope(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, int shape, hipStream_t st);
int ufv_launch_pp_rope_fp8(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, int shape, hipStream_t st);
