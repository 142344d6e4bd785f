// Fused (flash-style) backward of causal GQA self-attention for head_dim 128 (Qwen2 decoder; HF eager attention,
// modeling_qwen2.py:150-172, differentiated).  Nothing of size S x S is materialised: with the forward's log-sum-exp L (log2
// domain) and delta = rowsum(dO * O) per (head, query),
//     P = exp2(s * scale * log2e - L),   dS = scale * P * (dP - delta),   dP = dO V^T
// Two kernels, both built from the forward kernel's fragments (attn.hip: S^T = K Q^T with v_mfma_f32_32x32x16_bf16 so a lane
// owns one column; the exponentiated tile is re-used as the B operand of the second product; transposed operands come from
// row-major LDS tiles through ds_read_b64_tr_b16):
//   attn_bwd_dq   one wave = 32 queries (Q^T, dO^T fragments in registers), loops over 64-key tiles:
//                 S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T
//   attn_bwd_dkv  one wave = 32 keys of one kv head (K^T, V^T fragments in registers), one block per (128 keys, q head), loops
//                 over 64-query tiles from the diagonal on:  S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS;
//                 per-head fp32 partials, summed over the heads of the kv group by attn_bwd_reduce.
// Tiles go global -> registers (issued one tile ahead) -> LDS (two stages, one barrier per tile) in the two layouts the reads
// need: row pitch 272 B for the conflict-free ds_read_b128 row fragments, 320 B for the transposed reads.
#include "common.h"
#include "../../include/ufv.h"

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ bf16x8 pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    const bf16x2 p0 = __builtin_convertvector((f32x2){a0, a1}, bf16x2), p1 = __builtin_convertvector((f32x2){a2, a3}, bf16x2);
    const bf16x2 p2 = __builtin_convertvector((f32x2){a4, a5}, bf16x2), p3 = __builtin_convertvector((f32x2){a6, a7}, bf16x2);
    const bf16x4 q0 = __builtin_shufflevector(p0, p1, 0, 1, 2, 3), q1 = __builtin_shufflevector(p2, p3, 0, 1, 2, 3);
    return __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7);
}

constexpr int HD = 128, KS = 8, DT = 4, NCH = 16;            // k-steps of 16, d-tiles of 32, 16-byte chunks per row
constexpr int PR = 272, PT = 320;                            // row pitch for row fragments / for transposed reads
constexpr int NT = 256;
constexpr int STAGE_DQ = 2 * 64 * PR + 64 * PT, STAGE_DKV = 2 * 64 * PR + 2 * 64 * PT + 512;

struct BwdArgs {
    const bf16 *q, *k, *v, *dO;
    int64_t ldq, ldkv, lddo;
    const float *lse, *delta;      // [Hq, S]
    int S, Hq, Hkv;
    float scale;
};

// acc[dt] (+)= T^T[d][i] * B[i][col] over one 32-row half of a staged tile: T = 32 rows at `tb` (pitch PT), contracted over its
// rows in the forward kernel's chunk order; pf[c] = the B operand of 16-row chunk c (see attn.hip)
__device__ __forceinline__ void mma_transposed_half(const char* tb, int v_off, const bf16x8 (&pf)[2], f32x16 (&acc)[DT]) {
    bf16x8 vf[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const char* vp = tb + (16 * c) * PT + v_off + dt * 64;
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vp));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vp + 8 * PT));
            vf[dt][c] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][c], pf[c], acc[dt], 0, 0, 0);
}

// delta[h][s] = sum_d dO[s][h*128 + d] * O[s][h*128 + d]; one wave per (s, h)
__global__ __launch_bounds__(256) void attn_bwd_delta_k(const bf16* __restrict__ dO, int64_t lddo, const bf16* __restrict__ O, int64_t ldo, int S,
                                                        int Hq, float* __restrict__ delta) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (int64_t)S * Hq) return;
    const int h = w / S, s = w % S;                                  // consecutive waves -> consecutive floats of delta[h][*]
    const bf16x2 a = *reinterpret_cast<const bf16x2*>(dO + (int64_t)s * lddo + h * HD + lane * 2);
    const bf16x2 b = *reinterpret_cast<const bf16x2*>(O + (int64_t)s * ldo + h * HD + lane * 2);
    float t = (float)a[0] * (float)b[0] + (float)a[1] * (float)b[1];
    t = wave_sum(t);
    if (lane == 0) delta[(int64_t)h * S + s] = t;
}

// ---- dQ: grid (q tiles of 128, Hq) -----------------------------------------------------------------------------------
__global__ __launch_bounds__(NT, 1) void attn_bwd_dq_k(BwdArgs a, bf16* __restrict__ dq, int64_t lddq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // two stages of [K rows (pitch PR) | V rows (PR) | K rows again at the transposed-read pitch PT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int bi = blockIdx.x + gridDim.x * blockIdx.y;              // heaviest (last) q tiles of every head first
    const int qt = gridDim.x - 1 - bi / a.Hq;
    const int hq = bi % a.Hq, hkv = hq / (a.Hq / a.Hkv);
    const int q0 = qt * 128 + wave * 32, qi = q0 + l31;
    const int qc = min(qi, a.S - 1);
    bf16x8 qf[KS], dof[KS];
    {
        const bf16* qp = a.q + (int64_t)qc * a.ldq + hq * HD;
        const bf16* dp = a.dO + (int64_t)qc * a.lddo + hq * HD;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16 + h * 8);
            dof[ks] = *reinterpret_cast<const bf16x8*>(dp + ks * 16 + h * 8);
        }
    }
    const float lse = a.lse[(int64_t)hq * a.S + qc], dl = a.delta[(int64_t)hq * a.S + qc];
    const float sl2 = a.scale * 1.4426950408889634f, scale_ = a.scale;
    f32x16 acc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int k_off = l31 * PR + h * 16;
    const int v_off = (4 * h + ((lane & 15) >> 2)) * PT + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const bf16* kbase = a.k + hkv * HD;
    const bf16* vbase = a.v + hkv * HD;
    const int kmax = min(a.S, qt * 128 + 128);                       // causal: keys <= last query of the block
    const int ntiles = (kmax + 63) / 64;
    bf16x8 kreg[4], vreg[4];                                         // next tile in flight while this one is consumed
    auto issue_loads = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * NT, row = min(t * 64 + id / NCH, a.S - 1), c = id % NCH;
            kreg[i] = *reinterpret_cast<const bf16x8*>(kbase + (int64_t)row * a.ldkv + c * 8);
            vreg[i] = *reinterpret_cast<const bf16x8*>(vbase + (int64_t)row * a.ldkv + c * 8);
        }
    };
    auto write_lds = [&](int st) {
        char* b = smem + st * STAGE_DQ;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * NT, row = id / NCH, c = id % NCH;
            *reinterpret_cast<bf16x8*>(b + row * PR + c * 16) = kreg[i];
            *reinterpret_cast<bf16x8*>(b + 64 * PR + row * PR + c * 16) = vreg[i];
            *reinterpret_cast<bf16x8*>(b + 128 * PR + row * PT + c * 16) = kreg[i];
        }
    };
    issue_loads(0);
    write_lds(0);
    // pin the resident fragments here: a load still counted as pending at the loop header makes the compiler's vmcnt waits inside
    // the loop cover the freshly issued prefetch as well
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]), "+v"(dof[ks]));
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) issue_loads(t + 1);
        const char* k_row = smem + (t & 1) * STAGE_DQ;
        const char* v_row = k_row + 64 * PR;
        const char* k_tr = v_row + 64 * PR;
        {   // (tiles past this wave's diagonal are computed too: everything in them is masked to zero; a wave-level skip costs more in register shuffling than the one or two tiles it saves)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {                             // the two 32-key halves in turn (keeps the live tile at 32 VGPRs)
            f32x16 sv, pv;                                           // [key][q]: lane = query column l31, rows = keys
#pragma unroll
            for (int r = 0; r < 16; ++r) { sv[r] = 0.f; pv[r] = 0.f; }
            bf16x8 kfr[KS], vfr[KS];                                 // all row fragments in flight before the first MFMA
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kfr[ks] = *reinterpret_cast<const bf16x8*>(k_row + hh * 32 * PR + k_off + ks * 32);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) vfr[ks] = *reinterpret_cast<const bf16x8*>(v_row + hh * 32 * PR + k_off + ks * 32);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ks], qf[ks], sv, 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) pv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[ks], dof[ks], pv, 0, 0, 0);
            // dS^T[key][q] = scale * P * (dP - delta); key of register r: 32hh + (r&3) + 8(r>>2) + 4h
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kj = t * 64 + hh * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float x = __builtin_fmaf(sv[r], sl2, -lse);
                x = (kj <= qi) ? x : -INFINITY;                      // select, not a branch (qi < S for every stored row, so kj < S too)
                sv[r] = scale_ * __builtin_amdgcn_exp2f(x) * (pv[r] - dl);
            }
            bf16x8 pf[2];
            pf[0] = pack8(sv[0], sv[1], sv[2], sv[3], sv[4], sv[5], sv[6], sv[7]);
            pf[1] = pack8(sv[8], sv[9], sv[10], sv[11], sv[12], sv[13], sv[14], sv[15]);
            mma_transposed_half(k_tr + hh * 32 * PT, v_off, pf, acc);        // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
        }
        }
        if (t + 1 < ntiles) write_lds((t + 1) & 1);
        __syncthreads();
    }
    if (qi < a.S) {
        bf16* op = dq + (int64_t)qi * lddq + hq * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = dt * 32 + g4 * 8 + h * 4;
                bf16x4 ov = {(bf16)acc[dt][g4 * 4 + 0], (bf16)acc[dt][g4 * 4 + 1], (bf16)acc[dt][g4 * 4 + 2], (bf16)acc[dt][g4 * 4 + 3]};
                *reinterpret_cast<bf16x4*>(op + d0) = ov;
            }
    }
}

// ---- dK, dV partials per q head: grid (key tiles of 128, Hq) -----------------------------------------------------------
__global__ __launch_bounds__(NT, 1) void attn_bwd_dkv_k(BwdArgs a, float* __restrict__ dk_part, float* __restrict__ dv_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // two stages of [Q rows (pitch PR) | dO rows (PR) | Q rows (PT) | dO rows (PT) | lse[64] | delta[64]]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int bi = blockIdx.x + gridDim.x * blockIdx.y;              // heaviest (first) key tiles of every head first
    const int kt = bi / a.Hq, hq = bi % a.Hq, hkv = hq / (a.Hq / a.Hkv);
    const int k0 = kt * 128 + wave * 32, ki = k0 + l31;
    const int kc = min(ki, a.S - 1);
    bf16x8 kf[KS], vf[KS];                                           // B fragments: lane (col key = l31, k = d)
    {
        const bf16* kp = a.k + (int64_t)kc * a.ldkv + hkv * HD;
        const bf16* vp = a.v + (int64_t)kc * a.ldkv + hkv * HD;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(kp + ks * 16 + h * 8);
            vf[ks] = *reinterpret_cast<const bf16x8*>(vp + ks * 16 + h * 8);
        }
    }
    const float sl2 = a.scale * 1.4426950408889634f;
    f32x16 dka[DT], dva[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dka[i][r] = 0.f; dva[i][r] = 0.f; }
    const int k_off = l31 * PR + h * 16;
    const int v_off = (4 * h + ((lane & 15) >> 2)) * PT + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const bf16* qbase = a.q + hq * HD;
    const bf16* dobase = a.dO + hq * HD;
    const float* lse_h = a.lse + (int64_t)hq * a.S;
    const float* dl_h = a.delta + (int64_t)hq * a.S;
    const int nqt = (a.S + 63) / 64;
    bf16x8 qreg[4], dreg[4];
    float lreg = 0.f, dlreg = 0.f;
    auto issue_loads = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * NT, row = min(t * 64 + id / NCH, a.S - 1), c = id % NCH;
            qreg[i] = *reinterpret_cast<const bf16x8*>(qbase + (int64_t)row * a.ldq + c * 8);
            dreg[i] = *reinterpret_cast<const bf16x8*>(dobase + (int64_t)row * a.lddo + c * 8);
        }
        if (tid < 64) {
            const int qq = t * 64 + tid;
            lreg = qq < a.S ? lse_h[qq] : INFINITY;                  // exp2(-inf) = 0 for the rows past the sequence end
            dlreg = qq < a.S ? dl_h[qq] : 0.f;
        }
    };
    auto write_lds = [&](int st) {
        char* b = smem + st * STAGE_DKV;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * NT, row = id / NCH, c = id % NCH;
            *reinterpret_cast<bf16x8*>(b + row * PR + c * 16) = qreg[i];
            *reinterpret_cast<bf16x8*>(b + 64 * PR + row * PR + c * 16) = dreg[i];
            *reinterpret_cast<bf16x8*>(b + 128 * PR + row * PT + c * 16) = qreg[i];
            *reinterpret_cast<bf16x8*>(b + 128 * PR + 64 * PT + row * PT + c * 16) = dreg[i];
        }
        if (tid < 64) {
            float* f = reinterpret_cast<float*>(b + 128 * PR + 128 * PT);
            f[tid] = lreg;
            f[64 + tid] = dlreg;
        }
    };
    const int t0 = (kt * 128) / 64;                                  // causal: queries from this block's first key on
    issue_loads(t0);
    write_lds(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(kf[ks]), "+v"(vf[ks]));      // see attn_bwd_dq_k
    __syncthreads();
    for (int t = t0; t < nqt; ++t) {
        if (t + 1 < nqt) issue_loads(t + 1);
        const char* q_row = smem + ((t - t0) & 1) * STAGE_DKV;
        const char* do_row = q_row + 64 * PR;
        const char* q_tr = do_row + 64 * PR;
        const char* do_tr = q_tr + 64 * PT;
        const float* lse_s = reinterpret_cast<const float*>(do_tr + 64 * PT);
        const float* dl_s = lse_s + 64;
        {   // (a tile wholly before this wave's keys is computed too and masked to zero, see attn_bwd_dq_k)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {                             // the two 32-query halves in turn (keeps the live tile at 32 VGPRs)
            f32x16 sv, pv;                                           // [q][key]: lane = key column l31, rows = queries
#pragma unroll
            for (int r = 0; r < 16; ++r) { sv[r] = 0.f; pv[r] = 0.f; }
            bf16x8 qa[KS], da[KS];                                   // all row fragments in flight before the first MFMA
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qa[ks] = *reinterpret_cast<const bf16x8*>(q_row + hh * 32 * PR + k_off + ks * 32);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) da[ks] = *reinterpret_cast<const bf16x8*>(do_row + hh * 32 * PR + k_off + ks * 32);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[ks], kf[ks], sv, 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) pv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[ks], vf[ks], pv, 0, 0, 0);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {                         // registers 4g4..4g4+3 = 4 consecutive queries
                const int ql = hh * 32 + 8 * g4 + 4 * h;             // query row inside the tile
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + ql);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dl_s + ql);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = g4 * 4 + j;
                    float x = __builtin_fmaf(sv[r], sl2, -l4[j]);
                    x = (ki <= t * 64 + ql + j) ? x : -INFINITY;     // select, not a branch; rows past the end have lse = +inf
                    const float e = __builtin_amdgcn_exp2f(x);
                    sv[r] = e;
                    pv[r] = a.scale * e * (pv[r] - d4[j]);
                }
            }
            bf16x8 pf[2], df[2];
            pf[0] = pack8(sv[0], sv[1], sv[2], sv[3], sv[4], sv[5], sv[6], sv[7]);
            pf[1] = pack8(sv[8], sv[9], sv[10], sv[11], sv[12], sv[13], sv[14], sv[15]);
            df[0] = pack8(pv[0], pv[1], pv[2], pv[3], pv[4], pv[5], pv[6], pv[7]);
            df[1] = pack8(pv[8], pv[9], pv[10], pv[11], pv[12], pv[13], pv[14], pv[15]);
            mma_transposed_half(do_tr + hh * 32 * PT, v_off, pf, dva);       // dV^T[d][key] += dO^T[d][q] P[q][key]
            mma_transposed_half(q_tr + hh * 32 * PT, v_off, df, dka);        // dK^T[d][key] += Q^T[d][q] dS[q][key]
        }
        }
        if (t + 1 < nqt) write_lds(((t - t0) + 1) & 1);
        __syncthreads();
    }
    if (ki < a.S) {
        float* ok = dk_part + ((int64_t)hq * a.S + ki) * HD;
        float* ov = dv_part + ((int64_t)hq * a.S + ki) * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = dt * 32 + g4 * 8 + h * 4;
                *reinterpret_cast<f32x4*>(ok + d0) = f32x4{dka[dt][g4 * 4 + 0], dka[dt][g4 * 4 + 1], dka[dt][g4 * 4 + 2], dka[dt][g4 * 4 + 3]};
                *reinterpret_cast<f32x4*>(ov + d0) = f32x4{dva[dt][g4 * 4 + 0], dva[dt][g4 * 4 + 1], dva[dt][g4 * 4 + 2], dva[dt][g4 * 4 + 3]};
            }
    }
}

// dk[s][g*128 + d] = sum over the q heads of group g of the per-head partials (fixed order: deterministic); same for dv
__global__ __launch_bounds__(256) void attn_bwd_reduce_k(const float* __restrict__ dk_part, const float* __restrict__ dv_part, bf16* __restrict__ dk,
                                                         bf16* __restrict__ dv, int64_t lddkv, int S, int Hq, int Hkv) {
    const int rep = Hq / Hkv;
    const int64_t total = (int64_t)S * Hkv * (HD / 4);
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int c = id % (HD / 4);
        const int g = (id / (HD / 4)) % Hkv;
        const int s = id / ((int64_t)(HD / 4) * Hkv);
        f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = ak;
        for (int j = 0; j < rep; ++j) {
            const int64_t off = ((int64_t)(g * rep + j) * S + s) * HD + c * 4;
            ak += *reinterpret_cast<const f32x4*>(dk_part + off);
            av += *reinterpret_cast<const f32x4*>(dv_part + off);
        }
        *reinterpret_cast<bf16x4*>(dk + (int64_t)s * lddkv + g * HD + c * 4) = bf16x4{(bf16)ak[0], (bf16)ak[1], (bf16)ak[2], (bf16)ak[3]};
        *reinterpret_cast<bf16x4*>(dv + (int64_t)s * lddkv + g * HD + c * 4) = bf16x4{(bf16)av[0], (bf16)av[1], (bf16)av[2], (bf16)av[3]};
    }
}

}  // namespace

extern "C" int64_t ufv_attention_bwd_fused_ws_bytes(int S, int Hq) {
    return (int64_t)Hq * S * sizeof(float) + 2 * (int64_t)Hq * S * 128 * sizeof(float) + 1024;       // delta + dK / dV partials per q head
}

extern "C" int ufv_attention_bwd_fused(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* o, int64_t ldo,
                                       const void* dO, int64_t lddo, const float* lse, void* dq, int64_t lddq, void* dk, void* dv,
                                       int64_t lddkv, int S, int Hq, int Hkv, int hd, float scale, void* ws, void* stream) {
    UFV_REQUIRE(q && k && v && o && dO && lse && dq && dk && dv && ws && S > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0,
                "ufv_attention_bwd_fused: bad arguments");
    UFV_REQUIRE(hd == 128, "ufv_attention_bwd_fused: head_dim %d (128 only; use ufv_attention_bwd)", hd);
    UFV_REQUIRE(ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && lddq % 4 == 0 && lddkv % 4 == 0 &&
                ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)dO % 16 == 0) &&
                ((uintptr_t)o % 16 == 0) && ((uintptr_t)dq % 8 == 0) && ((uintptr_t)dk % 8 == 0) && ((uintptr_t)dv % 8 == 0),
                "ufv_attention_bwd_fused: rows must be 16-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    float* delta = reinterpret_cast<float*>(w);
    float* dk_part = delta + (((int64_t)Hq * S + 63) & ~(int64_t)63);
    float* dv_part = dk_part + (int64_t)Hq * S * 128;
    BwdArgs a;
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.dO = (const bf16*)dO;
    a.ldq = ldq; a.ldkv = ldkv; a.lddo = lddo; a.lse = lse; a.delta = delta; a.S = S; a.Hq = Hq; a.Hkv = Hkv; a.scale = scale;
    constexpr int SM_DQ = 2 * STAGE_DQ, SM_DKV = 2 * STAGE_DKV;
    UFV_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_k), hipFuncAttributeMaxDynamicSharedMemorySize, SM_DQ);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_k), hipFuncAttributeMaxDynamicSharedMemorySize, SM_DKV);
    );
    const int64_t nw = (int64_t)S * Hq;
    hipLaunchKernelGGL(attn_bwd_delta_k, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, st, (const bf16*)dO, lddo, (const bf16*)o, ldo, S, Hq, delta);
    UFV_CHECK_LAUNCH();
    const int nblk = (S + 127) / 128;
    hipLaunchKernelGGL(attn_bwd_dq_k, dim3(nblk, Hq), dim3(NT), SM_DQ, st, a, (bf16*)dq, lddq);
    UFV_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_bwd_dkv_k, dim3(nblk, Hq), dim3(NT), SM_DKV, st, a, dk_part, dv_part);
    UFV_CHECK_LAUNCH();
    const int64_t total = (int64_t)S * Hkv * 32;
    hipLaunchKernelGGL(attn_bwd_reduce_k, dim3((unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), dim3(256), 0, st, dk_part,
                       dv_part, (bf16*)dk, (bf16*)dv, lddkv, S, Hq, Hkv);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}
