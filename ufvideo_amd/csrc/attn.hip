// Attention for gfx950: softmax(q k^T * scale [+causal]) v with fp32 softmax statistics.
//
// MFMA kernel (attn_fwd_mfma): flash-style, one wave = 32 query rows, a block of NW waves shares
// 64-key K/V tiles staged through LDS (register-staged, double buffered, one barrier per tile).
// The score tile is computed TRANSPOSED, S^T = K * Q^T with v_mfma_f32_32x32x16_bf16, so that a
// lane owns one query column: row max / row sum are in-register reductions plus one lane^32
// exchange, and the exponentiated tile is directly the B operand of the second product
// O^T = V^T * P^T (accumulator-as-operand, no LDS round trip).  V^T fragments come from the
// row-major V tile with the hardware transpose read ds_read_b64_tr_b16.  head_dim is padded in
// LDS/registers only (72 -> 80 for QK^T, 96 for PV), never in HBM.
//
// Generic kernel (attn_generic): any head_dim / length, scores kept in LDS; used for decode
// (Sq = 1) and tiny test shapes.
#include "common.h"
#include "../../include/ufv.h"
#include "gemm_state.h"

#ifndef UFV_STAMP
#define UFV_STAMP(i)            // tools/attn_lab.hip defines it to record s_memtime at point i (block-level timeline)
#endif

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// 8 fp32 -> bf16x8 with packed converts (v_cvt_pk_bf16_f32), no byte permutes
__device__ __forceinline__ bf16x8 pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    const bf16x2 p0 = __builtin_convertvector((f32x2){a0, a1}, bf16x2), p1 = __builtin_convertvector((f32x2){a2, a3}, bf16x2);
    const bf16x2 p2 = __builtin_convertvector((f32x2){a4, a5}, bf16x2), p3 = __builtin_convertvector((f32x2){a6, a7}, bf16x2);
    const bf16x4 q0 = __builtin_shufflevector(p0, p1, 0, 1, 2, 3), q1 = __builtin_shufflevector(p2, p3, 0, 1, 2, 3);
    return __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Eight ds_read_b64_tr_b16 for one 32-row d-tile of V^T (4 chunks of 16 keys x {keys 0-3, keys 8-11} sub-blocks), issued
// from inline asm: hipcc's builtin for this instruction makes it wait vmcnt(0) for every LDS-DMA in flight (no alias
// info), which would drain the K/V prefetch ring each iteration.  The caller counts lgkmcnt by hand.
template <int PV_, int BASE>
__device__ __forceinline__ void tr_read8(unsigned addr, bf16x4 (&v)[8]) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8 offset:%9\n\t"
        "ds_read_b64_tr_b16 %1, %8 offset:%10\n\t"
        "ds_read_b64_tr_b16 %2, %8 offset:%11\n\t"
        "ds_read_b64_tr_b16 %3, %8 offset:%12\n\t"
        "ds_read_b64_tr_b16 %4, %8 offset:%13\n\t"
        "ds_read_b64_tr_b16 %5, %8 offset:%14\n\t"
        "ds_read_b64_tr_b16 %6, %8 offset:%15\n\t"
        "ds_read_b64_tr_b16 %7, %8 offset:%16"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(addr), "n"(BASE + 0 * 16 * PV_), "n"(BASE + 0 * 16 * PV_ + 8 * PV_), "n"(BASE + 1 * 16 * PV_),
          "n"(BASE + 1 * 16 * PV_ + 8 * PV_), "n"(BASE + 2 * 16 * PV_), "n"(BASE + 2 * 16 * PV_ + 8 * PV_),
          "n"(BASE + 3 * 16 * PV_), "n"(BASE + 3 * 16 * PV_ + 8 * PV_)
        : "memory");
}

// ds_read_b128 from inline asm: hipcc makes an ordinary LDS load wait vmcnt(0) while LDS-DMA writes are in flight (it
// cannot prove the ring stages disjoint), which drains the K/V prefetch every tile.  The caller counts lgkmcnt by hand.
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read128(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void lgkm_wait2(bf16x8& a, bf16x8& b) {      // ties the fragments to the wait so their MFMAs stay behind it
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}

struct AttnArgs {
    const bf16 *q, *k, *v;
    bf16* o;
    int64_t q_bs, q_ss, k_bs, k_ss, v_bs, v_ss, o_bs, o_ss;
    int B, Hq, Hkv, Sq, Sk, hd;
    float scale;
    int q_pos0;
    float* lse;        // optional [B, Hq, Sq]: log2-domain log-sum-exp of the scaled scores (m + log2 l), for the fused backward
};

template <int HD>
struct Cfg {
    static constexpr int KS = (HD + 15) / 16;              // k-steps of QK^T (K padded to 16)
    static constexpr int DT = (HD + 31) / 32;              // 32-row d tiles of O^T
    static constexpr int KC = KS * 2;                      // 16-byte chunks per K row (incl. zero pad)
    static constexpr int VC = HD / 8;                      // 16-byte chunks per V row
    static constexpr int PK = ((KC | 1)) * 16;             // K row pitch: odd number of 16-B slots -> conflict-free b128 reads
    static constexpr int PV = (DT * 64 <= 192) ? 192 : 320;  // V row pitch = 64 or 192 (mod 256): conflict-free tr reads
    static constexpr int STAGE = 64 * PK + 64 * PV;
};

// NG = 2 (key split): the block holds TWO groups of NW waves over the same 32*NW queries; group g takes the key tiles t = g, g + 2, ... with its
// own K/V staging buffers, and the groups' (m, l, O) are merged through LDS at the end (lane to lane: the same lane of the same wave holds the same
// query and output columns in both groups).  A causal launch is as long as its last q-tile's walk over all keys (38 tiles at S = 2399, with 532
// blocks for 512 slots the light blocks cannot fill the time the heavy ones need); the split halves that walk.
template <int HD, int NW, bool CAUSAL, int NG = 1>
__global__ __launch_bounds__(NW * NG * 64, (NG > 1 ? 1 : (HD <= 96 ? 3 : 2))) void attn_fwd_mfma(AttnArgs a) {
    using C = Cfg<HD>;
    constexpr int KS = C::KS, DT = C::DT, KC = C::KC, VC = C::VC, PK = C::PK, PV = C::PV, STAGE = C::STAGE;
    constexpr int NT = NW * 64;
    constexpr int CK = (64 * KC + NT - 1) / NT, CV = (64 * VC + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem_all[];

    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = NG > 1 ? wave_all / NW : 0, wave = NG > 1 ? wave_all % NW : wave_all;
    const int tid = wave * 64 + lane;                           // thread index inside the group: staging work is split per group
    char* smem = smem_all + grp * (2 * STAGE);
    const int h = lane >> 5, l31 = lane & 31;
    int qt = blockIdx.x, hq = blockIdx.y, b = blockIdx.z;
    if (CAUSAL) {
        // Causal work per block grows with its q-tile: hand the blocks out globally heaviest first (all heads' last q-tiles,
        // then the next ones, ...) so that the light blocks fill the slots the heavy ones free up.  (Pairing heavy with light
        // blocks per CU on top of this order measured no further gain.)
        const int nqt = gridDim.x, per = a.Hq * a.B;
        const int i = blockIdx.x + nqt * (blockIdx.y + a.Hq * blockIdx.z);
        qt = nqt - 1 - i / per;
        const int g = i % per;
        hq = g % a.Hq; b = g / a.Hq;
    }
    const int hkv = hq / (a.Hq / a.Hkv);
    const int qblk0 = qt * 32 * NW, q0 = qblk0 + wave * 32;
    const int qi = q0 + l31;                                                       // this lane's query row

    // ---- Q^T fragments (B operand): lane (col q = l31, k = 8h + j) <- Q[q][16ks + 8h + j]
    bf16x8 qf[KS];
    {
        const bf16* qp = a.q + b * a.q_bs + (int64_t)min(qi, a.Sq - 1) * a.q_ss + hq * HD;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = ks * 16 + h * 8;
            if (d0 < HD) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + d0);
            else qf[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    int kmax = a.Sk;
    if (CAUSAL) kmax = min(a.Sk, a.q_pos0 + min(qblk0 + 32 * NW, a.Sq));
    const int ntiles = (kmax + 63) / 64;

    // ---- tile staging (global -> registers -> LDS)
    const bf16* kbase = a.k + b * a.k_bs + hkv * HD;
    const bf16* vbase = a.v + b * a.v_bs + hkv * HD;
    bf16x8 kreg[CK], vreg[CV];
    auto issue_loads = [&](int tile) {
#pragma unroll
        for (int i = 0; i < CK; ++i) {
            const int id = tid + i * NT, row = id / KC, c = id % KC;
            if (id < 64 * KC) {
                if (c * 8 < HD)
                    kreg[i] = *reinterpret_cast<const bf16x8*>(kbase + (int64_t)min(tile * 64 + row, a.Sk - 1) * a.k_ss + c * 8);
                else
                    kreg[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int i = 0; i < CV; ++i) {
            const int id = tid + i * NT, row = id / VC, c = id % VC;
            if (id < 64 * VC)
                vreg[i] = *reinterpret_cast<const bf16x8*>(vbase + (int64_t)min(tile * 64 + row, a.Sk - 1) * a.v_ss + c * 8);
        }
    };
    auto write_lds = [&](int s) {
        char* kb = smem + s * STAGE;
        char* vb = kb + 64 * PK;
#pragma unroll
        for (int i = 0; i < CK; ++i) {
            const int id = tid + i * NT, row = id / KC, c = id % KC;
            if (id < 64 * KC) *reinterpret_cast<bf16x8*>(kb + row * PK + c * 16) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < CV; ++i) {
            const int id = tid + i * NT, row = id / VC, c = id % VC;
            if (id < 64 * VC) *reinterpret_cast<bf16x8*>(vb + row * PV + c * 16) = vreg[i];
        }
    };

    f32x16 oacc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;                 // running max (log2 domain, scaled) and partial row sum
    const float sl2 = a.scale * 1.4426950408889634f;
    constexpr float RESCALE_THR = 6.0f;                   // defer the O rescale while the row max grows by < 2^6

    // per-lane LDS offsets
    const int k_off = l31 * PK + h * 16;                                  // + hh*32*PK + ks*32
    const int v_off = (4 * h + ((lane & 15) >> 2)) * PV + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;   // + (hh*32+16s(+8))*PV + dt*64

    if (grp < ntiles) { issue_loads(grp); write_lds(0); }
    __syncthreads();
    for (int it = 0; it * NG < ntiles; ++it) {
        const int t = it * NG + grp;
        if (t < ntiles) {                                          // (wave-uniform; the last round of a key split may leave a group without a tile)
        if (t + NG < ntiles) issue_loads(t + NG);
        const char* kb = smem + (it & 1) * STAGE;
        const char* vb = kb + 64 * PK;
        const int kbase_idx = t * 64;
        // ---- S^T[key][q] for 64 keys: two 32x32 tiles
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(kb + k_off + ks * 32);
            const bf16x8 kf1 = *reinterpret_cast<const bf16x8*>(kb + 32 * PK + k_off + ks * 32);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0, qf[ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1, qf[ks], s1, 0, 0, 0);
        }
        // ---- masking only where a tile can contain invalid keys (sequence end / causal diagonal): wave-uniform
        bool need_mask = kbase_idx + 64 > a.Sk;
        if (CAUSAL) need_mask = need_mask || (kbase_idx + 63 > a.q_pos0 + q0);
        if (need_mask) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kj = kbase_idx + (r & 3) + 8 * (r >> 2) + 4 * h;
                bool ok0 = kj < a.Sk, ok1 = kj + 32 < a.Sk;
                if (CAUSAL) { ok0 = ok0 && (kj <= a.q_pos0 + qi); ok1 = ok1 && (kj + 32 <= a.q_pos0 + qi); }
                s0[r] = ok0 ? s0[r] : -INFINITY;
                s1[r] = ok1 ? s1[r] : -INFINITY;
            }
        }
        // ---- row max (raw scores), lane pair exchange, deferred rescale
        float tmax = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = max3f(tmax, s0[r], s1[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * sl2;
        if (__any(tmax > m_run + RESCALE_THR)) {
            const float mnew = fmaxf(m_run, tmax);
            const float alpha = (mnew == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run - mnew);   // exp2(-inf) = 0 on the first tile
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
            m_run = mnew;
        }
        // ---- P = exp2(s*scale*log2e - m): one FMA + one v_exp per element; masked keys give exp2(-inf) = 0
        const float msub = (m_run == -INFINITY) ? 0.f : m_run;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], sl2, -msub));
            s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], sl2, -msub));
            psum += s0[r] + s1[r];
        }
        l_run += psum;
        // ---- P^T as B operand: k-step sp of half hh uses registers 8sp..8sp+7
        bf16x8 pf[4];
        pf[0] = pack8(s0[0], s0[1], s0[2], s0[3], s0[4], s0[5], s0[6], s0[7]);
        pf[1] = pack8(s0[8], s0[9], s0[10], s0[11], s0[12], s0[13], s0[14], s0[15]);
        pf[2] = pack8(s1[0], s1[1], s1[2], s1[3], s1[4], s1[5], s1[6], s1[7]);
        pf[3] = pack8(s1[8], s1[9], s1[10], s1[11], s1[12], s1[13], s1[14], s1[15]);
        // ---- O^T[d][q] += sum_key V[key][d] P[key][q]; V^T fragments via transposed LDS reads
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {           // c = hh*2 + sp : 16-key chunk
                const char* vp = vb + (16 * c) * PV + v_off + dt * 64;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vp));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vp + 8 * PV));
                const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[c], oacc[dt], 0, 0, 0);
            }
        }
        if (t + NG < ntiles) write_lds((it + 1) & 1);
        }
        __syncthreads();
    }

    if constexpr (NG > 1) {
        // ---- merge the groups: group 1 parks (m, l, O) in LDS (its K/V buffers are idle now), group 0 folds them into its own
        float* mb = reinterpret_cast<float*>(smem_all) + (size_t)(wave * 64 + lane) * (DT * 16 + 2);
        if (grp == 1) {
            mb[0] = m_run; mb[1] = l_run;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) mb[2 + i * 16 + r] = oacc[i][r];
        }
        __syncthreads();
        if (grp == 1) return;
        const float m1 = mb[0], l1 = mb[1];
        const float mnew = fmaxf(m_run, m1);
        const float a0 = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - mnew);
        const float a1 = (m1 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m1 - mnew);
        l_run = l_run * a0 + l1 * a1;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[i][r] = oacc[i][r] * a0 + mb[2 + i * 16 + r] * a1;
        m_run = mnew;
    }

    // ---- normalise and store O[q][d], d = 32dt + (r&3) + 8(r>>2) + 4h  (4 consecutive d per register quad)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (a.lse && h == 0 && qi < a.Sq) a.lse[((int64_t)b * a.Hq + hq) * a.Sq + qi] = m_run + log2f(l_tot);
    if (qi < a.Sq) {
        bf16* op = a.o + b * a.o_bs + (int64_t)qi * a.o_ss + hq * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = dt * 32 + g4 * 8 + h * 4;
                if (d0 < HD) {
                    bf16x4 ov = {(bf16)(oacc[dt][g4 * 4 + 0] * inv), (bf16)(oacc[dt][g4 * 4 + 1] * inv),
                                 (bf16)(oacc[dt][g4 * 4 + 2] * inv), (bf16)(oacc[dt][g4 * 4 + 3] * inv)};
                    *reinterpret_cast<bf16x4*>(op + d0) = ov;
                }
            }
    }
}


// ---------------------------------------------------------------------------------------------------------
// LDS-DMA variant (head dims whose row is an odd number of 16-byte chunks, e.g. 72): K/V tiles are copied
// global -> LDS by global_load_lds (no staging registers) into a 3-stage ring, TWO tiles ahead of the compute,
// retired with a counted s_waitcnt vmcnt and a raw s_barrier (one per 64-key tile).  Rows keep their natural
// pitch (HD*2 bytes = odd number of 16-B slots -> conflict-free ds_read_b128 K fragments); the zero padding of
// the QK^T contraction (72 -> 80) lives in the Q fragment, so the K "pad" chunk may alias the next row.
// ---------------------------------------------------------------------------------------------------------
// PP (ping-pong): the waves of a SIMD come from two groups that run half a tile apart -- a second barrier splits every tile
// step into [QK^T MFMA, first half of the softmax] and [second half of the softmax, PV MFMA], and group 1 starts one barrier
// late, so one group's MFMAs overlap the other group's exp/convert VALU work instead of all waves alternating in lockstep
// between the two units.  Needs a 4-stage K/V ring (the late group still reads tile t-1 when tile t+2 is prefetched).
template <int HD, int NW, bool CAUSAL, bool PP>
__global__ __launch_bounds__(NW * 64, (NW > 8 ? 3 : 2)) void attn_fwd_mfma_dma(AttnArgs a) {
    constexpr int KS = (HD + 15) / 16, DT = (HD + 31) / 32;
    constexpr int PK = HD * 2, PV = HD * 2;                 // natural row pitch
    constexpr int TILEB = 64 * PK, NI = TILEB / 1024;       // bytes / 1-KiB DMA pieces per operand tile
    static_assert(TILEB % 1024 == 0 && (HD % 8) == 0 && ((HD / 8) & 1) == 1, "row must be an odd number of 16-B chunks");
    constexpr int STG = 2 * TILEB + 256;                    // [K tile][16 B zeros + pad][V tile][tail pad]
    constexpr int NST = PP ? 4 : 3;                         // ring stages
    constexpr int VOFF = TILEB + 128;
    constexpr int NPW = (2 * NI + NW - 1) / NW;             // DMA pieces per wave per tile (max)
    constexpr int NPW_MIN = (2 * NI) / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    UFV_STAMP(0);
    // 1-D grid; consecutive launch ids go to different XCDs, so give the q-tiles of one (batch, head) ids that are
    // 8 apart: they then share one XCD's L2 for their common K/V.
    const int nqt = (a.Sq + 32 * NW - 1) / (32 * NW);
    int qt, hq, b;
    {
        const int id = blockIdx.x, x = id & 7, slot = id >> 3;
        const int groups = a.Hq * a.B;                       // (head, batch) pairs
        const int gper = (groups + 7) >> 3;                  // pairs per XCD
        const int gi = slot / nqt;
        const int g = x * gper + gi;
        qt = slot % nqt;
        if (g >= groups || gi >= gper) return;
        hq = g % a.Hq; b = g / a.Hq;
        if (CAUSAL) qt = nqt - 1 - qt;
    }
    const int hkv = hq / (a.Hq / a.Hkv);
    const int qblk0 = qt * 32 * NW, q0 = qblk0 + wave * 32;
    const int qi = q0 + l31;

    // Q: the wave's 32 rows are fetched as whole rows (16 bytes per lane in row-major chunk order: a wave instruction covers 7
    // rows x 144 B; a per-lane fetch of "my row" touches 32 lines per instruction), parked in the wave's staging area behind the
    // K/V ring (row pitch HD*2 + 16 bytes) and read back as MFMA B fragments: lane (col q = l31, k = 8h + j) <- Q[q][16ks + 8h + j]
    constexpr int OP = HD * 2 + 16, CPRW = HD / 8;
    char* stg = smem + NST * STG + wave * (32 * OP);
    bf16x8 qf[KS];
    {
        const bf16* qbase = a.q + b * a.q_bs + hq * HD;
#pragma unroll
        for (int i = 0; i < (32 * CPRW + 63) / 64; ++i) {
            const int c = lane + 64 * i, row = c / CPRW, ch = c % CPRW;
            if (c < 32 * CPRW)
                *reinterpret_cast<bf16x8*>(stg + row * OP + ch * 16) =
                    *reinterpret_cast<const bf16x8*>(qbase + (int64_t)min(q0 + row, a.Sq - 1) * a.q_ss + ch * 8);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = ks * 16 + h * 8;
            if (d0 < HD) qf[ks] = *reinterpret_cast<const bf16x8*>(stg + l31 * OP + d0 * 2);
            else qf[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // Q in registers before any DMA is in flight
    UFV_STAMP(1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));   // ... and hipcc's own scoreboard must see them consumed here:
    // it does not understand the asm wait above, and with LDS-DMA issued in between it would otherwise put a vmcnt(0) in
    // front of the first MFMA of EVERY iteration (draining the K/V prefetch ring each tile)
    int kmax = a.Sk;
    if (CAUSAL) kmax = min(a.Sk, a.q_pos0 + min(qblk0 + 32 * NW, a.Sq));
    const int ntiles = (kmax + 63) / 64;

    const bf16* kbase = a.k + b * a.k_bs + hkv * HD;
    const bf16* vbase = a.v + b * a.v_bs + hkv * HD;
    // this wave's DMA pieces: piece j = wave + i*NW, j < 2*NI; op = j / NI (0 K, 1 V); lane -> (row, chunk)
    int p_row[NPW], p_chunk[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int j = wave + i * NW, jj = j % NI;
        const int c16 = jj * 64 + lane;
        p_row[i] = c16 / (HD / 8);
        p_chunk[i] = c16 % (HD / 8);
    }
    auto dma_tile = [&](int stage, int tile) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int j = wave + i * NW;
            if (j < 2 * NI) {
                const bool isv = j >= NI;
                const int jj = j - (isv ? NI : 0);
                const int row = min(tile * 64 + p_row[i], a.Sk - 1);
                const bf16* src = (isv ? vbase + (int64_t)row * a.v_ss : kbase + (int64_t)row * a.k_ss) + p_chunk[i] * 8;
                char* dst = smem + stage * STG + (isv ? VOFF : 0) + jj * 1024;
                __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(dst), 16, 0, 0);
            }
        }
    };
    // zero the 16 bytes after every K tile (read as the d>=HD tail of the last key row)
    if (tid < NST * 4) reinterpret_cast<float*>(smem + (tid >> 2) * STG + TILEB)[tid & 3] = 0.f;

    f32x16 oacc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sl2 = a.scale * 1.4426950408889634f;
    constexpr float RESCALE_THR = 6.0f;
    const int k_off = l31 * PK + h * 16;
    const int v_off = (4 * h + ((lane & 15) >> 2)) * PV + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

    dma_tile(0, 0);
    if (ntiles > 1) dma_tile(1, 1);
    if (ntiles > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW_MIN) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    UFV_STAMP(2);
    const bool grp1 = PP && ((wave >> 2) & 1);                // waves 4..7: one barrier behind (SIMD s holds waves s, s+4, s+8)
    if (grp1) __builtin_amdgcn_s_barrier();
    int st = 0;
    for (int t = 0; t < ntiles; ++t) {
        if (t + 2 < ntiles) dma_tile(st + 2 >= NST ? st + 2 - NST : st + 2, t + 2);       // ring slot (t+2) % NST
        const char* kb = smem + st * STG;
        const char* vb = kb + VOFF;
        const int kbase_idx = t * 64;
        // ---- S^T[key][q] for 64 keys: two 32x32 tiles
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
        {
            static_assert(KS == 5, "hand-counted lgkmcnt schedule below is written for 5 k-steps (head_dim 72)");
            const unsigned kaddr = (unsigned)(uintptr_t)LDS_PTR(kb + k_off);
            bf16x8 kf[KS][2];
            kf[0][0] = lds_read128<0 * 32>(kaddr); kf[0][1] = lds_read128<32 * PK + 0 * 32>(kaddr);
            kf[1][0] = lds_read128<1 * 32>(kaddr); kf[1][1] = lds_read128<32 * PK + 1 * 32>(kaddr);
            kf[2][0] = lds_read128<2 * 32>(kaddr); kf[2][1] = lds_read128<32 * PK + 2 * 32>(kaddr);
            kf[3][0] = lds_read128<3 * 32>(kaddr); kf[3][1] = lds_read128<32 * PK + 3 * 32>(kaddr);
            kf[4][0] = lds_read128<4 * 32>(kaddr); kf[4][1] = lds_read128<32 * PK + 4 * 32>(kaddr);
            lgkm_wait2<8>(kf[0][0], kf[0][1]);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0][0], qf[0], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0][1], qf[0], s1, 0, 0, 0);
            lgkm_wait2<6>(kf[1][0], kf[1][1]);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1][0], qf[1], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1][1], qf[1], s1, 0, 0, 0);
            lgkm_wait2<4>(kf[2][0], kf[2][1]);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2][0], qf[2], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2][1], qf[2], s1, 0, 0, 0);
            lgkm_wait2<2>(kf[3][0], kf[3][1]);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[3][0], qf[3], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[3][1], qf[3], s1, 0, 0, 0);
            lgkm_wait2<0>(kf[4][0], kf[4][1]);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[4][0], qf[4], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[4][1], qf[4], s1, 0, 0, 0);
        }
        // ---- V^T fragments for the first two d-tiles: issued now (K fragment reads have been consumed by the MFMAs
        //      above, so the LDS queue holds nothing else) and landing under the softmax arithmetic below.
        const unsigned vaddr = (unsigned)(uintptr_t)LDS_PTR(vb + v_off);
        bf16x4 vr[2][8];                                                // two register sets, d-tile dt uses set dt&1
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tr_read8<PV, 0>(vaddr, vr[0]);
        if (DT > 1) tr_read8<PV, 64>(vaddr, vr[1]);
        __builtin_amdgcn_sched_barrier(0);
        // ---- masking only where a tile can contain invalid keys (sequence end / causal diagonal): wave-uniform
        bool need_mask = kbase_idx + 64 > a.Sk;
        if (CAUSAL) need_mask = need_mask || (kbase_idx + 63 > a.q_pos0 + q0);
        if (need_mask) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kj = kbase_idx + (r & 3) + 8 * (r >> 2) + 4 * h;
                bool ok0 = kj < a.Sk, ok1 = kj + 32 < a.Sk;
                if (CAUSAL) { ok0 = ok0 && (kj <= a.q_pos0 + qi); ok1 = ok1 && (kj + 32 <= a.q_pos0 + qi); }
                s0[r] = ok0 ? s0[r] : -INFINITY;
                s1[r] = ok1 ? s1[r] : -INFINITY;
            }
        }
        // ---- row max (raw scores), lane pair exchange, deferred rescale
        float tmax = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = max3f(tmax, s0[r], s1[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * sl2;
        if (__any(tmax > m_run + RESCALE_THR)) {
            const float mnew = fmaxf(m_run, tmax);
            const float alpha = (mnew == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run - mnew);   // exp2(-inf) = 0 on the first tile
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
            m_run = mnew;
        }
        // ---- P = exp2(s*scale*log2e - m): one FMA + one v_exp per element; masked keys give exp2(-inf) = 0
        const float msub = (m_run == -INFINITY) ? 0.f : m_run;
        float psum = 0.f;
        bf16x8 pf[4];                 // P^T as B operand: k-step sp of half hh uses registers 8sp..8sp+7
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], sl2, -msub));
            psum += s0[r];
        }
        pf[0] = pack8(s0[0], s0[1], s0[2], s0[3], s0[4], s0[5], s0[6], s0[7]);
        pf[1] = pack8(s0[8], s0[9], s0[10], s0[11], s0[12], s0[13], s0[14], s0[15]);
        if constexpr (PP) {           // half-step seam: the other group is entering its MFMA-heavy half
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW_MIN) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], sl2, -msub));
            psum += s1[r];
        }
        l_run += psum;
        pf[2] = pack8(s1[0], s1[1], s1[2], s1[3], s1[4], s1[5], s1[6], s1[7]);
        pf[3] = pack8(s1[8], s1[9], s1[10], s1[11], s1[12], s1[13], s1[14], s1[15]);
        // ---- O^T[d][q] += sum_key V[key][d] P[key][q]; V^T fragments via transposed LDS reads
        {
            static_assert(DT <= 4, "V fragment register sets");
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (dt + 1 < DT) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");     // set dt landed, set dt+1 in flight
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {       // c = hh*2 + sp : 16-key chunk
                    const bf16x8 vf = __builtin_shufflevector(vr[dt & 1][2 * c], vr[dt & 1][2 * c + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[c], oacc[dt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (dt + 2 < DT) {                   // refill this set for d-tile dt+2 (its MFMAs above have issued)
                    if (dt == 0) tr_read8<PV, 128>(vaddr, vr[0]);
                    else tr_read8<PV, 192>(vaddr, vr[1]);
                }
            }
        }
        // tile t+1 must have landed (every wave waits for its own pieces, then the barrier publishes them);
        // tile t+2 stays in flight across the barrier.
        if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW_MIN) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        st = (st == NST - 1) ? 0 : st + 1;
    }
    if (PP && !grp1) __builtin_amdgcn_s_barrier();           // balance the stagger barrier
    UFV_STAMP(3);

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    // O^T accumulators -> this wave's 32 rows x HD in LDS (row pitch HD*2 + 16 bytes; every wave is past its last ring read) ->
    // whole rows out with 16 bytes per lane: a wave instruction covers 7 rows x 144 B instead of 64 scattered 8-byte pieces
    // (per-lane stores at a row stride touch 32-64 lines each)
    char* ob = stg;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d0 = dt * 32 + g4 * 8 + h * 4;
            if (d0 < HD) {
                bf16x4 ov = {(bf16)(oacc[dt][g4 * 4 + 0] * inv), (bf16)(oacc[dt][g4 * 4 + 1] * inv),
                             (bf16)(oacc[dt][g4 * 4 + 2] * inv), (bf16)(oacc[dt][g4 * 4 + 3] * inv)};
                *reinterpret_cast<bf16x4*>(ob + l31 * OP + d0 * 2) = ov;
            }
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bf16* obase = a.o + b * a.o_bs + hq * HD;
#pragma unroll
    for (int i = 0; i < (32 * CPRW + 63) / 64; ++i) {
        const int c = lane + 64 * i, row = c / CPRW, ch = c % CPRW;
        if (c < 32 * CPRW && q0 + row < a.Sq)
            *reinterpret_cast<bf16x8*>(obase + (int64_t)(q0 + row) * a.o_ss + ch * 8) = *reinterpret_cast<const bf16x8*>(ob + row * OP + ch * 16);
    }
    UFV_STAMP(4);
}

#include "attn_vit.inc"
#include "attn_vit_p2.inc"
#include "attn_c128.inc"

// ---------------------------------------------------------------------------------------------------------
// Head-pair variant of the LDS-DMA kernel (diagnostic, kernel ids 7 / 8).  With 168 VGPRs a CU holds 3 waves per SIMD, i.e. ONE
// 9-wave block of the kernel above (3/2/2/2 waves per SIMD, 4 sequential rounds per CU, each exposing ~8 us of Q / first-tile
// latency).  Here a block is 2*NW = 12 waves = 3 per SIMD: two heads of one image side by side (each half with its own K/V ring),
// every wave walking Sq / (32*NW) = 3 query blocks in turn, so the launch is one round of 256 blocks with balanced SIMDs.
// Measured (MI355X, 32 x 16 x 576 x 72): 118.6 us lockstep / 104.9 us ping-pong against 94.9 us for the 9-wave blocks: a tile
// step of 12 waves takes 3.2 / 2.7 us against 1.9 us for 9 -- the CU retires ~4.6 wave-tiles per us whatever the arrangement
// (QK^T MFMA, softmax VALU and PV MFMA of co-resident waves largely serialise), so filling the idle SIMD slots buys nothing
// and the three drained pass seams cost more than the four block prologues they replace.  Kept as a tested variant.
// ---------------------------------------------------------------------------------------------------------
template <int HD, int NW, bool PP>
__global__ __launch_bounds__(2 * NW * 64, 1) void attn_fwd_pair(AttnArgs a) {
    constexpr bool CAUSAL = false;
    constexpr int KS = (HD + 15) / 16, DT = (HD + 31) / 32;
    constexpr int PK = HD * 2, PV = HD * 2;                 // natural row pitch
    constexpr int TILEB = 64 * PK, NI = TILEB / 1024;       // bytes / 1-KiB DMA pieces per operand tile
    static_assert(TILEB % 1024 == 0 && (HD % 8) == 0 && ((HD / 8) & 1) == 1, "row must be an odd number of 16-B chunks");
    constexpr int STG = 2 * TILEB + 256;                    // [K tile][16 B zeros + pad][V tile][tail pad]
    constexpr int NST = PP ? 4 : 3;                         // ring stages
    constexpr int VOFF = TILEB + 128;
    constexpr int NPW = (2 * NI + NW - 1) / NW;             // DMA pieces per wave per tile (max)
    constexpr int NPW_MIN = (2 * NI) / NW;
    extern __shared__ __attribute__((aligned(16))) char smem_all[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    // block = one (batch, head PAIR); waves 0..NW-1 work on the even head, NW..2NW-1 on the odd one, each half with its own
    // K/V ring.  1-D grid; consecutive launch ids go to different XCDs: XCD x takes a contiguous run of pairs.
    const int half = wave >= NW ? 1 : 0;
    const int w = wave - half * NW;                              // wave index inside its half
    char* smem = smem_all + half * (NST * STG);
    int hq, b;
    {
        const int id = blockIdx.x, x = id & 7, slot = id >> 3;
        const int groups = (a.Hq >> 1) * a.B;
        const int gper = (groups + 7) >> 3;
        const int g = x * gper + slot;
        if (g >= groups || slot >= gper) return;
        hq = 2 * (g % (a.Hq >> 1)) + half; b = g / (a.Hq >> 1);
    }
    const int hkv = hq / (a.Hq / a.Hkv);
    const int npass = a.Sq / (32 * NW);                          // the launcher guarantees Sq % (32 * NW) == 0
    const int kmax = a.Sk;
    const int ntiles = (kmax + 63) / 64;
    const bf16* kbase = a.k + b * a.k_bs + hkv * HD;
    const bf16* vbase = a.v + b * a.v_bs + hkv * HD;
    // this wave's DMA pieces: piece j = w + i*NW, j < 2*NI; op = j / NI (0 K, 1 V); lane -> (row, chunk)
    int p_row[NPW], p_chunk[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int j = w + i * NW, jj = j % NI;
        const int c16 = jj * 64 + lane;
        p_row[i] = c16 / (HD / 8);
        p_chunk[i] = c16 % (HD / 8);
    }
    auto dma_tile = [&](int stage, int tile) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int j = w + i * NW;
            if (j < 2 * NI) {
                const bool isv = j >= NI;
                const int jj = j - (isv ? NI : 0);
                const int row = min(tile * 64 + p_row[i], a.Sk - 1);
                const bf16* src = (isv ? vbase + (int64_t)row * a.v_ss : kbase + (int64_t)row * a.k_ss) + p_chunk[i] * 8;
                char* dst = smem + stage * STG + (isv ? VOFF : 0) + jj * 1024;
                __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(dst), 16, 0, 0);
            }
        }
    };
    // zero the 16 bytes after every K tile (read as the d>=HD tail of the last key row)
    if (w == 0 && lane < NST * 4) reinterpret_cast<float*>(smem + (lane >> 2) * STG + TILEB)[lane & 3] = 0.f;
    const float sl2 = a.scale * 1.4426950408889634f;
    constexpr float RESCALE_THR = 6.0f;
    const int k_off = l31 * PK + h * 16;
    const int v_off = (4 * h + ((lane & 15) >> 2)) * PV + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const bool grp1 = PP && ((wave >> 2) & 1);                // waves 4..7: one barrier behind (SIMD s holds waves s, s+4, s+8)

    for (int pass = 0; pass < npass; ++pass) {
    const int q0 = (pass * NW + w) * 32;
    const int qi = q0 + l31;
    bf16x8 qf[KS];
    {
        const bf16* qp = a.q + b * a.q_bs + (int64_t)min(qi, a.Sq - 1) * a.q_ss + hq * HD;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = ks * 16 + h * 8;
            if (d0 < HD) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + d0);
            else qf[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // Q in registers before any DMA is in flight
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));   // ... and hipcc's own scoreboard must see them consumed here:
    // it does not understand the asm wait above, and with LDS-DMA issued in between it would otherwise put a vmcnt(0) in
    // front of the first MFMA of EVERY iteration (draining the K/V prefetch ring each tile)
    f32x16 oacc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    dma_tile(0, 0);
    if (ntiles > 1) dma_tile(1, 1);
    if (ntiles > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW_MIN) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp1) __builtin_amdgcn_s_barrier();
    int st = 0;
    for (int t = 0; t < ntiles; ++t) {
        if (t + 2 < ntiles) dma_tile(st + 2 >= NST ? st + 2 - NST : st + 2, t + 2);       // ring slot (t+2) % NST
        const char* kb = smem + st * STG;
        const char* vb = kb + VOFF;
        const int kbase_idx = t * 64;
        // ---- S^T[key][q] for 64 keys: two 32x32 tiles
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
        {
            static_assert(KS == 5, "hand-counted lgkmcnt schedule below is written for 5 k-steps (head_dim 72)");
            const unsigned kaddr = (unsigned)(uintptr_t)LDS_PTR(kb + k_off);
            bf16x8 kf[KS][2];
            kf[0][0] = lds_read128<0 * 32>(kaddr); kf[0][1] = lds_read128<32 * PK + 0 * 32>(kaddr);
            kf[1][0] = lds_read128<1 * 32>(kaddr); kf[1][1] = lds_read128<32 * PK + 1 * 32>(kaddr);
            kf[2][0] = lds_read128<2 * 32>(kaddr); kf[2][1] = lds_read128<32 * PK + 2 * 32>(kaddr);
            kf[3][0] = lds_read128<3 * 32>(kaddr); kf[3][1] = lds_read128<32 * PK + 3 * 32>(kaddr);
            kf[4][0] = lds_read128<4 * 32>(kaddr); kf[4][1] = lds_read128<32 * PK + 4 * 32>(kaddr);
            lgkm_wait2<8>(kf[0][0], kf[0][1]);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0][0], qf[0], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0][1], qf[0], s1, 0, 0, 0);
            lgkm_wait2<6>(kf[1][0], kf[1][1]);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1][0], qf[1], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1][1], qf[1], s1, 0, 0, 0);
            lgkm_wait2<4>(kf[2][0], kf[2][1]);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2][0], qf[2], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2][1], qf[2], s1, 0, 0, 0);
            lgkm_wait2<2>(kf[3][0], kf[3][1]);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[3][0], qf[3], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[3][1], qf[3], s1, 0, 0, 0);
            lgkm_wait2<0>(kf[4][0], kf[4][1]);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[4][0], qf[4], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[4][1], qf[4], s1, 0, 0, 0);
        }
        // ---- V^T fragments for the first two d-tiles: issued now (K fragment reads have been consumed by the MFMAs
        //      above, so the LDS queue holds nothing else) and landing under the softmax arithmetic below.
        const unsigned vaddr = (unsigned)(uintptr_t)LDS_PTR(vb + v_off);
        bf16x4 vr[2][8];                                                // two register sets, d-tile dt uses set dt&1
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tr_read8<PV, 0>(vaddr, vr[0]);
        if (DT > 1) tr_read8<PV, 64>(vaddr, vr[1]);
        __builtin_amdgcn_sched_barrier(0);
        // ---- masking only where a tile can contain invalid keys (sequence end / causal diagonal): wave-uniform
        bool need_mask = kbase_idx + 64 > a.Sk;
        if (CAUSAL) need_mask = need_mask || (kbase_idx + 63 > a.q_pos0 + q0);
        if (need_mask) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kj = kbase_idx + (r & 3) + 8 * (r >> 2) + 4 * h;
                bool ok0 = kj < a.Sk, ok1 = kj + 32 < a.Sk;
                if (CAUSAL) { ok0 = ok0 && (kj <= a.q_pos0 + qi); ok1 = ok1 && (kj + 32 <= a.q_pos0 + qi); }
                s0[r] = ok0 ? s0[r] : -INFINITY;
                s1[r] = ok1 ? s1[r] : -INFINITY;
            }
        }
        // ---- row max (raw scores), lane pair exchange, deferred rescale
        float tmax = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = max3f(tmax, s0[r], s1[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * sl2;
        if (__any(tmax > m_run + RESCALE_THR)) {
            const float mnew = fmaxf(m_run, tmax);
            const float alpha = (mnew == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run - mnew);   // exp2(-inf) = 0 on the first tile
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
            m_run = mnew;
        }
        // ---- P = exp2(s*scale*log2e - m): one FMA + one v_exp per element; masked keys give exp2(-inf) = 0
        const float msub = (m_run == -INFINITY) ? 0.f : m_run;
        float psum = 0.f;
        bf16x8 pf[4];                 // P^T as B operand: k-step sp of half hh uses registers 8sp..8sp+7
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], sl2, -msub));
            psum += s0[r];
        }
        pf[0] = pack8(s0[0], s0[1], s0[2], s0[3], s0[4], s0[5], s0[6], s0[7]);
        pf[1] = pack8(s0[8], s0[9], s0[10], s0[11], s0[12], s0[13], s0[14], s0[15]);
        if constexpr (PP) {           // half-step seam: the other group is entering its MFMA-heavy half
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW_MIN) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], sl2, -msub));
            psum += s1[r];
        }
        l_run += psum;
        pf[2] = pack8(s1[0], s1[1], s1[2], s1[3], s1[4], s1[5], s1[6], s1[7]);
        pf[3] = pack8(s1[8], s1[9], s1[10], s1[11], s1[12], s1[13], s1[14], s1[15]);
        // ---- O^T[d][q] += sum_key V[key][d] P[key][q]; V^T fragments via transposed LDS reads
        {
            static_assert(DT <= 4, "V fragment register sets");
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (dt + 1 < DT) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");     // set dt landed, set dt+1 in flight
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {       // c = hh*2 + sp : 16-key chunk
                    const bf16x8 vf = __builtin_shufflevector(vr[dt & 1][2 * c], vr[dt & 1][2 * c + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[c], oacc[dt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (dt + 2 < DT) {                   // refill this set for d-tile dt+2 (its MFMAs above have issued)
                    if (dt == 0) tr_read8<PV, 128>(vaddr, vr[0]);
                    else tr_read8<PV, 192>(vaddr, vr[1]);
                }
            }
        }
        // tile t+1 must have landed (every wave waits for its own pieces, then the barrier publishes them);
        // tile t+2 stays in flight across the barrier.
        if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW_MIN) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        st = (st == NST - 1) ? 0 : st + 1;
    }
    if (PP && !grp1) __builtin_amdgcn_s_barrier();           // balance the stagger barrier

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qi < a.Sq) {
        bf16* op = a.o + b * a.o_bs + (int64_t)qi * a.o_ss + hq * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = dt * 32 + g4 * 8 + h * 4;
                if (d0 < HD) {
                    bf16x4 ov = {(bf16)(oacc[dt][g4 * 4 + 0] * inv), (bf16)(oacc[dt][g4 * 4 + 1] * inv),
                                 (bf16)(oacc[dt][g4 * 4 + 2] * inv), (bf16)(oacc[dt][g4 * 4 + 3] * inv)};
                    *reinterpret_cast<bf16x4*>(op + d0) = ov;
                }
            }
    }
    }   // pass
}

// ---------------------------------------------------------------------------------------------------------
// Decode attention (Sq == 1): HBM/L2-bound streaming of the KV cache.  Keys are split over blocks (flash-decoding):
// grid (nsplit, Hq, B); a wave reads 64/CPR keys per 16-byte load instruction (CPR = hd/8 lanes per key row),
// scores go to LDS, each block writes an un-normalised partial (max, sum, o[hd]); attn_decode_combine merges them.
// ---------------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void attn_decode_split(AttnArgs a, float* ws, int nsplit, const int* __restrict__ pos_dev) {
    constexpr int CPR = HD / 8, KPI = 64 / CPR;            // lanes per key row, keys per wave instruction
    extern __shared__ __attribute__((aligned(16))) char smem_d[];
    float* sc = reinterpret_cast<float*>(smem_d);           // [keys in this split]
    __shared__ float red[16];
    __shared__ float opart[4][HD];
    const int split = blockIdx.x, hq = blockIdx.y, b = blockIdx.z, hkv = hq / (a.Hq / a.Hkv);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane / CPR, ch = lane % CPR;
    // Sq == 1 with q_pos0 = Sk-1: every cached key is visible.  pos_dev (graph-captured decode): the key count lives in
    // device memory (position of the new token + 1), so the launch arguments are the same for every token
    const int nk = pos_dev ? *pos_dev + 1 : a.Sk;
    const int per = (nk + nsplit - 1) / nsplit;
    const int k0 = split * per, k1 = min(nk, k0 + per), n = max(k1 - k0, 0);
    float* out = ws + ((size_t)(b * a.Hq + hq) * nsplit + split) * (HD + 2);
    if (n == 0) {
        if (tid == 0) { out[0] = -INFINITY; out[1] = 0.f; }
        for (int d = tid; d < HD; d += 256) out[2 + d] = 0.f;
        return;
    }
    const bf16* qp = a.q + b * a.q_bs + hq * HD;
    const bf16x8 qv = *reinterpret_cast<const bf16x8*>(qp + ch * 8);
    float qf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[j] = (float)qv[j] * a.scale;
    const bf16* kb = a.k + b * a.k_bs + hkv * HD + ch * 8;
    const bf16* vb = a.v + b * a.v_bs + hkv * HD + ch * 8;
    // ---- scores
    float mx = -INFINITY;
    for (int j0 = wave * KPI; j0 < n; j0 += 4 * KPI) {
        const int j = j0 + sub;
        float s = 0.f;
        if (j < n) {
            const bf16x8 kv = *reinterpret_cast<const bf16x8*>(kb + (int64_t)(k0 + j) * a.k_ss);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += qf[e] * (float)kv[e];
        }
#pragma unroll
        for (int o = 1; o < CPR; o <<= 1) s += __shfl_xor(s, o, 64);
        if (j < n && ch == 0) sc[j] = s;
        if (j < n) mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < n; j += 256) {
        const float p = __expf(sc[j] - mx);
        sc[j] = p;
        sum += p;
    }
    sum = block_sum(sum, red + 4);
    __syncthreads();
    // ---- o = sum_j p_j v_j : lane accumulates its 8-d chunk over keys j = j0 + sub
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j0 = wave * KPI; j0 < n; j0 += 4 * KPI) {
        const int j = j0 + sub;
        if (j < n) {
            const float p = sc[j];
            const bf16x8 vv = *reinterpret_cast<const bf16x8*>(vb + (int64_t)(k0 + j) * a.v_ss);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += p * (float)vv[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int o = CPR; o < 64; o <<= 1) acc[e] += __shfl_xor(acc[e], o, 64);
    if (sub == 0)
#pragma unroll
        for (int e = 0; e < 8; ++e) opart[wave][ch * 8 + e] = acc[e];
    __syncthreads();
    for (int d = tid; d < HD; d += 256) out[2 + d] = opart[0][d] + opart[1][d] + opart[2][d] + opart[3][d];
    if (tid == 0) { out[0] = mx; out[1] = sum; }
}

template <int HD>
__global__ void attn_decode_combine(const float* ws, bf16* o, int64_t o_bs, int Hq, int nsplit) {
    const int hq = blockIdx.x, b = blockIdx.y;
    const float* base = ws + (size_t)(b * Hq + hq) * nsplit * (HD + 2);
    float m = -INFINITY;
    for (int s = 0; s < nsplit; ++s) m = fmaxf(m, base[s * (HD + 2)]);
    for (int d = threadIdx.x; d < HD; d += blockDim.x) {
        float num = 0.f, den = 0.f;
        for (int s = 0; s < nsplit; ++s) {
            const float* p = base + s * (HD + 2);
            const float w = (p[0] == -INFINITY) ? 0.f : __expf(p[0] - m);
            num += w * p[2 + d];
            den += w * p[1];
        }
        o[b * o_bs + hq * HD + d] = (bf16)(num / den);
    }
}

// ---- generic: one block per (q row, head, batch); scores in LDS -----------------------------------
__global__ __launch_bounds__(256) void attn_generic(AttnArgs a, int causal) {
    extern __shared__ __attribute__((aligned(16))) char smem_g[];
    float* qv = reinterpret_cast<float*>(smem_g);            // [hd]
    float* red = qv + a.hd;                                  // [16]
    float* part = red + 16;                                  // [key groups][hd]: max(4 * hd, 256) floats
    float* sc = part + max(4 * a.hd, 256);                   // [nk]
    const int qi = blockIdx.x, hq = blockIdx.y, b = blockIdx.z, hkv = hq / (a.Hq / a.Hkv);
    const int tid = threadIdx.x;
    const int nk = causal ? min(a.Sk, a.q_pos0 + qi + 1) : a.Sk;
    const bf16* qp = a.q + b * a.q_bs + (int64_t)qi * a.q_ss + hq * a.hd;
    for (int d = tid; d < a.hd; d += 256) qv[d] = (float)qp[d];
    __syncthreads();
    const bf16* kb = a.k + b * a.k_bs + hkv * a.hd;
    const bf16* vb = a.v + b * a.v_bs + hkv * a.hd;
    float mx = -INFINITY;
    for (int j = tid; j < nk; j += 256) {
        const bf16* kr = kb + (int64_t)j * a.k_ss;
        float s = 0.f;
        for (int d = 0; d < a.hd; ++d) s += qv[d] * (float)kr[d];
        s *= a.scale;
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < nk; j += 256) {
        const float p = __expf(sc[j] - mx);
        sc[j] = p;
        sum += p;
    }
    sum = block_sum(sum, red);
    __syncthreads();
    // output: G = 256 / hd key groups (at least 4 x 64 lanes for hd >= 64), thread (g, d) sums keys g, g+G, ...
    if (a.hd <= 64) {
        const int G = 256 / a.hd, g = tid / a.hd, d = tid - g * a.hd;
        if (g < G) {
            float acc = 0.f;
            for (int j = g; j < nk; j += G) acc += sc[j] * (float)vb[(int64_t)j * a.v_ss + d];
            part[g * a.hd + d] = acc;
        }
        __syncthreads();
        bf16* op = a.o + b * a.o_bs + (int64_t)qi * a.o_ss + hq * a.hd;
        if (tid < a.hd) {
            float t = 0.f;
            for (int gg = 0; gg < G; ++gg) t += part[gg * a.hd + tid];
            op[tid] = (bf16)(t / sum);
        }
        return;
    }
    const int g = tid >> 6, ln = tid & 63;
    for (int d = ln; d < a.hd; d += 64) {
        float acc = 0.f;
        for (int j = g; j < nk; j += 4) acc += sc[j] * (float)vb[(int64_t)j * a.v_ss + d];
        part[g * a.hd + d] = acc;
    }
    __syncthreads();
    bf16* op = a.o + b * a.o_bs + (int64_t)qi * a.o_ss + hq * a.hd;
    for (int d = tid; d < a.hd; d += 256)
        op[d] = (bf16)((part[d] + part[a.hd + d] + part[2 * a.hd + d] + part[3 * a.hd + d]) / sum);
}

// ---- few keys (Sk <= 64, hd 16/32): one thread per (query row, head); K/V of the batch element in LDS as fp32 -------
// SAM2 image->token cross attention: 4096 image queries against 9 prompt tokens (sam2.py:1405-1412).
template <int HD>
__global__ __launch_bounds__(256) void attn_fewkeys(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_f[];
    float* ks = reinterpret_cast<float*>(smem_f);               // [Sk][Hkv*HD]
    const int b = blockIdx.y, C = a.Hkv * HD;
    float* vs = ks + a.Sk * C;
    for (int i = threadIdx.x; i < a.Sk * C; i += 256) {
        const int j = i / C, c = i - j * C;
        ks[i] = (float)a.k[b * a.k_bs + (int64_t)j * a.k_ss + c];
        vs[i] = (float)a.v[b * a.v_bs + (int64_t)j * a.v_ss + c];
    }
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= a.Sq * a.Hq) return;
    const int qi = t / a.Hq, hq = t - qi * a.Hq, hkv = hq / (a.Hq / a.Hkv);
    const bf16* qp = a.q + b * a.q_bs + (int64_t)qi * a.q_ss + hq * HD;
    float q[HD], acc[HD];
#pragma unroll
    for (int d = 0; d < HD; d += 8) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(qp + d);
#pragma unroll
        for (int j = 0; j < 8; ++j) { q[d + j] = (float)v[j] * a.scale; acc[d + j] = 0.f; }
    }
    float mx = -INFINITY, sum = 0.f;
    for (int j = 0; j < a.Sk; ++j) {
        const float* kr = ks + j * C + hkv * HD;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s += q[d] * kr[d];
        const float mn = fmaxf(mx, s), corr = __expf(mx - mn), p = __expf(s - mn);
        const float* vr = vs + j * C + hkv * HD;
        sum = sum * corr + p;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = acc[d] * corr + p * vr[d];
        mx = mn;
    }
    const float inv = 1.f / sum;
    bf16* op = a.o + b * a.o_bs + (int64_t)qi * a.o_ss + hq * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4)
        *reinterpret_cast<bf16x4*>(op + d) = (bf16x4){(bf16)(acc[d] * inv), (bf16)(acc[d + 1] * inv), (bf16)(acc[d + 2] * inv), (bf16)(acc[d + 3] * inv)};
}

// ---- small windows (Sq <= 16, Sk <= 16, head_dim 72, non-causal): Hiera's 4 x 4 windows and their q-pooled form.  One WAVE per (window, head): the 64-key tiles of
// the flash kernels spend 16 x 64 MFMA slots on 16 x 16 (or 4 x 16) problems and the launch becomes thousands of nearly empty blocks (350 us per call at 8192 windows x 4
// heads; 1 ms in the generic kernel for the pooled form).  Here Q, K and V^T of the pair sit in 7 KB of LDS; lane (q = l & 15, g = l >> 4) computes the scores of keys
// g, g + 4, g + 8, g + 12 with v_dot2c_f32_bf16, the softmax runs over the 4 lanes of a query, P (bf16, as in the flash kernels) goes back through LDS, and the lane
// produces head dims 18 g .. 18 g + 17 of its query.
__global__ __launch_bounds__(256) void attn_win16_k(AttnArgs a) {
    constexpr int HD = 72, CH = HD / 8;                              // 9 chunks of 8 per row
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    __shared__ __attribute__((aligned(16))) bf16 lq[4][16 * HD], lk[4][16 * HD], lvt[4][HD * 16], lp[4][16 * 16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= a.B * a.Hq) return;                                  // (no block-wide barrier below: a wave only touches its own LDS slices)
    const int b = pair / a.Hq, hq = pair % a.Hq, hkv = hq / (a.Hq / a.Hkv);
    const bf16* qp = a.q + b * a.q_bs + hq * HD;
    const bf16* kp = a.k + b * a.k_bs + hkv * HD;
    const bf16* vp = a.v + b * a.v_bs + hkv * HD;
    bf16 *q_ = lq[wave], *k_ = lk[wave], *vt = lvt[wave], *p_ = lp[wave];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = lane + 64 * i;
        if (c < 16 * CH) {
            const int r = c / CH, cc = c % CH;
            *reinterpret_cast<bf16x8*>(q_ + r * HD + cc * 8) = *reinterpret_cast<const bf16x8*>(qp + (int64_t)min(r, a.Sq - 1) * a.q_ss + cc * 8);
            *reinterpret_cast<bf16x8*>(k_ + r * HD + cc * 8) = *reinterpret_cast<const bf16x8*>(kp + (int64_t)min(r, a.Sk - 1) * a.k_ss + cc * 8);
            const bf16x8 vv = *reinterpret_cast<const bf16x8*>(vp + (int64_t)min(r, a.Sk - 1) * a.v_ss + cc * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) vt[(cc * 8 + j) * 16 + r] = r < a.Sk ? vv[j] : (bf16)0.f;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int q = lane & 15, g = lane >> 4;
    float sc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const bf16x8 qv = *reinterpret_cast<const bf16x8*>(q_ + q * HD + cc * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16x8 kv = *reinterpret_cast<const bf16x8*>(k_ + (g + 4 * j) * HD + cc * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                sc[j] = __builtin_amdgcn_fdot2_f32_bf16(bf2{qv[2 * e], qv[2 * e + 1]}, bf2{kv[2 * e], kv[2 * e + 1]}, sc[j], false);
        }
    }
    const float sl2 = a.scale * 1.4426950408889634f;
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sc[j] = (g + 4 * j < a.Sk) ? sc[j] * sl2 : -INFINITY;
        m = fmaxf(m, sc[j]);
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float p = __builtin_amdgcn_exp2f(sc[j] - m);
        l += p;
        p_[q * 16 + g + 4 * j] = (bf16)p;
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bf16x8 p0 = *reinterpret_cast<const bf16x8*>(p_ + q * 16), p1 = *reinterpret_cast<const bf16x8*>(p_ + q * 16 + 8);
    const float inv = 1.0f / l;
    if (q < a.Sq) {
        bf16* op = a.o + b * a.o_bs + (int64_t)q * a.o_ss + hq * HD + 18 * g;
#pragma unroll
        for (int dd = 0; dd < 18; dd += 2) {
            float o2[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16* vr = vt + (18 * g + dd + t) * 16;
                const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(vr), v1 = *reinterpret_cast<const bf16x8*>(vr + 8);
                float acc = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_fdot2_f32_bf16(bf2{p0[2 * e], p0[2 * e + 1]}, bf2{v0[2 * e], v0[2 * e + 1]}, acc, false);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_fdot2_f32_bf16(bf2{p1[2 * e], p1[2 * e + 1]}, bf2{v1[2 * e], v1[2 * e + 1]}, acc, false);
                o2[t] = acc * inv;
            }
            *reinterpret_cast<bf2*>(op + dd) = bf2{(bf16)o2[0], (bf16)o2[1]};
        }
    }
}

static bool win16_ok(const AttnArgs& a, bool aligned) {
    return aligned && a.hd == 72 && a.Sq <= 16 && a.Sk <= 16 && a.Sq >= 1 && a.Sk >= 1 && a.Hq % a.Hkv == 0;
}

static int launch_win16(const AttnArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(attn_win16_k, dim3(cdiv(a.B * a.Hq, 4)), dim3(256), 0, st, a);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

// causal head_dim 128 with a key split over two wave groups per block (see attn_fwd_mfma NG)
template <int NW>
int launch_mfma_split2(const AttnArgs& a, hipStream_t st) {
    constexpr int smem = 4 * Cfg<128>::STAGE;
    static_assert((size_t)NW * 64 * (Cfg<128>::DT * 16 + 2) * 4 <= (size_t)smem, "merge buffer must fit in the staging area");
    UFV_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_mfma<128, NW, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    );
    dim3 grid(cdiv(a.Sq, 32 * NW), a.Hq, a.B);
    hipLaunchKernelGGL((attn_fwd_mfma<128, NW, true, 2>), grid, dim3(NW * 128), smem, st, a);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

static inline bool split2_pays(int Sq, int Hq, int B, int q_pos0) {
    return Sq + q_pos0 >= 512 && (int64_t)cdiv(Sq, 128) * Hq * B <= 600;
}

template <int HD, int NW>
int launch_mfma(const AttnArgs& a, int causal, hipStream_t st) {
    constexpr int smem = 2 * Cfg<HD>::STAGE;
    UFV_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_mfma<HD, NW, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_mfma<HD, NW, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    );
    dim3 grid(cdiv(a.Sq, 32 * NW), a.Hq, a.B);
    if (causal)
        hipLaunchKernelGGL((attn_fwd_mfma<HD, NW, true>), grid, dim3(NW * 64), smem, st, a);
    else
        hipLaunchKernelGGL((attn_fwd_mfma<HD, NW, false>), grid, dim3(NW * 64), smem, st, a);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

template <int HD, int NW, bool PP>
int launch_mfma_dma(const AttnArgs& a, int causal, hipStream_t st) {
    constexpr int smem = (PP ? 4 : 3) * (2 * 64 * HD * 2 + 256) + NW * 32 * (HD * 2 + 16);      // K/V ring + per-wave Q / O staging
    UFV_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_mfma_dma<HD, NW, true, PP>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_mfma_dma<HD, NW, false, PP>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    );
    const int nqt = cdiv(a.Sq, 32 * NW), groups = a.Hq * a.B, gper = (groups + 7) / 8;
    dim3 grid(8 * gper * nqt);
    if (causal) hipLaunchKernelGGL((attn_fwd_mfma_dma<HD, NW, true, PP>), grid, dim3(NW * 64), smem, st, a);
    else hipLaunchKernelGGL((attn_fwd_mfma_dma<HD, NW, false, PP>), grid, dim3(NW * 64), smem, st, a);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

template <int HD, int NW, bool PP>
int launch_pair(const AttnArgs& a, hipStream_t st) {
    constexpr int smem = 2 * (PP ? 4 : 3) * (2 * 64 * HD * 2 + 256);
    UFV_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_pair<HD, NW, PP>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    );
    const int groups = (a.Hq / 2) * a.B, gper = (groups + 7) / 8;
    hipLaunchKernelGGL((attn_fwd_pair<HD, NW, PP>), dim3(8 * gper), dim3(2 * NW * 64), smem, st, a);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

}  // namespace

extern "C" int ufv_attention(const void* q, int64_t q_bs, int64_t q_ss, const void* k, int64_t k_bs, int64_t k_ss,
                             const void* v, int64_t v_bs, int64_t v_ss, void* o, int64_t o_bs, int64_t o_ss, int B, int Hq,
                             int Hkv, int Sq, int Sk, int hd, float scale, int causal, int q_pos0, int kernel, void* stream) {
    UFV_REQUIRE(q && k && v && o && B > 0 && Hq > 0 && Hkv > 0 && Sq > 0 && Sk > 0 && hd > 0, "ufv_attention: bad arguments");
    UFV_REQUIRE(Hq % Hkv == 0, "ufv_attention: Hq (%d) must be a multiple of Hkv (%d)", Hq, Hkv);
    AttnArgs a;
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (bf16*)o;
    a.q_bs = q_bs; a.q_ss = q_ss; a.k_bs = k_bs; a.k_ss = k_ss; a.v_bs = v_bs; a.v_ss = v_ss; a.o_bs = o_bs; a.o_ss = o_ss;
    a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.Sq = Sq; a.Sk = Sk; a.hd = hd; a.scale = scale; a.q_pos0 = q_pos0; a.lse = nullptr;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool aligned = ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) &&
                         ((uintptr_t)o % 8 == 0) && (q_ss % 8 == 0) && (k_ss % 8 == 0) && (v_ss % 8 == 0) && (o_ss % 4 == 0) &&
                         (q_bs % 8 == 0) && (k_bs % 8 == 0) && (v_bs % 8 == 0) && (o_bs % 4 == 0);
    const bool hd_ok = (hd == 64 || hd == 72 || hd == 80 || hd == 96 || hd == 128);
    const bool mfma_ok = aligned && hd_ok;
    if (kernel == 16 || (kernel == 0 && !causal && win16_ok(a, aligned) && (int64_t)B * Hq >= 256)) {
        if (causal || !win16_ok(a, aligned)) { ufv_set_error("ufv_attention: kernel 16 is built for non-causal head_dim 72 with Sq, Sk <= 16 and aligned rows"); return UFV_EUNSUPPORTED; }
        return launch_win16(a, st);
    }
    if ((kernel == 14 && hd != 72) || (kernel == 15 && hd != 128)) {      // the diagnostic ids of the two generated kernels name ONE head_dim each
        ufv_set_error("ufv_attention: kernel %d is built for head_dim %d (hd=%d)", kernel, kernel == 14 ? 72 : 128, hd);
        return UFV_EUNSUPPORTED;
    }
#ifndef UFV_LAB_KERNELS
    // ids 3, 4, 6, 7, 8, 10 name first-generation forms that no AUTO choice can reach (register-staged, lockstep, head-pair and 6-wave ping-pong ViT
    // kernels: measured, slower, LABNOTES.md): they are compiled into lab builds only (tools/lab/build_variant_lib.sh -DUFV_LAB_KERNELS)
    if (kernel == 3 || kernel == 4 || kernel == 6 || kernel == 7 || kernel == 8 || kernel == 10) {
        ufv_set_error("ufv_attention: kernel %d is a lab-only diagnostic form (library built without UFV_LAB_KERNELS)", kernel);
        return UFV_EUNSUPPORTED;
    }
#endif
    if ((kernel == 1 || kernel == 3) && !mfma_ok) {
        ufv_set_error("ufv_attention: MFMA kernel needs hd in {64,72,80,96,128} and 16-byte aligned rows (hd=%d)", hd);
        return UFV_EUNSUPPORTED;
    }
    if (kernel == 1 || kernel == 3 || kernel == 4 || kernel == 6 || kernel == 7 || kernel == 8 || kernel == 9 || kernel == 10 || kernel == 11 || kernel == 14 || kernel == 15 || ((kernel == 12 || kernel == 13) && mfma_ok) ||
        (kernel == 0 && mfma_ok && Sq >= 16)) {
        const bool six = (Sq % 192 == 0) && (Sq % 128 != 0);     // e.g. 576 ViT tokens: 3 blocks of 6 waves, no idle wave
        switch (hd) {
            case 64: return launch_mfma<64, 4>(a, causal, st);
            case 72:
#ifdef UFV_LAB_KERNELS
                     if (kernel == 3) return launch_mfma<72, 4>(a, causal, st);      // register-staged variant (diagnostic)
#endif
                     if (kernel == 14) {
                         if (causal || !vit72_p2_ok(a)) { ufv_set_error("ufv_attention: kernel 14 is built for non-causal hd 72, S = 576 or 729, an even head count and equal q / k / v row pitches"); return UFV_EUNSUPPORTED; }
                         return launch_vit72_p2(a, st);
                     }
#ifdef UFV_LAB_KERNELS
                     // head-pair variants (diagnostic: measured 10-25 % slower than the 9-wave blocks)
                     if (!causal && Sq % 192 == 0 && Hq % 2 == 0 && kernel == 7) return launch_pair<72, 6, false>(a, st);
                     if (!causal && Sq % 192 == 0 && Hq % 2 == 0 && kernel == 8) return launch_pair<72, 6, true>(a, st);
#endif
                     // third generation (attn_vit_p2.inc): SigLIP at 336 px (S = 576) and 384 px (S = 729); bit-identical to the second generation, which stays for other lengths
                     if (!causal && vit72_p2_ok(a) && (kernel == 0 || kernel == 1)) return launch_vit72_p2(a, st);
                     if (!causal && Sq % 288 == 0 && (int64_t)Sk * (k_ss > v_ss ? k_ss : v_ss) * 2 < (1ll << 31) && (kernel == 0 || kernel == 1 || kernel == 11))
                         return launch_vit72<9>(a, st);   // second-generation ViT kernel (attn_vit.inc)
                     // the same kernel in 6-wave blocks (192 rows; ragged query and key tails are its own: clamped rows, -inf scores) for the lengths above the 336-px
                     // tower's that are no multiple of 288 -- SigLIP at 384 px: 729 tokens when the generated kernel cannot take the call -- and for id 11 at any length:
                     // one numerics family (oracle `_flash_vit72_mirror`) for every long ViT sequence
                     if (!causal && (int64_t)Sk * (k_ss > v_ss ? k_ss : v_ss) * 2 < (1ll << 31) && (kernel == 11 || ((kernel == 0 || kernel == 1) && Sq > 576)))
                         return launch_vit72<6>(a, st);
#ifdef UFV_LAB_KERNELS
                     if (Sq % 288 == 0 && kernel == 6) return launch_mfma_dma<72, 9, false>(a, causal, st);   // lockstep variant (diagnostic)
                     if (six && kernel == 10) return launch_mfma_dma<72, 6, true>(a, causal, st);               // 6-wave blocks, ping-pong, 2 blocks / CU (diagnostic)
#endif
                     if (Sq % 288 == 0 && kernel != 4) return launch_mfma_dma<72, 9, true>(a, causal, st);     // 2 blocks of 9 waves (kernel 9: the former default; AUTO's for causal hd 72)
                     return six ? launch_mfma_dma<72, 6, false>(a, causal, st) : launch_mfma_dma<72, 4, false>(a, causal, st);
            case 80: return launch_mfma<80, 4>(a, causal, st);
            case 96: return launch_mfma<96, 4>(a, causal, st);
            case 128:
                if (kernel == 15) {
                    if (!(causal && mfma_ok && c128_ok(a))) { ufv_set_error("ufv_attention: kernel 15 is built for causal hd 128 prefill of one sequence (Sq = Sk >= 64, GQA with >= 2 kv heads)"); return UFV_EUNSUPPORTED; }
                    return launch_c128(a, st);
                }
                // kernel 12: key split over two wave groups per block; kernel 13: never.  Default when the launch has few blocks for the chip
                // (<= ~2.3 per CU: the causal S = 2399 prefill has 532 for 512 slots and ends with its heaviest blocks alone; measured 88 -> 82 us
                // there, 41 -> 31 us at S = 1200; with many blocks -- S = 4703: 1036 -- the plain kernel's two blocks per CU retire more tiles: 232 vs 254 us)
                // default for the prefill of one sequence: the 4 + 4-wave kernel of attn_c128.inc (bit-identical to the plain kernel below)
                if (causal && kernel == 0 && c128_ok(a) && Sq >= C128_MIN_S) return launch_c128(a, st);
                if (causal && kernel != 13 && (kernel == 12 || (kernel == 0 && split2_pays(Sq, Hq, B, q_pos0)))) return launch_mfma_split2<4>(a, st);
                return launch_mfma<128, 4>(a, causal, st);
        }
    }
    const bool few_ok = aligned && !causal && Sk <= 64 && (hd == 16 || hd == 32) && (size_t)Sk * Hkv * hd * 8 <= 64 * 1024;
    if (kernel == 5 && !few_ok) {
        ufv_set_error("ufv_attention: few-keys kernel needs Sk<=64, hd in {16,32}, non-causal, aligned rows (Sk=%d hd=%d)", Sk, hd);
        return UFV_EUNSUPPORTED;
    }
    if (kernel == 5 || (kernel == 0 && few_ok && Sq >= 64)) {
        const size_t sm = (size_t)Sk * Hkv * hd * 8;
        dim3 grid(cdiv(Sq * Hq, 256), B);
        if (hd == 16) hipLaunchKernelGGL((attn_fewkeys<16>), grid, dim3(256), sm, st, a);
        else hipLaunchKernelGGL((attn_fewkeys<32>), grid, dim3(256), sm, st, a);
        UFV_CHECK_LAUNCH();
        return UFV_OK;
    }
    const size_t smem = sizeof(float) * ((size_t)hd + 16 + (4 * hd > 256 ? 4 * hd : 256) + Sk);
    UFV_REQUIRE(smem <= 64 * 1024, "ufv_attention: generic kernel supports Sk <= ~16000 (Sk=%d hd=%d)", Sk, hd);
    hipLaunchKernelGGL(attn_generic, dim3(Sq, Hq, B), dim3(256), smem, st, a, causal);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

// causal self-attention forward that also returns the log2-domain log-sum-exp per (head, query): the training forward of the
// fused backward (ufv_attention_bwd_fused).  Same kernel as ufv_attention's hd = 128 path.
extern "C" int ufv_attention_causal_lse(const void* q, int64_t q_ss, const void* k, int64_t k_ss, const void* v, int64_t v_ss, void* o,
                                        int64_t o_ss, int Hq, int Hkv, int S, int hd, float scale, float* lse, void* stream) {
    UFV_REQUIRE(q && k && v && o && lse && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && S > 0, "ufv_attention_causal_lse: bad arguments");
    UFV_REQUIRE(hd == 128 && ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)o % 8 == 0) &&
                q_ss % 8 == 0 && k_ss % 8 == 0 && v_ss % 8 == 0 && o_ss % 4 == 0, "ufv_attention_causal_lse: needs head_dim 128 and 16-byte aligned rows");
    AttnArgs a;
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (bf16*)o;
    a.q_bs = 0; a.q_ss = q_ss; a.k_bs = 0; a.k_ss = k_ss; a.v_bs = 0; a.v_ss = v_ss; a.o_bs = 0; a.o_ss = o_ss;
    a.B = 1; a.Hq = Hq; a.Hkv = Hkv; a.Sq = S; a.Sk = S; a.hd = hd; a.scale = scale; a.q_pos0 = 0; a.lse = lse;
    // the same kernel choice (and bits) as ufv_attention's causal hd-128 path; attn_c128.inc writes the log-sum-exp from its epilogue
    if (c128_shape_ok(a) && S >= C128_MIN_S) return launch_c128(a, reinterpret_cast<hipStream_t>(stream));
    if (split2_pays(S, Hq, 1, 0)) return launch_mfma_split2<4>(a, reinterpret_cast<hipStream_t>(stream));
    return launch_mfma<128, 4>(a, 1, reinterpret_cast<hipStream_t>(stream));
}

template <int HD>
static int launch_decode(const AttnArgs& a, float* ws, int nsplit, const int* pos_dev, int max_keys, hipStream_t st) {
    const int per = cdiv(pos_dev ? max_keys : a.Sk, nsplit);          // score buffer: sized for the longest split this launch can see
    hipLaunchKernelGGL((attn_decode_split<HD>), dim3(nsplit, a.Hq, a.B), dim3(256), per * sizeof(float), st, a, ws, nsplit, pos_dev);
    hipLaunchKernelGGL((attn_decode_combine<HD>), dim3(a.Hq, a.B), dim3(HD < 64 ? 64 : HD), 0, st, ws, a.o, a.o_bs, a.Hq, nsplit);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

// ---------------------------------------------------------------------------------------------------------
// One-launch decode attention for one new token (the three kernels RoPE + KV append / split attention / combine of the decode step in one):
// grid (nsplit, Hq).  Every block rotates its head's q itself (the bf16 value rope_kv1 would have stored); the blocks of the last split
// take the new token's K (rotated) and V from the qkv row instead of the cache, and one of them per kv head appends that row to the cache for
// the tokens to come; the partial (m, l, o) go to `ws` with agent-scope accesses, and the block that arrives last at the head's counter
// merges them (same arithmetic as attn_decode_combine).  The counters (Hq ints behind the partials) must be zero before the first launch;
// they return to zero on their own.  Bit-identical to the three-kernel sequence.
// ---------------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void attn_decode_fused(const bf16* __restrict__ qkv, int Hq, int Hkv, const float* __restrict__ inv_freq, int pos_host,
                                                          const int* __restrict__ pos_dev, bf16* kv, int ldkv, bf16* o, float scale, float* ws, int nsplit) {
    constexpr int CPR = HD / 8, KPI = 64 / CPR, HALF = HD / 2;
    // The launch is a chain of memory round trips, not a stream (5 MB of cache per layer): every key / value row a lane will use is REQUESTED before the
    // first one is consumed -- NIT rows of K and NIT rows of V per lane and chunk of CHUNK keys (one chunk covers 256 keys per split: 4096 positions at
    // 16 splits) -- and the RoPE trigonometry runs underneath.  (With one load per loop iteration a split of 150 keys was 2 x 10 dependent HBM round trips:
    // 24 us per launch, 0.68 ms per token.)  The arithmetic and its order are those of the three-kernel sequence.
    constexpr int NIT = 64 / KPI, CHUNK = NIT * 4 * KPI;
    extern __shared__ __attribute__((aligned(16))) char smem_d[];
    float* sc = reinterpret_cast<float*>(smem_d);
    __shared__ float red[16];
    __shared__ float opart[4][HD];
    __shared__ int last_flag;
    const int split = blockIdx.x, hq = blockIdx.y, hkv = hq / (Hq / Hkv);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane / CPR, ch = lane % CPR;
    const int pos = pos_dev ? *pos_dev : pos_host;
    const int nk = pos + 1;
    const int per = (nk + nsplit - 1) / nsplit;
    const int k0 = split * per, k1 = min(nk, k0 + per), n = max(k1 - k0, 0);
    float* out = ws + ((size_t)hq * nsplit + split) * (HD + 2);
    int* counter = reinterpret_cast<int*>(ws + (size_t)Hq * nsplit * (HD + 2)) + hq;
    const bf16* kb = kv + hkv * HD + ch * 8;
    const bf16* vb = kv + (Hkv + hkv) * HD + ch * 8;
    bf16x8 kreg[NIT], vreg[NIT];
    // rows past the split's end are clamped to row `pos` (inside the cache; it may be the row another block is appending right now: such a value is never used)
    auto request = [&](bf16x8 (&r)[NIT], const bf16* base, int c0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = min(k0 + c0 + (it * 4 + wave) * KPI + sub, pos);
            r[it] = *reinterpret_cast<const bf16x8*>(base + (int64_t)row * ldkv);
        }
    };
    // the small operands first (memory returns in order: they arrive ahead of the cache rows and the trigonometry starts on them): the 8 frequencies of this
    // lane's chunk as two vector loads (one scalar load + wait per frequency was 8 dependent L2 round trips per rotated row), q's and the new k's halves, v
    const bool lo = ch * 8 < HALF;
    const int partner = lo ? ch * 8 + HALF : ch * 8 - HALF;
    const bool has_new = k1 == nk && n > 0;                    // this split ends with the new token
    const f32x4 fr0 = *reinterpret_cast<const f32x4*>(inv_freq + (ch * 8) % HALF), fr1 = *reinterpret_cast<const f32x4*>(inv_freq + (ch * 8) % HALF + 4);
    const bf16* qrow = qkv + hq * HD;
    const bf16* krow = qkv + (Hq + hkv) * HD;
    // unconditional (a split without keys, or without the new token, reads valid rows it does not use): a branch here makes the compiler wait for ALL of it at the join
    const bf16x8 q_mine = *reinterpret_cast<const bf16x8*>(qrow + ch * 8), q_other = *reinterpret_cast<const bf16x8*>(qrow + partner);
    const bf16x8 k_mine = *reinterpret_cast<const bf16x8*>(krow + ch * 8), k_other = *reinterpret_cast<const bf16x8*>(krow + partner);
    const bf16x8 vnew = *reinterpret_cast<const bf16x8*>(qkv + (Hq + Hkv + hkv) * HD + ch * 8);
    bf16x8 knew = {0, 0, 0, 0, 0, 0, 0, 0};
    asm volatile("" ::: "memory");                             // keep this order: the scheduler would otherwise put the 32 row requests first
    request(kreg, kb, 0);
    request(vreg, vb, 0);
    asm volatile("" ::: "memory");
    // rotate-half RoPE of one 8-element chunk of a head row (dims 8ch .. 8ch+7; the partner dims are HALF away), rounded to bf16 as the cache holds it
    auto rope_chunk = [&](const bf16x8 mine, const bf16x8 other) -> bf16x8 {
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float ang = (float)pos * (j < 4 ? fr0[j & 3] : fr1[j & 3]);
            const float c = cosf(ang), sn = sinf(ang);
            const float x1 = lo ? (float)mine[j] : (float)other[j], x2 = lo ? (float)other[j] : (float)mine[j];
            float r1, r2;
            rope_pair(x1, x2, c, sn, r1, r2);
            r[j] = lo ? (bf16)r1 : (bf16)r2;
        }
        return r;
    };
    if (has_new) {
        knew = rope_chunk(k_mine, k_other);
        if (hq % (Hq / Hkv) == 0 && wave == 0 && sub == 0) {      // one block per kv head appends the row for later tokens
            *reinterpret_cast<bf16x8*>(kv + (int64_t)pos * ldkv + hkv * HD + ch * 8) = knew;
            *reinterpret_cast<bf16x8*>(kv + (int64_t)pos * ldkv + (Hkv + hkv) * HD + ch * 8) = vnew;
        }
    }
    float mx = -INFINITY, sum = 0.f;
    if (n > 0) {
        const bf16x8 qv = rope_chunk(q_mine, q_other);
        float qf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[j] = (float)qv[j] * scale;
        for (int c0 = 0; c0 < n; c0 += CHUNK) {
            if (c0 > 0) request(kreg, kb, c0);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int j = c0 + (it * 4 + wave) * KPI + sub;
                if (c0 + (it * 4 + wave) * KPI < n) {                // wave-uniform: the iterations the one-load-per-iteration loop ran
                    float s = 0.f;
                    if (j < n) {
                        const bf16x8 kvv = (k0 + j == pos) ? knew : kreg[it];
#pragma unroll
                        for (int e = 0; e < 8; ++e) s += qf[e] * (float)kvv[e];
                    }
#pragma unroll
                    for (int oo = 1; oo < CPR; oo <<= 1) s += __shfl_xor(s, oo, 64);
                    if (j < n && ch == 0) sc[j] = s;
                    if (j < n) mx = fmaxf(mx, s);
                }
            }
        }
        mx = wave_max(mx);
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        for (int j = tid; j < n; j += 256) {
            const float pj = __expf(sc[j] - mx);
            sc[j] = pj;
            sum += pj;
        }
        sum = block_sum(sum, red + 4);
        __syncthreads();
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c0 = 0; c0 < n; c0 += CHUNK) {
            if (c0 > 0) request(vreg, vb, c0);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int j = c0 + (it * 4 + wave) * KPI + sub;
                if (j < n) {
                    const float pj = sc[j];
                    const bf16x8 vv = (k0 + j == pos) ? vnew : vreg[it];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += pj * (float)vv[e];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int oo = CPR; oo < 64; oo <<= 1) acc[e] += __shfl_xor(acc[e], oo, 64);
        if (sub == 0)
#pragma unroll
            for (int e = 0; e < 8; ++e) opart[wave][ch * 8 + e] = acc[e];
        __syncthreads();
    }
    // ---- publish the partial (agent scope: the merging block may sit on another XCD), then count in
    for (int d = tid; d < HD; d += 256)
        __hip_atomic_store(out + 2 + d, n > 0 ? opart[0][d] + opart[1][d] + opart[2][d] + opart[3][d] : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
        __hip_atomic_store(out, n > 0 ? mx : -INFINITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(out + 1, n > 0 ? sum : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the partials went out as write-through stores: acknowledged = visible device-wide, so a plain wait + barrier orders them before the count
    // (an agent-scope fence here writes back / invalidates the whole L2 of the XCD, 448 times per layer: measured 5.1 against 4.1 ms per token)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) last_flag = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsplit - 1;
    __syncthreads();
    if (!last_flag) return;
    if (tid == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next token
    // The merge: ONE round trip.  System-scope loads; thread d requests its dimension's partial of up to 16 splits and threads request the (m, l) pairs in the
    // same breath, one wait for all of it (the registers pass through the waiting asm so that nothing reads them early).  More than 16 splits: further batches.
    const float* base = ws + (size_t)hq * nsplit * (HD + 2);
    float* ml = sc;                                            // [nsplit][2] (m, l) shared by the block; the score buffer is free now
    const int d = tid < HD ? tid : HD - 1;
    float v16[16];
    {
        const float* q[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) q[j] = base + 2 + d + (size_t)min(j, nsplit - 1) * (HD + 2);
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=&v"(v16[j]) : "v"(q[j]) : "memory");
    }
    __syncthreads();                                           // every thread is done with sc[] as scores
    for (int i = tid; i < 2 * nsplit; i += 256) {
        float vv;
        const float* pp = base + (i >> 1) * (HD + 2) + (i & 1);
        asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(vv) : "v"(pp) : "memory");
        ml[i] = vv;
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v16[0]), "+v"(v16[1]), "+v"(v16[2]), "+v"(v16[3]), "+v"(v16[4]), "+v"(v16[5]), "+v"(v16[6]), "+v"(v16[7]),
                 "+v"(v16[8]), "+v"(v16[9]), "+v"(v16[10]), "+v"(v16[11]), "+v"(v16[12]), "+v"(v16[13]), "+v"(v16[14]), "+v"(v16[15]) :: "memory");
    __syncthreads();
    float m = -INFINITY;
    for (int s2 = 0; s2 < nsplit; ++s2) m = fmaxf(m, ml[2 * s2]);
    float num = 0.f, den = 0.f;
    for (int s0 = 0; s0 < nsplit; s0 += 16) {
        if (s0 > 0) {
            const float* q[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) q[j] = base + 2 + d + (size_t)min(s0 + j, nsplit - 1) * (HD + 2);
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=&v"(v16[j]) : "v"(q[j]) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v16[0]), "+v"(v16[1]), "+v"(v16[2]), "+v"(v16[3]), "+v"(v16[4]), "+v"(v16[5]), "+v"(v16[6]), "+v"(v16[7]),
                         "+v"(v16[8]), "+v"(v16[9]), "+v"(v16[10]), "+v"(v16[11]), "+v"(v16[12]), "+v"(v16[13]), "+v"(v16[14]), "+v"(v16[15]) :: "memory");
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (s0 + j < nsplit) {
                const float pm = ml[2 * (s0 + j)];
                const float w = (pm == -INFINITY) ? 0.f : __expf(pm - m);
                num += w * v16[j];
                den += w * ml[2 * (s0 + j) + 1];
            }
    }
    if (tid < HD) o[hq * HD + tid] = (bf16)(num / den);
}

extern "C" int64_t ufv_attention_decode_fused_ws_bytes(int Hq, int hd, int nsplit) {
    return (int64_t)(sizeof(float) * (size_t)Hq * nsplit * (hd + 2) + sizeof(int) * (size_t)Hq);
}

// pos_dev == NULL: position `pos` from the host.  ws: ufv_attention_decode_fused_ws_bytes(...) bytes whose trailing Hq ints are ZERO before the first call.
extern "C" int ufv_attention_decode_fused(const void* qkv, int Hq, int Hkv, int hd, const float* inv_freq, int pos, const int* pos_dev, void* kv_cache,
                                          int ldkv, int max_keys, void* o, float scale, void* ws, int nsplit, void* stream) {
    UFV_REQUIRE(qkv && inv_freq && kv_cache && o && ws && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && nsplit > 0, "ufv_attention_decode_fused: bad arguments");
    UFV_REQUIRE(hd == 64 || hd == 128, "ufv_attention_decode_fused: head_dim %d (built for 64 and 128)", hd);
    UFV_REQUIRE(((uintptr_t)qkv % 16 == 0) && ((uintptr_t)kv_cache % 16 == 0) && ldkv % 8 == 0, "ufv_attention_decode_fused: rows must be 16-byte aligned");
    const int keys = pos_dev ? max_keys : pos + 1;
    UFV_REQUIRE(keys > 0 && (pos_dev || pos < max_keys), "ufv_attention_decode_fused: position %d outside the cache (%d rows)", pos, max_keys);
    UFV_REQUIRE(cdiv(keys, nsplit) * sizeof(float) <= 48 * 1024 && nsplit <= 4096, "ufv_attention_decode_fused: too many keys per split (%d / %d)", keys, nsplit);
    // dynamic LDS: the scores of one split, re-used by the merging block as ml[nsplit][2] -- sized for whichever is larger (a short prompt has
    // fewer keys per split than 2 * nsplit merge words)
    const size_t sm = (size_t)(cdiv(keys, nsplit) > 2 * nsplit ? cdiv(keys, nsplit) : 2 * nsplit) * sizeof(float);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hd == 64)
        hipLaunchKernelGGL((attn_decode_fused<64>), dim3(nsplit, Hq), dim3(256), sm, st, (const bf16*)qkv, Hq, Hkv, inv_freq, pos, pos_dev, (bf16*)kv_cache, ldkv,
                           (bf16*)o, scale, (float*)ws, nsplit);
    else
        hipLaunchKernelGGL((attn_decode_fused<128>), dim3(nsplit, Hq), dim3(256), sm, st, (const bf16*)qkv, Hq, Hkv, inv_freq, pos, pos_dev, (bf16*)kv_cache, ldkv,
                           (bf16*)o, scale, (float*)ws, nsplit);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_attention_decode_ws_bytes(int B, int Hq, int hd, int nsplit) {
    return (int)(sizeof(float) * (size_t)B * Hq * nsplit * (hd + 2));
}

static int attention_decode_impl(const void* q, int64_t q_bs, const void* k, int64_t k_bs, int64_t k_ss, const void* v,
                                 int64_t v_bs, int64_t v_ss, void* o, int64_t o_bs, int B, int Hq, int Hkv, int Sk, int hd,
                                 float scale, void* ws, int nsplit, const int* pos_dev, void* stream) {
    UFV_REQUIRE(q && k && v && o && ws && B > 0 && Hq > 0 && Hkv > 0 && Sk > 0 && nsplit > 0, "ufv_attention_decode: bad arguments");
    UFV_REQUIRE(Hq % Hkv == 0, "ufv_attention_decode: Hq must be a multiple of Hkv");
    UFV_REQUIRE(cdiv(Sk, nsplit) * sizeof(float) <= 48 * 1024, "ufv_attention_decode: too many keys per split (Sk=%d nsplit=%d)", Sk, nsplit);
    AttnArgs a;
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (bf16*)o;
    a.q_bs = q_bs; a.q_ss = 0; a.k_bs = k_bs; a.k_ss = k_ss; a.v_bs = v_bs; a.v_ss = v_ss; a.o_bs = o_bs; a.o_ss = 0;
    a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.Sq = 1; a.Sk = Sk; a.hd = hd; a.scale = scale; a.q_pos0 = Sk - 1; a.lse = nullptr;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool aligned = ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) && (k_ss % 8 == 0) &&
                         (v_ss % 8 == 0) && (k_bs % 8 == 0) && (v_bs % 8 == 0) && (q_bs % 8 == 0);
    UFV_REQUIRE(aligned, "ufv_attention_decode: q/k/v rows must be 16-byte aligned");
    switch (hd) {
        case 16: return launch_decode<16>(a, (float*)ws, nsplit, pos_dev, Sk, st);
        case 32: return launch_decode<32>(a, (float*)ws, nsplit, pos_dev, Sk, st);
        case 64: return launch_decode<64>(a, (float*)ws, nsplit, pos_dev, Sk, st);
        case 128: return launch_decode<128>(a, (float*)ws, nsplit, pos_dev, Sk, st);
    }
    ufv_set_error("ufv_attention_decode: head_dim %d not supported (16/32/64/128)", hd);
    return UFV_EUNSUPPORTED;
}

extern "C" int ufv_attention_decode(const void* q, int64_t q_bs, const void* k, int64_t k_bs, int64_t k_ss, const void* v,
                                    int64_t v_bs, int64_t v_ss, void* o, int64_t o_bs, int B, int Hq, int Hkv, int Sk, int hd,
                                    float scale, void* ws, int nsplit, void* stream) {
    return attention_decode_impl(q, q_bs, k, k_bs, k_ss, v, v_bs, v_ss, o, o_bs, B, Hq, Hkv, Sk, hd, scale, ws, nsplit, nullptr, stream);
}

// the same with the key count read from device memory (*pos_dev + 1 <= max_keys): identical launch arguments for every
// token, so a decode step can be captured into a HIP graph once and replayed
extern "C" int ufv_attention_decode_dev(const void* q, int64_t q_bs, const void* k, int64_t k_bs, int64_t k_ss, const void* v,
                                        int64_t v_bs, int64_t v_ss, void* o, int64_t o_bs, int B, int Hq, int Hkv, const int* pos_dev,
                                        int max_keys, int hd, float scale, void* ws, int nsplit, void* stream) {
    UFV_REQUIRE(pos_dev && max_keys > 0, "ufv_attention_decode_dev: bad arguments");
    return attention_decode_impl(q, q_bs, k, k_bs, k_ss, v, v_bs, v_ss, o, o_bs, B, Hq, Hkv, max_keys, hd, scale, ws, nsplit, pos_dev, stream);
}
