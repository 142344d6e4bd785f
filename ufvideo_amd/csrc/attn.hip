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

namespace {

struct AttnArgs {
    const bf16 *q, *k, *v;
    bf16* o;
    int64_t q_bs, q_ss, k_bs, k_ss, v_bs, v_ss, o_bs, o_ss;
    int B, Hq, Hkv, Sq, Sk, hd;
    float scale;
    int q_pos0;
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

template <int HD, int NW, bool CAUSAL>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_mfma(AttnArgs a) {
    using C = Cfg<HD>;
    constexpr int KS = C::KS, DT = C::DT, KC = C::KC, VC = C::VC, PK = C::PK, PV = C::PV, STAGE = C::STAGE;
    constexpr int NT = NW * 64;
    constexpr int CK = (64 * KC + NT - 1) / NT, CV = (64 * VC + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int qt = CAUSAL ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;   // heavy causal tiles first
    const int hq = blockIdx.y, b = blockIdx.z, hkv = hq / (a.Hq / a.Hkv);
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
    float m_run = -INFINITY, l_run = 0.f;
    const float sl2 = a.scale * 1.4426950408889634f;     // scores kept in the log2 domain

    // per-lane LDS offsets
    const int k_off = l31 * PK + h * 16;                                  // + hh*32*PK + ks*32
    const int v_off = (4 * h + ((lane & 15) >> 2)) * PV + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;   // + (hh*32+16s(+8))*PV + dt*64

    issue_loads(0);
    write_lds(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) issue_loads(t + 1);
        const char* kb = smem + (t & 1) * STAGE;
        const char* vb = kb + 64 * PK;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int kbase_idx = t * 64 + hh * 32;
            if (kbase_idx >= kmax) break;                       // block-uniform
            // ---- S^T[key][q] = sum_d K[key][d] Q[q][d]
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + hh * 32 * PK + k_off + ks * 32);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
            }
            // ---- online softmax over keys (register index) for query column l31
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kj = kbase_idx + (r & 3) + 8 * (r >> 2) + 4 * h;
                bool ok = kj < a.Sk;
                if (CAUSAL) ok = ok && (kj <= a.q_pos0 + qi);
                s[r] = ok ? s[r] * sl2 : -INFINITY;
                tmax = fmaxf(tmax, s[r]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float mnew = fmaxf(m_run, tmax);
            const float alpha = (m_run == -INFINITY) ? 0.f : exp2f(m_run - mnew);
            const float msub = (mnew == -INFINITY) ? 0.f : mnew;
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = exp2f(s[r] - msub);     // exp2(-inf) = 0 for masked keys
                psum += s[r];
            }
            l_run = l_run * alpha + psum;
            m_run = mnew;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
            // ---- P^T as B operand: k-step sp uses registers 8sp..8sp+7
            bf16x8 pf[2];
#pragma unroll
            for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[sp][j] = (bf16)s[8 * sp + j];
            // ---- O^T[d][q] += sum_key V[key][d] P[key][q]; V^T fragment via transposed LDS reads
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    const char* vp = vb + (hh * 32 + 16 * sp) * PV + v_off + dt * 64;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (bf16x4 __attribute__((address_space(3)))*)(vp));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (bf16x4 __attribute__((address_space(3)))*)(vp + 8 * PV));
                    const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[sp], oacc[dt], 0, 0, 0);
                }
            }
        }
        if (t + 1 < ntiles) write_lds((t + 1) & 1);
        __syncthreads();
    }

    // ---- normalise and store O[q][d], d = 32dt + (r&3) + 8(r>>2) + 4h  (4 consecutive d per register quad)
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
}

// ---- generic: one block per (q row, head, batch); scores in LDS -----------------------------------
__global__ __launch_bounds__(256) void attn_generic(AttnArgs a, int causal) {
    extern __shared__ __attribute__((aligned(16))) char smem_g[];
    float* qv = reinterpret_cast<float*>(smem_g);            // [hd]
    float* red = qv + a.hd;                                  // [16]
    float* part = red + 16;                                  // [4][hd]
    float* sc = part + 4 * a.hd;                             // [nk]
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
    // output: 4 key groups x 64 lanes over d
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

template <int HD, int NW>
int launch_mfma(const AttnArgs& a, int causal, hipStream_t st) {
    constexpr int smem = 2 * Cfg<HD>::STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_mfma<HD, NW, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_mfma<HD, NW, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    dim3 grid(cdiv(a.Sq, 32 * NW), a.Hq, a.B);
    if (causal)
        hipLaunchKernelGGL((attn_fwd_mfma<HD, NW, true>), grid, dim3(NW * 64), smem, st, a);
    else
        hipLaunchKernelGGL((attn_fwd_mfma<HD, NW, false>), grid, dim3(NW * 64), smem, st, a);
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
    a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.Sq = Sq; a.Sk = Sk; a.hd = hd; a.scale = scale; a.q_pos0 = q_pos0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool aligned = ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) &&
                         ((uintptr_t)o % 8 == 0) && (q_ss % 8 == 0) && (k_ss % 8 == 0) && (v_ss % 8 == 0) && (o_ss % 4 == 0) &&
                         (q_bs % 8 == 0) && (k_bs % 8 == 0) && (v_bs % 8 == 0) && (o_bs % 4 == 0);
    const bool hd_ok = (hd == 64 || hd == 72 || hd == 80 || hd == 96 || hd == 128);
    const bool mfma_ok = aligned && hd_ok;
    if (kernel == 1 && !mfma_ok) {
        ufv_set_error("ufv_attention: MFMA kernel needs hd in {64,72,80,96,128} and 16-byte aligned rows (hd=%d)", hd);
        return UFV_EUNSUPPORTED;
    }
    if (kernel == 1 || (kernel == 0 && mfma_ok && Sq >= 16)) {
        switch (hd) {
            case 64: return launch_mfma<64, 4>(a, causal, st);
            case 72: return launch_mfma<72, 4>(a, causal, st);
            case 80: return launch_mfma<80, 4>(a, causal, st);
            case 96: return launch_mfma<96, 4>(a, causal, st);
            case 128: return launch_mfma<128, 4>(a, causal, st);
        }
    }
    const size_t smem = sizeof(float) * ((size_t)hd * 5 + 16 + Sk);
    UFV_REQUIRE(smem <= 64 * 1024, "ufv_attention: generic kernel supports Sk <= ~16000 (Sk=%d hd=%d)", Sk, hd);
    hipLaunchKernelGGL(attn_generic, dim3(Sq, Hq, B), dim3(256), smem, st, a, causal);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}
