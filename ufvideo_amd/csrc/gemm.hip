// bf16 GEMM  C[M,N] = epilogue(A[M,K] · W[N,K]^T)  for gfx950.
//
// Both operands are K-contiguous ("NT" form: activations row-major, nn.Linear weights as
// stored), which is exactly the MFMA A/B fragment order — no transposes anywhere.
//
// Fast kernel: 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 tiles of
// v_mfma_f32_16x16x32_bf16, operands staged global->LDS with 16-byte LDS-DMA
// (global_load_lds_dwordx4), two LDS stages, XOR-swizzled 16-byte chunks so every
// ds_read_b128 fragment read is bank-conflict free (swizzle applied on the per-lane SOURCE
// address because the LDS-DMA destination is lane-linear), XCD-aware tile order.
// The MFMA is issued as mfma(W-frag, A-frag) so that each lane ends up with 4 CONSECUTIVE
// output columns of one row -> 8/16-byte epilogue accesses for bias / residual / stores.
//
// Generic kernel: any shape/alignment, one thread per output, used for tiny problems
// (region encoder, SE gates, tiny test models).
#include "common.h"
#include <atomic>
#include <vector>
#include <cstdlib>
#include "../../include/ufv.h"
#include "gemm_epi.h"

namespace {

constexpr int BN = 128, BK = 64;
// block tile = (32*MT) x 128 x 64, MT in {4,5,6}: 128/160/192 rows.  The row count is picked per launch so
// that the tile count fills the 512 resident blocks (2 per CU) with as little tail as possible.
template <int MT> struct Tile {
    static constexpr int BM = 32 * MT;
    static constexpr int STAGE_BYTES = (BM + BN) * BK * 2;   // 32 / 36 / 40 KiB
    static constexpr int SMEM_BYTES = 2 * STAGE_BYTES;       // 64 / 72 / 80 KiB -> 2 blocks / CU
};

// acc[nt][mt][j] = C[m0+wm*(BM/2)+mt*16+(lane&15)][n0+wn*64+nt*16+(lane>>4)*4+j]
template <bool OUT_F32, bool SWIGLU, int MT, int ACT>
__device__ __forceinline__ void epilogue128(const f32x4 (&acc)[4][MT], const Epi& e, int M, int m0, int n0, int wm, int wn,
                                            int frow, int fq) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + wm * (16 * MT) + mt * 16 + frow;
        if (m < M) {
            if (SWIGLU) {
                // weight rows were packed [16 gate | 16 up] alternating: nt even = gate, nt odd = up
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int n = ((n0 + wn * 64) >> 1) + p * 16 + fq * 4;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float gte = acc[2 * p][mt][j], up = acc[2 * p + 1][mt][j];
                        v[j] = swiglu_f(gte, up);
                    }
                    epi_store4<OUT_F32, ACT_NONE>(e, m, n, v[0], v[1], v[2], v[3]);
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int n = n0 + wn * 64 + nt * 16 + fq * 4;
                    epi_store4<OUT_F32, ACT>(e, m, n, acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]);
                }
            }
        }
    }
}


// The residual epilogue with the rows pipelined by hand and the bf16 epilogue with 16-byte stores: the 128-wide kernel's forms of
// epilogue256_resid / epilogue256_wide (gemm256_kernel.h, where the reasons are written down).  A wave's 64 columns are contiguous here.
template <bool OUT_F32, int MT>
__device__ __forceinline__ void epilogue128_resid(const f32x4 (&acc)[4][MT], const Epi& e, int M, int m0, int n0, int wm, int wn, int frow, int fq) {
    const int ncol = n0 + wn * 64 + fq * 4;
    f32x4 bias[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bias[nt] = e.bias ? *reinterpret_cast<const f32x4*>(e.bias + ncol + 16 * nt) : f32x4{0, 0, 0, 0};
    f32x4 r[2][4];
    auto request = [&](int mt, f32x4 (&dst)[4]) {
        const int m = min(m0 + wm * (16 * MT) + mt * 16 + frow, M - 1);
        const int mr = e.resid_rows > 0 ? m % e.resid_rows : m;
        const float* rp = e.resid + (size_t)mr * e.ldr + ncol;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[nt]) : "v"(rp + 16 * nt) : "memory");
    };
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2]), "+v"(bias[3])::"memory");
    request(0, r[0]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x4 (&cur)[4] = r[mt & 1];
        if (mt + 1 < MT) request(mt + 1, r[(mt + 1) & 1]);
        if (mt == 0 && MT > 1) asm volatile("s_waitcnt vmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3])::"memory");
        else if (mt + 1 < MT) asm volatile("s_waitcnt vmcnt(8)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3])::"memory");
        else if (MT > 1) asm volatile("s_waitcnt vmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3])::"memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3])::"memory");
        const int m = m0 + wm * (16 * MT) + mt * 16 + frow;
        const bool ok = m < M;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 v = acc[nt][mt] + bias[nt];
            v += cur[nt];
            if constexpr (OUT_F32) {
                float* pp = reinterpret_cast<float*>(e.out) + (size_t)min(m, M - 1) * e.ldc + ncol + 16 * nt;
                asm volatile("s_mov_b64 s[2:3], exec\n\ts_and_b64 exec, exec, %2\n\tglobal_store_dwordx4 %0, %1, off\n\ts_mov_b64 exec, s[2:3]\n\ts_nop 1"
                             ::"v"(pp), "v"(v), "s"(__builtin_amdgcn_ballot_w64(ok)) : "memory", "s2", "s3", "scc");
            } else {
                bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                bf16* pp = reinterpret_cast<bf16*>(e.out) + (size_t)min(m, M - 1) * e.ldc + ncol + 16 * nt;
                asm volatile("s_mov_b64 s[2:3], exec\n\ts_and_b64 exec, exec, %2\n\tglobal_store_dwordx2 %0, %1, off\n\ts_mov_b64 exec, s[2:3]\n\ts_nop 1"
                             ::"v"(pp), "v"(o), "s"(__builtin_amdgcn_ballot_w64(ok)) : "memory", "s2", "s3", "scc");
            }
        }
    }
}

template <int MT, int ACT>
__device__ __forceinline__ void epilogue128_wide(const f32x4 (&acc)[4][MT], const Epi& e, int M, int m0, int n0, int wm, int wn, int frow, int fq) {
    bf16* out = reinterpret_cast<bf16*>(e.out);
    f32x4 bias[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bias[nt] = e.bias ? *reinterpret_cast<const f32x4*>(e.bias + n0 + wn * 64 + nt * 16 + fq * 4) : f32x4{0, 0, 0, 0};
    // (the bias registers leave this wait as asm outputs: otherwise hipcc waits vmcnt(0) for them in front of every (row, run) block, i.e. behind every store)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2]), "+v"(bias[3])::"memory");
    auto pack2 = [&](float a, float b) -> unsigned {
        const bf16x2 p = {(bf16)a, (bf16)b};
        return __builtin_bit_cast(unsigned, p);
    };
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + wm * (16 * MT) + mt * 16 + frow;
#pragma unroll
        for (int run = 0; run < 2; ++run) {
            const f32x4 va = acc[2 * run][mt] + bias[2 * run], vb = acc[2 * run + 1][mt] + bias[2 * run + 1];
            unsigned x0 = pack2(act_apply_t<ACT>(va[0]), act_apply_t<ACT>(va[1])), x1 = pack2(act_apply_t<ACT>(va[2]), act_apply_t<ACT>(va[3]));
            unsigned y0 = pack2(act_apply_t<ACT>(vb[0]), act_apply_t<ACT>(vb[1])), y1 = pack2(act_apply_t<ACT>(vb[2]), act_apply_t<ACT>(vb[3]));
            auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
            auto t0 = __builtin_amdgcn_permlane16_swap((unsigned)s0[0], (unsigned)s0[1], false, false);
            auto t1 = __builtin_amdgcn_permlane16_swap((unsigned)s1[0], (unsigned)s1[1], false, false);
            if (m < M) {
                const i32x4 o = {(int)t0[0], (int)t1[0], (int)t0[1], (int)t1[1]};       // columns 8 fq .. 8 fq + 7 of the 32-column run
                *reinterpret_cast<i32x4*>(out + (size_t)m * e.ldc + n0 + wn * 64 + run * 32 + fq * 8) = o;
            }
        }
    }
}

// FP8: operands are OCP e4m3 bytes (K-tile = 128 elements = the same 128-byte LDS rows), one
// v_mfma_f32_16x16x128_f8f6f4 per fragment pair (2x the bf16 MFMA rate), per-row scales applied to the accumulators.
template <bool OUT_F32, bool SWIGLU, int MT, bool FP8>
__global__ __launch_bounds__(256, 2) void gemm_nt_128(const void* __restrict__ Av, const void* __restrict__ Wv, Epi e,
                                                       int M, int N, int K, int lda, int ldw) {
    constexpr int BM = Tile<MT>::BM, STAGE_BYTES = Tile<MT>::STAGE_BYTES;
    constexpr int ES = FP8 ? 1 : 2;                          // bytes per operand element
    const char* A = reinterpret_cast<const char*>(Av);
    const char* W = reinterpret_cast<const char*>(Wv);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware tile order: consecutive launch ids round-robin over 8 XCDs, so give each XCD
    //      a contiguous run of the tile sequence, and order the sequence in 16-row-tile groups.
    const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
    const int nwg = tiles_m * tiles_n;
    int id;
    {
        const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = bid & 7;
        id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    }
    constexpr int GM = 16;
    const int gsz = GM * tiles_n, g = id / gsz, first_m = g * GM;
    const int gm = min(tiles_m - first_m, GM);
    const int tm = first_m + (id % gsz) % gm, tn = (id % gsz) / gm;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- LDS-DMA source addresses. One wave-instruction fills 8 rows x 128 B; lane l writes
    //      LDS chunk (l&7) of row (l>>3) and therefore must FETCH chunk (l&7)^(row&7).
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
    const char* a_src[MT];
    const char* b_src[4];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r = (wave * MT + i) * 8 + lrow;
        a_src[i] = A + (size_t)min(m0 + r, M - 1) * lda * ES + lchunk * 16;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + lrow;
        b_src[i] = W + (size_t)(n0 + r) * ldw * ES + lchunk * 16;
    }
    auto stage = [&](int s, int k0) {      // k0 = byte offset of the K-tile inside a row
        char* abase = smem + s * STAGE_BYTES + wave * (MT * 1024);
        char* bbase = smem + s * STAGE_BYTES + BM * BK * 2 + wave * 4096;
#pragma unroll
        for (int i = 0; i < MT; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(a_src[i] + k0), LDS_PTR(abase + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(b_src[i] + k0), LDS_PTR(bbase + i * 1024), 16, 0, 0);
    };

    f32x4 acc[4][MT];   // [nt][mt]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row (lane&15) of a 16-row tile, 16-byte chunk kk*4+(lane>>4), XOR (row&7)
    const int frow = lane & 15, fq = lane >> 4, fx = lane & 7;
    const int a_off = (wm * (BM / 2) + frow) * 128, b_off = BM * BK * 2 + (wn * 64 + frow) * 128;

    int nk = K * ES / 128, t0 = 0;
    // split-K (only the fp32-output 192-row instantiation carries it; training's thin "reduce over tokens" products):
    // blockIdx.y takes K-tiles [t0, nk) of its slice and stores its partial tile to slice blockIdx.y of a workspace.
    constexpr bool CAN_SPLIT = OUT_F32 && !SWIGLU && MT == 6 && !FP8;
    if constexpr (CAN_SPLIT) {
        if (e.ksplit > 1) {
            const int per = (nk + e.ksplit - 1) / e.ksplit;
            t0 = blockIdx.y * per;
            nk = min(nk, t0 + per);
            if (t0 >= nk) return;
        }
    }
    stage(0, t0 * 128);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = t0; t < nk; ++t) {
        if (t + 1 < nk) stage((t + 1 - t0) & 1, (t + 1) * 128);
        const char* sb = smem + ((t - t0) & 1) * STAGE_BYTES;
        if constexpr (FP8) {
            // lane (row r, k-group g) feeds bytes 32g..32g+31 of its row = 16-byte chunks 2g and 2g+1
            const int c0 = ((2 * fq) ^ fx) << 4, c1 = ((2 * fq + 1) ^ fx) << 4;
            i32x8 af[MT], bf[4];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const i32x4 lo = *reinterpret_cast<const i32x4*>(sb + a_off + i * 16 * 128 + c0);
                const i32x4 hi = *reinterpret_cast<const i32x4*>(sb + a_off + i * 16 * 128 + c1);
                af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const i32x4 lo = *reinterpret_cast<const i32x4*>(sb + b_off + i * 16 * 128 + c0);
                const i32x4 hi = *reinterpret_cast<const i32x4*>(sb + b_off + i * 16 * 128 + c1);
                bf[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[nt], af[mt], acc[nt][mt], 0, 0, 0, 0, 0, 0);
        } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int coff = ((kk * 4 + fq) ^ fx) << 4;
            bf16x8 af[MT], bf[4];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sb + a_off + i * 16 * 128 + coff);
#pragma unroll
            for (int i = 0; i < 4; ++i) bf[i] = *reinterpret_cast<const bf16x8*>(sb + b_off + i * 16 * 128 + coff);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if constexpr (FP8) {     // de-quantise: acc[nt][mt][j] *= scale_m[row] * scale_n[col]
        f32x4 sn[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) sn[nt] = *reinterpret_cast<const f32x4*>(e.scale_n + n0 + wn * 64 + nt * 16 + fq * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float sm = e.scale_m[min(m0 + wm * (16 * MT) + mt * 16 + frow, M - 1)];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt][mt] *= sn[nt] * sm;
        }
    }

    if constexpr (CAN_SPLIT) {
        if (e.ksplit > 1) {       // partial tile of K-slice blockIdx.y -> slice blockIdx.y of the fp32 workspace [ksplit][M][ldc]
            Epi pe = e;
            pe.out = reinterpret_cast<float*>(e.out) + (size_t)blockIdx.y * M * e.ldc;
            epilogue128<OUT_F32, SWIGLU, MT, ACT_NONE>(acc, pe, M, m0, n0, wm, wn, frow, fq);
            return;
        }
    }
    // ---- epilogue (activation resolved once so the body stays unrolled)
    if constexpr (!SWIGLU) {
        if (e.resid != nullptr && e.act == ACT_NONE && !e.resid_bf16) {
            epilogue128_resid<OUT_F32, MT>(acc, e, M, m0, n0, wm, wn, frow, fq);
            return;
        }
        if constexpr (!OUT_F32) {
            if (e.resid == nullptr && e.ldc % 8 == 0 && ((uintptr_t)e.out & 15) == 0) {
                UFV_ACT_SWITCH(e.act, (epilogue128_wide<MT, ACT_>(acc, e, M, m0, n0, wm, wn, frow, fq)))
                return;
            }
        }
    }
    UFV_ACT_SWITCH(e.act, (epilogue128<OUT_F32, SWIGLU, MT, ACT_>(acc, e, M, m0, n0, wm, wn, frow, fq)))
}

// ---- generic kernel: one thread per output element -------------------------------------------
template <bool OUT_F32, bool SWIGLU>
__global__ void gemm_nt_generic(const bf16* __restrict__ A, const bf16* __restrict__ W, Epi e, int M, int N, int K,
                                int lda, int ldw) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;      // output column (post-SWIGLU index when SWIGLU)
    const int m = blockIdx.y;
    const int n_out = SWIGLU ? N / 2 : N;
    if (n >= n_out || m >= M) return;
    const bf16* a = A + (size_t)m * lda;
    float v;
    if (SWIGLU) {
        const int pg = (n / 16) * 32 + (n % 16);
        const bf16* wg = W + (size_t)pg * ldw;
        const bf16* wu = W + (size_t)(pg + 16) * ldw;
        float g = 0.f, u = 0.f;
        for (int k = 0; k < K; ++k) {
            const float x = (float)a[k];
            g += x * (float)wg[k];
            u += x * (float)wu[k];
        }
        v = swiglu_f(g, u);
    } else {
        const bf16* w = W + (size_t)n * ldw;
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += (float)a[k] * (float)w[k];
        v = s;
        if (e.bias) v += e.bias[n];
        v = act_apply(v, e.act);
    }
    if (e.resid) v += resid_load1(e, (size_t)(e.resid_rows > 0 ? m % e.resid_rows : m) * e.ldr + n);
    if (OUT_F32)
        reinterpret_cast<float*>(e.out)[(size_t)m * e.ldc + n] = v;
    else
        reinterpret_cast<bf16*>(e.out)[(size_t)m * e.ldc + n] = (bf16)v;
}

// ---- GEMV (M <= 8): one wave per output column, weights streamed once with 16-byte loads -----
template <bool OUT_F32, bool SWIGLU, int MR>
__global__ __launch_bounds__(256) void gemv_nt(const bf16* __restrict__ A, const bf16* __restrict__ W, Epi e, int M, int N,
                                               int K, int lda, int ldw) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);       // output index (post-SWIGLU when SWIGLU)
    const int n_out = SWIGLU ? N / 2 : N;
    if (n >= n_out) return;
    const int row0 = SWIGLU ? (n / 16) * 32 + (n % 16) : n;
    const bf16* w0 = W + (size_t)row0 * ldw;
    const bf16* w1 = W + (size_t)(row0 + 16) * ldw;
    const int r0 = blockIdx.y * MR;
    A += (size_t)r0 * lda;
    M = min(M - r0, MR);
    float s0[MR], s1[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) s0[r] = s1[r] = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w0 + k);
        bf16x8 uv;
        if (SWIGLU) uv = *reinterpret_cast<const bf16x8*>(w1 + k);
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            if (r < M) {
                // two products per instruction (v_dot2_f32_bf16, fp32 accumulate): with a cvt + cvt + fma per element the kernel was VALU-bound on the
                // 32-row SE products of the connector (896 x 3584 and 3584 x 896: 26 us each, 16 per clip)
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(A + (size_t)r * lda + k);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s0[r] = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{av[2 * j], av[2 * j + 1]}, bf16x2{wv[2 * j], wv[2 * j + 1]}, s0[r], false);
                    if (SWIGLU) s1[r] = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{av[2 * j], av[2 * j + 1]}, bf16x2{uv[2 * j], uv[2 * j + 1]}, s1[r], false);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        if (r >= M) break;
        float v = wave_sum(s0[r]);
        if (SWIGLU) {
            const float u = wave_sum(s1[r]);
            v = swiglu_f(v, u);
        } else {
            if (e.bias) v += e.bias[n];
            v = act_apply(v, e.act);
        }
        if (lane == 0) {
            const int m = r0 + r;
            if (e.resid) v += resid_load1(e, (size_t)(e.resid_rows > 0 ? m % e.resid_rows : m) * e.ldr + n);
            if (OUT_F32)
                reinterpret_cast<float*>(e.out)[(size_t)m * e.ldc + n] = v;
            else
                reinterpret_cast<bf16*>(e.out)[(size_t)m * e.ldc + n] = (bf16)v;
        }
    }
}

// ---- small-M product on the matrix cores (8 < M <= 64: the connector's squeeze-excite 32 x 3584 -> 896 -> 3584) ----------------------------
// A block owns 16 output columns and 32 rows; its NW waves split K into contiguous parts and stream both operands straight from global memory into
// v_mfma_f32_16x16x32_bf16 fragments (a lane's 16 bytes = 8 consecutive k of one row: no LDS staging, every load of a wave is independent of the others, so
// the whole K range of the wave is in flight at once).  The partial tiles meet in LDS and wave 0 adds them in wave order (deterministic).  The one-wave-per-
// column gemv_nt above spent 19 us on each of these products in a chain of K / 512 dependent round trips per wave.
template <bool OUT_F32, int NW, int UB>
__global__ __launch_bounds__(64 * NW) void gemm_small_m(const bf16* __restrict__ A, const bf16* __restrict__ W, Epi e, int M, int N, int K, int lda, int ldw) {
    __shared__ float part[NW - 1][2][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16, r0 = blockIdx.y * 32;
    const int ksteps = K >> 5, per = (ksteps + NW - 1) / NW;
    const int s0 = wave * per, s1 = min(ksteps, s0 + per);
    const int lr = lane & 15, kq = (lane >> 4) * 8;
    const bf16* wp = W + (size_t)min(n0 + lr, N - 1) * ldw + kq;
    const bf16* ap0 = A + (size_t)min(r0 + lr, M - 1) * lda + kq;
    const bf16* ap1 = A + (size_t)min(r0 + 16 + lr, M - 1) * lda + kq;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int s = s0;
    for (; s + UB <= s1; s += UB) {
        bf16x8 wv[UB], a0[UB], a1[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            wv[u] = *reinterpret_cast<const bf16x8*>(wp + (s + u) * 32);
            a0[u] = *reinterpret_cast<const bf16x8*>(ap0 + (s + u) * 32);
            a1[u] = *reinterpret_cast<const bf16x8*>(ap1 + (s + u) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);          // all 3 UB loads issued before the first product waits (the scheduler otherwise interleaves them in groups)
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[u], a0[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[u], a1[u], acc1, 0, 0, 0);
        }
    }
    for (; s < s1; ++s) {
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(wp + s * 32);
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ap0 + s * 32), a1 = *reinterpret_cast<const bf16x8*>(ap1 + s * 32);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, a0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, a1, acc1, 0, 0, 0);
    }
    if (wave > 0) {
        *reinterpret_cast<f32x4*>(part[wave - 1][0][lane]) = acc0;
        *reinterpret_cast<f32x4*>(part[wave - 1][1][lane]) = acc1;
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) {
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(part[w][0][lane]), p1 = *reinterpret_cast<const f32x4*>(part[w][1][lane]);
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc0[i] += p0[i]; acc1[i] += p1[i]; }
    }
    // lane holds out[m = r0 + 16 h + (lane & 15)][n0 + 4 (lane >> 4) + i], i = 0..3
    const int nb = n0 + (lane >> 4) * 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m = r0 + 16 * h + lr;
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = nb + i;
            if (n >= N) continue;
            float v = h ? acc1[i] : acc0[i];
            if (e.bias) v += e.bias[n];
            v = act_apply(v, e.act);
            if (e.resid) v += resid_load1(e, (size_t)(e.resid_rows > 0 ? m % e.resid_rows : m) * e.ldr + n);
            if (OUT_F32)
                reinterpret_cast<float*>(e.out)[(size_t)m * e.ldc + n] = v;
            else
                reinterpret_cast<bf16*>(e.out)[(size_t)m * e.ldc + n] = (bf16)v;
        }
    }
}

// ---- GEMV with fp8 operands (M <= 64): weights streamed once, 16 e4m3 bytes per lane per load; dequantised in registers
template <bool OUT_F32, bool SWIGLU, int MR>
__global__ __launch_bounds__(256) void gemv_nt_fp8(const uint8_t* __restrict__ A, const uint8_t* __restrict__ W, Epi e, int M, int N,
                                                   int K, int lda, int ldw) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_out = SWIGLU ? N / 2 : N;
    if (n >= n_out) return;
    const int row0 = SWIGLU ? (n / 16) * 32 + (n % 16) : n;
    const uint8_t* w0 = W + (size_t)row0 * ldw;
    const uint8_t* w1 = W + (size_t)(row0 + 16) * ldw;
    const int r0 = blockIdx.y * MR;
    A += (size_t)r0 * lda;
    const int Mb = min(M - r0, MR);
    float s0[MR], s1[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) s0[r] = s1[r] = 0.f;
    auto unpack = [](const i32x4 v, float (&f)[16]) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            f[4 * d + 0] = __builtin_amdgcn_cvt_f32_fp8(v[d], 0);
            f[4 * d + 1] = __builtin_amdgcn_cvt_f32_fp8(v[d], 1);
            f[4 * d + 2] = __builtin_amdgcn_cvt_f32_fp8(v[d], 2);
            f[4 * d + 3] = __builtin_amdgcn_cvt_f32_fp8(v[d], 3);
        }
    };
    for (int k = lane * 16; k < K; k += 1024) {
        float wv[16], uv[16];
        unpack(*reinterpret_cast<const i32x4*>(w0 + k), wv);
        if (SWIGLU) unpack(*reinterpret_cast<const i32x4*>(w1 + k), uv);
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            if (r < Mb) {
                float av[16];
                unpack(*reinterpret_cast<const i32x4*>(A + (size_t)r * lda + k), av);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    s0[r] += av[j] * wv[j];
                    if (SWIGLU) s1[r] += av[j] * uv[j];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        if (r >= Mb) break;
        const int m = r0 + r;
        float v = wave_sum(s0[r]) * e.scale_m[m] * e.scale_n[row0];
        if (SWIGLU) {
            const float u = wave_sum(s1[r]) * e.scale_m[m] * e.scale_n[row0 + 16];
            v = swiglu_f(v, u);
        } else {
            if (e.bias) v += e.bias[n];
            v = act_apply(v, e.act);
        }
        if (lane == 0) {
            if (e.resid) v += resid_load1(e, (size_t)(e.resid_rows > 0 ? m % e.resid_rows : m) * e.ldr + n);
            if (OUT_F32)
                reinterpret_cast<float*>(e.out)[(size_t)m * e.ldc + n] = v;
            else
                reinterpret_cast<bf16*>(e.out)[(size_t)m * e.ldc + n] = (bf16)v;
        }
    }
}

// ---- single-row GEMV for the decode step: y[N] = epilogue(h[K] . W[N,K]^T), h = the bf16 row `a`, or (RMS) the RMSNorm of
//      an fp32 row computed in the prologue exactly as rmsnorm_k rounds it (so fusing the norm changes no bit).  h lives in
//      LDS; a wave produces 4 outputs at once (4 or 8 weight rows in flight per lane: latency hidden, 4x fewer blocks).
template <bool OUT_F32, bool SWIGLU, bool RMS, bool W8, int NO = 4>
__global__ __launch_bounds__(256) void gemv1_nt(const bf16* __restrict__ a, const float* __restrict__ x, const float* __restrict__ g,
                                                float eps, const void* __restrict__ Wv, const float* __restrict__ w_scale, Epi e, int N,
                                                int K, int ldw) {
    extern __shared__ __attribute__((aligned(16))) char smem_v[];
    bf16* hs = reinterpret_cast<bf16*>(smem_v);
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (RMS) {
        // every wave reduces the whole row with rmsnorm_k's own order (lane owns float4 chunks lane + 64 i, wave_sum), so
        // r -- and with it every rounded h[k] = bf16(w * (x * r)) -- is the value the stand-alone kernel produces
        float q = 0.f;
        for (int c = lane; c < (K >> 2); c += 64) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * c);
            q += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
        const float r = rsqrtf(wave_sum(q) / K + eps);
        for (int k = tid * 4; k < K; k += 1024) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + k), w = *reinterpret_cast<const f32x4*>(g + k);
            bf16x4 o = {(bf16)(w[0] * (v[0] * r)), (bf16)(w[1] * (v[1] * r)), (bf16)(w[2] * (v[2] * r)), (bf16)(w[3] * (v[3] * r))};
            *reinterpret_cast<bf16x4*>(hs + k) = o;
        }
    } else {
        for (int k = tid * 8; k < K; k += 2048) *reinterpret_cast<bf16x8*>(hs + k) = *reinterpret_cast<const bf16x8*>(a + k);
    }
    __syncthreads();
    // W8: quantise the row exactly as ufv_quantize_fp8 does (scale = amax/448, q = rne_e4m3(h * (1/scale))); codes behind hs
    uint8_t* hq = reinterpret_cast<uint8_t*>(smem_v) + (size_t)K * 2;
    float a_scale = 1.f;
    if (W8) {
        float amax = 0.f;
        for (int k = tid; k < K; k += 256) amax = fmaxf(amax, fabsf((float)hs[k]));
        amax = wave_max(amax);
        if (lane == 0) red[wave] = amax;
        __syncthreads();
        amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        a_scale = amax > 0.f ? amax / 448.0f : 1.0f;
        const float inv = 1.0f / a_scale;
        for (int k = tid * 4; k < K; k += 1024) {
            int w = 0;
            w = __builtin_amdgcn_cvt_pk_fp8_f32((float)hs[k] * inv, (float)hs[k + 1] * inv, w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32((float)hs[k + 2] * inv, (float)hs[k + 3] * inv, w, true);
            *reinterpret_cast<int*>(hq + k) = w;
        }
        __syncthreads();
    }
    // NO = outputs per wave: 4 for the wide layers (4x fewer blocks, 4-8 weight rows in flight per lane), 1 for N <= 8192 (QKV, o,
    // down: with 4 the grid is 224-288 blocks = one 4-wave block per CU and the weight stream runs at 2 TB/s)
    constexpr int ES = W8 ? 1 : 2;
    const char* W = reinterpret_cast<const char*>(Wv);
    const int n_out = SWIGLU ? N / 2 : N;
    const int n0 = (blockIdx.x * 4 + wave) * NO;
    if (n0 >= n_out) return;
    const char* wr[NO];
    const char* wu[NO];
    int rows[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int n = min(n0 + j, n_out - 1);
        rows[j] = SWIGLU ? (n / 16) * 32 + (n % 16) : n;
        wr[j] = W + (size_t)rows[j] * ldw * ES;
        wu[j] = W + (size_t)(rows[j] + 16) * ldw * ES;
    }
    float s0[NO], s1[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) s0[j] = s1[j] = 0.f;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 p0[NO], p1[NO];                                     // fp8 path: packed partial sums (v_cvt_pk_f32_fp8 + v_pk_fma_f32)
#pragma unroll
    for (int j = 0; j < NO; ++j) p0[j] = p1[j] = f32x2{0.f, 0.f};
    if constexpr (W8) {
        auto unpack = [](const i32x4 v, f32x2 (&f)[8]) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                f[2 * d + 0] = __builtin_amdgcn_cvt_pk_f32_fp8(v[d], false);
                f[2 * d + 1] = __builtin_amdgcn_cvt_pk_f32_fp8(v[d], true);
            }
        };
        for (int k = lane * 16; k < K; k += 1024) {
            f32x2 hv[8];
            unpack(*reinterpret_cast<const i32x4*>(hq + k), hv);
            i32x4 wv[NO], uv[NO];
#pragma unroll
            for (int j = 0; j < NO; ++j) {
                wv[j] = *reinterpret_cast<const i32x4*>(wr[j] + k);
                if (SWIGLU) uv[j] = *reinterpret_cast<const i32x4*>(wu[j] + k);
            }
#pragma unroll
            for (int j = 0; j < NO; ++j) {
                f32x2 wf[8];
                unpack(wv[j], wf);
#pragma unroll
                for (int i = 0; i < 8; ++i) p0[j] = __builtin_elementwise_fma(hv[i], wf[i], p0[j]);
                if (SWIGLU) {
                    unpack(uv[j], wf);
#pragma unroll
                    for (int i = 0; i < 8; ++i) p1[j] = __builtin_elementwise_fma(hv[i], wf[i], p1[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NO; ++j) { s0[j] = p0[j][0] + p0[j][1]; s1[j] = p1[j][0] + p1[j][1]; }
    } else {
        for (int k = lane * 8; k < K; k += 512) {
            const bf16x8 hv = *reinterpret_cast<const bf16x8*>(hs + k);
            bf16x8 wv[NO], uv[NO];
#pragma unroll
            for (int j = 0; j < NO; ++j) {
                wv[j] = *reinterpret_cast<const bf16x8*>(wr[j] + (size_t)k * 2);
                if (SWIGLU) uv[j] = *reinterpret_cast<const bf16x8*>(wu[j] + (size_t)k * 2);
            }
#pragma unroll
            for (int j = 0; j < NO; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    s0[j] += (float)hv[i] * (float)wv[j][i];
                    if (SWIGLU) s1[j] += (float)hv[i] * (float)uv[j][i];
                }
        }
    }
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int n = n0 + j;
        float v = wave_sum(s0[j]);
        if (W8) v *= a_scale * w_scale[rows[j]];
        if (SWIGLU) {
            float u = wave_sum(s1[j]);
            if (W8) u *= a_scale * w_scale[rows[j] + 16];
            v = swiglu_f(v, u);
        } else {
            if (e.bias) v += e.bias[min(n, n_out - 1)];
            v = act_apply(v, e.act);
        }
        if (lane == 0 && n < n_out) {
            if (e.resid) v += resid_load1(e, n);
            if (OUT_F32) reinterpret_cast<float*>(e.out)[n] = v;
            else reinterpret_cast<bf16*>(e.out)[n] = (bf16)v;
        }
    }
}

template <bool F, bool S, int MT, bool Q>
int launch_fast_mt(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, hipStream_t st) {
    UFV_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_128<F, S, MT, Q>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, Tile<MT>::SMEM_BYTES);
    );
    const int tiles = cdiv(M, Tile<MT>::BM) * (N / BN);
    hipLaunchKernelGGL((gemm_nt_128<F, S, MT, Q>), dim3(tiles), dim3(256), Tile<MT>::SMEM_BYTES, st, A, W, e, M, N, K, lda, ldw);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

// rows per block tile: minimise (rounds of 512 resident blocks) x (rows per tile)
inline int pick_mt(int M, int N) {
    int best = 4;
    long best_cost = -1;
    for (int mt = 4; mt <= 6; ++mt) {
        const long tiles = (long)cdiv(M, 32 * mt) * (N / BN);
        const long cost = ((tiles + 511) / 512) * 32 * mt;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = mt; }
    }
    return best;
}

// Kernel choice: a cost model in shader ticks, fitted to the in-kernel timeline of tools/lab/gemm_lab.hip and checked against
// tools/gemm_shapes.py on MI355X (it picks the measured-fastest kernel, or one within 0.3 % of it, on every config-#2 shape).
//   ping-pong kernel at tile (32*(MA0+MA1)) x (128+64*NB1), one block per CU, persistent:
//       rounds = ceil(tiles / 256);  K-tile = 2 * sum over the two phases of max(18.5 * MFMAs of the phase, 500)
//       (a phase's MFMA step runs beside the other wave group's load step, which costs ~500 ticks whatever the tile: 8 LDS-DMA
//       pieces per wave and K-tile); seam (prologue + epilogue, partly overlapped) = 2500 + area * (0.33 fp32+residual | 0.183 bf16)
//   128-wide kernel, (32*MT) x 128 tiles, two blocks per CU, not persistent:
//       rounds = ceil(tiles / 512);  K-tile = 2 * area / 21.4;  seam = 9000 + 2 * area * (0.33 | 0.183)
// returns 0 for the 128-wide kernel, else the ping-pong shape code (ufv.h UFV_GEMM_PP).  K in elements of a bf16 K-tile (fp8: K / 2).
struct PPShape { int code, ma0, ma1, nb1; };
constexpr PPShape PP_SHAPES[] = {{1442, 4, 4, 2}, {1432, 4, 3, 2}, {1332, 3, 3, 2}, {1322, 3, 2, 2}, {1441, 4, 4, 1}, {1431, 4, 3, 1}, {1331, 3, 3, 1}};

// split-K forms (aligned split, gemm256_kernel.h KSPL; fp32 output without activation only): items = tiles x parts in rounds of 256, a part's seam
// carries the turn-ordered read-modify-write of its tile (+12000 ticks); taken when the model sees at least 5 % over the best unsplit kernel --
// in practice the decoder's `down` projection (M = 2399: 140 tiles of 256x256 for 296 K-tiles; 348 -> ~270 us) and its 64-frame form.
// rope_cost (optional out): the model's ticks of the shape the fused QKV + RoPE kernel is built for (1332) and that shape's code in rope_pick
// pp_only: N is no multiple of 128 (but N % 192 is 0 or 128: Hiera-L's widths since round 6): the 128-wide kernel is no candidate
inline int choose_kernel(int M, int N, int K, bool out_f32, bool swiglu, bool can_split = false, double* best_cost = nullptr, double* rope_cost = nullptr,
                         int* rope_pick = nullptr, bool pp_only = false) {
    const double nk = K / 64.0, c_out = out_f32 ? 0.21 : 0.183;      // (fp32 + residual: 0.33 before the row-pipelined residual epilogue of round 3)
    double best = 1e30, rbest = 1e30;
    int pick = 0, rpick = 0;
    for (int mt = 4; mt <= 6 && !pp_only; ++mt) {
        const long t = (long)cdiv(M, 32 * mt) * (N / BN);
        const double area = 32.0 * mt * 128;
        const double c = (double)((t + 511) / 512) * (nk * 2.0 * area / 21.4 + 9000.0 + 2.0 * area * c_out);
        if (c < best) best = c;
    }
    if (M < 256) return 0;
    for (const PPShape& s : PP_SHAPES) {
        const int bm = 32 * (s.ma0 + s.ma1), bn = 128 + 64 * s.nb1, nt = 2 + s.nb1;
        if (swiglu && s.code != 1442) continue;
        if (N % bn != 0 && N % bn != 128) continue;
        const long t = (long)cdiv(M, bm) * cdiv(N, bn);
        const double pa = s.ma0 * nt * 2 * 18.5, pb = s.ma1 * nt * 2 * 18.5;
        // the step floor (a group's load step): 500 ticks; 415 where the fragment reads are prefetched between the MFMAs (gemm256_kernel.h FPF: the 192 x 192
        // shape runs 1650 ticks per K-tile instead of 2000; the 224 x 192 shape gains less and keeps the old figure)
        const double fl = (s.ma0 + s.ma1 <= 6 && s.nb1 == 1) ? 415.0 : 500.0;
        const double tk = 2.0 * ((pa > fl ? pa : fl) + (pb > fl ? pb : fl));
        const double c = (double)((t + 255) / 256) * (nk * tk + 2500.0 + (double)bm * bn * c_out);
        if (c < best) { best = c; pick = s.code; }
        if (s.code == 1332 && c < rbest) { rbest = c; rpick = s.code; }
    }
    if (best_cost) *best_cost = best;
    if (rope_cost) *rope_cost = rbest;
    if (rope_pick) *rope_pick = rpick;
    // Long-K fp32 products whose 192x192 tiling fills the chip (the decoder's `down` at S = 2399: 247 tiles for 256 CUs; 475 at 64 frames): since the
    // row-pipelined residual epilogue (round 3) the unsplit 192x192 kernel beats both the split-K form and the 128-wide kernel there (292 us against
    // 314 / 353, tools/gemm_shapes.py); the tick model above still carries the old epilogue's K-tile floor for small tiles, so this case is decided here.
    // (round 4: bf16 outputs too -- the connector's Conv3d as a GEMM, 2304 x 3584 x 28672, 228 tiles: 427 us against 580 on the 128-wide kernel the model picked)
    // (round 5: bf16 outputs from K = 2048 when ONE round of 192 x 192 tiles fills the chip -- the connector's stage-2 1x1 convolutions and readout, 2304 x 3584 x 3584,
    // 228 tiles: 57.2 us against 61.3 on the 128-wide kernel the model picked, tools/gemm_shapes.py proj_ro)
    if (pp_only) return pick;
    if (!swiglu && K >= (out_f32 ? 8192 : 2048) && M >= 256 && (N % 192 == 0 || N % 192 == 128)) {
        const long t = (long)cdiv(M, 192) * cdiv(N, 192);
        if ((out_f32 || t <= 256) && (double)t / (256.0 * ((t + 255) / 256)) >= (out_f32 ? 0.9 : 0.85)) return 1331;
    }
    if (can_split && out_f32 && !swiglu && K % 64 == 0) {
        // regression over 31 measured (shape, parts) runs of tools/gemm_splitk.py (within 5 % of all but the 3-part splits, which it flatters
        // by 16 %: parts >= 4 only): us = rounds * (K-tiles per part * (1.114 a + 0.328 r) + 2.73 a + 14.63), a = tile area / 256^2, r = operand
        // rows staged per K-tile / 512 (a 96-row half is staged as 128); 1389 ticks per us on the box the unsplit model's ticks were fitted on
        const int nkt = K / 64;
        double best_split = best * 0.95;
        for (const PPShape& s : PP_SHAPES) {
            const int bm = 32 * (s.ma0 + s.ma1), bn = 128 + 64 * s.nb1;
            if (N % bn != 0 && N % bn != 128) continue;
            const long t = (long)cdiv(M, bm) * cdiv(N, bn);
            const double a = (double)bm * bn / 65536.0, r = ((s.ma0 > 2 ? 128 : 64) + (s.ma1 > 2 ? 128 : 64) + 128 + 64 * s.nb1) / 512.0;
            for (int parts = 4; parts <= 8; ++parts) {
                const int per = cdiv(nkt, parts);
                if ((parts - 1) * per >= nkt || per < 24) continue;
                const double c = 1389.0 * (double)((t * parts + 255) / 256) * (per * (1.114 * a + 0.328 * r) + 2.73 * a + 14.63);
                if (c < best_split) { best_split = c; pick = s.code + 10000 * parts; }
            }
        }
    }
    return pick;
}

// Whether UFV_GEMM_AUTO may pick the split-K form (ufv_gemm_set_splitk).  The environment variable UFV_GEMM_NO_SPLITK is read ONCE, when the
// library is loaded, as the initial value; after that only the setter changes it (an atomic: no getenv on the launch path).
static std::atomic<int> g_splitk{getenv("UFV_GEMM_NO_SPLITK") == nullptr ? 1 : 0};
static inline bool splitk_enabled() { return g_splitk.load(std::memory_order_relaxed) != 0; }

template <bool F, bool S, bool Q>
int launch_fast(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, hipStream_t st) {
    if (!S) {
        switch (pick_mt(M, N)) {
            case 5: return launch_fast_mt<F, S, 5, Q>(A, W, e, M, N, K, lda, ldw, st);
            case 6: return launch_fast_mt<F, S, 6, Q>(A, W, e, M, N, K, lda, ldw, st);
        }
    }
    return launch_fast_mt<F, S, 4, Q>(A, W, e, M, N, K, lda, ldw, st);
}

template <bool F, bool S, bool Q>
int launch_any(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, int force_shape, hipStream_t st) {
    const int force = force_shape & 0xff, shape = force_shape >> 8;       // UFV_GEMM_PP(shape): the ping-pong kernel at a named tile shape
    constexpr int KE = Q ? 128 : BK;        // elements per K-tile; row pitches must keep rows 16-byte aligned
    constexpr int LDA = Q ? 16 : 8;
    const bool aligned = ((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0) && (lda % LDA == 0) && (ldw % LDA == 0) &&
                         ((uintptr_t)e.out % 16 == 0) && (e.ldc % 4 == 0) && (!e.bias || (uintptr_t)e.bias % 16 == 0) &&
                         (!e.resid || ((uintptr_t)e.resid % 16 == 0 && e.ldr % 4 == 0)) &&
                         (!Q || ((uintptr_t)e.scale_n % 16 == 0));
    const bool fast_ok = aligned && (N % BN == 0) && (K % KE == 0) && M >= 1;
    const bool gemv_ok = aligned && (K % (Q ? 16 : 8) == 0) && M <= 64 && (!S || N % 32 == 0);
    if (force == UFV_GEMM_FAST && !fast_ok) {
        ufv_set_error("ufv_gemm: fast kernel needs N%%128==0, K%%%d==0, 16-byte aligned operands (M=%d N=%d K=%d)", KE, M, N, K);
        return UFV_EUNSUPPORTED;
    }
    if ((force == UFV_GEMM_AUTO && gemv_ok) || force == UFV_GEMM_GEMV) {
        if (!gemv_ok) {
            ufv_set_error("ufv_gemm: gemv kernel needs M<=64, K%%%d==0 (M=%d N=%d K=%d)", Q ? 16 : 8, M, N, K);
            return UFV_EUNSUPPORTED;
        }
        const int n_out = S ? N / 2 : N;
        if constexpr (!Q && !S) {
            static const bool no_small_m = getenv("UFV_NO_SMALL_M") != nullptr;          // lab switch (same-box A/B)
            // chosen by (N, K) alone, never by M: a frame chunk of a video (M = its frames) must get the arithmetic the whole video gets (frame-sharded encode is
            // bit-identical to the single-process pass); products with a small weight matrix (the SAM heads' 256-wide layers) and the vocabulary-sized lm_head row stay on the GEMV
            if (K % 32 == 0 && N >= 512 && K >= 512 && N <= 8192 && !no_small_m) {
                if (K >= 2048)          // 8 waves x 14 K-steps in flight each: the whole K range of the block requested at once
                    hipLaunchKernelGGL((gemm_small_m<F, 8, 14>), dim3(cdiv(N, 16), cdiv(M, 32)), dim3(512), 0, st, (const bf16*)A, (const bf16*)W, e, M, N, K, lda, ldw);
                else
                    hipLaunchKernelGGL((gemm_small_m<F, 4, 7>), dim3(cdiv(N, 16), cdiv(M, 32)), dim3(256), 0, st, (const bf16*)A, (const bf16*)W, e, M, N, K, lda, ldw);
                UFV_CHECK_LAUNCH();
                return UFV_OK;
            }
        }
        if constexpr (Q)
            hipLaunchKernelGGL((gemv_nt_fp8<F, S, 8>), dim3(cdiv(n_out, 4), cdiv(M, 8)), dim3(256), 0, st, (const uint8_t*)A,
                               (const uint8_t*)W, e, M, N, K, lda, ldw);
        else
            hipLaunchKernelGGL((gemv_nt<F, S, 8>), dim3(cdiv(n_out, 4), cdiv(M, 8)), dim3(256), 0, st, (const bf16*)A, (const bf16*)W,
                               e, M, N, K, lda, ldw);
        UFV_CHECK_LAUNCH();
        return UFV_OK;
    }
    const bool big_ok = fast_ok && M >= 256;
    if (force == UFV_GEMM_FAST256 && !big_ok) {
        ufv_set_error("ufv_gemm: 256-tile kernel needs M>=256, N%%128==0, K%%%d==0 (M=%d N=%d K=%d)", KE, M, N, K);
        return UFV_EUNSUPPORTED;
    }
    if (force == UFV_GEMM_STREAMK && !big_ok) {
        ufv_set_error("ufv_gemm: stream-K kernel needs M>=256, N%%128==0, K%%%d==0 (M=%d N=%d K=%d)", KE, M, N, K);
        return UFV_EUNSUPPORTED;
    }
    if (force == UFV_GEMM_STREAMK) return ufv_launch_gemm256(A, W, e, M, N, K, lda, ldw, F, S, Q, true, 0, st);
    if (force == UFV_GEMM_FAST256) return ufv_launch_gemm256(A, W, e, M, N, K, lda, ldw, F, S, Q, false, shape, st);
    if (force == UFV_GEMM_AUTO && big_ok) {
        const int pick = choose_kernel(M, N, Q ? K / 2 : K, F, S, !Q && e.act == ACT_NONE && e.resid_rows == 0 && splitk_enabled());
        if (pick) return ufv_launch_gemm256(A, W, e, M, N, K, lda, ldw, F, S, Q, false, pick, st);
    }
    // N no multiple of 128 but a multiple of 192, or 128 past one (192, 320, 576, 960, 1728: the widths of Hiera-L's stages since round 6, ufvideo_amd/model/sam2.py
    // _pad_dim): the ping-pong kernel's 192-wide tile shapes take it (their N condition is N % 192 in {0, 128}); bf16 operands only (e4m3 K-tiles are 128 deep: that
    // mode keeps its widths padded to 128)
    if constexpr (!Q && !S) {
        if (force == UFV_GEMM_AUTO && aligned && !fast_ok && (N % 192 == 0 || N % 192 == 128) && K % KE == 0 && M >= 256) {
            const int pick = choose_kernel(M, N, K, F, false, false, nullptr, nullptr, nullptr, true);
            if (pick) return ufv_launch_gemm256(A, W, e, M, N, K, lda, ldw, F, false, false, false, pick, st);
        }
    }
    if ((force == UFV_GEMM_AUTO && fast_ok && M > 64) || force == UFV_GEMM_FAST)
        return launch_fast<F, S, Q>(A, W, e, M, N, K, lda, ldw, st);
    if constexpr (Q) {
        ufv_set_error("ufv_gemm_fp8: needs N%%128==0 and K%%128==0 (or M<=64, K%%16==0) and 16-byte aligned rows (M=%d N=%d K=%d)", M, N, K);
        return UFV_EUNSUPPORTED;
    } else {
        if (S && N % 32 != 0) {
            ufv_set_error("ufv_gemm: SWIGLU needs N%%32==0");
            return UFV_EINVAL;
        }
        const int n_out = S ? N / 2 : N;
        hipLaunchKernelGGL((gemm_nt_generic<F, S>), dim3(cdiv(n_out, 64), M), dim3(64), 0, st, (const bf16*)A, (const bf16*)W, e, M, N, K,
                           lda, ldw);
        UFV_CHECK_LAUNCH();
        return UFV_OK;
    }
}

// Launch timing of the dominant kernel for bench.py's roofline entry (ufv_gemm_timing): while enabled, every tile-kernel GEMM with the SwiGLU
// epilogue (the decoder's gate/up projection) is bracketed by a HIP event pair on ITS stream, wherever the call comes from (op-level or a stage call).
struct TimedLaunch { hipEvent_t s, e; int M, N, K; };
static int g_timing = 0;                 // 0 = off; n > 0: every n-th eligible launch is bracketed (each bracket costs the stream ~11 us of idle: two event packets)
static unsigned g_timing_seen = 0;
static std::vector<TimedLaunch>& timed_launches() { static std::vector<TimedLaunch> v; return v; }

template <bool Q>
int gemm_entry(const void* A, int lda, const float* a_scale, const void* W, int ldw, const float* w_scale, void* C, int ldc, int out_f32,
               int M, int N, int K, const float* bias, int act, const float* resid, int ldr, int resid_rows, int swiglu, int kernel,
               void* stream, int resid_bf16 = 0) {
    Epi e;
    e.bias = bias; e.resid = resid; e.out = C; e.ldr = ldr; e.ldc = ldc; e.act = act; e.resid_rows = resid_rows; e.resid_bf16 = resid_bf16;
    e.scale_m = a_scale; e.scale_n = w_scale; e.dump_f32 = 0; e.ksplit = 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    TimedLaunch tl{};
    const bool timed = g_timing > 0 && swiglu && M > 64 && (g_timing_seen++ % (unsigned)g_timing) == 0 && hipEventCreate(&tl.s) == hipSuccess &&
                       hipEventCreate(&tl.e) == hipSuccess;
    if (timed) (void)hipEventRecord(tl.s, st);
    int rc;
    if (out_f32)
        rc = swiglu ? launch_any<true, true, Q>(A, W, e, M, N, K, lda, ldw, kernel, st)
                    : launch_any<true, false, Q>(A, W, e, M, N, K, lda, ldw, kernel, st);
    else
        rc = swiglu ? launch_any<false, true, Q>(A, W, e, M, N, K, lda, ldw, kernel, st)
                    : launch_any<false, false, Q>(A, W, e, M, N, K, lda, ldw, kernel, st);
    if (timed) {
        (void)hipEventRecord(tl.e, st);
        tl.M = M; tl.N = N; tl.K = K;
        timed_launches().push_back(tl);
    }
    return rc;
}

}  // namespace

// what UFV_GEMM_AUTO would launch for a bf16 GEMM of this shape: 0 = the 128-wide kernel (or GEMV / generic for tiny shapes), else the ping-pong
// kernel's shape code as in UFV_GEMM_PP (1442, 1431, ..., + 10000 * parts for the split-K form).  Host arithmetic only (no GPU needed).
extern "C" int ufv_gemm_choice(int M, int N, int K, int out_f32, int swiglu, int act_none) {
    if (M < 256 || N % BN != 0 || K % BK != 0) return 0;
    return choose_kernel(M, N, K, out_f32 != 0, swiglu != 0, act_none != 0 && splitk_enabled());
}

extern "C" int ufv_gemm_set_splitk(int enable) { return g_splitk.exchange(enable != 0 ? 1 : 0); }

// ---- fused QKV projection + RoPE + KV-cache append (round 5) ------------------------------------------------------------------------------
// Which tile shape the fused kernel would run for S rows, or 0 when the unfused pair (ufv_gemm, ufv_rope_kv_table) is the better or the only choice:
// head_dim 128 (a 256-column tile = two heads), S >= 256, K a multiple of 64, and the cost model (choose_kernel) must not see the fused kernel's best
// shape (192 x 256) more than ~12 us (17 k ticks: what the separate RoPE launch costs at S = 2399) behind the shape AUTO would take.
extern "C" int ufv_gemm_qkv_rope_shape(int S, int Hq, int Hkv, int hd, int K) {
    if (hd != 128 || S < 256 || Hq < 1 || Hkv < 1 || K % BK != 0 || getenv("UFV_NO_FUSED_ROPE") != nullptr) return 0;
    const int N = (Hq + 2 * Hkv) * hd;
    double best = 0, rc = 0;
    int rp = 0;
    (void)choose_kernel(S, N, K, false, false, false, &best, &rc, &rp);
    const double rope_ticks = 17000.0 * (double)S / 2399.0;
    return (rp != 0 && rc <= best + rope_ticks) ? rp : 0;
}

// e4m3 operands: A codes [S, K] with fp32 row scales, W codes with fp32 per-channel scales (ufv_gemm_fp8's operands); the same fused epilogue after the de-quantising scale
extern "C" int ufv_gemm_qkv_rope_fp8(const void* A, int lda, const float* a_scale, const void* W, int ldw, const float* w_scale, const float* bias, void* q_out, int ldq,
                                     void* kv_row0, int ldkv, int S, int Hq, int Hkv, int hd, int K, const float* rope_table, void* stream) {
    UFV_REQUIRE(A && a_scale && W && w_scale && q_out && kv_row0 && rope_table && S >= 256 && hd == 128 && K % 128 == 0,
                "ufv_gemm_qkv_rope_fp8: needs head_dim 128, S >= 256, K %% 128 == 0 (S=%d hd=%d K=%d)", S, hd, K);
    const int N = (Hq + 2 * Hkv) * hd;
    UFV_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0) && (lda % 16 == 0) && (ldw % 16 == 0) && ((uintptr_t)q_out % 16 == 0) && ((uintptr_t)kv_row0 % 16 == 0) &&
                (ldq % 8 == 0) && (ldkv % 8 == 0) && ((uintptr_t)rope_table % 16 == 0) && (!bias || (uintptr_t)bias % 16 == 0) && ((uintptr_t)w_scale % 16 == 0) &&
                ldq >= Hq * hd && ldkv >= 2 * Hkv * hd && (int64_t)S * ldq < (1ll << 30) && (int64_t)S * ldkv < (1ll << 30),
                "ufv_gemm_qkv_rope_fp8: alignment / pitch (S=%d ldq=%d ldkv=%d)", S, ldq, ldkv);
    Epi e;
    e.bias = bias; e.resid = nullptr; e.out = q_out; e.ldr = 0; e.ldc = ldq; e.act = ACT_NONE; e.resid_rows = 0;
    e.scale_m = a_scale; e.scale_n = w_scale; e.dump_f32 = 0; e.ksplit = 0;
    e.rope_tab = rope_table; e.out_kv = kv_row0; e.ldkv = ldkv; e.rope_hq = Hq; e.rope_hkv = Hkv;
    return ufv_launch_pp_rope_fp8(A, W, e, S, N, K, lda, ldw, 1332, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int ufv_gemm_qkv_rope(const void* A, int lda, const void* W, int ldw, const float* bias, void* q_out, int ldq, void* kv_row0, int ldkv, int S,
                                 int Hq, int Hkv, int hd, int K, const float* rope_table, int shape, void* stream) {
    UFV_REQUIRE(A && W && q_out && kv_row0 && rope_table && S > 0 && K > 0, "ufv_gemm_qkv_rope: bad arguments");
    if (shape == 0) shape = ufv_gemm_qkv_rope_shape(S, Hq, Hkv, hd, K);
    UFV_REQUIRE(shape == 1332, "ufv_gemm_qkv_rope: no fused kernel for S=%d heads %d/%d head_dim %d K=%d (ufv_gemm_qkv_rope_shape returned 0: "
                "use ufv_gemm + ufv_rope_kv_table)", S, Hq, Hkv, hd, K);
    UFV_REQUIRE(hd == 128 && S >= 256 && K % BK == 0, "ufv_gemm_qkv_rope: needs head_dim 128, S >= 256, K %% 64 == 0 (S=%d hd=%d K=%d)", S, hd, K);
    const int N = (Hq + 2 * Hkv) * hd;
    UFV_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0) && (lda % 8 == 0) && (ldw % 8 == 0) && ((uintptr_t)q_out % 16 == 0) && ((uintptr_t)kv_row0 % 16 == 0) &&
                (ldq % 8 == 0) && (ldkv % 8 == 0) && ((uintptr_t)rope_table % 16 == 0) && (!bias || (uintptr_t)bias % 16 == 0),
                "ufv_gemm_qkv_rope: operands, outputs and the table must be 16-byte aligned, pitches multiples of 8");
    UFV_REQUIRE(ldq >= Hq * hd && ldkv >= 2 * Hkv * hd && (int64_t)S * ldq < (1ll << 30) && (int64_t)S * ldkv < (1ll << 30),
                "ufv_gemm_qkv_rope: q / kv pitches too small or an output beyond 2^31 bytes (S=%d ldq=%d ldkv=%d)", S, ldq, ldkv);
    Epi e;
    e.bias = bias; e.resid = nullptr; e.out = q_out; e.ldr = 0; e.ldc = ldq; e.act = ACT_NONE; e.resid_rows = 0;
    e.scale_m = nullptr; e.scale_n = nullptr; e.dump_f32 = 0; e.ksplit = 0;
    e.rope_tab = rope_table; e.out_kv = kv_row0; e.ldkv = ldkv; e.rope_hq = Hq; e.rope_hkv = Hkv;
    return ufv_launch_pp_rope(A, W, e, S, N, K, lda, ldw, shape, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int ufv_gemm_timing(int enable) {
    g_timing = enable < 0 ? 0 : enable;
    g_timing_seen = 0;
    return UFV_OK;
}

// waits for the recorded launches, writes up to `cap` durations (ms) and shapes (M, N, K per launch), forgets them; returns how many there were
extern "C" int ufv_gemm_timing_read(float* ms, int32_t* mnk, int cap) {
    std::vector<TimedLaunch>& v = timed_launches();
    const int n = (int)v.size();
    for (int i = 0; i < n; ++i) {
        float t = 0.f;
        (void)hipEventSynchronize(v[i].e);
        (void)hipEventElapsedTime(&t, v[i].s, v[i].e);
        if (i < cap) {
            if (ms) ms[i] = t;
            if (mnk) { mnk[3 * i] = v[i].M; mnk[3 * i + 1] = v[i].N; mnk[3 * i + 2] = v[i].K; }
        }
        (void)hipEventDestroy(v[i].s); (void)hipEventDestroy(v[i].e);
    }
    v.clear();
    return n;
}

extern "C" int ufv_gemm(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int out_f32, int M, int N, int K,
                        const float* bias, int act, const float* resid, int ldr, int resid_rows, int swiglu, int kernel,
                        void* stream) {
    UFV_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0, "ufv_gemm: bad arguments (M=%d N=%d K=%d)", M, N, K);
    UFV_REQUIRE(!(swiglu && (bias || act != ACT_NONE)), "ufv_gemm: swiglu epilogue takes no bias/activation");
    return gemm_entry<false>(A, lda, nullptr, W, ldw, nullptr, C, ldc, out_f32, M, N, K, bias, act, resid, ldr, resid_rows, swiglu, kernel,
                             stream);
}

extern "C" int ufv_gemm_stream_bf16(const void* A, int lda, const void* W, int ldw, void* x, int ldx, int M, int N, int K, const float* bias, int kernel,
                                    void* stream) {
    UFV_REQUIRE(A && W && x && M > 0 && N > 0 && K > 0 && ldx % 4 == 0 && (uintptr_t)x % 16 == 0, "ufv_gemm_stream_bf16: bad arguments (M=%d N=%d K=%d ldx=%d)", M, N, K,
                ldx);
    return gemm_entry<false>(A, lda, nullptr, W, ldw, nullptr, x, ldx, 0, M, N, K, bias, ACT_NONE, reinterpret_cast<const float*>(x), ldx, 0, 0, kernel, stream, 1);
}

// C[M,N] (+)= A[M,K] * W[N,K]^T with the K range split over up to `nsplit` blocks per output tile: for products whose output
// is small and whose K is long (attention backward's dV = P^T dO, dK = dS^T Q: 2399 x 128 outputs over K = 7 x 2399), where
// whole-K tiles would occupy 13 of the 256 CUs.  Partial tiles go to the fp32 workspace [nsplit][M][N] with plain stores
// and are summed in slice order by a second kernel (deterministic; device-scope fp32 atomics measured 4x slower: they are
// served memory-side because the XCDs' L2s are not coherent).
namespace {
template <bool OUT_F32>
__global__ __launch_bounds__(256) void sum_partials_k(const float* __restrict__ part, int nsplit, int M, int N, void* __restrict__ out, int ldc,
                                                      int accumulate) {
    const int64_t total = (int64_t)M * N / 4;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int m = (id * 4) / N, n = (id * 4) % N;
        f32x4 s = *reinterpret_cast<const f32x4*>(part + (size_t)m * N + n);
        for (int k = 1; k < nsplit; ++k) s += *reinterpret_cast<const f32x4*>(part + ((size_t)k * M + m) * N + n);
        if (OUT_F32) {
            float* o = reinterpret_cast<float*>(out) + (size_t)m * ldc + n;
            if (accumulate) s += *reinterpret_cast<const f32x4*>(o);
            *reinterpret_cast<f32x4*>(o) = s;
        } else {
            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(out) + (size_t)m * ldc + n) = bf16x4{(bf16)s[0], (bf16)s[1], (bf16)s[2], (bf16)s[3]};
        }
    }
}
}  // namespace

extern "C" int ufv_gemm_splitk(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int out_f32, int accumulate, int M, int N,
                               int K, int nsplit, void* ws, void* stream) {
    UFV_REQUIRE(A && W && C && ws && M > 0 && N > 0 && K > 0 && nsplit >= 1, "ufv_gemm_splitk: bad arguments (M=%d N=%d K=%d)", M, N, K);
    UFV_REQUIRE(N % 4 == 0 && ldc % 4 == 0 && (uintptr_t)C % 16 == 0 && (uintptr_t)ws % 16 == 0 && !(accumulate && !out_f32),
                "ufv_gemm_splitk: N, ldc multiples of 4, 16-byte aligned C / ws, accumulation only into fp32 (N=%d ldc=%d)", N, ldc);
    const bool ok = ((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0) && (lda % 8 == 0) && (ldw % 8 == 0) && (N % BN == 0) && (K % BK == 0);
    Epi e;
    e.bias = nullptr; e.resid = nullptr; e.out = ws; e.ldr = 0; e.ldc = N; e.act = ACT_NONE; e.resid_rows = 0;
    e.scale_m = nullptr; e.scale_n = nullptr; e.dump_f32 = 0; e.ksplit = 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int splits = 1;
    if (!ok || nsplit == 1) {               // small / unaligned shapes: one ordinary GEMM into slice 0
        const int rc = launch_any<true, false, false>(A, W, e, M, N, K, lda, ldw, UFV_GEMM_AUTO, st);
        if (rc != UFV_OK) return rc;
    } else {
        const int nk = K / BK, per = cdiv(nk, nsplit);
        splits = cdiv(nk, per);                                    // no empty slice
        e.ksplit = splits > 1 ? splits : 0;
        UFV_ONCE_PER_DEVICE(
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_128<true, false, 6, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, Tile<6>::SMEM_BYTES);
        );
        const int tiles = cdiv(M, Tile<6>::BM) * (N / BN);
        hipLaunchKernelGGL((gemm_nt_128<true, false, 6, false>), dim3(tiles, splits), dim3(256), Tile<6>::SMEM_BYTES, st, A, W, e, M, N, K, lda, ldw);
        UFV_CHECK_LAUNCH();
    }
    const int64_t total = (int64_t)M * N / 4;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (out_f32) hipLaunchKernelGGL((sum_partials_k<true>), dim3(grid), dim3(256), 0, st, (const float*)ws, splits, M, N, C, ldc, accumulate);
    else hipLaunchKernelGGL((sum_partials_k<false>), dim3(grid), dim3(256), 0, st, (const float*)ws, splits, M, N, C, ldc, 0);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_gemm_fp8(const void* A, int lda, const float* a_scale, const void* W, int ldw, const float* w_scale, void* C, int ldc,
                            int out_f32, int M, int N, int K, const float* bias, int act, const float* resid, int ldr, int resid_rows,
                            int swiglu, int kernel, void* stream) {
    UFV_REQUIRE(A && W && C && a_scale && w_scale && M > 0 && N > 0 && K > 0, "ufv_gemm_fp8: bad arguments (M=%d N=%d K=%d)", M, N, K);
    UFV_REQUIRE(!(swiglu && (bias || act != ACT_NONE)), "ufv_gemm_fp8: swiglu epilogue takes no bias/activation");
    UFV_REQUIRE(kernel != UFV_GEMM_GENERIC, "ufv_gemm_fp8: no generic kernel for fp8 operands");
    return gemm_entry<true>(A, lda, a_scale, W, ldw, w_scale, C, ldc, out_f32, M, N, K, bias, act, resid, ldr, resid_rows, swiglu, kernel,
                            stream);
}

// ---- W8A8 with MX block scales on the activation side (round 5) -------------------------------------------------------------------------------------
// a_bscale != NULL: A carries one e8m0 scale byte per (row, 32 K-elements) instead of a_scale's fp32 per row, stored as [M / 64][K / 512][64][16] (ld_abs bytes per 64-row block; include/ufv.h ufv_quantize_mx).  out_bscale != NULL: C receives e4m3 codes
// (1 byte per element, ldc in bytes) and out_bscale their block scales [M, ld_obs] -- the input of the next e4m3 GEMM, written by this GEMM's own epilogue
// (256 x 256 tile; swiglu: the 32-column blocks are the permuted ones of epilogue256_swiglu_mx).  w_scale: fp32 per output channel, as in ufv_gemm_fp8.
extern "C" int ufv_gemm_fp8_mx(const void* A, int lda, const float* a_scale, const void* a_bscale, int ld_abs, const void* W, int ldw, const float* w_scale, void* C,
                               int ldc, int out_f32, void* out_bscale, int ld_obs, int M, int N, int K, const float* bias, int act, const void* resid, int ldr,
                               int resid_bf16, int swiglu, void* stream) {
    UFV_REQUIRE(A && W && C && w_scale && M > 0 && N > 0 && K > 0 && ((a_scale != nullptr) != (a_bscale != nullptr)),
                "ufv_gemm_fp8_mx: give exactly one of a_scale (fp32 per row) / a_bscale (e8m0 per row and 32 elements) (M=%d N=%d K=%d)", M, N, K);
    UFV_REQUIRE(a_bscale || out_bscale, "ufv_gemm_fp8_mx: neither a block-scaled input nor a block-scaled output: use ufv_gemm_fp8");
    UFV_REQUIRE(!(a_bscale && out_bscale), "ufv_gemm_fp8_mx: block scales on BOTH sides of one GEMM are not built (register budget of the 256 x 256 tile)");
    UFV_REQUIRE(M >= 256 && N % 128 == 0 && K % 128 == 0 && ((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0) && lda % 16 == 0 && ldw % 16 == 0 &&
                ((uintptr_t)C % 16 == 0) && (!bias || (uintptr_t)bias % 16 == 0) && ((uintptr_t)w_scale % 16 == 0),
                "ufv_gemm_fp8_mx: needs M >= 256, N %% 128 == 0, K %% 128 == 0 and 16-byte aligned operands (M=%d N=%d K=%d)", M, N, K);
    UFV_REQUIRE(!a_bscale || (ld_abs % 1024 == 0 && ld_abs >= 1024 * ((K + 511) / 512) && (uintptr_t)a_bscale % 16 == 0),
                "ufv_gemm_fp8_mx: A block scales [M / 64][K / 512][64][16]: block pitch %% 1024 == 0, >= 1024 ceil(K / 512)");
    UFV_REQUIRE(!(swiglu && (bias || act != ACT_NONE)), "ufv_gemm_fp8_mx: swiglu epilogue takes no bias/activation");
    Epi e;
    UFV_REQUIRE(!(resid_bf16 && out_f32), "ufv_gemm_fp8_mx: a bf16 residual goes with a bf16 output");
    e.bias = bias; e.resid = reinterpret_cast<const float*>(resid); e.out = C; e.ldr = ldr; e.ldc = ldc; e.act = act; e.resid_rows = 0; e.resid_bf16 = resid && resid_bf16 ? 1 : 0;
    e.scale_m = a_scale; e.scale_n = w_scale; e.dump_f32 = 0; e.ksplit = 0;
    e.a_bscale = reinterpret_cast<const unsigned char*>(a_bscale); e.ld_abs = ld_abs;
    e.out_bscale = reinterpret_cast<unsigned char*>(out_bscale); e.ld_obs = ld_obs;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (out_bscale) {
        const int n_out = swiglu ? N / 2 : N;
        UFV_REQUIRE(!out_f32 && !resid && N % 256 == 0 && ldc % 8 == 0 && ldc >= n_out && ld_obs % 1024 == 0 && ld_obs >= 1024 * ((n_out + 511) / 512) && (int64_t)M * ldc < (1ll << 31) &&
                    (int64_t)((M + 63) / 64) * ld_obs < (1ll << 31),
                    "ufv_gemm_fp8_mx: the MX-emitting epilogue needs N %% 256 == 0, no residual, a byte pitch %% 8 == 0 (N=%d ldc=%d)", N, ldc);
        // bench.py's roofline entry for --fp8: the same every-n-th-launch HIP event bracket ufv_gemm puts around the bf16 gate/up GEMM (ufv_gemm_timing)
        TimedLaunch tl{};
        const bool timed = g_timing > 0 && swiglu && (g_timing_seen++ % (unsigned)g_timing) == 0 && hipEventCreate(&tl.s) == hipSuccess && hipEventCreate(&tl.e) == hipSuccess;
        if (timed) (void)hipEventRecord(tl.s, st);
        const int rc = ufv_launch_pp_mx(A, W, e, M, N, K, lda, ldw, false, swiglu != 0, 1442, 2, st);
        if (timed) {
            (void)hipEventRecord(tl.e, st);
            tl.M = M; tl.N = N; tl.K = K;
            timed_launches().push_back(tl);
        }
        return rc;
    }
    UFV_REQUIRE(!swiglu, "ufv_gemm_fp8_mx: SwiGLU with a block-scaled input is not built");
    UFV_REQUIRE(e.ldc % 4 == 0 && e.ldc >= N && (!resid || ((uintptr_t)resid % 16 == 0 && ldr >= N && ldr % (resid_bf16 ? 8 : 4) == 0)), "ufv_gemm_fp8_mx: output / residual pitch >= N, %% 4 == 0 (N=%d ldc=%d)", N, ldc);
    int pick = choose_kernel(M, N, K / 2, out_f32 != 0, false, false);
    // (the block-scaled A operand costs 14 registers: the 256 x 256 and 224 x 256 tiles spill with it -- 124 / 8 bytes of scratch -- and are not offered)
    if (pick == 0 || pick >= 10000 || pick == 1442) pick = N % 192 == 0 || N % 192 == 128 ? 1441 : 1332;
    if (pick == 1432) pick = 1332;
    return ufv_launch_pp_mx(A, W, e, M, N, K, lda, ldw, out_f32 != 0, false, pick, 1, st);
}

extern "C" int ufv_gemv1(const void* a, const float* x, const float* ln_w, float eps, const void* W, int ldw, const float* w_scale,
                         void* C, int out_f32, int N, int K, const float* bias, int act, const float* resid, int swiglu, void* stream) {
    UFV_REQUIRE(W && C && N > 0 && K > 0 && ((a != nullptr) != (x != nullptr)), "ufv_gemv1: give exactly one of a (bf16 row) / x (fp32 row)");
    UFV_REQUIRE(!x || ln_w, "ufv_gemv1: the fp32 row form needs the RMSNorm weight");
    UFV_REQUIRE(K % (w_scale ? 16 : 8) == 0 && ldw % (w_scale ? 16 : 8) == 0 && (size_t)K * 3 <= 64 * 1024 && (!swiglu || N % 32 == 0),
                "ufv_gemv1: K %% 8 (16 for fp8 weights), K <= 21845 (K=%d)", K);
    UFV_REQUIRE(!(swiglu && (bias || act != ACT_NONE)), "ufv_gemv1: swiglu epilogue takes no bias/activation");
    Epi e;
    e.bias = bias; e.resid = resid; e.out = C; e.ldr = 0; e.ldc = 0; e.act = act; e.resid_rows = 0; e.scale_m = nullptr; e.scale_n = nullptr;
    e.dump_f32 = 0; e.ksplit = 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int n_out = swiglu ? N / 2 : N;
    const bool narrow = n_out <= 8192 && !w_scale;      // fp8 rows are half as long and every block re-quantises the input row: 4 outputs per wave stays faster there
    dim3 grid(cdiv(n_out, narrow ? 4 : 16)), blk(256);
    const size_t sm = (size_t)K * (w_scale ? 3 : 2);
    const bf16* ab = reinterpret_cast<const bf16*>(a);
#define G1(F_, S_, R_, Q_) do { if (narrow) hipLaunchKernelGGL((gemv1_nt<F_, S_, R_, Q_, 1>), grid, blk, sm, st, ab, x, ln_w, eps, W, w_scale, e, N, K, ldw); \
                                else hipLaunchKernelGGL((gemv1_nt<F_, S_, R_, Q_, 4>), grid, blk, sm, st, ab, x, ln_w, eps, W, w_scale, e, N, K, ldw); } while (0)
#define G1Q(F_, S_, R_) do { if (w_scale) G1(F_, S_, R_, true); else G1(F_, S_, R_, false); } while (0)
    if (x) {
        if (out_f32) { if (swiglu) G1Q(true, true, true); else G1Q(true, false, true); }
        else { if (swiglu) G1Q(false, true, true); else G1Q(false, false, true); }
    } else {
        if (out_f32) { if (swiglu) G1Q(true, true, false); else G1Q(true, false, false); }
        else { if (swiglu) G1Q(false, true, false); else G1Q(false, false, false); }
    }
#undef G1Q
#undef G1
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}
