// bf16 GEMM, 256x256x64 block tile, 8 waves, ping-pong schedule ("8 phases" per two K-tiles) for gfx950.
//
//   C[M,N] = epilogue(A[M,K] * W[N,K]^T), same epilogues and operand conventions as gemm.hip.
//
// Structure (one block per CU, 128 KiB LDS = 2 K-tile buffers x {A0,A1,B0,B1} half-tiles of 128 rows x 64 k):
//  * 8 waves = 2 (rows) x 4 (cols); wave (wr,wc) owns 64 rows of EACH A half and 32 columns of EACH B half, so a
//    wave's 128x64 output splits into 4 quadrants (A half x B half) and every LDS half-tile is read in few phases:
//        phase 0: read A0,B0 -> quadrant (0,0)      phase 1: read B1 -> (0,1)
//        phase 2: read A1    -> quadrant (1,1)      phase 3: read B0 -> (1,0)
//  * each phase = [ds_read fragments; issue ONE half-tile of LDS-DMA prefetch] s_barrier [16 MFMA] s_barrier.
//    Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA segment while its
//    partner is in its load segment.
//  * prefetch runs through the whole loop with a COUNTED s_waitcnt vmcnt(4) once per K-tile (never 0 inside the
//    loop): staging order A1[t+1], B0[t+1], A0[t+2], B1[t+2] in phases 0..3 of K-tile t.
//      WAR: a half-tile buffer is re-staged exactly 2 phases after its last ds_read (safe for the lagging group);
//      RAW: the wait in phase 3 retires everything up to B0[t+1]; first read is in the next phase, two barriers later.
//  * persistent: one block per CU walks tiles b, b+G, ...; the next tile's prologue DMA is issued before the
//    current tile's epilogue stores (its bias is fetched first so no ordinary load queues behind the DMA).
//  * operands: 128-byte rows, 16-byte chunks XOR-swizzled by (row & 7) through the LDS-DMA SOURCE address.
#include "common.h"
#include "../../include/ufv.h"
#include "gemm_epi.h"

namespace {

constexpr int SMEM256 = 131072;

__device__ __forceinline__ i32x8 cat8(bf16x8 lo, bf16x8 hi) {      // two 16-byte LDS chunks -> the 32-byte fp8 operand
    const i32x4 a = __builtin_bit_cast(i32x4, lo), b = __builtin_bit_cast(i32x4, hi);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// acc[nt][mt][j] = C[m0 + (mt>>2)*128 + wr*64 + (mt&3)*16 + frow][n0 + (nt>>1)*128 + wc*32 + (nt&1)*16 + fq*4 + j]
template <bool OUT_F32, bool SWIGLU, int ACT, bool DUMP = false>
__device__ __forceinline__ void epilogue256(const f32x4 (&acc)[4][8], const Epi& e, int M, int N, int m0, int n0, int wr, int wc,
                                            int frow, int fq, f32x4 bias0, f32x4 bias1, f32x4 bias2, f32x4 bias3) {
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int m = m0 + (mt >> 2) * 128 + wr * 64 + (mt & 3) * 16 + frow;
        if (m < M) {
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
                const int nb = n0 + nh * 128;
                if (nb < N) {
                    if (SWIGLU) {
                        const int n = ((nb + wc * 32) >> 1) + fq * 4;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float gte = acc[2 * nh][mt][j], up = acc[2 * nh + 1][mt][j];
                            v[j] = gte / (1.0f + __expf(-gte)) * up;
                        }
                        epi_store4b<OUT_F32, ACT_NONE>(e, m, n, v[0], v[1], v[2], v[3], f32x4{0, 0, 0, 0});
                    } else {
                        const int n = nb + wc * 32 + fq * 4;
                        epi_store4b<OUT_F32, ACT, DUMP>(e, m, n, acc[2 * nh][mt][0], acc[2 * nh][mt][1], acc[2 * nh][mt][2],
                                                        acc[2 * nh][mt][3], nh ? bias2 : bias0);
                        epi_store4b<OUT_F32, ACT, DUMP>(e, m, n + 16, acc[2 * nh + 1][mt][0], acc[2 * nh + 1][mt][1],
                                                        acc[2 * nh + 1][mt][2], acc[2 * nh + 1][mt][3], nh ? bias3 : bias1);
                    }
                }
            }
        }
    }
}

// FP8: e4m3 operands, K-tile = 128 elements (the same 128-byte LDS rows and DMA pattern), 8 x v_mfma_f32_16x16x128_f8f6f4
// per phase instead of 16 x 16x16x32_bf16; accumulators are scaled by scale_m[row] * scale_n[col] before the epilogue.
//
// Stream-K (sk_ws != nullptr): instead of whole tiles, block `pos` takes the contiguous range [lo, hi) of the
// tiles x K-tiles iteration space (tile-major), so every CU does the same number of K-tile iterations whatever the tile
// count.  A block's range is: [tail of a tile] [whole tiles ...] [head of a tile].  A tail / middle part (k0 > 0) dumps its
// raw accumulators to workspace slot `pos` and raises flag[pos] = epoch; the block holding a tile's head (k0 == 0) is the
// tile's owner: it adds the slots of the following blocks in order (deterministic) and runs the epilogue.  Owners hold the
// head as their LAST item and the other parts are their blocks' FIRST items, so an owner practically never waits.
// All blocks must be co-resident (grid <= number of CUs, one block per CU) -- guaranteed when nothing else runs on the GPU.
struct StreamK {
    float* ws;          // [grid][32][512] f32x4 accumulator dumps (256 KiB per block)
    int* flags;         // [grid]
    int epoch;          // value that marks "slot written during THIS launch"
    int gm;             // row-tiles per group of the tile order (concurrent tiles of a group share A / W panels in L2)
};

template <bool OUT_F32, bool SWIGLU, bool FP8, bool SKT>
__global__ __launch_bounds__(512, 2) void gemm_nt_256(const void* __restrict__ Av, const void* __restrict__ Wv, Epi e, int M,
                                                       int N, int K, int lda, int ldw, StreamK sk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ES = FP8 ? 1 : 2;
    const char* A = reinterpret_cast<const char*>(Av);
    const char* W = reinterpret_cast<const char*>(Wv);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
    const int nwg = tiles_m * tiles_n;
    const int nk = K * ES / 128;
    const int G = gridDim.x;                       // persistent: block b walks tiles b, b+G, ...
    // tile sequence position -> (tm, tn): within a round of G tiles give each XCD (launch id % 8) a contiguous run,
    // and order the sequence in groups of 8 row-tiles so that concurrent tiles share A/W panels in L2.
    auto tile_coords = [&](int id, int& m0_, int& n0_) {
        const int GM = sk.gm;
        const int gsz = GM * tiles_n, g = id / gsz, first_m = g * GM;
        const int gm = min(tiles_m - first_m, GM);
        m0_ = (first_m + (id % gsz) % gm) * 256;
        n0_ = ((id % gsz) / gm) * 256;
    };
    constexpr bool SK = SKT;
    // stream-K: logical position of this block (each XCD = launch id % 8 gets a contiguous run of positions) and its range
    const int pos = (G % 8 == 0) ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const long long total = (long long)nwg * nk;
    const int per = (int)(total / G), rem = (int)(total % G);
    auto range_lo = [&](int p) -> long long { return (long long)p * per + min(p, rem); };
    long long cur = SK ? range_lo(pos) : 0;
    const long long hi = SK ? range_lo(pos + 1) : 0;
    int round = 0;
    // next work item: tile (m0_, n0_) and its K-tile range [k0_, k1_)
    auto next_item = [&](int& m0_, int& n0_, int& k0_, int& k1_) -> bool {
        if (SK) {
            if (cur >= hi) return false;
            const int tile = (int)(cur / nk);
            k0_ = (int)(cur - (long long)tile * nk);
            k1_ = (int)min((long long)nk, k0_ + (hi - cur));
            cur += k1_ - k0_;
            tile_coords(tile, m0_, n0_);
            return true;
        }
        const int base = round * G;
        const int cnt = min(G, nwg - base);          // tiles in this round
        const int bid = blockIdx.x;
        ++round;
        if (bid >= cnt) return false;
        const int q = cnt >> 3, r = cnt & 7, x = bid & 7;
        tile_coords(base + (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3), m0_, n0_);
        k0_ = 0; k1_ = nk;
        return true;
    };

    // ---- LDS-DMA sources: half-tile `which` (0=A0 1=A1 2=B0 3=B1), two 8-row pieces per wave
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
    const char* src[4][2];
    auto set_src = [&](int m0_, int n0_) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = h * 128 + (wave * 2 + i) * 8 + lrow;
                src[h][i] = A + (size_t)min(m0_ + r, M - 1) * lda * ES + lchunk * 16;
                src[2 + h][i] = W + (size_t)min(n0_ + r, N - 1) * ldw * ES + lchunk * 16;
            }
    };
    int kbeg = 0, kend = 0;                        // K-tile range of the item being loaded
    auto stage = [&](int d, int which, int kt) {
        if (kt < kend) {
            char* dst = smem + d * 65536 + which * 16384 + wave * 2048;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src[which][0] + kt * 128), LDS_PTR(dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(src[which][1] + kt * 128), LDS_PTR(dst + 1024), 16, 0, 0);
        }
    };
    auto prologue_loads = [&]() {      // first K-tile of the item complete + A0/B1 of its second K-tile
        stage(0, 0, kbeg); stage(0, 2, kbeg); stage(0, 3, kbeg); stage(0, 1, kbeg);
        stage(1, 0, kbeg + 1); stage(1, 3, kbeg + 1);
    };

    const int frow = lane & 15, fq = lane >> 4, fx = lane & 7;
    const int a_row_off = (wr * 64 + frow) * 128;     // + (mt&3)*2048 inside the half
    const int b_row_off = (wc * 32 + frow) * 128;     // + (nt&1)*2048 inside the half
    // bf16: k-step kk reads chunk 4*kk + fq;  fp8: the lane's 32 bytes are chunks 2*fq and 2*fq + 1
    const int coff0 = ((FP8 ? 2 * fq : fq) ^ fx) << 4, coff1 = ((FP8 ? 2 * fq + 1 : 4 + fq) ^ fx) << 4;

    int m0, n0, k0, k1;
    bool have = next_item(m0, n0, k0, k1);
    if (have) { set_src(m0, n0); kbeg = k0; kend = k1; prologue_loads(); }
    while (have) {
    const int len = k1 - k0;
    f32x4 acc[4][8];   // [nt][mt]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 afr[4][2], bfr[2][2];
    auto read_a = [&](const char* half) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            afr[i][0] = *reinterpret_cast<const bf16x8*>(half + a_row_off + i * 2048 + coff0);
            afr[i][1] = *reinterpret_cast<const bf16x8*>(half + a_row_off + i * 2048 + coff1);
        }
    };
    auto read_b = [&](const char* half) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bfr[i][0] = *reinterpret_cast<const bf16x8*>(half + b_row_off + i * 2048 + coff0);
            bfr[i][1] = *reinterpret_cast<const bf16x8*>(half + b_row_off + i * 2048 + coff1);
        }
    };
#define UFV_SYNC_THEN_MMA(NTB, MTB)                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    __builtin_amdgcn_s_barrier();                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    __builtin_amdgcn_s_setprio(1);                                                                           \
    if constexpr (FP8) {                                                                                     \
        _Pragma("unroll") for (int n_ = 0; n_ < 2; ++n_)                                                     \
            _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_)                                                 \
                acc[NTB + n_][MTB + m_] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                  \
                    cat8(bfr[n_][0], bfr[n_][1]), cat8(afr[m_][0], afr[m_][1]), acc[NTB + n_][MTB + m_], 0, 0, 0, 0, 0, 0); \
    } else {                                                                                                 \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                         \
        _Pragma("unroll") for (int n_ = 0; n_ < 2; ++n_)                                                     \
            _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_)                                                 \
                acc[NTB + n_][MTB + m_] =                                                                    \
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[n_][kk], afr[m_][kk], acc[NTB + n_][MTB + m_], 0, 0, 0); \
    }                                                                                                        \
    __builtin_amdgcn_s_setprio(0);                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    __builtin_amdgcn_s_barrier();

    if (len > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();      // waves 4-7 run one barrier behind

    for (int tt = 0; tt < len; ++tt) {
        const int t = k0 + tt, d = tt & 1;
        const char* buf = smem + d * 65536;
        // phase 0: A0, B0 -> quadrant (0,0); prefetch A1[t+1]
        read_b(buf + 32768);
        read_a(buf);
        stage(d ^ 1, 1, t + 1);
        UFV_SYNC_THEN_MMA(0, 0)
        // phase 1: B1 -> quadrant (0,1); prefetch B0[t+1]
        read_b(buf + 49152);
        stage(d ^ 1, 2, t + 1);
        UFV_SYNC_THEN_MMA(2, 0)
        // phase 2: A1 -> quadrant (1,1); prefetch A0[t+2]
        read_a(buf + 16384);
        stage(d, 0, t + 2);
        UFV_SYNC_THEN_MMA(2, 4)
        // phase 3: B0 -> quadrant (1,0); prefetch B1[t+2]; retire K-tile t+1
        read_b(buf + 32768);
        stage(d, 3, t + 2);
        if (tt + 2 < len) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        UFV_SYNC_THEN_MMA(0, 4)
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();      // balance the stagger barrier
#undef UFV_SYNC_THEN_MMA


    // ---- item seam: every wave has finished its LDS reads (final barrier above) -> start the NEXT item's
    //      LDS-DMA prologue now so that it lands under this item's epilogue stores.  The bias / scales of THIS tile are
    //      fetched first so that no ordinary load has to wait behind the DMA queue.
    const bool part_tail = SK && k0 > 0;                 // not the tile's owner: dump the accumulators
    const bool part_head = SK && k0 == 0 && k1 < nk;     // owner of a split tile: add the other parts first
    if constexpr (FP8) {     // de-quantise in place: acc *= scale_m[row] * scale_n[col]; per element, so it distributes over
        f32x4 sn[4];         // the stream-K partial sums (every part scales its own accumulators)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            sn[nt] = *reinterpret_cast<const f32x4*>(e.scale_n + min(n0 + (nt >> 1) * 128, N - 128) + wc * 32 + (nt & 1) * 16 + fq * 4);
        float sm[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) sm[mt] = e.scale_m[min(m0 + (mt >> 2) * 128 + wr * 64 + (mt & 3) * 16 + frow, M - 1)];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt][mt] *= sn[nt] * sm[mt];
    }
    f32x4 bias0 = {0, 0, 0, 0}, bias1 = bias0, bias2 = bias0, bias3 = bias0;
    if (!part_tail && !SWIGLU && e.bias) {
        const int nA = min(n0, N - 128) + wc * 32 + fq * 4, nB = min(n0 + 128, N - 128) + wc * 32 + fq * 4;
        bias0 = *reinterpret_cast<const f32x4*>(e.bias + nA);
        bias1 = *reinterpret_cast<const f32x4*>(e.bias + nA + 16);
        bias2 = *reinterpret_cast<const f32x4*>(e.bias + nB);
        bias3 = *reinterpret_cast<const f32x4*>(e.bias + nB + 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int cm0 = m0, cn0 = n0;
    have = next_item(m0, n0, k0, k1);
    if (have) { set_src(m0, n0); kbeg = k0; kend = k1; prologue_loads(); }
    if (part_head) {
        // this block's range ended inside the tile: the following blocks hold the rest, in order
        int covered = len;                          // K-tiles of the tile accounted for so far (this block's k0 was 0)
        for (int q = pos + 1; covered < nk; ++q) {
            if (tid == 0)
                while (__hip_atomic_load(sk.flags + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk.epoch) __builtin_amdgcn_s_sleep(2);
            __syncthreads();
            const float* img = sk.ws + (size_t)q * 65536;          // 256x256 fp32 tile image written by block q's epilogue
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const float* rowp = img + ((mt >> 2) * 128 + wr * 64 + (mt & 3) * 16 + frow) * 256 + wc * 32 + fq * 4;
                f32x4 p0, p1, p2, p3;        // system-scope loads: bypass this XCD's (non-coherent) L2
                asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\t"
                             "global_load_dwordx4 %1, %4, off offset:64 sc0 sc1\n\t"
                             "global_load_dwordx4 %2, %4, off offset:512 sc0 sc1\n\t"
                             "global_load_dwordx4 %3, %4, off offset:576 sc0 sc1\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(rowp) : "memory");
                acc[0][mt] += p0; acc[1][mt] += p1; acc[2][mt] += p2; acc[3][mt] += p3;
            }
            covered += (int)min((long long)(nk - covered), range_lo(q + 1) - range_lo(q));
        }
    }
    // ---- epilogue (activation resolved once per tile so the body unrolls with acc in registers).  A stream-K tail / middle
    //      part goes through the same code with the output redirected to its fp32 workspace slot (a 256x256 tile image).
    if constexpr (SK) {
        Epi pe = e;
        int eM = M, eN = N, em0 = cm0, en0 = cn0;
        if (part_tail) {
            pe.out = reinterpret_cast<char*>(sk.ws) + (size_t)pos * 262144;
            pe.ldc = 256; pe.bias = nullptr; pe.resid = nullptr; pe.act = ACT_NONE; pe.dump_f32 = 1;
            eM = 256; eN = 256; em0 = 0; en0 = 0;
        }
        UFV_ACT_SWITCH(pe.act, (epilogue256<OUT_F32, SWIGLU, ACT_, true>(acc, pe, eM, eN, em0, en0, wr, wc, frow, fq, bias0, bias1, bias2, bias3)))
        if (part_tail) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave's write-through stores are acknowledged ...
            __syncthreads();                                        // ... before one thread publishes the slot
            if (tid == 0) __hip_atomic_store(sk.flags + pos, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        UFV_ACT_SWITCH(e.act, (epilogue256<OUT_F32, SWIGLU, ACT_>(acc, e, M, N, cm0, cn0, wr, wc, frow, fq, bias0, bias1, bias2, bias3)))
    }
    }   // persistent tile loop
}

}  // namespace

// stream-K workspace: one 256 KiB accumulator slot + one flag per block, allocated on first use and kept (grow-only).
// One GEMM at a time may use it (launches are ordered on the caller's stream).
static int streamk_state(int grid, StreamK* out) {
    static float* ws = nullptr;
    static int* flags = nullptr;
    static int cap = 0, epoch = 0;
    if (grid > cap) {
        if (ws) (void)hipFree(ws);
        if (flags) (void)hipFree(flags);
        ws = nullptr; flags = nullptr; cap = 0;
        if (hipMalloc(&ws, (size_t)grid * 262144) != hipSuccess || hipMalloc(&flags, (size_t)grid * sizeof(int)) != hipSuccess) {
            ufv_set_error("ufv_gemm: could not allocate the stream-K workspace (%d x 256 KiB)", grid);
            return UFV_EHIP;
        }
        if (hipMemset(flags, 0, (size_t)grid * sizeof(int)) != hipSuccess) return UFV_EHIP;
        cap = grid; epoch = 0;
    }
    out->ws = ws; out->flags = flags; out->epoch = ++epoch;
    return UFV_OK;
}

template <bool F, bool S, bool Q>
static int launch256_t(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, bool streamk, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_256<F, S, Q, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  SMEM256);
        if constexpr (!S)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_256<F, S, Q, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      SMEM256);
        attr_set = true;
    }
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    const int tiles = cdiv(M, 256) * cdiv(N, 256);
    const int nk = K / (Q ? 128 : 64);
    // row-tiles per tile-order group: all of them when there are few (M = 2399 -> 10 row tiles: a ragged second group of 2 rows
    // made its XCDs fetch 16 W panels per round instead of 4; 1203 -> 1244 TF/s on gate/up), else ~8 in equal groups
    const int tiles_m = cdiv(M, 256);
    const int gm = tiles_m <= 16 ? tiles_m : cdiv(tiles_m, cdiv(tiles_m, 8));
    StreamK sk = {nullptr, nullptr, 0, gm};
    int grid = tiles < n_cu ? tiles : n_cu;          // persistent: one block per CU walks the tiles
    if constexpr (!S) {
        if (streamk && (long long)tiles * nk >= n_cu) {
            grid = n_cu;
            const int rc = streamk_state(grid, &sk);
            if (rc != UFV_OK) return rc;
            hipLaunchKernelGGL((gemm_nt_256<F, S, Q, true>), dim3(grid), dim3(512), SMEM256, st, A, W, e, M, N, K, lda, ldw, sk);
            UFV_CHECK_LAUNCH();
            return UFV_OK;
        }
    } else if (streamk) {
        ufv_set_error("ufv_gemm: the stream-K split is not built for the SwiGLU epilogue (its tile counts are large anyway)");
        return UFV_EUNSUPPORTED;
    }
    hipLaunchKernelGGL((gemm_nt_256<F, S, Q, false>), dim3(grid), dim3(512), SMEM256, st, A, W, e, M, N, K, lda, ldw, sk);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

template <bool Q>
static int launch256_q(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, bool out_f32, bool swiglu,
                       bool streamk, hipStream_t st) {
    if (out_f32) return swiglu ? launch256_t<true, true, Q>(A, W, e, M, N, K, lda, ldw, streamk, st) : launch256_t<true, false, Q>(A, W, e, M, N, K, lda, ldw, streamk, st);
    return swiglu ? launch256_t<false, true, Q>(A, W, e, M, N, K, lda, ldw, streamk, st) : launch256_t<false, false, Q>(A, W, e, M, N, K, lda, ldw, streamk, st);
}

int ufv_launch_gemm256(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, bool out_f32,
                       bool swiglu, bool fp8, bool streamk, hipStream_t st) {
    return fp8 ? launch256_q<true>(A, W, e, M, N, K, lda, ldw, out_f32, swiglu, streamk, st)
               : launch256_q<false>(A, W, e, M, N, K, lda, ldw, out_f32, swiglu, streamk, st);
}
