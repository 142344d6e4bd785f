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
#pragma once
#include "common.h"
#include "../../include/ufv.h"
#include "gemm_epi.h"
#include "gemm_state.h"
#include <type_traits>

#ifndef UFV_GSTAMP
#define UFV_GSTAMP(i)      /* lab builds (tools/lab/gemm_lab.hip) record s_memtime here */
#define UFV_GSTAMP_DECL
#define UFV_GSTAMP_FLUSH
#endif
#ifndef UFV_TSTAMP         /* lab builds (tools/lab/gemm_tile_lab.hip): a per-TILE timeline (K loop start, first K-tiles, K loop end, seam, epilogue end) */
#define UFV_TSTAMP(slot)
#define UFV_TSTAMP_K(tt)
#define UFV_TSTAMP_DECL
#define UFV_TSTAMP_NEXT
#endif

namespace {

constexpr int SMEM256 = 131072;

__device__ __forceinline__ i32x8 cat8(bf16x8 lo, bf16x8 hi) {      // two 16-byte LDS chunks -> the 32-byte fp8 operand
    const i32x4 a = __builtin_bit_cast(i32x4, lo), b = __builtin_bit_cast(i32x4, hi);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Tile shapes.  The A operand is staged as two half-tiles of 32*MA0 and 32*MA1 rows (wave row `wr` owns 16*MAh consecutive rows
// of each), the B operand as halves of 128 and 64*NB1 columns (wave column `wc` owns 32 and 16*NB1 of them), so the block tile
// is (32*(MA0+MA1)) x (128 + 64*NB1): <4,4,2> = 256x256 (the original), <3,2,2> = 160x256, <4,3,1> = 224x192, ...  The phase
// schedule, the LDS slots (16 KiB per half) and the DMA pattern are the same for every shape; a 96-row half is staged as 128 rows
// (the extra 32 land in the unused part of its slot).
template <int MA0, int MA1, int NB1> struct PP {
    static constexpr int MT = MA0 + MA1, NT = 2 + NB1;
    static constexpr int BM = 32 * MT, BN = 64 * NT;
    static constexpr int LA0 = MA0 > 2 ? 2 : 1, LA1 = MA1 > 2 ? 2 : 1, LB0 = 2, LB1 = NB1;     // DMA instructions per wave per half-tile
};


// Residual epilogue (o_proj / out_proj / fc2 / down / patch-embed: out = acc + bias + resid, no activation), rows pipelined by hand.
// Written as plain C++ the residual load of every 4-column group is followed by `s_waitcnt vmcnt(0)` and its store (the output may alias
// the residual -- it IS the residual stream, updated in place -- so the compiler keeps every load behind the previous store): 32 dependent
// memory round trips per wave and tile, ~16 us of a 35 us out_proj tile.  Here the residual rows of accumulator row mt + 1 are requested
// before row mt is stored, the loads are inline asm (no compiler-inserted waits) and the waits are counted: one round trip is exposed per
// tile, the rest run under each other.  Same arithmetic, same element order: bit-identical output.
// ISA markers (tests/test_isa_epilogue_stores.py): every epilogue form brackets its body with two comment lines in the listing.  The relaxed item-seam waits of
// the K loop (NST / NSTW / NSTS below) are exact only if the epilogue that ran last issued EXACTLY that many vector-memory stores; the CPU test compiles the kernels
// to gfx950 assembly, walks every path between a BEGIN and its END and counts the vector-memory instructions on it.  ("memory": nothing that touches memory moves
// across a marker; the markers emit no instruction.)
#ifndef UFV_MX_LAB
#define UFV_MX_LAB 0           /* lab: 1 = the MX-A kernels issue the UNSCALED MFMA (wrong sums): what the scale operand itself costs */
#endif
#ifndef UFV_RESID_SC
#define UFV_RESID_SC ""        /* lab (tools/lab/build_variant_lib.sh): " sc0 sc1" = write-through stores of the fp32 stream; measured free (LABNOTES round 5) */
#endif
#define UFV_EPI_MARK(text) asm volatile("; UFV_EPI_" text ::: "memory")

// RB: the residual is bf16 (Epi::resid_bf16: the stream of a bf16 module, updated in place as bf16 -- the reference's own `residual + hidden_states` on bf16
// tensors); 8-byte loads, same number of load and store instructions as the fp32 form (the counted waits do not change).
template <bool OUT_F32, int MA0, int MA1, int NB1, bool RB = false>
__device__ __forceinline__ void epilogue256_resid(const f32x4 (&acc)[2 + NB1][MA0 + MA1], const Epi& e, int M, int N, int m0, int n0, int wr, int wc,
                                                  int frow, int fq, const f32x4 (&bias)[2 + NB1]) {
    constexpr int MT = MA0 + MA1, NT = 2 + NB1;
    int ncol[NT];
    bool nok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        ncol[nt] = n0 + (nt < 2 ? wc * 32 + nt * 16 : 128 + wc * 16 * NB1 + (nt - 2) * 16) + fq * 4;
        nok[nt] = n0 + (nt < 2 ? 0 : 128) < N;
        ncol[nt] = nok[nt] ? ncol[nt] : 0;            // a half-tile past N: loads a valid address, stores nothing
    }
    auto row_of = [&](int mt) { return m0 + (mt < MA0 ? wr * 16 * MA0 + mt * 16 : 32 * MA0 + wr * 16 * MA1 + (mt - MA0) * 16) + frow; };
    UFV_EPI_MARK("BEGIN resid");
    typedef typename std::conditional<RB, u32x2, f32x4>::type rvec;          // what one residual load returns: 4 bf16 or 4 floats
    rvec r[2][NT];
    auto request = [&](int mt, rvec (&dst)[NT]) {
        const int m = min(row_of(mt), M - 1);
        const int mr = e.resid_rows > 0 ? m % e.resid_rows : m;
        if constexpr (RB) {
            const bf16* rp = reinterpret_cast<const bf16*>(e.resid) + (size_t)mr * e.ldr;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst[nt]) : "v"(rp + ncol[nt]) : "memory");
        } else {
            const float* rp = e.resid + (size_t)mr * e.ldr;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[nt]) : "v"(rp + ncol[nt]) : "memory");
        }
    };
    request(0, r[0]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        rvec (&cur)[NT] = r[mt & 1];
        if (mt + 1 < MT) request(mt + 1, r[(mt + 1) & 1]);
        // in flight behind row mt's loads: the stores of row mt - 1 (NT, when there was one) and the loads of row mt + 1 (NT, when there is one)
        if (mt == 0) {
            if constexpr (NT == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3])::"memory");
            else asm volatile("s_waitcnt vmcnt(3)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2])::"memory");
        } else if (mt + 1 < MT) {
            if constexpr (NT == 4) asm volatile("s_waitcnt vmcnt(8)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3])::"memory");
            else asm volatile("s_waitcnt vmcnt(6)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2])::"memory");
        } else {                                       // last row: only the stores of the row before it are younger
            if constexpr (NT == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3])::"memory");
            else asm volatile("s_waitcnt vmcnt(3)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2])::"memory");
        }
        const int m = row_of(mt);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // the element order of epi_store4b: (acc + bias) then + resid
            f32x4 v = acc[nt][mt] + bias[nt];
            if constexpr (RB) {
                const u32x2 c = cur[nt];
                const f32x4 rf = {__builtin_bit_cast(float, c[0] << 16), __builtin_bit_cast(float, c[0] & 0xffff0000u), __builtin_bit_cast(float, c[1] << 16),
                                  __builtin_bit_cast(float, c[1] & 0xffff0000u)};
                v += rf;
            } else {
                v += cur[nt];
            }
            // stores are ALWAYS issued (the wait counts above rely on it); rows / half-tiles past the edge go to a clamped address with no lane enabled
            const bool ok = m < M && nok[nt];
            if constexpr (OUT_F32) {
                float* pp = reinterpret_cast<float*>(e.out) + (size_t)min(m, M - 1) * e.ldc + ncol[nt];
                asm volatile("s_mov_b64 s[2:3], exec\n\ts_and_b64 exec, exec, %2\n\tglobal_store_dwordx4 %0, %1, off" UFV_RESID_SC "\n\ts_mov_b64 exec, s[2:3]\n\ts_nop 1"
                             ::"v"(pp), "v"(v), "s"(__builtin_amdgcn_ballot_w64(ok)) : "memory", "s2", "s3", "scc");
            } else {
                bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                bf16* pp = reinterpret_cast<bf16*>(e.out) + (size_t)min(m, M - 1) * e.ldc + ncol[nt];
                asm volatile("s_mov_b64 s[2:3], exec\n\ts_and_b64 exec, exec, %2\n\tglobal_store_dwordx2 %0, %1, off\n\ts_mov_b64 exec, s[2:3]\n\ts_nop 1"
                             ::"v"(pp), "v"(o), "s"(__builtin_amdgcn_ballot_w64(ok)) : "memory", "s2", "s3", "scc");
            }
        }
    }
    UFV_EPI_MARK("END resid");
}


// bf16 epilogue with 16-byte stores (no residual): the two 16-column n-tiles of a wave's 32-column run are regrouped across the four 16-lane rows
// with v_permlane32_swap + v_permlane16_swap (two per register pair), after which lane row r of accumulator row `frow` holds columns 8r .. 8r + 7
// of the run: one global_store_dwordx4 per (accumulator row, run) -- 16 store instructions per wave and tile instead of 32, 64-byte segments
// instead of 32-byte ones.  The tile round's store burst is issue-bound: same values, same rounding, bit-identical output.
template <int ACT, int MA0, int MA1, int NB1>
__device__ __forceinline__ void epilogue256_wide(const f32x4 (&acc)[2 + NB1][MA0 + MA1], const Epi& e, int M, int N, int m0, int n0, int wr, int wc,
                                                 int frow, int fq, const f32x4 (&bias)[2 + NB1]) {
    constexpr int MT = MA0 + MA1;
    // Round 4: the stores go through ONE buffer descriptor over the whole output (num_records = M rows): a row past M is dropped by the hardware, so there is no
    // exec masking, and an address is a 32-bit per-lane offset + an immediate -- the flat form cost five 64-bit vector operations per store (the epilogue is
    // VALU-bound: ~35 vector instructions per store, two waves per SIMD, 7 - 9 k ticks of a 61 k-tick K = 1152 tile).  The caller guarantees M * ldc * 2 < 2^31.
    // (sizes and offsets in UNSIGNED arithmetic: a row past M of an output close to 2^31 bytes lands above 2^31 -- still beyond num_records, dropped)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(e.out, 0, (int)((unsigned)M * (unsigned)e.ldc * 2u), 0x20000);
    const unsigned step = 32u * (unsigned)e.ldc;                                                   // bytes between accumulator rows mt and mt + 1 (16 matrix rows)
    const unsigned col0 = n0 + wc * 32 + fq * 8;
    const unsigned off_h0 = ((unsigned)(m0 + wr * 16 * MA0 + frow) * (unsigned)e.ldc + col0) * 2u;              // this lane's row of accumulator row 0 (A0 half), run 0
    const unsigned off_h1 = ((unsigned)(m0 + 32 * MA0 + wr * 16 * MA1 + frow) * (unsigned)e.ldc + col0) * 2u;   // ... of accumulator row MA0 (A1 half)
    auto pack2 = [&](float a, float b) -> unsigned {
        const bf16x2 p = {(bf16)a, (bf16)b};
        return __builtin_bit_cast(unsigned, p);
    };
    // Round 5: EVERY store instruction is issued on EVERY path -- a run past N (the tile cut at N % BN == 128) is sent to an offset with bit 31 set, beyond any
    // descriptor this epilogue accepts (M * ldc * 2 < 2^31), where the hardware drops it like a row past M.  No branch between the stores, and the count the relaxed
    // seam waits rest on (NSTW) holds on edge tiles too; tests/test_isa_epilogue_stores.py walks the listing.
    const unsigned drop1 = n0 + 128 >= N ? 0x80000000u : 0u;
    UFV_EPI_MARK("BEGIN wide");
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const unsigned off = mt < MA0 ? off_h0 + mt * step : off_h1 + (mt - MA0) * step;
#pragma unroll
        for (int run = 0; run < 2; ++run) {
            const unsigned drop = run ? drop1 : 0u;
            if (run == 1 && NB1 == 1) {               // a 16-column second half has no partner n-tile: the 8-byte store (columns 128 + 16 wc + 4 fq ..)
                const f32x4 v = acc[2][mt] + bias[2];
                const bf16x4 o = {(bf16)act_apply_t<ACT>(v[0]), (bf16)act_apply_t<ACT>(v[1]), (bf16)act_apply_t<ACT>(v[2]), (bf16)act_apply_t<ACT>(v[3])};
                typedef __attribute__((ext_vector_type(2))) int i32x2;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, o), rs, (int)((off + (unsigned)(128 - wc * 16 - fq * 4) * 2u) | drop), 0, 0);
                continue;
            }
            const f32x4 va = acc[2 * run][mt] + bias[2 * run], vb = acc[2 * run + 1][mt] + bias[2 * run + 1];
            unsigned x0 = pack2(act_apply_t<ACT>(va[0]), act_apply_t<ACT>(va[1])), x1 = pack2(act_apply_t<ACT>(va[2]), act_apply_t<ACT>(va[3]));
            unsigned y0 = pack2(act_apply_t<ACT>(vb[0]), act_apply_t<ACT>(vb[1])), y1 = pack2(act_apply_t<ACT>(vb[2]), act_apply_t<ACT>(vb[3]));
            auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
            auto t0 = __builtin_amdgcn_permlane16_swap((unsigned)s0[0], (unsigned)s0[1], false, false);
            auto t1 = __builtin_amdgcn_permlane16_swap((unsigned)s1[0], (unsigned)s1[1], false, false);
            const i32x4 o = {(int)t0[0], (int)t1[0], (int)t0[1], (int)t1[1]};       // columns 8 fq .. 8 fq + 7 of the run
            __builtin_amdgcn_raw_buffer_store_b128(o, rs, (int)((off + run * 256u) | drop), 0, 0);
        }
    }
    UFV_EPI_MARK("END wide");
}

// SwiGLU epilogue with 16-byte stores (round 4).  A lane holds 4 output columns (8 bytes of bf16) per (accumulator row, half): columns 4 fq .. 4 fq + 3 of the
// 16-column output tile, so the plain form issues 16 eight-byte stores per wave and tile.  One v_permlane16_swap per register over a PAIR of accumulator rows
// (x = row mt, y = row mt + 1: the swap exchanges x's odd 16-lane rows with y's even ones) leaves lane rows 0 / 2 with columns 0..7 / 8..15 of row mt and lane
// rows 1 / 3 with the same columns of row mt + 1: 8 sixteen-byte stores, through a buffer descriptor (rows >= M dropped by the hardware, 32-bit offsets).
// Same values, same rounding: bit-identical output.
template <int MA0, int MA1>
__device__ __forceinline__ void epilogue256_swiglu_wide(const f32x4 (&acc)[4][MA0 + MA1], const Epi& e, int M, int N, int m0, int n0, int wr, int wc, int frow, int fq) {
    static_assert(MA0 % 2 == 0 && MA1 % 2 == 0, "accumulator rows are stored in pairs inside each A half");
    constexpr int MT = MA0 + MA1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(e.out, 0, (int)((unsigned)M * (unsigned)e.ldc * 2u), 0x20000);
    const unsigned step2 = 64u * (unsigned)e.ldc;                                                  // bytes between accumulator-row pairs (32 matrix rows)
    const unsigned col0 = (n0 >> 1) + wc * 16 + 8 * (fq >> 1);
    const int rsel = 16 * (fq & 1);                                                                // lane rows 1 / 3 carry accumulator row mt + 1
    const unsigned off_h0 = ((unsigned)(m0 + wr * 16 * MA0 + frow + rsel) * (unsigned)e.ldc + col0) * 2u;
    const unsigned off_h1 = ((unsigned)(m0 + 32 * MA0 + wr * 16 * MA1 + frow + rsel) * (unsigned)e.ldc + col0) * 2u;
    auto pack2 = [&](float a, float b) -> unsigned {
        const bf16x2 p = {(bf16)a, (bf16)b};
        return __builtin_bit_cast(unsigned, p);
    };
    const unsigned drop1 = n0 + 128 >= N ? 0x80000000u : 0u;   // as in epilogue256_wide: a half past N is stored to a dropped offset, every store is issued
    UFV_EPI_MARK("BEGIN swiglu_wide");
#pragma unroll
    for (int mp = 0; mp < MT; mp += 2) {
        const unsigned off = mp < MA0 ? off_h0 + (mp >> 1) * step2 : off_h1 + ((mp - MA0) >> 1) * step2;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            unsigned x[2], y[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                x[k] = pack2(swiglu_f(acc[2 * nh][mp][2 * k], acc[2 * nh + 1][mp][2 * k]), swiglu_f(acc[2 * nh][mp][2 * k + 1], acc[2 * nh + 1][mp][2 * k + 1]));
                y[k] = pack2(swiglu_f(acc[2 * nh][mp + 1][2 * k], acc[2 * nh + 1][mp + 1][2 * k]), swiglu_f(acc[2 * nh][mp + 1][2 * k + 1], acc[2 * nh + 1][mp + 1][2 * k + 1]));
            }
            auto t0 = __builtin_amdgcn_permlane16_swap(x[0], y[0], false, false);
            auto t1 = __builtin_amdgcn_permlane16_swap(x[1], y[1], false, false);
            const i32x4 o = {(int)t0[0], (int)t1[0], (int)t0[1], (int)t1[1]};
            __builtin_amdgcn_raw_buffer_store_b128(o, rs, (int)((off + nh * 128u) | (nh ? drop1 : 0u)), 0, 0);
        }
    }
    UFV_EPI_MARK("END swiglu_wide");
}

// Fused QKV projection + rotary embedding + KV-cache append (round 5; modeling_qwen2.py:176-205: q / k / v Linear, apply_rotary_pos_emb, cache update).
// ROPE kernels read their B fragments so that n-tile 2 h + t of a wave holds head-dims t * 64 + wc * 16 .. + 15 of the tile's head h (h = 0, 1: a 256-column
// tile is two heads of 128): the rotate-half partners (i, i + 64) of a head sit in ONE lane, accumulators acc[2 h][mt][j] and acc[2 h + 1][mt][j].  Nothing
// about W changes -- which LDS rows a wave multiplies is the wave's own choice.  Per (row pair, head, t): bias, round to bf16 (the value the unfused path stored
// and read back), RoPE with the row's cos / sin (the table ufv_rope_kv_table reads), round, regroup the pair's two rows with one v_permlane16_swap per register
// (as epilogue256_swiglu_wide) into 16-byte runs and store: q heads to `out` [M, Hq * 128], k / v heads to the KV cache row, v untouched.  Same operations in the
// same order as ufv_gemm + ufv_rope_kv_table: bit-identical.  2 MT store instructions per wave and tile (= NSTW), every one issued on every path.
template <int MA0, int MA1>
__device__ __forceinline__ void epilogue256_rope(const f32x4 (&acc)[4][MA0 + MA1], const Epi& e, int M, int N, int m0, int n0, int wr, int wc, int frow, int fq,
                                                 const f32x4 (&bias)[4], const f32x4 (&rcos)[MA0 + MA1], const f32x4 (&rsin)[MA0 + MA1]) {
    constexpr int MT = MA0 + MA1;
    static_assert(MT % 2 == 0, "accumulator rows are stored in pairs");
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(e.out, 0, (int)((unsigned)M * (unsigned)e.ldc * 2u), 0x20000);
    const __amdgpu_buffer_rsrc_t rs_kv = __builtin_amdgcn_make_buffer_rsrc(e.out_kv, 0, (int)((unsigned)M * (unsigned)e.ldkv * 2u), 0x20000);
    auto row_of = [&](int mt) { return m0 + (mt < MA0 ? wr * 16 * MA0 + mt * 16 : 32 * MA0 + wr * 16 * MA1 + (mt - MA0) * 16) + frow; };
    auto pack2 = [&](float a, float b) -> unsigned {
        const bf16x2 p = {(bf16)a, (bf16)b};
        return __builtin_bit_cast(unsigned, p);
    };
    auto rt = [](float v) { return (float)(bf16)v; };             // the bf16 the unfused GEMM stored
    const int head0 = n0 >> 7;
    const unsigned colw = wc * 16 + 8 * (fq >> 1);                 // this lane's 8-column run inside a 16-column n-tile, after the regroup
    UFV_EPI_MARK("BEGIN rope");
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
        const int head = head0 + nh;
        const bool is_q = head < e.rope_hq, roped = head < e.rope_hq + e.rope_hkv;
        const __amdgpu_buffer_rsrc_t rs = is_q ? rs_q : rs_kv;
        const unsigned ld = is_q ? (unsigned)e.ldc : (unsigned)e.ldkv;
        const unsigned colbase = (unsigned)(is_q ? head : head - e.rope_hq) * 128u + colw;      // kv row = [k heads | v heads]: head - Hq indexes both
        const unsigned drop = n0 + nh * 128 >= N ? 0x80000000u : 0u;
#pragma unroll
        for (int mp = 0; mp < MT; mp += 2) {
            const unsigned row = (unsigned)((fq & 1) ? row_of(mp + 1) : row_of(mp));             // lane rows 1 / 3 carry accumulator row mp + 1 after the swap
            float lo[2][4], hi[2][4];                                                            // [row of the pair][j]: dims i and i + 64
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x1 = rt(acc[2 * nh][mp + r][j] + bias[2 * nh][j]), x2 = rt(acc[2 * nh + 1][mp + r][j] + bias[2 * nh + 1][j]);
                    float y1, y2;
                    rope_pair(x1, x2, rcos[mp + r][j], rsin[mp + r][j], y1, y2);
                    lo[r][j] = roped ? y1 : x1;
                    hi[r][j] = roped ? y2 : x2;
                }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float (&v)[2][4] = t ? hi : lo;
                const unsigned x0 = pack2(v[0][0], v[0][1]), x1 = pack2(v[0][2], v[0][3]), y0 = pack2(v[1][0], v[1][1]), y1 = pack2(v[1][2], v[1][3]);
                auto t0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                auto t1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                const i32x4 o = {(int)t0[0], (int)t1[0], (int)t0[1], (int)t1[1]};
                __builtin_amdgcn_raw_buffer_store_b128(o, rs, (int)(((row * ld + colbase + t * 64u) * 2u) | drop), 0, 0);
            }
        }
    }
    UFV_EPI_MARK("END rope");
}

// ---- MX-emitting epilogues (e4m3 kernels, 256 x 256 tile; round 5) --------------------------------------------------------------------------------------
// The consumer of these activations is another e4m3 GEMM: instead of a bf16 tensor that a separate kernel reads back, scans for the row maximum and quantises
// (3.8 ms of stand-alone quantise launches per W8A8 clip), the epilogue stores e4m3 codes with ONE power-of-two scale per (row, 32 consecutive columns) -- the
// block a lane of v_mfma_scale_f32_16x16x128_f8f6f4 owns.  A wave's 32-column run of an accumulator row sits in its four 16-lane rows: block maximum = in-lane
// maximum of 8 values + two cross-row exchanges; scale byte e = biased exponent of amax / 448 rounded UP to a power of two (no value saturates;
// oracle.mx_quantize restates it bit for bit); codes = rne_e4m3(v * 2^(127 - e)); one v_permlane32_swap + one v_permlane16_swap leave lane row r with bytes
// 8 r .. 8 r + 7 of the run: ONE 8-byte store per (row, run).  The scale bytes of FOUR accumulator rows go out in one store: after the block maximum every lane of a
// row's four (fq = 0..3) holds the byte, so lane row fq takes accumulator row 4 g + fq -- the CU's write-out is paced by store INSTRUCTIONS (37 cycles each with
// every CU storing, LABNOTES round 4): a byte store per (row, run) made fc1's MX epilogue 11 us slower than the bf16 one.  2 MT + MT / 4 (SwiGLU: MT + MT / 4) store
// instructions per wave and tile (NSTM), every one issued on every path.
__device__ __forceinline__ unsigned mx_store_run(const float (&va)[4], const float (&vb)[4], const __amdgpu_buffer_rsrc_t rs_q, unsigned off_q) {
    float am = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) am = fmaxf(am, fmaxf(fabsf(va[j]), fabsf(vb[j])));
    {   // maximum over the four 16-lane rows without the LDS crossbar (two __shfl_xor = two ds_bpermute round trips per run, 32 per tile, each behind its own wait:
        // the MX epilogue of fc1 ran 10 us over the bf16 one): v_permlane16_swap(a, a) = ([r0 r0 r2 r2], [r1 r1 r3 r3]), v_permlane32_swap(c, c) = ([c0 c1 c0 c1], [c2 c3 c2 c3])
        const unsigned ab = __builtin_bit_cast(unsigned, am);
        auto p = __builtin_amdgcn_permlane16_swap(ab, ab, false, false);
        const float c = fmaxf(__builtin_bit_cast(float, (unsigned)p[0]), __builtin_bit_cast(float, (unsigned)p[1]));
        const unsigned cb = __builtin_bit_cast(unsigned, c);
        auto q = __builtin_amdgcn_permlane32_swap(cb, cb, false, false);
        am = fmaxf(__builtin_bit_cast(float, (unsigned)q[0]), __builtin_bit_cast(float, (unsigned)q[1]));
    }
    const unsigned e = mx_scale_byte(am);
    // v_cvt_scalef32_pk_fp8_f32: two values divided by the (power-of-two) scale and rounded to e4m3 in ONE instruction -- the epilogue is vector-issue-bound, and
    // eight multiplies by 2^(127 - e) in front of four plain conversions were a fifth of its extra work (same codes: the scaling is exact)
    const float sc = __builtin_bit_cast(float, e << 23);                                     // 2^(e - 127)
    typedef __attribute__((ext_vector_type(2))) short s16x2;
    s16x2 xs = {0, 0}, ys = {0, 0};
    xs = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(xs, va[0], va[1], sc, false);
    xs = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(xs, va[2], va[3], sc, true);
    ys = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(ys, vb[0], vb[1], sc, false);
    ys = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(ys, vb[2], vb[3], sc, true);
    const int x = __builtin_bit_cast(int, xs), y = __builtin_bit_cast(int, ys);
    // rows after the swaps: A = [x.r0, x.r2, y.r0, y.r2], B = [x.r1, x.r3, y.r1, y.r3]: lane row r holds bytes 8 r .. 8 r + 7 of the 32-byte run as {A, B}
    auto s = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)y, false, false);
    auto t = __builtin_amdgcn_permlane16_swap((unsigned)s[0], (unsigned)s[1], false, false);
    typedef __attribute__((ext_vector_type(2))) int i32x2;
    const i32x2 o = {(int)t[0], (int)t[1]};
    __builtin_amdgcn_raw_buffer_store_b64(o, rs_q, (int)off_q, 0, 0);
    return e;                                                  // every lane of the row's four holds the block's scale byte: the caller stores four rows' at once
}

// plain / activation form: the run = the wave's two adjacent n-tiles (32 consecutive columns), as in epilogue256_wide
template <int ACT, int MA0, int MA1>
__device__ __forceinline__ void epilogue256_wide_mx(const f32x4 (&acc)[4][MA0 + MA1], const Epi& e, int M, int N, int m0, int n0, int wr, int wc, int frow, int fq,
                                                    const f32x4 (&bias)[4]) {
    constexpr int MT = MA0 + MA1;
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(e.out, 0, (int)((unsigned)M * (unsigned)e.ldc), 0x20000);
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(e.out_bscale, 0, (int)((unsigned)((M + 63) >> 6) * (unsigned)e.ld_obs), 0x20000);   // [ceil(M / 64)][ceil(N / 512)][64][16]
    const unsigned drop1 = n0 + 128 >= N ? 0x80000000u : 0u;
    UFV_EPI_MARK("BEGIN wide_mx");
    static_assert(MT % 4 == 0, "the scale bytes of four accumulator rows leave in one store");
    auto row_of = [&](int mt) { return (unsigned)(m0 + (mt < MA0 ? wr * 16 * MA0 + mt * 16 : 32 * MA0 + wr * 16 * MA1 + (mt - MA0) * 16) + frow); };
    const unsigned col0 = (unsigned)(n0 + wc * 32);
#pragma unroll
    for (int g4 = 0; g4 < MT / 4; ++g4) {
        unsigned es = 0;                                       // this lane's pick of the group's scale bytes: row 4 g4 + fq, run 0 | run 1 << 8
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int mt = 4 * g4 + k;
            const unsigned row = row_of(mt);
            unsigned e2 = 0;
#pragma unroll
            for (int run = 0; run < 2; ++run) {
                const f32x4 a4 = acc[2 * run][mt] + bias[2 * run], b4 = acc[2 * run + 1][mt] + bias[2 * run + 1];
                float va[4], vb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { va[j] = act_apply_t<ACT>(a4[j]); vb[j] = act_apply_t<ACT>(b4[j]); }
                e2 |= mx_store_run(va, vb, rs_q, (row * (unsigned)e.ldc + col0 + run * 128u + 8u * fq) | (run ? drop1 : 0u)) << (8 * run);
            }
            es = fq == k ? e2 : es;
        }
        // run 0 / run 1 = K-tiles (n0 >> 7) & 3 (even) and + 1 of the consumer: adjacent bytes 4 wc + t of the row's 16 (a half past N: its byte is written too, nobody reads it)
        const unsigned row = row_of(4 * g4 + fq);
        const unsigned off_s = (row >> 6) * (unsigned)e.ld_obs + (col0 >> 9) * 1024u + (row & 63u) * 16u + ((col0 >> 5) & 3u) * 4u + ((col0 >> 7) & 3u);
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)es, rs_s, (int)(off_s | (row >= (unsigned)M ? 0x80000000u : 0u)), 0, 0);
    }
    UFV_EPI_MARK("END wide_mx");
}

// SwiGLU form.  A wave's outputs of an accumulator row are 16 columns per half (nh): 32 values, but 64 columns apart in the natural order (n0 / 2 + nh * 64 +
// wc * 16).  They are stored as ONE 32-column block at physical column n0 / 2 + wc * 32 + nh * 16: a fixed permutation of the activation's columns inside every
// group of 128, which the consumer's weight matrix (down_proj) carries on its K axis (ops.Fp8Weight(..., mx_swiglu_cols=True)) -- the product does not change.
template <int MA0, int MA1>
__device__ __forceinline__ void epilogue256_swiglu_mx(const f32x4 (&acc)[4][MA0 + MA1], const Epi& e, int M, int N, int m0, int n0, int wr, int wc, int frow, int fq) {
    constexpr int MT = MA0 + MA1;
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(e.out, 0, (int)((unsigned)M * (unsigned)e.ldc), 0x20000);
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(e.out_bscale, 0, (int)((unsigned)((M + 63) >> 6) * (unsigned)e.ld_obs), 0x20000);   // [ceil(M / 64)][ceil(N_out / 512)][64][16]
    const unsigned drop1 = n0 + 128 >= N ? 0x80000000u : 0u;      // (half a tile past N: both halves feed one block, so the whole block is dropped -- N % 256 == 0 is required by the host)
    UFV_EPI_MARK("BEGIN swiglu_mx");
    static_assert(MT % 4 == 0, "the scale bytes of four accumulator rows leave in one store");
    auto row_of = [&](int mt) { return (unsigned)(m0 + (mt < MA0 ? wr * 16 * MA0 + mt * 16 : 32 * MA0 + wr * 16 * MA1 + (mt - MA0) * 16) + frow); };
    const unsigned col = (unsigned)((n0 >> 1) + wc * 32);
#pragma unroll
    for (int g4 = 0; g4 < MT / 4; ++g4) {
        unsigned es = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int mt = 4 * g4 + k;
            float va[4], vb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { va[j] = swiglu_f(acc[0][mt][j], acc[1][mt][j]); vb[j] = swiglu_f(acc[2][mt][j], acc[3][mt][j]); }
            const unsigned e1 = mx_store_run(va, vb, rs_q, (row_of(mt) * (unsigned)e.ldc + col + 8u * fq) | drop1);
            es = fq == k ? e1 : es;
        }
        const unsigned row = row_of(4 * g4 + fq);
        const unsigned off_s = (row >> 6) * (unsigned)e.ld_obs + (col >> 9) * 1024u + (row & 63u) * 16u + ((col >> 5) & 3u) * 4u + ((col >> 7) & 3u);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)es, rs_s, (int)(off_s | drop1 | (row >= (unsigned)M ? 0x80000000u : 0u)), 0, 0);
    }
    UFV_EPI_MARK("END swiglu_mx");
}

// acc[nt][mt][j] = C[m0 + row(mt)][n0 + col(nt) + fq*4 + j] with
//   row(mt) = mt < MA0 ? wr*16*MA0 + mt*16 + frow : 32*MA0 + wr*16*MA1 + (mt-MA0)*16 + frow
//   col(nt) = nt < 2 ? wc*32 + nt*16 : 128 + wc*16*NB1 + (nt-2)*16
template <bool OUT_F32, bool SWIGLU, int ACT, bool DUMP, int MA0, int MA1, int NB1>
__device__ __forceinline__ void epilogue256(const f32x4 (&acc)[2 + NB1][MA0 + MA1], const Epi& e, int M, int N, int m0, int n0, int wr, int wc,
                                            int frow, int fq, const f32x4 (&bias)[2 + NB1]) {
    static_assert(!SWIGLU || NB1 == 2, "the SwiGLU epilogue pairs n-tiles (gate, up) inside each half");
    UFV_EPI_MARK("BEGIN plain");
#pragma unroll
    for (int mt = 0; mt < MA0 + MA1; ++mt) {
        const int m = m0 + (mt < MA0 ? wr * 16 * MA0 + mt * 16 : 32 * MA0 + wr * 16 * MA1 + (mt - MA0) * 16) + frow;
        if (m < M) {
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
                const int nb = n0 + nh * 128;
                if (nb < N) {
                    if constexpr (SWIGLU) {
                        const int n = ((nb + wc * 32) >> 1) + fq * 4;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float gte = acc[2 * nh][mt][j], up = acc[2 * nh + 1][mt][j];
                            v[j] = swiglu_f(gte, up);
                        }
                        epi_store4b<OUT_F32, ACT_NONE>(e, m, n, v[0], v[1], v[2], v[3], f32x4{0, 0, 0, 0});
                    } else {
                        const int n = nb + wc * 16 * (nh ? NB1 : 2) + fq * 4;
#pragma unroll
                        for (int j = 0; j < (nh ? NB1 : 2); ++j)
                            epi_store4b<OUT_F32, ACT, DUMP>(e, m, n + 16 * j, acc[2 * nh + j][mt][0], acc[2 * nh + j][mt][1], acc[2 * nh + j][mt][2],
                                                            acc[2 * nh + j][mt][3], bias[2 * nh + j]);
                    }
                }
            }
        }
    }
    UFV_EPI_MARK("END plain");
}

// Split-K part of a tile (KSPL kernels): out (+)= acc in TURN order part 0, 1, ... (deterministic sum).  Part 0 stores acc + bias + residual;
// a later part adds its accumulators to what the earlier parts left in `out`.  The parts of a tile run on different XCDs, whose L2s are not
// coherent: the tile is read and written with system-scope (sc0 sc1) accesses, two accumulator rows per round trip.
template <int MA0, int MA1, int NB1>
__device__ __forceinline__ void epilogue_split(const f32x4 (&acc)[2 + NB1][MA0 + MA1], const Epi& e, int M, int N, int m0, int n0, int wr, int wc, int frow,
                                               int fq, const f32x4 (&bias)[2 + NB1], bool first) {
    constexpr int MT = MA0 + MA1, NT = 2 + NB1;
    float* out = reinterpret_cast<float*>(e.out);
    int ncol[NT];
    bool nok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        ncol[nt] = n0 + (nt < 2 ? wc * 32 + nt * 16 : 128 + wc * 16 * NB1 + (nt - 2) * 16) + fq * 4;
        nok[nt] = n0 + (nt < 2 ? 0 : 128) < N;
    }
    auto row_of = [&](int mt) { return m0 + (mt < MA0 ? wr * 16 * MA0 + mt * 16 : 32 * MA0 + wr * 16 * MA1 + (mt - MA0) * 16) + frow; };
    auto store_row = [&](int m, const f32x4 (&v)[NT]) {
        if (m < M) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                if (nok[nt]) {
                    float* pp = out + (size_t)m * e.ldc + ncol[nt];
                    // (s_nop: a store wider than 64 bits still reads its data registers for a cycle after issue; inside inline asm the
                    //  compiler cannot see that and re-used them for the next address at once -- corrupted rows on hardware)
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" ::"v"(pp), "v"(v[nt]) : "memory");
                }
        }
    };
    if (first) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = row_of(mt);
            f32x4 v[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                v[nt] = acc[nt][mt] + bias[nt];
                if (e.resid && m < M && nok[nt]) {
                    const int mr = e.resid_rows > 0 ? m % e.resid_rows : m;
                    v[nt] += *reinterpret_cast<const f32x4*>(e.resid + (size_t)mr * e.ldr + ncol[nt]);
                }
            }
            store_row(m, v);
        }
        return;
    }
    // later parts: read-modify-write, TWO accumulator rows per memory round trip (clamped rows / half-tiles past the edge are loaded, never stored)
    const int c2 = min(ncol[2], N - 4), c3 = NT == 4 ? min(ncol[NT - 1], N - 4) : 0;
#pragma unroll
    for (int mp = 0; mp < MT; mp += 2) {
        const int ma = row_of(mp), mb = mp + 1 < MT ? row_of(mp + 1) : ma;
        const float* ra = out + (size_t)min(ma, M - 1) * e.ldc;
        const float* rb = out + (size_t)min(mb, M - 1) * e.ldc;
        f32x4 a0, a1, a2, a3 = {0, 0, 0, 0}, b0, b1, b2, b3 = {0, 0, 0, 0};
        if constexpr (NT == 4)
            asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %9, off sc0 sc1\n\t"
                         "global_load_dwordx4 %2, %10, off sc0 sc1\n\tglobal_load_dwordx4 %3, %11, off sc0 sc1\n\t"
                         "global_load_dwordx4 %4, %12, off sc0 sc1\n\tglobal_load_dwordx4 %5, %13, off sc0 sc1\n\t"
                         "global_load_dwordx4 %6, %14, off sc0 sc1\n\tglobal_load_dwordx4 %7, %15, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
                         : "v"(ra + ncol[0]), "v"(ra + ncol[1]), "v"(ra + c2), "v"(ra + c3), "v"(rb + ncol[0]), "v"(rb + ncol[1]), "v"(rb + c2), "v"(rb + c3)
                         : "memory");
        else
            asm volatile("global_load_dwordx4 %0, %6, off sc0 sc1\n\tglobal_load_dwordx4 %1, %7, off sc0 sc1\n\t"
                         "global_load_dwordx4 %2, %8, off sc0 sc1\n\tglobal_load_dwordx4 %3, %9, off sc0 sc1\n\t"
                         "global_load_dwordx4 %4, %10, off sc0 sc1\n\tglobal_load_dwordx4 %5, %11, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(b0), "=&v"(b1), "=&v"(b2)
                         : "v"(ra + ncol[0]), "v"(ra + ncol[1]), "v"(ra + c2), "v"(rb + ncol[0]), "v"(rb + ncol[1]), "v"(rb + c2) : "memory");
        f32x4 va[NT], vb[NT];
        va[0] = acc[0][mp] + a0; va[1] = acc[1][mp] + a1; va[2] = acc[2][mp] + a2;
        if constexpr (NT == 4) va[3] = acc[3][mp] + a3;
        store_row(ma, va);
        if (mp + 1 < MT) {
            vb[0] = acc[0][mp + 1] + b0; vb[1] = acc[1][mp + 1] + b1; vb[2] = acc[2][mp + 1] + b2;
            if constexpr (NT == 4) vb[3] = acc[3][mp + 1] + b3;
            store_row(mb, vb);
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
    int ksplit;         // KSPL kernels: K parts per tile (items = tiles x parts, part-major), turn flags in `flags` (this launch's own slice), `epoch` = this launch's ticket base
    int* err;           // pinned host word: set when a bounded turn wait expires (gemm_state.hip)
    int half;           // HALF kernels: 1 = the launch ends with ONE round of half-tile items (see gemm_nt_256: HALF); 0 = whole tiles only
};

// Bounded waits on another block's flag.  A turn can only fail to come when the launch's blocks are not all resident (some other kernel
// holds CUs for seconds); instead of hanging the device the waiter gives up after ~2^21 polls (seconds), reports through `err` and goes
// on -- that launch's tile is then wrong and the next ufv_gemm call on the device returns an error (gemm_state.hip).
__device__ __forceinline__ void turn_wait_ge(const int* flag, int target, int* err) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 21)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
}
__device__ __forceinline__ void turn_wait_eq(const int* flag, int value, int* err) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != value) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 21)) { __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PH2: two phases per K-tile instead of four -- phase A reads A0 and ALL of B and multiplies the top half of the wave's tile, phase B
// reads A1 and multiplies the bottom half with the B fragments still in registers.  Half as many barrier steps per K-tile, each twice
// as long; staging: A1[t+1] in phase A of K-tile t, A0 / B0 / B1 [t+2] in phase B (every half-tile is re-staged two barrier steps after its
// last reader and lands six steps = 1.5 K-tiles before its first).
// KSPL: aligned split-K for outputs with few tiles and a long K (the decoder's `down` projection: 140 tiles of 256x256 on 256 CUs).  Items
// are (part, tile) in part-major order, every part a contiguous range of K-tiles, so the blocks of a round run the SAME K range of
// neighbouring tiles and keep sharing A / W panels in L2 (free-running stream-K ranges do not: they re-read every panel from HBM).  The parts of
// a tile add into the fp32 output in turn order (epilogue_split) -- deterministic, no workspace; needs act == none and one block per CU.
// ROPE: the fused QKV + RoPE + KV-append form (epilogue256_rope): B fragments re-mapped so that the rotate-half partners of a head share a lane.
// MX (e4m3 operands only): bit 0 = the A operand carries MX block scales (one e8m0 byte per row and 32 K-elements, staged per K-tile beside the operand tiles and
// handed to v_mfma_scale_f32_16x16x128_f8f6f4, whose lanes own exactly one such block each); bit 1 = the epilogue emits e4m3 codes + block scales (epilogue256_*_mx).
// HALF (round 6; the decoder's gate/up projection at M = 2399: 10 x 148 = 1480 tiles on 256 CUs are 5.78 rounds, and the 148 tiles of the last row band hold 95 valid rows
// -- 9.7 % of the launch's MFMA work was padding plus an idle last round): the launch's tiles are dealt as whole tiles for as many FULL rounds as there are, and the
// rest as ONE round of half-tile items when they fit one round: a leftover whole tile becomes two items of 32 MA0 rows each, a tile of a row band with <= 32 MA0 valid
// rows one item.  A half-tile item runs the same two-phase K loop with phase B's fragment reads and MFMAs left out (no A1 half is staged: the counted waits take L_ALL - LA1 operations per K-tile; the barriers
// and the ring's stages are the tile's own) and the same epilogue with the row limit at the item's
// end, so that the untouched A1 accumulators are stored nowhere (every store instruction is still issued: the relaxed seam waits hold).  Every output element is
// the same sum in the same order as in a whole tile: bit-identical.  M = 2399 x N = 37888: 5 rounds + one round of 2 x 52 + 148 = 252 half items instead of 6 rounds.
template <bool OUT_F32, bool SWIGLU, bool FP8, bool SKT, int MA0 = 4, int MA1 = 4, int NB1 = 2, bool PH2 = false, bool KSPL = false, bool ROPE = false, int MX = 0, bool HALF = false>
__global__ __launch_bounds__(512, 2) void gemm_nt_256(const void* __restrict__ Av, const void* __restrict__ Wv, Epi e, int M,
                                                       int N, int K, int lda, int ldw, StreamK sk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ES = FP8 ? 1 : 2;
    using T = PP<MA0, MA1, NB1>;
    constexpr int MT = T::MT, NT = T::NT, BM = T::BM, BN = T::BN;
    static_assert(!SKT || (MA0 == 4 && MA1 == 4 && NB1 == 2), "the stream-K fix-up is written for the 256x256 tile image");
    static_assert(!KSPL || (OUT_F32 && !SWIGLU && !FP8 && !SKT && PH2), "split-K parts accumulate into an fp32 output");
    static_assert(MX == 0 || (FP8 && PH2 && !SKT && !KSPL && !ROPE), "MX block scales belong to the e4m3 kernels");
    static_assert(!HALF || (PH2 && !SKT && !KSPL && (MX & 1) == 0), "half-tile items: the two-phase whole-tile kernels");
    static_assert(!(ROPE && FP8) || MX == 0, "the fused RoPE form of the e4m3 QKV GEMM takes per-row-scaled activations");
    static_assert(!(MX & 2) || (!OUT_F32 && MA0 == 4 && MA1 == 4 && NB1 == 2), "the MX-emitting epilogues are built for the 256 x 256 tile");
    constexpr bool MXA = (MX & 1) != 0;
    // MXA: two 4 KiB slots of A-operand block scales (256 rows x 16 bytes per FOUR K-tiles).  NB1 == 1 shapes: in the unused upper half of the B1 half-tile slots of the
    // operand ring (slot s at s * 65536 + 57344); else above the ring
    constexpr int SCL_OFF = NB1 == 1 ? 57344 : 131072, SCL_STRIDE = NB1 == 1 ? 65536 : 4096;
    static_assert(!ROPE || (!OUT_F32 && !SWIGLU && !SKT && !KSPL && PH2 && NB1 == 2 && (MA0 + MA1) % 2 == 0),
                  "the fused RoPE epilogue: bf16 output, 256-column tiles (two heads of 128), accumulator rows in pairs");
    const char* A = reinterpret_cast<const char*>(Av);
    const char* W = reinterpret_cast<const char*>(Wv);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    const int nk = K * ES / 128;
    const int G = gridDim.x;                       // persistent: block b walks tiles b, b+G, ...
    // tile sequence position -> (tm, tn): within a round of G tiles give each XCD (launch id % 8) a contiguous run,
    // and order the sequence in groups of 8 row-tiles so that concurrent tiles share A/W panels in L2.
    // HALF: row bands dealt as whole tiles (all but a last band with <= 32 MA0 valid rows), whole tiles, full rounds of them, leftover whole tiles, band tiles
    const bool half_on = HALF && sk.half != 0;
    const int light = (half_on && M - (tiles_m - 1) * BM <= 32 * MA0) ? 1 : 0;
    const int rows_full = half_on ? tiles_m - light : tiles_m, n_full = rows_full * tiles_n, n_light = light * tiles_n;
    const int full_rounds = half_on ? n_full / (int)gridDim.x : 0, left_full = half_on ? n_full - full_rounds * (int)gridDim.x : 0;
    bool item_half = false;                    // the item next_item() handed out last is a half-tile item
    auto tile_coords = [&](int id, int& m0_, int& n0_) {
        const int GM = sk.gm;
        const int gsz = GM * tiles_n, g = id / gsz, first_m = g * GM;
        const int gm = min(rows_full - first_m, GM);
        m0_ = (first_m + (id % gsz) % gm) * BM;
        n0_ = ((id % gsz) / gm) * BN;
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
    int item_part = 0, item_tile = 0;          // KSPL: the item next_item() handed out last
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
        const bool last_half_round = half_on && round == full_rounds;
        if (half_on && round > full_rounds) return false;
        const int items = KSPL ? nwg * sk.ksplit : nwg;
        const int base = round * G;
        const int cnt = last_half_round ? 2 * left_full + n_light : min(G, items - base);        // items in this round
        const int bid = blockIdx.x;
        ++round;
        if (bid >= cnt) return false;
        const int q = cnt >> 3, r = cnt & 7, x = bid & 7;
        const int id = base + (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
        if constexpr (HALF) {
            item_half = last_half_round;
            if (last_half_round) {                   // item i of the round: the two halves of a leftover tile are neighbours on ONE XCD (they share the tile's W panel)
                const int i = id - base;
                if (i < 2 * left_full) {
                    tile_coords(base + (i >> 1), m0_, n0_);
                    m0_ += (i & 1) * 32 * MA0;
                } else {
                    m0_ = (tiles_m - 1) * BM;
                    n0_ = (i - 2 * left_full) * BN;
                }
                k0_ = 0; k1_ = nk;
                return true;
            }
        }
        if constexpr (KSPL) {
            const int per = (nk + sk.ksplit - 1) / sk.ksplit;
            item_part = id / nwg; item_tile = id - item_part * nwg;
            tile_coords(item_tile, m0_, n0_);
            k0_ = item_part * per; k1_ = min(nk, k0_ + per);
            return true;
        }
        tile_coords(id, m0_, n0_);
        k0_ = 0; k1_ = nk;
        return true;
    };

    // ---- LDS-DMA sources: half-tile `which` (0=A0 1=A1 2=B0 3=B1), one or two 8-row pieces per wave (T::L*)
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
#ifndef UFV_FPF_MODE
#define UFV_FPF_MODE 1       /* lab: 0 = never */
#endif
#ifndef UFV_FPF_LEVEL         /* per shape: 2 = A and B fragments prefetched (two B sets), 1 = A only (B read at the classic place by both groups), 0 = classic schedule.
                                 The shapes measured faster with it inside the clip (LABNOTES round 4); lab: -D'UFV_FPF_LEVEL(a,b,c)=...' */
#define UFV_FPF_LEVEL(MA0_, MA1_, NB1_) (((MA0_) + (MA1_) <= 7 && (NB1_) == 1) ? 2 : 0)
#endif
    constexpr int FPL = (PH2 && !FP8 && !SKT && UFV_FPF_MODE == 1) ? UFV_FPF_LEVEL(MA0, MA1, NB1) : 0;
    constexpr bool FPF = FPL > 0;                              // fragment prefetch: see the K loop
    // X2 (with FPF): 96-row A halves staged EXACTLY -- 12 pieces of 8 rows instead of 16 (a 96-row half was staged as 128 rows), compact slots.  Per group (A stays
    // group-local) the 6 pieces of such a half go to 4 waves as 2 + 2 + 1 + 1; where both halves have 96 rows (192 x 192) the second one takes 1 + 1 + 2 + 2, so every wave
    // issues six DMA instructions per K-tile and the counted waits stay uniform; with one such half (224 x 192) waves 0, 1 of a group issue seven, waves 2, 3 six, and the
    // waits take the wave's own count.
#ifndef UFV_X2_MODE
#define UFV_X2_MODE 1        /* lab: 0 = padded staging */
#endif
    constexpr bool X2 = FPF && FPL == 2 && UFV_X2_MODE == 1 && NB1 == 1 && (MA0 == 3 || MA1 == 3) && MA0 >= 3 && MA1 >= 3;
    constexpr int OA1 = X2 ? (MA0 == 3 ? 12288 : 16384) : 16384, OB0 = X2 ? OA1 + (MA1 == 3 ? 12288 : 16384) : 32768, OB1 = OB0 + 16384, STG = X2 ? OB1 + 8192 : 65536;
    const int gj = wave & 3, gg = wave >> 2;
    const bool x_hi = gj < 2;
    // pieces of this wave in each A half, and its first piece: a 96-row half as 2 2 1 1 (or 1 1 2 2: the second such half), a 128-row half as always
    const int la0x = MA0 == 3 ? (x_hi ? 2 : 1) : 2, la1x = MA1 == 3 ? ((MA0 == 3) != x_hi ? 2 : 1) : 2;
    const int pa0x = MA0 == 3 ? 6 * gg + (x_hi ? 2 * gj : 4 + (gj - 2)) : 2 * wave;
    const int pa1x = MA1 == 3 ? 6 * gg + (MA0 == 3 ? (x_hi ? gj : 2 + 2 * (gj - 2)) : (x_hi ? 2 * gj : 4 + (gj - 2))) : 2 * wave;
    constexpr bool XCLS = X2 && (MA0 == 3) != (MA1 == 3);          // two classes of waves (7 / 6 instructions per K-tile)
    // what FPF rests on: the leading group (waves 0-3) reads A rows [0, 16 MA) of each half, and waves 0-3 stage rows [0, 32 LA) of it
    static_assert(!FPF || (16 * MA0 <= 32 * T::LA0 && 16 * MA1 <= 32 * T::LA1), "fragment prefetch: the leading group's A rows must be staged by its own waves");
    const char* src[4][2];
    const unsigned char* ssrc = nullptr;           // MXA: this lane's row of the A block scales (rows 64 (wave & 3) + lane of the tile; waves 4-7 stage the same bytes again)
    auto set_src = [&](int m0_, int n0_) {
        const int Mc = (HALF && item_half) ? min(M, m0_ + 32 * MA0) : M;        // a half-tile item: its A1 pieces fetch (and re-fetch) the item's last row
        if constexpr (X2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                src[0][i] = A + (size_t)min(m0_ + (pa0x + min(i, la0x - 1)) * 8 + lrow, M - 1) * lda * ES + lchunk * 16;
                src[1][i] = A + (size_t)min(m0_ + 32 * MA0 + (pa1x + min(i, la1x - 1)) * 8 + lrow, M - 1) * lda * ES + lchunk * 16;
                src[2][i] = W + (size_t)min(n0_ + (wave * 2 + i) * 8 + lrow, N - 1) * ldw * ES + lchunk * 16;
                src[3][i] = W + (size_t)min(n0_ + 128 + wave * 8 + lrow, N - 1) * ldw * ES + lchunk * 16;
            }
            return;
        }
        // block scales are stored in GROUPS OF FOUR K-TILES, 16 bytes per (row, group), byte 4 fq + (kt & 3) of a row = the scale of its K-block fq of K-tile kt: the 64
        // rows a wave stages are 1 KiB contiguous, ONE piece per four K-tiles (row-major [M][K / 32], 64 lanes fetched 64 different cache lines per K-tile: the scale
        // piece cost more address work than an operand half-tile; and one piece per K-tile was still +8 % on the 192 x 192 kernel), and a lane's four K-tile scales
        // of a fragment row are ONE dword (one ds_read_b32 per fragment per four K-tiles + a shift per K-tile)
        // ... and per BLOCK OF 64 ROWS all K-groups follow each other, [M / 64][K / 512][64 rows][16]: the pieces a wave fetches along K are consecutive KiBs.  (With the
        // groups as outer planes, [K / 512][M][16], every piece of an item touched a new page 16 M bytes further on: 0.75 us per piece, 6 us for the prologue's -- fc2 ran
        // 125 instead of 101 us, lab 5 / 6 / 7.)
        if constexpr (MXA) {
            const int srow = min(m0_ + (wave & 3) * 64 + lane, M - 1);
            ssrc = e.a_bscale + (size_t)(srow >> 6) * e.ld_abs + (srow & 63) * 16;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int la = h ? T::LA1 : T::LA0, lb = h ? T::LB1 : T::LB0;
                const int ra = h * 32 * MA0 + (wave * la + i) * 8 + lrow, rb = h * 128 + (wave * lb + i) * 8 + lrow;
                src[h][i] = A + (size_t)min(m0_ + ra, Mc - 1) * lda * ES + lchunk * 16;
                src[2 + h][i] = W + (size_t)min(n0_ + rb, N - 1) * ldw * ES + lchunk * 16;
            }
    };
    int kbeg = 0, kend = 0;                        // K-tile range of the item being loaded
    auto stage = [&](int d, int which, int kt) {
        if constexpr (X2) {
            if (kt < kend) {
                const int l = which == 0 ? la0x : which == 1 ? la1x : which == 2 ? 2 : 1;
                char* dst = smem + d * STG + (which == 0 ? pa0x * 1024 : which == 1 ? OA1 + pa1x * 1024 : which == 2 ? OB0 + wave * 2048 : OB1 + wave * 1024);
                __builtin_amdgcn_global_load_lds(GLB_PTR(src[which][0] + kt * 128), LDS_PTR(dst), 16, 0, 0);
                if (l == 2) __builtin_amdgcn_global_load_lds(GLB_PTR(src[which][1] + kt * 128), LDS_PTR(dst + 1024), 16, 0, 0);
            }
            return;
        }
        if constexpr (MXA) {
            if (which == 4) {                      // the K-tile's 4 scale bytes of 64 rows per wave: one dword per lane
                // one piece = the scales of FOUR K-tiles (kt .. kt + 3, kt % 4 == 0) of 64 rows per wave, 16 bytes per row; slot = (kt / 4) & 1
                if (kt < kend) __builtin_amdgcn_global_load_lds(GLB_PTR(ssrc + (size_t)(kt >> 2) * 1024), LDS_PTR(smem + SCL_OFF + ((kt >> 2) & 1) * SCL_STRIDE + (wave & 3) * 1024), 16, 0, 0);
                return;
            }
        }
        const int l = which == 0 ? T::LA0 : which == 1 ? T::LA1 : which == 2 ? T::LB0 : T::LB1;
        if constexpr (HALF) { if (which == 1 && item_half) return; }      // a half-tile item stages no A1 half: its K loop's waits count L_ALL - LA1 operations per K-tile
        if (kt < kend) {
            char* dst = smem + d * 65536 + which * 16384 + wave * (1024 * l);
            __builtin_amdgcn_global_load_lds(GLB_PTR(src[which][0] + kt * 128), LDS_PTR(dst), 16, 0, 0);
            if (l == 2) __builtin_amdgcn_global_load_lds(GLB_PTR(src[which][1] + kt * 128), LDS_PTR(dst + 1024), 16, 0, 0);
        }
    };
    auto prologue_loads = [&]() {      // first K-tile of the item complete + A0/B1 (PH2: A0/B0/B1) of its second K-tile
        if constexpr (MXA) stage(0, 4, kbeg);      // the scales of K-tiles 0 .. 3 FIRST: the item's first wait needs them (every MX-A item starts at K-tile 0): the first wait of the item needs them (every MXA item starts at K-tile 0)
        stage(0, 0, kbeg); stage(0, 2, kbeg); stage(0, 3, kbeg);
        stage(0, 1, kbeg);
        stage(1, 0, kbeg + 1);
        if constexpr (PH2) stage(1, 2, kbeg + 1);
        stage(1, 3, kbeg + 1);
    };
    constexpr int L_ALL = T::LA0 + T::LA1 + T::LB0 + T::LB1;      // PH2: DMA instructions in flight behind the half-tiles that are needed next
    // the counted wait that leaves exactly the last two staged half-tiles (A0, B1 of the K-tile after next) in flight
#define UFV_WAIT_KEEP_A0_B1()                                                                              \
    do {                                                                                                     \
        if constexpr (T::LA0 + T::LB1 == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");              \
        else if constexpr (T::LA0 + T::LB1 == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");         \
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                              \
    } while (0)

    const int frow = lane & 15, fq = lane >> 4, fx = lane & 7;
    const int a_row_off[2] = {(wr * 16 * MA0 + frow) * 128, (wr * 16 * MA1 + frow) * 128};     // + i*2048 inside the half
    // ROPE: n-tile i of a half = rows wc * 16 + 64 i .. (head-dims wc * 16 + 64 i ..): the rotate-half partners land in one lane (epilogue256_rope)
    const int b_row_off[2] = {ROPE ? (wc * 16 + frow) * 128 : (wc * 32 + frow) * 128, ROPE ? (wc * 16 + frow) * 128 : (wc * 16 * NB1 + frow) * 128};
    constexpr int BSTR = ROPE ? 8192 : 2048;                  // bytes between a wave's two n-tiles of a half (64 rows | 16 rows)
    // bf16: k-step kk reads chunk 4*kk + fq;  fp8: the lane's 32 bytes are chunks 2*fq and 2*fq + 1
    // MXA: the scaled MFMA applies lane group g's scale byte to K-block g, and its K-block j is the FIRST 16 bytes of lane groups 2 (j & 1), 2 (j & 1) + 1 for j < 2 and their
    // SECOND 16 bytes for j >= 2 (measured with one non-zero element per position, tools/scratch history in LABNOTES round 5): block j = chunks 2 j, 2 j + 1 of the row when lane
    // group g reads chunks g and 4 + g -- the bf16 mapping.  (Without scales any K order shared by both operands is right, which is all the plain e4m3 path needs.)
    // The plain e4m3 kernels take the same order (round 5; they used chunks 2 fq, 2 fq + 1 -- 32 contiguous bytes per lane): any K order shared by both operands gives the
    // same sums, and this one reads LDS faster -- the 16 lanes of a row group then cover 8 different 16-byte chunks per read instead of pairs of adjacent ones
    // (fc2 18432 x 1152 x 4352: 112 -> 101 us; down: 185 -> 177 us, lab 4 / 7).
    const int coff0 = (fq ^ fx) << 4, coff1 = ((4 + fq) ^ fx) << 4;

    int m0, n0, k0, k1;
    bool have = next_item(m0, n0, k0, k1);
    if (have) { set_src(m0, n0); kbeg = k0; kend = k1; prologue_loads(); }
    // Stores of the previous item's epilogue that may still be in flight when this item's first K-tile is consumed.  vmcnt is one in-order
    // counter: this item's prologue DMA was issued BEFORE those stores, so a wait that must only cover the prologue may leave them (NST more
    // operations) outstanding -- the first K-tile's two phases then run while the store burst drains instead of behind it.  Exact only when
    // every store instruction of the epilogue was issued (interior tile); 0 otherwise and for a block's first item.
    constexpr int NST = SWIGLU ? 2 * MT : MT * NT;           // store instructions of the 8-byte / residual epilogues; the 16-byte bf16 epilogue issues 2 * MT
    constexpr int NSTS = MT;                                 // the 16-byte SwiGLU epilogue: one store per accumulator-row pair and half
    // INVARIANT the relaxed waits rest on: the epilogue that ran last issued EXACTLY this many vector-memory instructions after the next tile's prologue DMA.
    // epilogue256_resid issues its loads / stores from inline asm (count fixed by construction, every store issued on edge tiles too); epilogue256_wide and
    // epilogue256_swiglu_wide issue one __builtin_amdgcn_raw_buffer_store per (row, run) / (row pair, half) -- buffer stores of different rows with no branch
    // between them (rows >= M and a half past N are dropped by the descriptor); the plain epilogue256 relaxes only on interior tiles.  A toolchain that changed one
    // of these counts would under-wait the first K-tile: tests/test_isa_epilogue_stores.py (CPU) compiles the kernels to assembly and counts the vector-memory
    // instructions on every path between the UFV_EPI_MARK comments of each epilogue form against NST / NSTW / NSTS; the bit-identity tests against the 128-wide
    // kernel (tests/test_kernels_gpu.py) are the second guard.
    constexpr int NSTW = 2 * MT;
    constexpr int NSTM = (SWIGLU ? MT : 2 * MT) + MT / 4;     // the MX-emitting epilogues: a code store per (row, run) | per row + one scale store per four rows
    constexpr bool RELAX_OK = PH2 && !SKT && !KSPL && (L_ALL + NST <= 63);
    int relax = 0;                                           // 0: strict waits; 1: NST stores may stay in flight; 2: NSTW; 3: NSTS
    UFV_TSTAMP_DECL
    while (have) {
    const int len = k1 - k0;
    f32x4 acc[NT][MT];   // [nt][mt]
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 afr[4][2], bfr[PH2 ? 4 : 2][2];
    int asc[MXA ? MT : 1];                                    // MXA: e8m0 block scale of each A fragment's row for this lane's 32-element K block of THIS K-tile (low byte)
    unsigned asc_raw[MXA ? MT : 1];                           //      ... of four K-tiles, one byte each
    int asc_sh = 0;
    auto read_a = [&](const char* half, auto hsel) {
        constexpr int h = decltype(hsel)::value;
#pragma unroll
        for (int i = 0; i < (h ? MA1 : MA0); ++i) {
            afr[i][0] = *reinterpret_cast<const bf16x8*>(half + a_row_off[h] + i * 2048 + coff0);
            afr[i][1] = *reinterpret_cast<const bf16x8*>(half + a_row_off[h] + i * 2048 + coff1);
        }
    };
    auto read_b = [&](const char* half, auto hsel) {
        constexpr int h = decltype(hsel)::value, d0 = PH2 ? 2 * h : 0;
#pragma unroll
        for (int i = 0; i < (h ? NB1 : 2); ++i) {
            bfr[d0 + i][0] = *reinterpret_cast<const bf16x8*>(half + b_row_off[h] + i * BSTR + coff0);
            bfr[d0 + i][1] = *reinterpret_cast<const bf16x8*>(half + b_row_off[h] + i * BSTR + coff1);
        }
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
#define UFV_SYNC_THEN_MMA(NTB, NCNT, MTB, MCNT, S0, S1, S2)                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    UFV_GSTAMP(S0);                                                                                          \
    __builtin_amdgcn_s_barrier();                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
    UFV_GSTAMP(S1);                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    __builtin_amdgcn_s_setprio(1);                                                                           \
    if constexpr (FP8) {                                                                                     \
        if constexpr (MXA && (MTB) == 0) {                                                                   \
            _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_) asc[m_] = (int)(asc_raw[m_] >> asc_sh);        \
        }                                                                                                    \
        _Pragma("unroll") for (int n_ = 0; n_ < NCNT; ++n_)                                                  \
            _Pragma("unroll") for (int m_ = 0; m_ < MCNT; ++m_)                                              \
                if constexpr (MXA && UFV_MX_LAB == 0)                                                        \
                    acc[NTB + n_][MTB + m_] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(              \
                        cat8(bfr[n_][0], bfr[n_][1]), cat8(afr[m_][0], afr[m_][1]), acc[NTB + n_][MTB + m_], 0, 0, 0, 0x7f, 0, asc[MTB + m_]); \
                else                                                                                         \
                acc[NTB + n_][MTB + m_] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                  \
                    cat8(bfr[n_][0], bfr[n_][1]), cat8(afr[m_][0], afr[m_][1]), acc[NTB + n_][MTB + m_], 0, 0, 0, 0, 0, 0); \
    } else {                                                                                                 \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                         \
        _Pragma("unroll") for (int n_ = 0; n_ < NCNT; ++n_)                                                  \
            _Pragma("unroll") for (int m_ = 0; m_ < MCNT; ++m_)                                              \
                acc[NTB + n_][MTB + m_] =                                                                    \
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[n_][kk], afr[m_][kk], acc[NTB + n_][MTB + m_], 0, 0, 0); \
    }                                                                                                        \
    if constexpr (MXA) {                                                                                     \
        /* anchor: with the scale shifts in front of them hipcc placed this step's MFMAs BEHIND the closing barrier, at the end of the next load step (the    */ \
        /* listing showed `s_setprio 1; s_setprio 0` with nothing between: both wave groups then multiply and wait in the same phases).  An empty volatile asm */ \
        /* that names the accumulators keeps their producers in front of it. */ \
        _Pragma("unroll") for (int n_ = 0; n_ < NCNT; ++n_)                                                  \
            _Pragma("unroll") for (int m_ = 0; m_ < MCNT; ++m_) asm volatile("" : "+v"(acc[NTB + n_][MTB + m_]));  \
    }                                                                                                        \
    __builtin_amdgcn_s_setprio(0);                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    UFV_GSTAMP(S2);                                                                                          \
    __builtin_amdgcn_s_barrier();

    // FPF (fragment prefetch): the ds_reads of a phase are issued BETWEEN THE MFMAs OF THE PHASE BEFORE IT, into a second set of fragment registers, instead of in
    // a burst at the head of the phase's load step.  Timeline of the 192 x 192 shape (tools/lab/gemm_lab.hip, ticks per K-tile and wave): MFMA 2 x 350, fragment reads
    // 250 + 85 (the four waves of a group read at once: the LDS array serves 48 KB in one burst), DMA issue 80 + 245, counted waits 140, barriers 120 -- a wave's own
    // instruction stream is 1630 ticks, 920 of them not MFMA, and two groups can only hide each other's halves: 2000 ticks per K-tile for 4 x 350 of MFMA.  Reads issued one
    // or two at a time behind a group of MFMAs cost a few cycles each (the LDS array is idle then) and the load steps shrink to DMA issue + wait.
    // WHICH reads may move: the two wave groups run one barrier apart and every wave stages its own pieces of every half-tile.  A read at the head of a load step follows the
    // covering wait by TWO barriers -- the lagging group's wait is behind it too.  A read inside the MFMA step follows it by ONE: for the leading group (waves 0-3) the
    // pieces the lagging waves staged are not covered yet (seen as wrong, varying sums once every CU streams from HBM: M 2399, K >= 8192, tools/scratch history in
    // LABNOTES).  The A halves are group-local (wave w stages rows 16 w .. of a half, group g reads rows 16 MA g ..: its own waves' or the leading group's), so both groups
    // prefetch A; the B halves are read by everybody, so only the LAGGING group keeps its prefetched B fragments -- the leading group reads B again at the head of its
    // phase A, where the classic schedule read it (its prefetch is issued all the same: no branch between the MFMAs, the second read overwrites it).
    // Costs (MA1 + NT) x 8 registers.
    bf16x8 aF0[FPF ? MA0 : 1][2], aF1[FPF ? MA1 : 1][2], bF[FPL == 2 ? 2 : 1][FPF ? NT : 1][2];
    // read op r of a fragment list: A-half h frag i, k-half c  /  B frag j (0,1 = the 128-column half, 2.. = the other), k-half c
    auto rd_a0 = [&](const char* stg, int r) { aF0[r >> 1][r & 1] = *reinterpret_cast<const bf16x8*>(stg + a_row_off[0] + (r >> 1) * 2048 + ((r & 1) ? coff1 : coff0)); };
    auto rd_a1 = [&](const char* stg, int r) { aF1[r >> 1][r & 1] = *reinterpret_cast<const bf16x8*>(stg + OA1 + a_row_off[1] + (r >> 1) * 2048 + ((r & 1) ? coff1 : coff0)); };
    auto rd_b = [&](auto pc, const char* stg, int r) {
        constexpr int P = decltype(pc)::value;
        const int j = r >> 1;
        const char* half = j < 2 ? stg + OB0 + b_row_off[0] + j * BSTR : stg + OB1 + b_row_off[1] + (j - 2) * BSTR;
        bF[FPL == 2 ? P : 0][FPF ? j : 0][r & 1] = *reinterpret_cast<const bf16x8*>(half + ((r & 1) ? coff1 : coff0));
    };
#define UFV_MMA_STEP(AF, BF, MTB, MCNT, S0, S1, S2, ISSUE, TAILWAIT)                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    UFV_GSTAMP(S0);                                                                                          \
    __builtin_amdgcn_s_barrier();                                                                            \
    UFV_GSTAMP(S1);                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    __builtin_amdgcn_s_setprio(1);                                                                           \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                         \
        _Pragma("unroll") for (int n_ = 0; n_ < NT; ++n_) {                                                  \
            _Pragma("unroll") for (int m_ = 0; m_ < MCNT; ++m_)                                              \
                acc[n_][MTB + m_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[n_][kk], AF[m_][kk], acc[n_][MTB + m_], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                               \
            ISSUE(kk * NT + n_);                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                               \
        }                                                                                                    \
    __builtin_amdgcn_s_setprio(0);                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    UFV_GSTAMP(S2);                                                                                          \
    TAILWAIT();                                                                                              \
    __builtin_amdgcn_s_barrier();

    constexpr int L_F = X2 ? (MA0 == 3 && MA1 == 3 ? 6 : 7) : L_ALL;      // DMA instructions per wave and K-tile in the prefetch forms (XCLS: of waves 0, 1 of a group; 2, 3: one less)
#define UFV_WAIT_G(EXTRA)                                                                                    \
    do {                                                                                                     \
        if (XCLS && !x_hi) wait_vmcnt<FPF ? L_F - (XCLS ? 1 : 0) + (EXTRA) : 0>();                           \
        else wait_vmcnt<FPF ? L_F + (EXTRA) : 0>();                                                          \
    } while (0)
    if constexpr (FPF) {
        if (len > 1) {                                 // as below: [A0 B0 B1] of the first K-tile; its A1 and the second tile's three stay in flight
            if (RELAX_OK && relax == 1) UFV_WAIT_G(RELAX_OK ? NST : 0);
            else if (RELAX_OK && relax == 2) UFV_WAIT_G(RELAX_OK ? NSTW : 0);
            else if (RELAX_OK && relax == 3) UFV_WAIT_G(RELAX_OK ? NSTS : 0);
            else UFV_WAIT_G(0);
        } else if constexpr (X2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the A1 count differs by wave there: wait for all of it
        else wait_vmcnt<T::LA1>();
    } else if constexpr (PH2) {
        constexpr int LH = L_ALL - T::LA1;             // half-tile items: no A1 pieces in flight
        if (HALF && item_half) {
            if (len > 1) {
                if (RELAX_OK && relax == 1) wait_vmcnt<RELAX_OK ? LH + NST : 0>();
                else if (RELAX_OK && relax == 2) wait_vmcnt<RELAX_OK ? LH + NSTW : 0>();
                else if (RELAX_OK && relax == 3) wait_vmcnt<RELAX_OK ? LH + NSTS : 0>();
                else if (RELAX_OK && relax == 4) wait_vmcnt<(RELAX_OK && L_ALL + NSTM <= 63) ? LH + NSTM : LH>();
                else wait_vmcnt<LH>();
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else
        if (len > 1) {                                 // A0 / B0 / B1 of the first K-tile; its A1 and the second tile's three stay in flight
            if (RELAX_OK && relax == 1) wait_vmcnt<RELAX_OK ? L_ALL + NST : 0>();
            else if (RELAX_OK && relax == 2) wait_vmcnt<RELAX_OK ? L_ALL + NSTW : 0>();
            else if (RELAX_OK && relax == 3) wait_vmcnt<RELAX_OK ? L_ALL + NSTS : 0>();
            else if (RELAX_OK && relax == 4) wait_vmcnt<(RELAX_OK && L_ALL + NSTM <= 63) ? L_ALL + NSTM : L_ALL>();
            else wait_vmcnt<L_ALL>();
        } else wait_vmcnt<T::LA1>();
    } else {
        if (len > 1) UFV_WAIT_KEEP_A0_B1();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();      // waves 4-7 run one barrier behind
    UFV_TSTAMP(0);

    UFV_GSTAMP_DECL
    if constexpr (FPF) {
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // the first phase's fragments (the pre-loop wait + barrier above guard them)
#pragma unroll
    for (int r = 0; r < 2 * MA0; ++r) rd_a0(smem, r);
    if constexpr (FPL == 2) {
#pragma unroll
        for (int r = 0; r < 2 * NT; ++r) rd_b(P0{}, smem, r);
    }
    auto ktile = [&](auto pc, int tt) {
        constexpr int P = decltype(pc)::value;                // = tt & 1: the ring stage and the B fragment set of this K-tile
        using PN = std::integral_constant<int, P ^ 1>;
        const int t = k0 + tt, d = P;
        const char* buf = smem + d * STG;
        const char* nbuf = smem + (d ^ 1) * STG;
        // phase A: load step = request A1[t+1], retire A1[t] (behind it: [A0 B0 B1][t+1], A1[t+1] and, at an item's first K-tile, the previous epilogue's stores);
        // MFMAs on (A0, B) with the reads of A1[t] between them
        UFV_GSTAMP(0);
        UFV_GSTAMP(9);
        stage(d ^ 1, 1, t + 1);
        UFV_GSTAMP(10);
        if (tt + 1 < len) {
            if (RELAX_OK && relax == 1 && tt == 0) UFV_WAIT_G(RELAX_OK ? NST : 0);
            else if (RELAX_OK && relax == 2 && tt == 0) UFV_WAIT_G(RELAX_OK ? NSTW : 0);
            else if (RELAX_OK && relax == 3 && tt == 0) UFV_WAIT_G(RELAX_OK ? NSTS : 0);
            else UFV_WAIT_G(0);
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (FPL == 1 || wave < 4) {                    // (level 1: every wave) leading group: B fragments at the classic place (two barriers behind every wave's wait for them); the MFMA
#pragma unroll                                         // step waits for them one fragment at a time (no lgkmcnt(0) at its head: the compiler counts, LDS returns in order)
            for (int r = 0; r < 2 * NT; ++r) rd_b(pc, buf, r);
        }
        constexpr int GRP = 2 * NT, RA = 2 * MA1, PER_A = (RA + GRP - 1) / GRP;
        auto issue_a = [&](int g) {
#pragma unroll
            for (int r = g * PER_A; r < (g + 1) * PER_A; ++r)
                if (r < RA) rd_a1(buf, r);
        };
        auto no_tail = [&]() {};
        UFV_MMA_STEP(aF0, bF[FPL == 2 ? P : 0], 0, MA0, 1, 2, 3, issue_a, no_tail)
        // phase B: load step = request [A0 B0 B1][t+2], retire those of [t+1] (behind them: A1[t+1] and the new three); MFMAs on (A1, B) with the reads of K-tile t+1's
        // A0 and B between them (past the item's last K-tile they read stale LDS into registers nobody uses)
        UFV_GSTAMP(4);
        UFV_GSTAMP(11);
        stage(d, 0, t + 2);
        stage(d, 2, t + 2);
        stage(d, 3, t + 2);
        UFV_GSTAMP(12);
        if (tt + 2 < len) {
            if (RELAX_OK && relax == 1 && tt == 0) UFV_WAIT_G(RELAX_OK ? NST : 0);
            else if (RELAX_OK && relax == 2 && tt == 0) UFV_WAIT_G(RELAX_OK ? NSTW : 0);
            else if (RELAX_OK && relax == 3 && tt == 0) UFV_WAIT_G(RELAX_OK ? NSTS : 0);
            else UFV_WAIT_G(0);
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        constexpr int RB = FPL == 2 ? 2 * (MA0 + NT) : 2 * MA0, PER_B = (RB + GRP - 1) / GRP;
        auto issue_b = [&](int g) {
#pragma unroll
            for (int r = g * PER_B; r < (g + 1) * PER_B; ++r) {
                if (r < 2 * MA0) rd_a0(nbuf, r);
                else if (r < RB) rd_b(PN{}, nbuf, r - 2 * MA0);
            }
        };
        UFV_MMA_STEP(aF1, bF[FPL == 2 ? P : 0], MA0, MA1, 5, 6, 7, issue_b, no_tail)
        UFV_GSTAMP(8);
        UFV_TSTAMP_K(tt);
    };
    for (int tt = 0; tt < len; tt += 2) {
        ktile(P0{}, tt);
        if (tt + 1 < len) ktile(P1{}, tt + 1);
    }
    } else if constexpr (PH2) {
    // One K-tile.  Q = t & 3 as a COMPILE-TIME value in the MX-A kernels (the loop below is unrolled by four there), -1 elsewhere.  MX-A: every fourth K-tile the scales
    // of the next four join the [A0 B0 B1] group two K-tiles ahead (Q == 2) as its YOUNGEST piece; that phase's wait and the next phase A's (Q == 3) keep one more
    // operation in flight.  Why compile-time: with `if ((t + 2) & 3 == 0) stage(...)` as a run-time branch inside the load step hipcc split the step's basic block and
    // SANK phase A's MFMAs below the next load step's waits -- the two wave groups then multiplied at the same time and waited at the same time (fc2: 101 -> 130 us;
    // the listing showed `s_setprio 1; s_setprio 0` with nothing between them).  Left uncounted the piece made two waits per four K-tiles retire an operand piece issued
    // one phase earlier: a memory latency each.
    // SKIPB (HALF kernels, half-tile items): phase B keeps its load step (DMA, counted wait) and its two barriers, and has no fragment reads and no MFMAs
    auto ktile2 = [&](auto qc, auto skipc, int tt) {
        constexpr int Q = decltype(qc)::value;
        constexpr bool SKIPB = decltype(skipc)::value;
        constexpr int LW = L_ALL - (SKIPB ? T::LA1 : 0);          // DMA instructions of one K-tile of this kind of item
        const int t = k0 + tt, d = tt & 1;
        const char* buf = smem + d * 65536;
        // phase A: A0, B0, B1 -> top half; prefetch A1[t+1]; retire A1[t]
        UFV_GSTAMP(0);
        read_b(buf + 32768, H0{});
        read_a(buf, H0{});
        read_b(buf + 49152, H1{});
        if constexpr (MXA) {
            // every fourth K-tile: the dword of each A fragment's row (this lane's K-block, four K-tiles) from the slot of the group; read HERE, in phase A -- the slot is
            // re-staged in a phase B, which the lagging wave group enters one barrier later.  The byte of THIS K-tile is shifted down behind the barrier (asc, in
            // UFV_SYNC_THEN_MMA): a vector instruction that depends on an LDS read, placed in the load step, would pull the compiler's lgkmcnt wait for every fragment
            // read in front of the DMA issue (measured: 159 -> 219 us on the 192 x 192 kernel).
            if constexpr (Q == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int trow = mt < MA0 ? wr * 16 * MA0 + mt * 16 + frow : 32 * MA0 + wr * 16 * MA1 + (mt - MA0) * 16 + frow;
                    asc_raw[mt] = *reinterpret_cast<const unsigned*>(smem + SCL_OFF + ((t >> 2) & 1) * SCL_STRIDE + trow * 16 + fq * 4);
                }
            }
            asc_sh = 8 * (Q < 0 ? 0 : Q);
        }
        UFV_GSTAMP(9);
        stage(d ^ 1, 1, t + 1);
        UFV_GSTAMP(10);
        if (tt + 1 < len) {                            // behind A1[t]: A0/B0/B1[t+1] and A1[t+1] (+ at the item's first K-tile the previous epilogue's stores)
            if (RELAX_OK && relax == 1 && tt == 0) wait_vmcnt<RELAX_OK ? LW + NST : 0>();
            else if (RELAX_OK && relax == 2 && tt == 0) wait_vmcnt<RELAX_OK ? LW + NSTW : 0>();
            else if (RELAX_OK && relax == 3 && tt == 0) wait_vmcnt<RELAX_OK ? LW + NSTS : 0>();
            else if (RELAX_OK && relax == 4 && tt == 0) wait_vmcnt<(RELAX_OK && L_ALL + NSTM <= 63) ? LW + NSTM : LW>();
            else wait_vmcnt<LW + ((MXA && Q == 3) ? 1 : 0)>();          // (Q == 3: the scale piece that joined [A0 B0 B1][t+1]; it was issued iff tt + 1 < len)
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        UFV_SYNC_THEN_MMA(0, NT, 0, MA0, 1, 2, 3)
        // phase B: A1 -> bottom half; prefetch A0 / B0 / B1 [t+2]; retire A0 / B0 / B1 [t+1]
        UFV_GSTAMP(4);
        if constexpr (!SKIPB) read_a(buf + 16384, H1{});
        UFV_GSTAMP(11);
        stage(d, 0, t + 2);
        stage(d, 2, t + 2);
        stage(d, 3, t + 2);
        if constexpr (MXA && Q == 2) stage(d, 4, t + 2);           // K-tiles t + 2 .. t + 5 (issued iff t + 2 < kend, i.e. tt + 2 < len: the branch of the wait below)
        UFV_GSTAMP(12);
        if (tt + 2 < len) {                            // behind them: A1[t+1] and A0/B0/B1[t+2]
            if (RELAX_OK && relax == 1 && tt == 0) wait_vmcnt<RELAX_OK ? LW + NST : 0>();
            else if (RELAX_OK && relax == 2 && tt == 0) wait_vmcnt<RELAX_OK ? LW + NSTW : 0>();
            else if (RELAX_OK && relax == 3 && tt == 0) wait_vmcnt<RELAX_OK ? LW + NSTS : 0>();
            else if (RELAX_OK && relax == 4 && tt == 0) wait_vmcnt<(RELAX_OK && L_ALL + NSTM <= 63) ? LW + NSTM : LW>();
            else wait_vmcnt<LW + ((MXA && Q == 2) ? 1 : 0)>();
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (SKIPB) {
            // the step's two barriers (the other wave group is in its load step between them), nothing to multiply
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        } else {
            UFV_SYNC_THEN_MMA(0, NT, MA0, MA1, 5, 6, 7)
        }
        UFV_GSTAMP(8);
        UFV_TSTAMP_K(tt);
    };
    using NOSKIP = std::integral_constant<bool, false>;
    if constexpr (MXA) {
        for (int tt = 0; tt < len; tt += 4) {           // (every MX-A item starts at K-tile 0: tt & 3 == t & 3)
            ktile2(std::integral_constant<int, 0>{}, NOSKIP{}, tt);
            if (tt + 1 < len) ktile2(std::integral_constant<int, 1>{}, NOSKIP{}, tt + 1);
            if (tt + 2 < len) ktile2(std::integral_constant<int, 2>{}, NOSKIP{}, tt + 2);
            if (tt + 3 < len) ktile2(std::integral_constant<int, 3>{}, NOSKIP{}, tt + 3);
        }
    } else if (HALF && item_half) {                      // (block-uniform; a loop of its own, so that the whole-tile loop's code is what it was)
        if constexpr (HALF)
            for (int tt = 0; tt < len; ++tt) ktile2(std::integral_constant<int, -1>{}, std::integral_constant<bool, true>{}, tt);
    } else {
        for (int tt = 0; tt < len; ++tt) ktile2(std::integral_constant<int, -1>{}, NOSKIP{}, tt);
    }
    } else {
    for (int tt = 0; tt < len; ++tt) {
        const int t = k0 + tt, d = tt & 1;
        const char* buf = smem + d * 65536;
        // phase 0: A0, B0 -> quadrant (0,0); prefetch A1[t+1]
        UFV_GSTAMP(0);
        read_b(buf + 32768, H0{});
        read_a(buf, H0{});
        stage(d ^ 1, 1, t + 1);
        UFV_SYNC_THEN_MMA(0, 2, 0, MA0, 1, 2, 3)
        // phase 1: B1 -> quadrant (0,1); prefetch B0[t+1]
        read_b(buf + 49152, H1{});
        stage(d ^ 1, 2, t + 1);
        UFV_SYNC_THEN_MMA(2, NB1, 0, MA0, 4, 5, 6)
        // phase 2: A1 -> quadrant (1,1); prefetch A0[t+2]
        read_a(buf + 16384, H1{});
        stage(d, 0, t + 2);
        UFV_SYNC_THEN_MMA(2, NB1, MA0, MA1, 7, 8, 9)
        // phase 3: B0 -> quadrant (1,0); prefetch B1[t+2]; retire K-tile t+1
        read_b(buf + 32768, H0{});
        stage(d, 3, t + 2);
        if (tt + 2 < len) UFV_WAIT_KEEP_A0_B1();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        UFV_SYNC_THEN_MMA(0, 2, MA0, MA1, 10, 11, 12)
        UFV_GSTAMP(13);
    }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();      // balance the stagger barrier
    UFV_TSTAMP(5);
    UFV_GSTAMP_FLUSH;
#undef UFV_SYNC_THEN_MMA
#undef UFV_MMA_STEP
#undef UFV_WAIT_G


    // ---- item seam: every wave has finished its LDS reads (final barrier above) -> start the NEXT item's
    //      LDS-DMA prologue now so that it lands under this item's epilogue stores.  The bias / scales of THIS tile are
    //      fetched first so that no ordinary load has to wait behind the DMA queue.
    const bool part_tail = SK && k0 > 0;                 // not the tile's owner: dump the accumulators
    const bool part_head = SK && k0 == 0 && k1 < nk;     // owner of a split tile: add the other parts first
    auto row_of = [&](int mt) { return (mt < MA0 ? wr * 16 * MA0 + mt * 16 : 32 * MA0 + wr * 16 * MA1 + (mt - MA0) * 16) + frow; };
    // column of n-tile nt (clamped so that a half-tile past N reads valid memory; its outputs are not stored)
    auto col_of = [&](int nt) {
        if constexpr (ROPE) return min(n0 + (nt >> 1) * 128, N - 128) + (nt & 1) * 64 + wc * 16 + fq * 4;
        return nt < 2 ? min(n0, N - 128) + wc * 32 + nt * 16 + fq * 4 : min(n0 + 128, N - 64 * NB1) + wc * 16 * NB1 + (nt - 2) * 16 + fq * 4;
    };
    if constexpr (FP8) {     // de-quantise in place: acc *= scale_m[row] * scale_n[col]; per element, so it distributes over
        f32x4 sn[NT];        // the stream-K partial sums (every part scales its own accumulators)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) sn[nt] = *reinterpret_cast<const f32x4*>(e.scale_n + col_of(nt));
        float sm[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) sm[mt] = MXA ? 1.0f : e.scale_m[min(m0 + row_of(mt), M - 1)];      // (MXA: the block scales were applied by the MFMAs)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt][mt] *= sn[nt] * sm[mt];
    }
    f32x4 bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = f32x4{0, 0, 0, 0};
    if (!part_tail && !SWIGLU && e.bias && (!KSPL || item_part == 0)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bias[nt] = *reinterpret_cast<const f32x4*>(e.bias + col_of(nt));
        // The wait names the bias registers as its outputs: to hipcc they are then values an asm statement produced, not results of loads that may still be
        // in flight.  Without that the compiler, which cannot see this wait, put its own `s_waitcnt vmcnt(0)` in front of the first use of a bias register
        // in EVERY (row, run) block of the epilogue -- behind the previous block's store, so each of a wave's 16 output stores was drained before the next
        // one was even computed (round 4: 16 x vmcnt(0) per tile in the ISA of every epilogue256_wide variant; the epilogue took 7.4 - 9.2 k ticks of a
        // 61 k-tick tile where the chip's write rate allows 4.8 k).
        if constexpr (NT == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2]), "+v"(bias[3])::"memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2])::"memory");
    }
    // ROPE: cos / sin of this lane's accumulator rows (dims wc * 16 + fq * 4 .. + 3), requested -- like the bias -- BEFORE the next tile's prologue DMA so that no
    // ordinary load queues behind it, and handed on as asm outputs so that the compiler puts no wait of its own in front of their first use in the epilogue
    f32x4 rcos[ROPE ? MT : 1], rsin[ROPE ? MT : 1];
    if constexpr (ROPE) {
        const bool any_roped = (n0 >> 7) < e.rope_hq + e.rope_hkv;          // a tile of two v heads rotates nothing
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            rcos[mt] = f32x4{1.f, 1.f, 1.f, 1.f}; rsin[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (any_roped) {
                const float* tr = e.rope_tab + (size_t)min(m0 + row_of(mt), M - 1) * 128 + wc * 16 + fq * 4;
                rcos[mt] = *reinterpret_cast<const f32x4*>(tr);
                rsin[mt] = *reinterpret_cast<const f32x4*>(tr + 64);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(rcos[mt]), "+v"(rsin[mt]));
    }
    const int cm0 = m0, cn0 = n0, cpart = item_part, ctile = item_tile, clen = len;
    const int rowlim = (HALF && item_half) ? min(M, m0 + 32 * MA0) : M;      // row limit of THIS item's epilogue: a half-tile item ends behind its 32 MA0 rows (the A1 accumulators are dropped)
    UFV_TSTAMP(8);
    have = next_item(m0, n0, k0, k1);
    if (have) { set_src(m0, n0); kbeg = k0; kend = k1; UFV_TSTAMP(9); prologue_loads(); }
    UFV_TSTAMP(6);
    if constexpr (SK) if (part_head) {
        // this block's range ended inside the tile: the following blocks hold the rest, in order
        int covered = len;                          // K-tiles of the tile accounted for so far (this block's k0 was 0)
        for (int q = pos + 1; covered < nk; ++q) {
            if (tid == 0) turn_wait_eq(sk.flags + q, sk.epoch, sk.err);
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
        UFV_ACT_SWITCH(pe.act, (epilogue256<OUT_F32, SWIGLU, ACT_, true, MA0, MA1, NB1>(acc, pe, eM, eN, em0, en0, wr, wc, frow, fq, bias)))
        if (part_tail) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave's write-through stores are acknowledged ...
            __syncthreads();                                        // ... before one thread publishes the slot
            if (tid == 0) __hip_atomic_store(sk.flags + pos, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if constexpr (KSPL) {
        // turn order: part p of a tile stores after part p - 1 has (flag = base + p); a part with an empty K range only passes the turn on
        if (cpart > 0) {
            if (tid == 0) turn_wait_ge(sk.flags + ctile, sk.epoch + cpart, sk.err);
            __syncthreads();
        }
        if (clen > 0 || cpart == 0) epilogue_split<MA0, MA1, NB1>(acc, e, M, N, cm0, cn0, wr, wc, frow, fq, bias, cpart == 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every wave's write-through stores are acknowledged ...
        __syncthreads();                                            // ... before one thread passes the turn on
        if (tid == 0) __hip_atomic_store(sk.flags + ctile, sk.epoch + cpart + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        bool by_rows = false;
        if constexpr (!SWIGLU && !FP8) by_rows = e.resid != nullptr && e.act == ACT_NONE;
        bool swide = false;
        if constexpr (!OUT_F32 && SWIGLU && MA0 % 2 == 0 && MA1 % 2 == 0 && NB1 == 2)
            swide = e.resid == nullptr && e.ldc % 8 == 0 && ((uintptr_t)e.out & 15) == 0 && (int64_t)M * e.ldc < (1ll << 30);
        bool wide = false;
        if constexpr (!OUT_F32 && !SWIGLU) wide = e.resid == nullptr && e.ldc % 8 == 0 && ((uintptr_t)e.out & 15) == 0 && (int64_t)M * e.ldc < (1ll << 30);
        const bool interior = cm0 + BM <= rowlim && cn0 + BN <= N;     // every store instruction of the epilogue is issued
        if constexpr (ROPE) {
            epilogue256_rope<MA0, MA1>(acc, e, rowlim, N, cm0, cn0, wr, wc, frow, fq, bias, rcos, rsin);
            relax = RELAX_OK ? 2 : 0;                             // 2 MT sixteen-byte buffer stores, every one issued on every tile
        } else if constexpr ((MX & 2) != 0) {
            if constexpr (SWIGLU) epilogue256_swiglu_mx<MA0, MA1>(acc, e, rowlim, N, cm0, cn0, wr, wc, frow, fq);
            else { UFV_ACT_SWITCH(e.act, (epilogue256_wide_mx<ACT_, MA0, MA1>(acc, e, rowlim, N, cm0, cn0, wr, wc, frow, fq, bias))) }
            relax = RELAX_OK ? 4 : 0;
        } else if (by_rows) {
            if constexpr (!SWIGLU && !FP8) {
                if constexpr (!OUT_F32) {
                    if (e.resid_bf16) epilogue256_resid<OUT_F32, MA0, MA1, NB1, true>(acc, e, rowlim, N, cm0, cn0, wr, wc, frow, fq, bias);
                    else epilogue256_resid<OUT_F32, MA0, MA1, NB1>(acc, e, rowlim, N, cm0, cn0, wr, wc, frow, fq, bias);
                } else {
                    epilogue256_resid<OUT_F32, MA0, MA1, NB1>(acc, e, rowlim, N, cm0, cn0, wr, wc, frow, fq, bias);
                }
            }
            relax = RELAX_OK ? 1 : 0;                             // this form issues every store instruction, edge tiles included
        } else if (swide) {
            if constexpr (!OUT_F32 && SWIGLU && MA0 % 2 == 0 && MA1 % 2 == 0 && NB1 == 2) epilogue256_swiglu_wide<MA0, MA1>(acc, e, rowlim, N, cm0, cn0, wr, wc, frow, fq);
            relax = RELAX_OK ? 3 : 0;                             // every store instruction is issued on every tile (round 5: dropped offsets, no branches)
        } else if (wide) {
            if constexpr (!OUT_F32 && !SWIGLU) { UFV_ACT_SWITCH(e.act, (epilogue256_wide<ACT_, MA0, MA1, NB1>(acc, e, rowlim, N, cm0, cn0, wr, wc, frow, fq, bias))) }
            relax = RELAX_OK ? 2 : 0;                             // buffer stores: every store instruction is issued on row- and column-edge tiles too (rows >= M and a half past N are dropped by the descriptor)
        } else {
            UFV_ACT_SWITCH(e.act, (epilogue256<OUT_F32, SWIGLU, ACT_, false, MA0, MA1, NB1>(acc, e, rowlim, N, cm0, cn0, wr, wc, frow, fq, bias)))
            relax = RELAX_OK && interior ? 1 : 0;
        }
    }
    UFV_TSTAMP(7);
    UFV_TSTAMP_NEXT
    }   // persistent tile loop
#undef UFV_WAIT_KEEP_A0_B1
}

}  // namespace


static inline int pp_n_cu() { return ufv_dev_n_cu(); }

// row-tiles per tile-order group: all of them when there are few (M = 2399 -> 10 row tiles: a ragged second group of 2 rows
// made its XCDs fetch 16 W panels per round instead of 4; 1203 -> 1244 TF/s on gate/up), else ~8 in equal groups
static inline int pp_group(int tiles_m) { return tiles_m <= 16 ? tiles_m : cdiv(tiles_m, cdiv(tiles_m, 8)); }

template <bool F, bool S, bool Q, int MA0, int MA1, int NB1, bool PH2 = false, bool KSPL = false, bool ROPE = false, int MX = 0, bool HALF = false>
static int launch_pp(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, hipStream_t st, int ksplit = 1) {
    constexpr int SMEM = SMEM256 + (((MX & 1) && NB1 != 1) ? 8192 : 0);          // MX & 1: two slots of A block scales (four K-tiles each) above the operand ring unless they fit inside it
    UFV_ONCE_PER_DEVICE(
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_256<F, S, Q, false, MA0, MA1, NB1, PH2, KSPL, ROPE, MX, HALF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    );
    using T = PP<MA0, MA1, NB1>;
    const int rem = N % T::BN;
    if (rem != 0 && rem != 128) {
        ufv_set_error("ufv_gemm: the %dx%d tile needs N %% %d in {0, 128} (N=%d)", T::BM, T::BN, T::BN, N);
        return UFV_EUNSUPPORTED;
    }
    const int tiles_m = cdiv(M, T::BM), tiles = tiles_m * cdiv(N, T::BN), n_cu = pp_n_cu();
    StreamK sk = {nullptr, nullptr, 0, pp_group(tiles_m), 1, nullptr, 0};
    int items = tiles;
    if constexpr (HALF) {
        // the launch's last round as half-tile items (gemm_nt_256: HALF) when the leftover whole tiles (two items each) and the tiles of a last row band with <= 32 MA0
        // valid rows (one item each) fit ONE round: the round then costs what half a tile costs instead of a whole one.  
        const int tiles_n = cdiv(N, T::BN), light = (M - (tiles_m - 1) * T::BM <= 32 * MA0) ? 1 : 0;
        const int n_full = (tiles_m - light) * tiles_n, n_light = light * tiles_n, left = n_full % n_cu;
        // OPT-IN (UFV_GEMM_HALF=1, read per call): measured -1.2 % on the gate/up launch at M = 2399 for +14 % fabric traffic (the band's 148 tiles leave their W panels'
        // rounds: 1443 against 1265 MB per launch by the PMC counters) -- a half-tile item costs 0.85 of a whole tile, not 0.5, because the K loop's issue-to-wait distance is one
        // K-tile (LABNOTES round 6).  Not the default.
        const char* half_env = getenv("UFV_GEMM_HALF");
        if (tiles >= n_cu && left + n_light > 0 && 2 * left + n_light <= n_cu && half_env != nullptr && half_env[0] == '1') {
            sk.half = 1;
            sk.gm = pp_group(tiles_m - light);
        }
    }
    if constexpr (KSPL) {
        const int nk = K / 64;
        if (e.act != ACT_NONE || ksplit < 2 || ksplit > 32 || (ksplit - 1) * cdiv(nk, ksplit) >= nk) {
            ufv_set_error("ufv_gemm: split-K needs an activation-free epilogue and 2 <= parts <= 32 non-empty K ranges (parts=%d, K-tiles=%d)", ksplit, nk);
            return UFV_EUNSUPPORTED;
        }
        const int rc = ufv_splitk_acquire(tiles, ksplit, &sk.flags, &sk.epoch, &sk.err);      // this launch's own flag slice and ticket base
        if (rc != UFV_OK) return rc;
        sk.ksplit = ksplit;
        items = tiles * ksplit;
    }
    hipLaunchKernelGGL((gemm_nt_256<F, S, Q, false, MA0, MA1, NB1, PH2, KSPL, ROPE, MX, HALF>), dim3(items < n_cu ? items : n_cu), dim3(512), SMEM, st, A, W, e, M, N, K,
                       lda, ldw, sk);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

// the named tile shapes live in their own translation units (gemm256_b.hip: bf16 operands, gemm256_q.hip: e4m3), one build job each
int ufv_launch_pp_shape_bf16(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, bool out_f32, int shape, hipStream_t st);
int ufv_launch_pp_shape_fp8(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, bool out_f32, int shape, hipStream_t st);
// gemm256_s.hip: the split-K instantiations (bf16 operands, fp32 output)
int ufv_launch_pp_split(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, int shape, int ksplit, hipStream_t st);

template <bool F, bool Q>
static int launch_pp_shape(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, int shape, hipStream_t st) {
    switch (shape) {                      // 1000 + MA0 MA1 NB1 as decimal digits (two phases per K-tile)
        case 1322: return launch_pp<F, false, Q, 3, 2, 2, true>(A, W, e, M, N, K, lda, ldw, st);      // 160 x 256
        case 1332: return launch_pp<F, false, Q, 3, 3, 2, true>(A, W, e, M, N, K, lda, ldw, st);      // 192 x 256
        case 1432: return launch_pp<F, false, Q, 4, 3, 2, true>(A, W, e, M, N, K, lda, ldw, st);      // 224 x 256
        case 1331: return launch_pp<F, false, Q, 3, 3, 1, true>(A, W, e, M, N, K, lda, ldw, st);      // 192 x 192
        case 1431: return launch_pp<F, false, Q, 4, 3, 1, true>(A, W, e, M, N, K, lda, ldw, st);      // 224 x 192
        case 1441: return launch_pp<F, false, Q, 4, 4, 1, true>(A, W, e, M, N, K, lda, ldw, st);      // 256 x 192
    }
    ufv_set_error("ufv_gemm: unknown ping-pong tile shape %d", shape);
    return UFV_EINVAL;
}

