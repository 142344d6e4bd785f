#!/usr/bin/env python3
"""Generator of the causal GQA attention kernel for head_dim 128 (the decoder's prefill): writes ufvideo_amd/csrc/attn_c128_asm.inc -- one
inline-asm statement per wave role, 256 registers per wave, 8 waves per block:

  waves 0..3  COMPUTE: one 32-query unit (one q head, one 32-row slice) each, Q^T fragments in registers, key tiles of 64 streamed from a
              4-stage LDS ring; software pipeline ACROSS TILES: phase A = [QK^T of tile t+1] with the exp / pack / row-sum of tile t in the MFMA
              gaps, phase B = [PV of tile t] with the row max and rescale decision of tile t+1 in the gaps; one s_barrier per tile.
  waves 4..7  LOADERS: issue every K / V LDS-DMA piece of the block (buffer_load ... lds, 10 pieces per wave and tile: K rows at pitch 272 B, V rows
              at pitch 320 B = the conflict-free images of attn_fwd_mfma), two tiles ahead; the compute waves never issue a vector-memory
              instruction inside the key loop.

Why this shape (LABNOTES.md, round 3): at S = 2399 / 28 q heads the launch has 40 432 (32-row x 64-key) unit-tiles for 1024 SIMDs = 39.5 each, and
the longest unit is 38 tiles long: a SIMD has to spend the whole launch on ONE heavy unit (plus a light one), so a unit must run alone on its SIMD
at the MFMA rate, and the K/V tile it shares with the other three SIMDs' units (same kv head: GQA) has to be loaded by somebody else.
Work items = quads of 4 units of one kv head with (almost) equal length, sorted by length, dealt to the blocks in serpentine order (block b takes
items b, 2G-1-b, 2G+b, ...): every block ends within ~3 tiles of the others without a queue.

Numerics = attn_fwd_mfma<128, 4, causal, NG = 1> exactly (raw scores, max * scale*log2e, deferred rescale at 2^6 per 32-query unit with a per-lane
max, p = exp2(fma(s, c, -m)), row sums in fp32 in the same order, bf16 P, IEEE 1/l): outputs are bit-identical to that kernel.
"""
import os
import sys

HD = 128
PK, PV = 272, 320                      # K / V row pitch in LDS (17 / 20 chunks of 16 B; the last 1 / 4 chunks of a row are padding)
KT, VT = 64 * PK, 64 * PV              # 17408, 20480
STG = KT + VT                          # 37888 = 37 pieces of 1 KiB
NSTAGE = 4
SCRATCH = NSTAGE * STG                 # 3 KiB: destination of the 3 padding pieces that make 40 = 4 x 10 per tile
LDS_BYTES = SCRATCH + 3 * 1024 + 64
THR = 0x40C00000                       # 6.0f
OPT = os.environ.get("UFV_C128_OPT", "").split(",")
NOVPRE = "novpre" in OPT
PRE_A = int(os.environ.get("UFV_C128_PRE", "28"))
DROP = os.environ.get("UFV_C128_DROP", "").split(",")
STAMPS = "--stamps" in sys.argv       # lab build: s_memtime deltas of the compute waves summed per segment kind -> [block][wave][8] dwords at %[stp]
S_NOW, S_PREV, S_DLT = 88, 90, 91      # the V descriptor's SGPRs (loader-only); sums in a120..a127      # lab only: timing experiments with parts of the work removed (results are wrong)


def vr(n, c=1): return f"v{n}" if c == 1 else f"v[{n}:{n + c - 1}]"
def ar(n, c=1): return f"a{n}" if c == 1 else f"a[{n}:{n + c - 1}]"
def sr(n, c=1): return f"s{n}" if c == 1 else f"s[{n}:{n + c - 1}]"


# ---- compute-wave registers: VGPR 0..143, AGPR 0..111 ------------------------------------------------------------------------
def S(slot, r): return 32 * slot + r                 # v0..63: scores, two slots
def P(k): return 64 + k                              # v64..79: packed bf16 P of the tile being multiplied
def Q(ks, i=0): return 80 + 4 * ks + i               # v80..111: Q^T fragments, 8 k-steps
T0, T1, TMAX, MRUN, LRUN, PSUM, MSUB, TA, TB, TC, TD = range(112, 123)
KADDR, VADDR, QI, KJ, TE = range(123, 128)           # KJ: key index of register 0 of this lane inside a tile: 4 h (+ (r & 3) + 8 (r >> 2))
INV = PSUM
# prologue-only values (computed in score registers, parked in AGPRs: a compute wave has 128 + 128 registers at 2 waves / SIMD)
LANE, L31, HH, QOFF, OOFF, TG, TF = range(40, 47)
A_K0, A_V0, A_L31, A_QOFF, A_OOFF = range(112, 117)
def O(dt, r=0): return 16 * dt + r                   # a0..63
NSLOT = 12                                                          # a64..111: ONE ring of 12 four-register fragment slots shared by the K and the V^T stream
def RING(i, j=0): return 64 + 4 * (i % NSLOT) + j
LEAD = int(os.environ.get("UFV_C128_LEAD", "10"))                  # fragments requested ahead of the one being multiplied (<= NSLOT - 2: a slot is rewritten two MFMAs after its reader)
MAXFLY = 15                                                         # LDS instructions in flight per wave (lgkmcnt has 4 bits)

# ---- loader-wave registers: v0..9 source offsets of its 10 pieces, v10.. temps
# ---- fixed SGPRs (s40..s99)
(S_Q0, S_Q1, S_K0, S_K1, S_V0, S_V1, S_O0, S_O1, S_QSS, S_KSS, S_VSS, S_OSS, S_S, S_R, S_HKV, S_NS, S_NI, S_MR, S_MHKV, S_SL2, S_LDS,
 S_WAVE, S_BID, S_G, S_ROUND, S_ITEM, S_QD, S_GRP, S_TQ, S_TW, S_T, S_U, S_SLICE, S_HEAD, S_Q0ROW, S_VALID, S_STAGE, S_DELTA, S_TMP, S_TMP2,
 S_TMP3, S_SOFF) = range(40, 82)
S_KR, S_VR, S_QR, S_OR = 84, 88, 92, 96
S_CNT = 38
S_REM, S_OSOFF, S_PAR, S_QV = 84, 85, 86, 87       # compute-only (the K descriptor's registers of the loader waves)
S_LSE = 88                                         # s[88:89], compute-only (the V descriptor's registers of the loader waves): lse pointer
S_LSEOFF, S_ROWS = 80, 77                          # (= S_TMP3, S_DELTA: free between the last tile's set-up and the next item)
S_STG = 39                                      # holds STG        # buffer descriptors (4 SGPRs each)
S_ANY = 82                                      # s[82:83]: rescale decision mask


class Gen:
    def __init__(self):
        self.lines = []
        self.n = 0

    def e(self, s):
        self.lines.append(s)

    def label(self, base):
        self.n += 1
        return f"{base}_{self.n}%="


def spread(instrs, g0, g1):
    n = len(instrs)
    return [(g0 + (k * (g1 - g0 + 1)) // max(n, 1), ins) for k, ins in enumerate(instrs)]


def emit(G, ins):
    if isinstance(ins, (list, tuple)):
        for x in ins:
            emit(G, x)
    else:
        G.e(ins)


# ---- phase A: QK^T of the next tile (K fragments streamed from LDS) --------------------------------------------------------
def frag_stream(slot, want_qk, want_pv, i0=0):
    """the fragment stream of one iteration: 16 K fragments (k-step ks, key half) each feeding one QK^T MFMA into S[slot], then 16 V^T fragments
    (c outer, dt inner: per accumulator the chunks in the order c = 0..3 of attn_fwd_mfma, consecutive MFMAs independent) each feeding one PV MFMA.
    Returns (head, groups): head = the reads issued before anything is multiplied, groups[i] = [wait, MFMA, reads issued after it]."""
    fr = []                                            # (read instructions, mfma)
    n = i0
    if want_qk:
        for ks in range(8):
            for half in range(2):
                d = vr(S(slot, 16 * half), 16)
                fr.append(([f"ds_read_b128 {ar(RING(n), 4)}, {vr(KADDR)} offset:{ks * 32 + half * 32 * PK}"],
                           f"v_mfma_f32_32x32x16_bf16 {d}, {ar(RING(n), 4)}, {vr(Q(ks), 4)}, {'0' if ks == 0 else d}"))
                n += 1
    if want_pv:
        for c in range(4):
            for dt in range(4):
                off = dt * 64 + c * 16 * PV
                fr.append(([f"ds_read_b64_tr_b16 {ar(RING(n, 0), 2)}, {vr(VADDR)} offset:{off}", f"ds_read_b64_tr_b16 {ar(RING(n, 2), 2)}, {vr(VADDR)} offset:{off + 8 * PV}"],
                           f"v_mfma_f32_32x32x16_bf16 {ar(O(dt), 16)}, {ar(RING(n), 4)}, {vr(P(4 * c), 4)}, {ar(O(dt), 16)}"))
                n += 1
    nf = len(fr)
    issued = 0
    fly = []                                           # instruction counts of the fragments in flight, oldest first
    def pump(i):
        nonlocal issued
        out = []
        while issued < nf and issued - i < LEAD and sum(fly) + len(fr[issued][0]) <= MAXFLY:
            out += fr[issued][0]; fly.append(len(fr[issued][0])); issued += 1
        return out
    head = pump(0)
    groups = []
    for i in range(nf):
        if issued <= i:                                # (cannot happen with LEAD >= 1)
            raise RuntimeError("fragment not requested")
        later = sum(fly[1:])
        g = [f"s_waitcnt lgkmcnt({later})", fr[i][1]]
        fly.pop(0)
        g += pump(i + 1)
        groups.append(g)
    if "lds" in DROP:
        head = []
        groups = [[x for x in g if not x.startswith("ds_read")] for g in groups]
    return head, groups


def finish_softmax(slot):
    """P = exp2(fma(s, c, -msub)), psum in the order of attn_fwd_mfma (psum += s0[r] + s1[r], r = 0..15), l_run += psum, bf16 pack -- one sequence
    (the wave's last tile and --simple)"""
    a, b1, b2, b3 = finish_parts(slot)
    return a + b1 + b2 + b3


def finish_parts(slot):
    """the same work in four parts: A (beside the QK^T of the next tile) = all the FMAs, the exps and packs of the first key half (P chunks 0, 1);
    B1 / B2 (beside the first PV MFMAs, which use chunks 0, 1) = exps and packs of chunk 2 / chunk 3; B3 = the row sums, in the order of the reference kernel"""
    s = lambda r: vr(S(slot, r))
    a = [f"v_cmp_lg_f32 vcc, 0xff800000, {vr(MRUN)}", "s_nop 1", f"v_cndmask_b32 {vr(MSUB)}, 0, {vr(MRUN)}, vcc"]
    for r in range(16):
        a.append(f"v_fma_f32 {s(r)}, {s(r)}, {sr(S_SL2)}, -{vr(MSUB)}")
    for r in range(16):
        a.append(f"v_exp_f32 {s(r)}, {s(r)}")
        a.append(f"v_fma_f32 {s(16 + r)}, {s(16 + r)}, {sr(S_SL2)}, -{vr(MSUB)}")
    a += [f"v_cvt_pk_bf16_f32 {vr(P(j))}, {vr(S(slot, 2 * j))}, {vr(S(slot, 2 * j + 1))}" for j in range(8)]
    b1 = [f"v_exp_f32 {s(16 + r)}, {s(16 + r)}" for r in range(8)] + [f"v_cvt_pk_bf16_f32 {vr(P(8 + j))}, {vr(S(slot, 16 + 2 * j))}, {vr(S(slot, 17 + 2 * j))}" for j in range(4)]
    b2 = [f"v_exp_f32 {s(24 + r)}, {s(24 + r)}" for r in range(8)] + [f"v_cvt_pk_bf16_f32 {vr(P(12 + j))}, {vr(S(slot, 24 + 2 * j))}, {vr(S(slot, 25 + 2 * j))}" for j in range(4)]
    # psum = ((s0[0] + s1[0]) + (s0[1] + s1[1])) + ... : the pair sums run one ahead of the chain, in two temporaries, so that no instruction reads
    # the result of the one right before it
    tmp = [TA, TD]
    b3 = [f"v_add_f32 {vr(PSUM)}, {s(0)}, {s(16)}", f"v_add_f32 {vr(tmp[1])}, {s(1)}, {s(17)}"]
    for r in range(1, 16):
        if r + 1 < 16:
            b3.append(f"v_add_f32 {vr(tmp[(r + 1) % 2])}, {s(r + 1)}, {s(17 + r)}")
        b3.append(f"v_add_f32 {vr(PSUM)}, {vr(PSUM)}, {vr(tmp[r % 2])}")
    b3.append(f"v_add_f32 {vr(LRUN)}, {vr(LRUN)}, {vr(PSUM)}")
    return a, b1, b2, b3


def mask_tile(slot):
    """diagonal tile: key index kj = key0 + (r & 3) + 8 (r >> 2) + 4 h (+ 32 for the second half) must be <= the lane's query row (QI holds qi - key0)"""
    o = []
    for r in range(16):
        kk = (r & 3) + 8 * (r >> 2)
        o += [f"v_add_u32 {vr(TC)}, {kk}, {vr(KJ)}", f"v_cmp_le_i32 vcc, {vr(TC)}, {vr(QI)}", "s_nop 1",
              f"v_cndmask_b32 {vr(S(slot, r))}, {vr(TE)}, {vr(S(slot, r))}, vcc",
              f"v_add_u32 {vr(TC)}, {kk + 32}, {vr(KJ)}", f"v_cmp_le_i32 vcc, {vr(TC)}, {vr(QI)}", "s_nop 1",
              f"v_cndmask_b32 {vr(S(slot, 16 + r))}, {vr(TE)}, {vr(S(slot, 16 + r))}, vcc"]
    return o          # TE holds -inf


def start_softmax(slot):
    """tile max (scaled) of the scores in `slot` -> TMAX; decision mask of the deferred rescale -> S_ANY"""
    s = lambda r: vr(S(slot, r))
    o = [f"v_max_f32 {vr(T0)}, {s(0)}, {s(16)}", f"v_max_f32 {vr(T1)}, {s(1)}, {s(17)}"]
    for r in range(2, 16):
        t = T0 if r % 2 == 0 else T1
        o.append(f"v_max3_f32 {vr(t)}, {vr(t)}, {s(r)}, {s(16 + r)}")
    o += [f"v_max_f32 {vr(T0)}, {vr(T0)}, {vr(T1)}", f"v_mov_b32 {vr(T1)}, {vr(T0)}", "s_nop 1", f"v_permlane32_swap_b32 {vr(T0)}, {vr(T1)}",
          f"v_max_f32 {vr(T0)}, {vr(T0)}, {vr(T1)}", f"v_mul_f32 {vr(TMAX)}, {sr(S_SL2)}, {vr(T0)}",
          f"v_add_f32 {vr(TB)}, 0x{THR:08x}, {vr(MRUN)}", f"v_cmp_gt_f32 {sr(S_ANY, 2)}, {vr(TMAX)}, {vr(TB)}"]
    return o


def rescale_stub():
    """m_new = max(m_run, tmax); alpha = exp2(m_run - m_new); l_run *= alpha; O *= alpha; m_run = m_new (per lane)"""
    o = ["s_nop 7", "s_nop 7", "s_nop 7",                                  # the PV MFMAs in front have written O
         f"v_max_f32 {vr(TB)}, {vr(MRUN)}, {vr(TMAX)}", f"v_sub_f32 {vr(TC)}, {vr(MRUN)}, {vr(TB)}", f"v_exp_f32 {vr(TC)}, {vr(TC)}",
         f"v_mov_b32 {vr(MRUN)}, {vr(TB)}", f"v_mul_f32 {vr(LRUN)}, {vr(LRUN)}, {vr(TC)}"]
    for r in range(64):
        o += [f"v_accvgpr_read_b32 {vr(TD)}, {ar(r)}", "s_nop 0", f"v_mul_f32 {vr(TD)}, {vr(TD)}, {vr(TC)}", "s_nop 0", f"v_accvgpr_write_b32 {ar(r)}, {vr(TD)}"]
    o += ["s_nop 1"]
    return o


def stamp(k):
    if not STAMPS:
        return []
    return [f"s_memtime {sr(S_NOW, 2)}", "s_waitcnt lgkmcnt(0)", f"s_sub_u32 {sr(S_DLT)}, {sr(S_NOW)}, {sr(S_PREV)}", f"v_accvgpr_read_b32 {vr(TD)}, {ar(120 + k)}", "s_nop 0",
            f"v_add_u32 {vr(TD)}, {sr(S_DLT)}, {vr(TD)}", "s_nop 0", f"v_accvgpr_write_b32 {ar(120 + k)}, {vr(TD)}", f"s_mov_b32 {sr(S_PREV)}, {sr(S_NOW)}"]


def build(simple=False):
    G = Gen()
    e = G.e
    # ============ common prologue =====================================================================================================
    ins = [f"s_mov_b32 {sr(d)}, %[{n}]" for d, n in (
        (S_Q0, "q0"), (S_Q1, "q1"), (S_K0, "k0"), (S_K1, "k1"), (S_V0, "v0"), (S_V1, "v1"), (S_O0, "o0"), (S_O1, "o1"), (S_QSS, "qss"), (S_KSS, "kss"),
        (S_VSS, "vss"), (S_OSS, "oss"), (S_S, "seq"), (S_R, "rr"), (S_HKV, "hkv"), (S_NS, "ns"), (S_NI, "ni"), (S_MR, "mr"), (S_MHKV, "mhkv"), (S_LDS, "lds"),
        (S_WAVE, "wave"), (S_BID, "bid"), (S_G, "grid"))]
    ins += [f"v_mov_b32 {vr(TA)}, %[scale]", f"v_mul_f32 {vr(TA)}, 0x3fb8aa3b, {vr(TA)}", "s_nop 0", f"v_readfirstlane_b32 {sr(S_SL2)}, {vr(TA)}",
            f"s_mov_b32 {sr(S_ROUND)}, 0", f"s_mov_b32 {sr(S_STG)}, {STG}", f"s_mov_b32 {sr(S_VALID)}, 0",
            # optional fp32 [Hq, S] output: log2-domain log-sum-exp per (head, query) for the fused backward (0 = none; the stamps build uses the operand for its buffer)
            f"s_mov_b32 {sr(S_LSE)}, {'0' if STAMPS else '%[stp0]'}", f"s_mov_b32 {sr(S_LSE + 1)}, {'0' if STAMPS else '%[stp1]'}",
            f"v_mbcnt_lo_u32_b32 {vr(LANE)}, -1, 0", f"v_mbcnt_hi_u32_b32 {vr(LANE)}, -1, {vr(LANE)}"]
    for i in ins:
        e(i)
    e(f"s_cmp_ge_u32 {sr(S_WAVE)}, 4")
    e("s_cbranch_scc1 LOADER%=")

    # ============ compute waves =======================================================================================================
    for i in [f"v_and_b32 {vr(L31)}, 31, {vr(LANE)}", f"v_lshrrev_b32 {vr(HH)}, 5, {vr(LANE)}",
              f"v_mul_u32_u24 {vr(TA)}, {PK}, {vr(L31)}", f"v_lshlrev_b32 {vr(TB)}, 4, {vr(HH)}", f"v_add3_u32 {vr(TG)}, {vr(TA)}, {vr(TB)}, {sr(S_LDS)}",   # TG = K address in stage 0
              # V: (4 h + ((lane & 15) >> 2)) * PV + (16 ((lane >> 4) & 1) + 4 (lane & 3)) * 2 + KT
              f"v_and_b32 {vr(TA)}, 15, {vr(LANE)}", f"v_lshrrev_b32 {vr(TA)}, 2, {vr(TA)}", f"v_lshl_add_u32 {vr(TA)}, {vr(HH)}, 2, {vr(TA)}",
              f"v_mul_u32_u24 {vr(TA)}, {PV}, {vr(TA)}", f"v_bfe_u32 {vr(TB)}, {vr(LANE)}, 4, 1", f"v_lshlrev_b32 {vr(TB)}, 5, {vr(TB)}",
              f"v_and_b32 {vr(TC)}, 3, {vr(LANE)}", f"v_lshl_add_u32 {vr(TB)}, {vr(TC)}, 3, {vr(TB)}", f"v_add3_u32 {vr(TF)}, {vr(TA)}, {vr(TB)}, {sr(S_LDS)}",
              f"v_add_u32 {vr(TF)}, {KT}, {vr(TF)}",                                                                                                  # TF = V address in stage 0
              f"v_mul_lo_u32 {vr(TA)}, {vr(L31)}, {sr(S_QSS)}", f"v_lshl_add_u32 {vr(QOFF)}, {vr(HH)}, 4, {vr(TA)}",
              f"v_mul_lo_u32 {vr(TA)}, {vr(L31)}, {sr(S_OSS)}", f"v_lshl_add_u32 {vr(OOFF)}, {vr(HH)}, 4, {vr(TA)}",
              f"v_lshlrev_b32 {vr(KJ)}, 2, {vr(HH)}", f"v_mov_b32 {vr(TE)}, 0xff800000",
              f"v_accvgpr_write_b32 {ar(A_K0)}, {vr(TG)}", f"v_accvgpr_write_b32 {ar(A_V0)}, {vr(TF)}", f"v_accvgpr_write_b32 {ar(A_L31)}, {vr(L31)}",
              f"v_accvgpr_write_b32 {ar(A_QOFF)}, {vr(QOFF)}", f"v_accvgpr_write_b32 {ar(A_OOFF)}, {vr(OOFF)}"]:
        e(i)
    # KADDR / VADDR are rebuilt from the parked stage-0 addresses at every item start
    if STAMPS:
        for k in range(8):
            e(f"v_accvgpr_write_b32 {ar(120 + k)}, 0")
        e(f"s_memtime {sr(S_NOW, 2)}"); e("s_waitcnt lgkmcnt(0)"); e(f"s_mov_b32 {sr(S_PREV)}, {sr(S_NOW)}")
    def unit_decode():
        """unit of this wave in the decoded item: u = 4 q + wave; slice = NS - 1 - u / R; head = g R + u % R; T_w = tiles of the slice (0: no unit)"""
        for i in [f"s_lshl_b32 {sr(S_U)}, {sr(S_QD)}, 2", f"s_add_u32 {sr(S_U)}, {sr(S_U)}, {sr(S_WAVE)}",
                  f"s_mul_hi_u32 {sr(S_TMP)}, {sr(S_U)}, {sr(S_MR)}",                    # u / R
                  f"s_mul_i32 {sr(S_TMP2)}, {sr(S_TMP)}, {sr(S_R)}", f"s_sub_u32 {sr(S_TMP2)}, {sr(S_U)}, {sr(S_TMP2)}",       # u % R
                  f"s_mul_i32 {sr(S_HEAD)}, {sr(S_GRP)}, {sr(S_R)}", f"s_add_u32 {sr(S_HEAD)}, {sr(S_HEAD)}, {sr(S_TMP2)}",
                  f"s_sub_u32 {sr(S_SLICE)}, {sr(S_NS)}, 1", f"s_sub_i32 {sr(S_SLICE)}, {sr(S_SLICE)}, {sr(S_TMP)}",            # may go negative: no unit
                  f"s_lshl_b32 {sr(S_Q0ROW)}, {sr(S_SLICE)}, 5",
                  f"s_add_u32 {sr(S_TW)}, {sr(S_Q0ROW)}, 31", f"s_lshr_b32 {sr(S_TW)}, {sr(S_TW)}, 6", f"s_add_u32 {sr(S_TW)}, {sr(S_TW)}, 1",
                  f"s_cmp_lt_i32 {sr(S_SLICE)}, 0", f"s_cselect_b32 {sr(S_TW)}, 0, {sr(S_TW)}"]:
            e(i)

    def desc(lo, hi, ss, dst):
        """rows of head S_HEAD: base + head * 256 bytes; range = whole tensor rows (S - 1) * stride + 256"""
        for i in [f"s_lshl_b32 {sr(S_TMP)}, {sr(S_HEAD)}, 8", f"s_add_u32 {sr(dst)}, {sr(lo)}, {sr(S_TMP)}", f"s_addc_u32 {sr(dst + 1)}, {sr(hi)}, 0",
                  f"s_and_b32 {sr(dst + 1)}, {sr(dst + 1)}, 0xffff", f"s_sub_u32 {sr(S_TMP)}, {sr(S_S)}, 1", f"s_mul_i32 {sr(S_TMP)}, {sr(S_TMP)}, {sr(ss)}",
                  f"s_add_u32 {sr(dst + 2)}, {sr(S_TMP)}, 256", f"s_mov_b32 {sr(dst + 3)}, 0x20000"]:
            e(i)

    def q_load():
        """Q^T fragments of the decoded unit: lane (q = l31, h) <- 16 bytes of row q0 + l31 at column 16 ks + 8 h (rows past the end read as zeros: range-checked)"""
        desc(S_Q0, S_Q1, S_QSS, S_QR)
        e(f"s_mul_i32 {sr(S_SOFF)}, {sr(S_Q0ROW)}, {sr(S_QSS)}")
        e(f"v_accvgpr_read_b32 {vr(TD)}, {ar(A_QOFF)}"); e("s_nop 1")
        for ks in range(8):
            e(f"buffer_load_dwordx4 {vr(Q(ks), 4)}, {vr(TD)}, {sr(S_QR, 4)}, {sr(S_SOFF)} offen offset:{ks * 32}")

    for r in range(64):
        e(f"v_accvgpr_write_b32 {ar(r)}, 0")                # (later items: zeroed again by the epilogue as it reads them)
    e("C_ITEM%=:")
    item_decode(G, loader=False)
    e("s_cbranch_scc1 C_DONE%=")                      # no item left
    unit_decode()
    e(f"s_cmp_eq_u32 {sr(S_TW)}, 0")
    e("s_cbranch_scc1 C_IDLE%=")
    q_load()
    e("C_ITEM_GO%=:")                                  # the item is decoded and the unit's Q rows are on their way (asked for during the previous unit's last tile)
    desc(S_O0, S_O1, S_OSS, S_OR)
    for i in [f"v_mov_b32 {vr(MRUN)}, 0xff800000", f"v_mov_b32 {vr(LRUN)}, 0", f"v_accvgpr_read_b32 {vr(KADDR)}, {ar(A_K0)}", f"v_accvgpr_read_b32 {vr(VADDR)}, {ar(A_V0)}",
              f"v_accvgpr_read_b32 {vr(TC)}, {ar(A_L31)}",
              # the ring runs on across items: the item's first tile is global tile S_NBASE -> stage S_NBASE % 4
              f"s_and_b32 {sr(S_STAGE)}, {sr(S_NBASE)}, 3", f"s_mul_i32 {sr(S_TMP)}, {sr(S_STAGE)}, {STG}", f"s_mov_b32 {sr(S_T)}, 0",
              f"v_add_u32 {vr(KADDR)}, {sr(S_TMP)}, {vr(KADDR)}", f"v_add_u32 {vr(VADDR)}, {sr(S_TMP)}, {vr(VADDR)}",
              # QI = (q0 + l31) - key0 of the diagonal tile = rows relative to the last tile's first key
              f"s_sub_u32 {sr(S_TMP)}, {sr(S_TW)}, 1", f"s_lshl_b32 {sr(S_TMP)}, {sr(S_TMP)}, 6", f"s_sub_u32 {sr(S_TMP)}, {sr(S_Q0ROW)}, {sr(S_TMP)}",
              f"v_add_u32 {vr(QI)}, {sr(S_TMP)}, {vr(TC)}",
              "s_waitcnt vmcnt(0)"]:
        e(i)
    # ---- tile 0: B_0, K reads + QK^T, (mask if T_w == 1), first max: m_run = tmax
    e("s_barrier")
    head, groups = frag_stream(0, True, False)
    emit(G, head)
    for grp in groups:
        emit(G, grp)
    e("s_nop 7"); e("s_nop 7")
    lbl = G.label("NOMASK0")
    e(f"s_cmp_gt_u32 {sr(S_TW)}, 1"); e(f"s_cbranch_scc1 {lbl}")
    for i in mask_tile(0):
        e(i)
    e(f"{lbl}:")
    for i in start_softmax(0):
        e(i)
    e(f"v_max_f32 {vr(MRUN)}, {vr(MRUN)}, {vr(TMAX)}")          # first tile: m_run = max(-inf, tmax); l = 0, O = 0: nothing to rescale

    def advance_stage():
        """KADDR / VADDR -> the next ring stage (wraps after 4)"""
        return [f"s_add_u32 {sr(S_STAGE)}, {sr(S_STAGE)}, 1", f"s_cmp_eq_u32 {sr(S_STAGE)}, {NSTAGE}", f"s_cselect_b32 {sr(S_DELTA)}, {(-(NSTAGE - 1) * STG) & 0xffffffff}, {sr(S_STG)}",
                f"s_cselect_b32 {sr(S_STAGE)}, 0, {sr(S_STAGE)}"]

    def full_iter(slot, masked):
        """tile t in S[slot] (max / decision done): A = barrier, K(t+1) -> S[1-slot] with finish(t) in the gaps; B = PV(t) with [mask +] start(t+1)"""
        nxt = 1 - slot
        emit(G, stamp(0))
        e("s_barrier")                                              # B_{t+1}: tile t + 1 has landed
        emit(G, stamp(1))
        for i in advance_stage():
            e(i)
        e(f"v_add_u32 {vr(KADDR)}, {sr(S_DELTA)}, {vr(KADDR)}")     # K of tile t + 1; V of tile t is still at VADDR
        head, groups = frag_stream(nxt, True, True)
        fa, fb1, fb2, fb3 = finish_parts(slot)
        st = (mask_tile(nxt) if masked else []) + start_softmax(nxt)
        if "sm" in DROP:
            fa, fb1, fb2, fb3, st = [], [], [], [], [f"s_mov_b64 {sr(S_ANY, 2)}, 0"]
        emit(G, head)
        npre = 0 if simple else min(PRE_A, len(fa))
        for ins in fa[:npre]:                                       # VALU work while the first K fragments are on their way
            e(ins)
        # groups 0..15 = QK^T of tile t + 1, 16..31 = PV of tile t.  P chunks 0, 1 (fa) are first used by group 16, chunk 2 (fb1) by 24, chunk 3 (fb2) by 28;
        # the scores of tile t + 1 are complete a few MFMAs after group 15
        if simple:             # --simple (bring-up form): the whole softmax of tile t between the two MFMA phases, the next tile's max after the PV
            placed = [(15, ins) for ins in fa + fb1 + fb2 + fb3] + [(31, ins) for ins in st]
        else:
            placed = spread(fa[npre:], 0, 13) + spread(fb1, 14, 19) + spread(fb2, 18, 23) + spread(fb3, 24, 31) + spread(st, 19, 31)
        for gi, grp in enumerate(groups):
            if gi == 16:
                emit(G, stamp(2))
            emit(G, grp)
            for g, ins in placed:
                if g == gi:
                    e(ins)
        e(f"v_add_u32 {vr(VADDR)}, {sr(S_DELTA)}, {vr(VADDR)}")     # V of tile t + 1
        emit(G, stamp(3))
        lb, back = G.label("RS"), G.label("BK")
        e(f"s_cmp_lg_u64 {sr(S_ANY, 2)}, 0"); e(f"s_cbranch_scc1 {lb}"); e(f"{back}:")
        G.stubs.append((lb, back))
        e(f"s_add_u32 {sr(S_T)}, {sr(S_T)}, 1")

    def last_iter(slot):
        """the wave's last tile (scores in S[slot], masked, max / decision done): finish + PV, no next tile"""
        fin = finish_softmax(slot)
        for i in fin:
            e(i)
        e("s_nop 1")
        head, groups = frag_stream(slot, False, True)
        emit(G, head)
        for grp in groups:
            emit(G, grp)

    G.stubs = []
    # ---- loop: full iterations while t + 2 < T_w (unmasked), then the masked one (t + 2 == T_w), then the last tile
    emit(G, stamp(4))
    e("C_LOOP_E%=:")
    e(f"s_add_u32 {sr(S_TMP)}, {sr(S_T)}, 2"); e(f"s_cmp_lt_u32 {sr(S_TMP)}, {sr(S_TW)}"); e("s_cbranch_scc0 C_TAIL_E%=")
    full_iter(0, False)
    e(f"s_add_u32 {sr(S_TMP)}, {sr(S_T)}, 2"); e(f"s_cmp_lt_u32 {sr(S_TMP)}, {sr(S_TW)}"); e("s_cbranch_scc0 C_TAIL_O%=")
    full_iter(1, False)
    e("s_branch C_LOOP_E%=")
    for par, slot in (("E", 0), ("O", 1)):
        e(f"C_TAIL_{par}%=:")
        lab_last = G.label("TOLAST")
        e(f"s_mov_b32 {sr(S_PAR)}, {slot}")
        e(f"s_add_u32 {sr(S_TMP)}, {sr(S_T)}, 1"); e(f"s_cmp_lt_u32 {sr(S_TMP)}, {sr(S_TW)}"); e(f"s_cbranch_scc0 C_PRELAST%=")
        full_iter(slot, True)                       # t + 2 == T_w: the next tile is the diagonal one
        e(f"s_mov_b32 {sr(S_PAR)}, {1 - slot}")
        e("s_branch C_PRELAST%=")
    # ---- before the last tile: the barrier a longer unit of the quad still needs, the book-keeping of this item, and the NEXT item's decode + Q rows
    # (the Q registers are free from here on: the last QK^T has been issued)
    e("C_PRELAST%=:")
    nob = G.label("NOB")
    e(f"s_mov_b32 {sr(S_CNT)}, {sr(S_TW)}")
    e(f"s_cmp_lt_u32 {sr(S_TW)}, {sr(S_TQ)}"); e(f"s_cbranch_scc0 {nob}")
    e("s_barrier"); e(f"s_add_u32 {sr(S_CNT)}, {sr(S_CNT)}, 1")      # B_{T_w}: frees tile T_w - 2's stage only
    e(f"{nob}:")
    e(f"s_sub_u32 {sr(S_REM)}, {sr(S_TQ)}, {sr(S_CNT)}")             # barriers this wave still owes the item after its last tile
    e(f"s_add_u32 {sr(S_NBASE)}, {sr(S_NBASE)}, {sr(S_TQ)}")
    e(f"s_mul_i32 {sr(S_OSOFF)}, {sr(S_Q0ROW)}, {sr(S_OSS)}")
    e(f"s_mul_i32 {sr(S_LSEOFF)}, {sr(S_HEAD)}, {sr(S_S)}"); e(f"s_add_u32 {sr(S_LSEOFF)}, {sr(S_LSEOFF)}, {sr(S_Q0ROW)}"); e(f"s_lshl_b32 {sr(S_LSEOFF)}, {sr(S_LSEOFF)}, 2")
    e(f"s_sub_u32 {sr(S_ROWS)}, {sr(S_S)}, {sr(S_Q0ROW)}")                # valid rows of this unit (>= 32 except in the last slice)
    e(f"s_add_u32 {sr(S_ROUND)}, {sr(S_ROUND)}, 1")
    item_decode(G, loader=False)
    nonext, pdone = G.label("NONEXT"), G.label("PDONE")
    e(f"s_mov_b32 {sr(S_QV)}, 0")
    e(f"s_cbranch_scc1 {pdone}")
    unit_decode()
    e(f"s_mov_b32 {sr(S_QV)}, 2")
    e(f"s_cmp_eq_u32 {sr(S_TW)}, 0"); e(f"s_cbranch_scc1 {pdone}")
    e(f"s_mov_b32 {sr(S_QV)}, 1")
    q_load()
    e(f"{pdone}:")
    e(f"s_cmp_eq_u32 {sr(S_PAR)}, 0"); e("s_cbranch_scc0 C_LAST_O%=")
    for par, slot in (("E", 0), ("O", 1)):
        e(f"C_LAST_{par}%=:")
        last_iter(slot)
        e("s_branch C_STORE%=")
    # ---- normalise and store: d = 32 dt + (r & 3) + 8 (r >> 2) + 4 h
    e("C_STORE%=:")
    emit(G, stamp(5))
    e("s_nop 7"); e("s_nop 7"); e("s_nop 7")
    d = [TA, TB, TC, TD, T0]
    for i in [f"v_mov_b32 {vr(T1)}, {vr(LRUN)}", "s_nop 1", f"v_permlane32_swap_b32 {vr(LRUN)}, {vr(T1)}", f"v_add_f32 {vr(LRUN)}, {vr(LRUN)}, {vr(T1)}",
              # (lanes 0-31 now hold lo + hi in LRUN?  swap: LRUN = [lo, lo'], T1 = [hi, hi'] where primed = the other half's value -> lo + hi in both)
              "LSE_HERE",
              f"v_div_scale_f32 {vr(d[0])}, {sr(S_ANY, 2)}, {vr(LRUN)}, {vr(LRUN)}, 1.0", f"v_rcp_f32 {vr(d[1])}, {vr(d[0])}", "s_nop 0",
              f"v_fma_f32 {vr(d[2])}, -{vr(d[0])}, {vr(d[1])}, 1.0", f"v_fmac_f32 {vr(d[1])}, {vr(d[2])}, {vr(d[1])}",
              f"v_div_scale_f32 {vr(d[2])}, vcc, 1.0, {vr(LRUN)}, 1.0", f"v_mul_f32 {vr(d[3])}, {vr(d[2])}, {vr(d[1])}",
              f"v_fma_f32 {vr(d[4])}, -{vr(d[0])}, {vr(d[3])}, {vr(d[2])}", f"v_fmac_f32 {vr(d[3])}, {vr(d[4])}, {vr(d[1])}",
              f"v_fma_f32 {vr(d[0])}, -{vr(d[0])}, {vr(d[3])}, {vr(d[2])}", "s_nop 1",
              f"v_div_fmas_f32 {vr(d[0])}, {vr(d[0])}, {vr(d[1])}, {vr(d[3])}", f"v_div_fixup_f32 {vr(INV)}, {vr(d[0])}, {vr(LRUN)}, 1.0",
              f"v_cmp_lt_f32 vcc, 0, {vr(LRUN)}", "s_nop 1", f"v_cndmask_b32 {vr(INV)}, 0, {vr(INV)}, vcc",
              f"v_accvgpr_read_b32 {vr(S(0, 6))}, {ar(A_OOFF)}"]:
        if i == "LSE_HERE":
            nolse = G.label("NOLSE")
            for j in [f"s_cmp_eq_u64 {sr(S_LSE, 2)}, 0", f"s_cbranch_scc1 {nolse}",
                      # lse[head][q0 + l31] = m + log2(l) from the lanes of the first key half, rows < S
                      f"s_mov_b32 {sr(S_QR)}, {sr(S_LSE)}", f"s_and_b32 {sr(S_QR + 1)}, {sr(S_LSE + 1)}, 0xffff", f"s_mul_i32 {sr(S_TMP)}, {sr(S_R)}, {sr(S_HKV)}",
                      f"s_mul_i32 {sr(S_TMP)}, {sr(S_TMP)}, {sr(S_S)}", f"s_lshl_b32 {sr(S_QR + 2)}, {sr(S_TMP)}, 2", f"s_mov_b32 {sr(S_QR + 3)}, 0x20000",
                      f"v_log_f32 {vr(T0)}, {vr(LRUN)}", f"v_accvgpr_read_b32 {vr(TB)}, {ar(A_L31)}", "s_nop 0", f"v_add_f32 {vr(T0)}, {vr(MRUN)}, {vr(T0)}",
                      f"v_cmp_gt_u32 {sr(S_ANY, 2)}, {sr(S_ROWS)}, {vr(TB)}", f"v_lshlrev_b32 {vr(TB)}, 2, {vr(TB)}", "s_nop 3", f"s_mov_b32 {sr(S_ANY + 1)}, 0",
                      f"s_mov_b64 {sr(S_TMP, 2)}, exec", f"s_mov_b64 exec, {sr(S_ANY, 2)}", "s_nop 1",
                      f"buffer_store_dword {vr(T0)}, {vr(TB)}, {sr(S_QR, 4)}, {sr(S_LSEOFF)} offen", f"s_mov_b64 exec, {sr(S_TMP, 2)}", "s_nop 1", f"{nolse}:"]:
                e(j)
            continue
        e(i)
    # 16-byte stores: the two key halves of a query (lanes l and l + 32) hold interleaved groups of 4 head-dim values; one permlane32 swap per register pair
    # gives the low lane d = 8 j' .. 8 j' + 7 of the even group and the high lane those of the odd group (two banks of temporaries, alternating)
    for dt in range(4):
        for j in range(2):
            bank = 8 + 16 * ((2 * dt + j) % 2)
            x = [S(0, bank + i) for i in range(8)]
            y = [S(0, bank + 8 + i) for i in range(4)]          # A (even group) -> y[0:2], B (odd group) -> y[2:4]
            for i in range(8):
                e(f"v_accvgpr_read_b32 {vr(x[i])}, {ar(O(dt, 8 * j + i))}")
            for i in range(8):
                e(f"v_accvgpr_write_b32 {ar(O(dt, 8 * j + i))}, 0")           # the next unit's sums start from 0
            for i in range(8):
                e(f"v_mul_f32 {vr(x[i])}, {vr(x[i])}, {vr(INV)}")
            for i in range(4):
                e(f"v_cvt_pk_bf16_f32 {vr(y[i])}, {vr(x[2 * i])}, {vr(x[2 * i + 1])}")
            e("s_nop 1")
            e(f"v_permlane32_swap_b32 {vr(y[0])}, {vr(y[2])}")
            e(f"v_permlane32_swap_b32 {vr(y[1])}, {vr(y[3])}")
            # after the swaps: low lanes y[0:2] = own even group, y[2:4] = partner's even group -> registers must be ordered (y0, y1, y2, y3) = (A_lo, A_lo', A_hi, A_hi')
            e("s_nop 0")
            e(f"buffer_store_dwordx4 {vr(y[0], 4)}, {vr(S(0, 6))}, {sr(S_OR, 4)}, {sr(S_OSOFF)} offen offset:{dt * 64 + j * 32}")
    # ---- the barriers of the tiles this wave does not have, then the next item (decoded before the last tile)
    rl, rd = G.label("REM"), G.label("REMD")
    e(f"{rl}:")
    e(f"s_cmp_eq_u32 {sr(S_REM)}, 0"); e(f"s_cbranch_scc1 {rd}")
    e("s_barrier"); e(f"s_sub_u32 {sr(S_REM)}, {sr(S_REM)}, 1"); e(f"s_branch {rl}")
    e(f"{rd}:")
    emit(G, stamp(6))
    e(f"s_cmp_eq_u32 {sr(S_QV)}, 1"); e("s_cbranch_scc1 C_ITEM_GO%=")
    e(f"s_cmp_eq_u32 {sr(S_QV)}, 0"); e("s_cbranch_scc1 C_DONE%=")
    # a wave without a unit in the (decoded) item: its T_q barriers, then a fresh decode
    e("C_IDLE%=:")
    e(f"s_mov_b32 {sr(S_CNT)}, 0")
    il, idn = G.label("IDL"), G.label("IDD")
    e(f"{il}:")
    e(f"s_cmp_lt_u32 {sr(S_CNT)}, {sr(S_TQ)}"); e(f"s_cbranch_scc0 {idn}")
    e("s_barrier"); e(f"s_add_u32 {sr(S_CNT)}, {sr(S_CNT)}, 1"); e(f"s_branch {il}")
    e(f"{idn}:")
    e(f"s_add_u32 {sr(S_NBASE)}, {sr(S_NBASE)}, {sr(S_TQ)}")
    e(f"s_add_u32 {sr(S_ROUND)}, {sr(S_ROUND)}, 1")
    e("s_branch C_ITEM%=")
    for lb, back in G.stubs:
        e(f"{lb}:")
        for i in rescale_stub():
            e(i)
        e(f"s_branch {back}")
    e("C_DONE%=:")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    if STAMPS:
        for k in range(8):
            e(f"v_accvgpr_read_b32 {vr(k)}, {ar(120 + k)}")
        for ins in [f"s_lshl_b32 {sr(S_TMP)}, {sr(S_BID)}, 2", f"s_add_u32 {sr(S_TMP)}, {sr(S_TMP)}, {sr(S_WAVE)}", f"s_lshl_b32 {sr(S_TMP)}, {sr(S_TMP)}, 5",
                    "s_mov_b32 s88, %[stp0]", "s_mov_b32 s89, %[stp1]", f"s_add_u32 s88, s88, {sr(S_TMP)}", "s_addc_u32 s89, s89, 0",
                    "v_mov_b32 v8, s88", "v_mov_b32 v9, s89", "s_mov_b64 exec, 1", "s_nop 1",
                    "global_store_dwordx4 v[8:9], v[0:3], off", "global_store_dwordx4 v[8:9], v[4:7], off offset:16", "s_waitcnt vmcnt(0)", "s_mov_b64 exec, -1"]:
            e(ins)
    e("s_branch END%=")

    # ============ loader waves ========================================================================================================
    # One code path per loader wave (the piece list of a wave is static: slot j of loader i = global slot 4 j + i; piece p = slot (< 37: K pieces 0..16,
    # V pieces 17..36) else a padding piece -> scratch).  The tile stream of a block runs on across its items: global tile n -> ring stage n % 4,
    # issued two barriers ahead of its use, so the first tiles of an item are in flight while the compute waves finish the item before.
    e("LOADER%=:")
    for i in range(1, 4):
        e(f"s_cmp_eq_u32 {sr(S_WAVE)}, {4 + i}"); e(f"s_cbranch_scc1 LOADER{i}%=")
    for i in range(4):
        loader_body(G, i)
    e("L_DONE%=:")
    e("s_waitcnt vmcnt(0)")
    e("END%=:")
    return G.lines

S_NISS, S_NWAIT, S_IV = S_HEAD, S_Q0ROW, S_VALID
S_NBASE = S_VALID                                         # compute-only name: global index of the item's first tile          # loader-only names: tiles issued, barriers passed, issue cursor valid


def loader_body(G, i):
    e = G.e
    e(f"LOADER{i}%=:")
    pieces = []                                           # (kind, piece index) per slot: loader 0 has 10 pieces per tile, the others 9
    for j in range(10):
        p = 4 * j + i
        if p < 37 or "padpieces" in OPT:
            pieces.append(("K", p) if p < 17 else ("V", p) if p < 37 else ("P", p - 37))
    npc = len(pieces)
    # per-lane source offsets: padded chunk n = 64 p + lane; K: row n / 17, c n % 17 (c = 16 is padding: fetch c = 0); V: n' = n - 1088, row n' / 20, c n' % 20
    for j, (kind, p) in enumerate(pieces):
        pp = p if kind != "V" else p - 17
        div, magic, ss = (17, 0xf0f0f10, S_KSS) if kind != "V" else (20, 0xccccccd, S_VSS)
        for ins in [f"v_add_u32 {vr(10)}, {64 * pp}, {vr(LANE)}" if 64 * pp <= 64 else f"v_add_u32 {vr(10)}, {64 * pp}, {vr(LANE)}",
                    f"s_mov_b32 {sr(S_TMP2)}, 0x{magic:x}", f"v_mul_hi_u32 {vr(11)}, {vr(10)}, {sr(S_TMP2)}", f"v_mul_u32_u24 {vr(12)}, {div}, {vr(11)}",
                    f"v_sub_u32 {vr(12)}, {vr(10)}, {vr(12)}", f"v_cmp_gt_u32 vcc, 16, {vr(12)}", "s_nop 1", f"v_cndmask_b32 {vr(12)}, 0, {vr(12)}, vcc",
                    f"v_lshlrev_b32 {vr(12)}, 4, {vr(12)}", f"v_mul_lo_u32 {vr(13)}, {vr(11)}, {sr(ss)}", f"v_add_u32 {vr(j)}, {vr(13)}, {vr(12)}"]:
            e(ins)

    def cursor_item():
        """decode the item of round S_ROUND into the issue cursor: S_IV (0: none left), S_TQ, descriptors, S_T = 0"""
        item_decode(G, loader=True)
        e(f"s_cselect_b32 {sr(S_IV)}, 0, 1")
        for (lo, hi, ss, dst) in ((S_K0, S_K1, S_KSS, S_KR), (S_V0, S_V1, S_VSS, S_VR)):
            for ins in [f"s_lshl_b32 {sr(S_TMP)}, {sr(S_GRP)}, 8", f"s_add_u32 {sr(dst)}, {sr(lo)}, {sr(S_TMP)}", f"s_addc_u32 {sr(dst + 1)}, {sr(hi)}, 0",
                        f"s_and_b32 {sr(dst + 1)}, {sr(dst + 1)}, 0xffff", f"s_sub_u32 {sr(S_TMP)}, {sr(S_S)}, 1", f"s_mul_i32 {sr(S_TMP)}, {sr(S_TMP)}, {sr(ss)}",
                        f"s_add_u32 {sr(dst + 2)}, {sr(S_TMP)}, 256", f"s_mov_b32 {sr(dst + 3)}, 0x20000"]:
                e(ins)
        e(f"s_mov_b32 {sr(S_T)}, 0")

    def issue_next():
        skip, same = G.label("LSK"), G.label("LSAME")
        e(f"s_cmp_eq_u32 {sr(S_IV)}, 0"); e(f"s_cbranch_scc1 {skip}")
        for ins in [f"s_and_b32 {sr(S_STAGE)}, {sr(S_NISS)}, 3", f"s_mul_i32 {sr(S_STAGE)}, {sr(S_STAGE)}, {STG}", f"s_add_u32 {sr(S_STAGE)}, {sr(S_STAGE)}, {sr(S_LDS)}",
                    f"s_lshl_b32 {sr(S_SOFF)}, {sr(S_T)}, 6", f"s_mul_i32 {sr(S_DELTA)}, {sr(S_SOFF)}, {sr(S_KSS)}", f"s_mul_i32 {sr(S_SOFF)}, {sr(S_SOFF)}, {sr(S_VSS)}"]:
            e(ins)
        for j, (kind, p) in enumerate(pieces):
            if kind == "P":
                e(f"s_add_u32 m0, {sr(S_LDS)}, {SCRATCH + 1024 * p}")
            else:
                e(f"s_add_u32 m0, {sr(S_STAGE)}, {1024 * p}")
            if "dma" not in DROP:
                e("s_nop 0")
                e(f"buffer_load_dwordx4 {vr(j)}, {sr(S_VR if kind == 'V' else S_KR, 4)}, {sr(S_SOFF if kind == 'V' else S_DELTA)} offen lds")
        e(f"s_add_u32 {sr(S_NISS)}, {sr(S_NISS)}, 1"); e(f"s_add_u32 {sr(S_T)}, {sr(S_T)}, 1")
        e(f"s_cmp_lt_u32 {sr(S_T)}, {sr(S_TQ)}"); e(f"s_cbranch_scc1 {same}")
        e(f"s_add_u32 {sr(S_ROUND)}, {sr(S_ROUND)}, 1")
        cursor_item()
        e(f"{same}:")
        e(f"{skip}:")

    e(f"s_mov_b32 {sr(S_NISS)}, 0"); e(f"s_mov_b32 {sr(S_NWAIT)}, 0")
    cursor_item()
    issue_next()
    issue_next()
    loop, l2, l3 = G.label("LLOOP"), G.label("LW2"), G.label("LWD")
    e(f"{loop}:")
    e(f"s_cmp_eq_u32 {sr(S_NWAIT)}, {sr(S_NISS)}"); e("s_cbranch_scc1 L_DONE%=")
    e(f"s_add_u32 {sr(S_TMP)}, {sr(S_NWAIT)}, 1"); e(f"s_cmp_lt_u32 {sr(S_TMP)}, {sr(S_NISS)}"); e(f"s_cbranch_scc0 {l2}")
    e(f"s_waitcnt vmcnt({npc})"); e(f"s_branch {l3}")
    e(f"{l2}:")
    e("s_waitcnt vmcnt(0)")
    e(f"{l3}:")
    e("s_barrier")                                     # tile n is in; the compute waves are done with tile n - 2, whose stage tile n + 2 takes
    e(f"s_add_u32 {sr(S_NWAIT)}, {sr(S_NWAIT)}, 1")
    issue_next()
    e(f"s_branch {loop}")


def item_decode(G, loader):
    """serpentine deal: round r even -> item r G + bid, odd -> (r + 1) G - 1 - bid.  Sets SCC = 1 when no item is left; else S_ITEM, S_QD (quad), S_GRP
    (kv head), S_TQ (tiles of the quad's first = longest unit)"""
    e = G.e
    for i in [f"s_mul_i32 {sr(S_ITEM)}, {sr(S_ROUND)}, {sr(S_G)}", f"s_add_u32 {sr(S_TMP)}, {sr(S_ITEM)}, {sr(S_BID)}",
              f"s_add_u32 {sr(S_TMP2)}, {sr(S_ITEM)}, {sr(S_G)}", f"s_sub_u32 {sr(S_TMP2)}, {sr(S_TMP2)}, 1", f"s_sub_u32 {sr(S_TMP2)}, {sr(S_TMP2)}, {sr(S_BID)}",
              f"s_bitcmp1_b32 {sr(S_ROUND)}, 0", f"s_cselect_b32 {sr(S_ITEM)}, {sr(S_TMP2)}, {sr(S_TMP)}",
              # q = item / Hkv, g = item % Hkv
              f"s_mul_hi_u32 {sr(S_QD)}, {sr(S_ITEM)}, {sr(S_MHKV)}", f"s_mul_i32 {sr(S_TMP)}, {sr(S_QD)}, {sr(S_HKV)}", f"s_sub_u32 {sr(S_GRP)}, {sr(S_ITEM)}, {sr(S_TMP)}",
              # first unit of the quad: u0 = 4 q -> slice NS - 1 - u0 / R -> T_q
              f"s_lshl_b32 {sr(S_TMP)}, {sr(S_QD)}, 2", f"s_mul_hi_u32 {sr(S_TMP)}, {sr(S_TMP)}, {sr(S_MR)}", f"s_sub_u32 {sr(S_TMP2)}, {sr(S_NS)}, 1",
              f"s_sub_u32 {sr(S_TMP)}, {sr(S_TMP2)}, {sr(S_TMP)}", f"s_lshl_b32 {sr(S_TMP)}, {sr(S_TMP)}, 5", f"s_add_u32 {sr(S_TMP)}, {sr(S_TMP)}, 31",
              f"s_lshr_b32 {sr(S_TMP)}, {sr(S_TMP)}, 6", f"s_add_u32 {sr(S_TQ)}, {sr(S_TMP)}, 1",
              f"s_cmp_ge_u32 {sr(S_ITEM)}, {sr(S_NI)}"]:
        e(i)


def main():
    simple = "--simple" in sys.argv
    lines = build(simple)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "ufvideo_amd", "csrc", "attn_c128_asm.inc")
    clob = [f"v{i}" for i in range(128)] + [f"a{i}" for i in range(128 if STAMPS else 120)] + [f"s{i}" for i in range(38, 100)] + ["vcc", "scc", "memory"]
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_attn_c128.py%s -- do not edit.  %d instructions.\n" % (" --simple" if simple else "", len(lines)))
        f.write("#define UFV_ATTN_C128_ASM \\\n")
        for l in lines:
            f.write('    "%s\\n\\t" \\\n' % l)
        f.write('    ""\n')
        f.write("#define UFV_ATTN_C128_CLOBBERS " + ", ".join('"%s"' % c for c in clob) + "\n")
        f.write("#define UFV_ATTN_C128_LDS %d\n" % LDS_BYTES)
        if STAMPS:
            f.write("#define UFV_ATTN_C128_STAMPS 1\n")
    print(out, len(lines), "lines")


if __name__ == "__main__":
    main()
