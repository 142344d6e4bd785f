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
def KFR(i, half, j=0): return 64 + 8 * (i % 3) + 4 * half + j      # a64..87: ring of 3 k-steps
def VFR(i, j=0): return 88 + 4 * (i % 6) + j                       # a88..111: ring of 6 V^T fragments

# ---- loader-wave registers: v0..9 source offsets of its 10 pieces, v10.. temps
# ---- fixed SGPRs (s40..s99)
(S_Q0, S_Q1, S_K0, S_K1, S_V0, S_V1, S_O0, S_O1, S_QSS, S_KSS, S_VSS, S_OSS, S_S, S_R, S_HKV, S_NS, S_NI, S_MR, S_MHKV, S_SL2, S_LDS,
 S_WAVE, S_BID, S_G, S_ROUND, S_ITEM, S_QD, S_GRP, S_TQ, S_TW, S_T, S_U, S_SLICE, S_HEAD, S_Q0ROW, S_VALID, S_STAGE, S_DELTA, S_TMP, S_TMP2,
 S_TMP3, S_SOFF) = range(40, 82)
S_KR, S_VR, S_QR, S_OR = 84, 88, 92, 96
S_CNT = 38
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


def merge(G, mfmas, placed, simple=False):
    """emit MFMAs with fillers in the gaps (gap g = after MFMA g-1; gap 0 = before the first)"""
    gaps = [[] for _ in range(len(mfmas) + 1)]
    for g, ins in placed:
        gaps[len(mfmas) if simple else min(max(g, 0), len(mfmas))].append(ins)
    for ins in gaps[0]:
        emit(G, ins)
    for i, m in enumerate(mfmas):
        emit(G, m)
        for ins in gaps[i + 1]:
            emit(G, ins)


def emit(G, ins):
    if isinstance(ins, (list, tuple)):
        for x in ins:
            emit(G, x)
    else:
        G.e(ins)


# ---- phase A: QK^T of the next tile (K fragments streamed from LDS) --------------------------------------------------------
def phase_a_mfma(slot):
    """returns the list of 'MFMA groups': each entry = instructions that must be adjacent (waits + the MFMA)"""
    out = []
    reads = []
    for ks in range(8):
        for half in range(2):
            reads.append(f"ds_read_b128 {ar(KFR(ks, half), 4)}, {vr(KADDR)} offset:{ks * 32 + half * 32 * PK}")
    # issue order: reads of k-steps 0,1 up front, then after the MFMAs of k-step ks the reads of ks + 2
    seq = []
    seq.append(reads[0:4])
    for ks in range(8):
        # outstanding reads allowed when k-step ks is consumed: those of ks+1 (2) [and none beyond, since ks+2's are issued AFTER these MFMAs]
        allowed = 2 if ks < 7 else 0
        d0, d1 = vr(S(slot, 0), 16), vr(S(slot, 16), 16)
        c0 = "0" if ks == 0 else d0
        c1 = "0" if ks == 0 else d1
        grp = [f"s_waitcnt lgkmcnt({allowed})",
               f"v_mfma_f32_32x32x16_bf16 {d0}, {ar(KFR(ks, 0), 4)}, {vr(Q(ks), 4)}, {c0}",
               f"v_mfma_f32_32x32x16_bf16 {d1}, {ar(KFR(ks, 1), 4)}, {vr(Q(ks), 4)}, {c1}"]
        if ks + 2 < 8:
            grp += reads[2 * (ks + 2): 2 * (ks + 2) + 2]
        seq.append(grp)
    return seq


def finish_softmax(slot):
    """P = exp2(fma(s, c, -msub)), psum in the order of attn_fwd_mfma (psum += s0[r] + s1[r], r = 0..15), l_run += psum, bf16 pack"""
    s = lambda r: vr(S(slot, r))
    o = [f"v_cmp_lg_f32 vcc, 0xff800000, {vr(MRUN)}", "s_nop 1", f"v_cndmask_b32 {vr(MSUB)}, 0, {vr(MRUN)}, vcc", f"v_mov_b32 {vr(PSUM)}, 0"]
    for r in range(16):
        o += [f"v_fma_f32 {s(r)}, {s(r)}, {sr(S_SL2)}, -{vr(MSUB)}", f"v_fma_f32 {s(16 + r)}, {s(16 + r)}, {sr(S_SL2)}, -{vr(MSUB)}"]
    ex = []
    for r in range(16):
        ex += [f"v_exp_f32 {s(r)}, {s(r)}", f"v_exp_f32 {s(16 + r)}, {s(16 + r)}"]
    o += ex[:6]
    k = 6
    for r in range(16):
        o += [f"v_add_f32 {vr(TA)}, {s(r)}, {s(16 + r)}", f"v_add_f32 {vr(PSUM)}, {vr(PSUM)}, {vr(TA)}"]
        o += ex[k:k + 2]; k += 2
    assert k >= 32
    o.append(f"v_add_f32 {vr(LRUN)}, {vr(LRUN)}, {vr(PSUM)}")
    o += [f"v_cvt_pk_bf16_f32 {vr(P(j))}, {vr(S(slot, 2 * j))}, {vr(S(slot, 2 * j + 1))}" for j in range(16)]
    return o


# ---- phase B: PV of the current tile (V^T fragments streamed) --------------------------------------------------------------
def phase_b_mfma():
    frags = []
    for dt in range(4):
        for c in range(4):
            i = dt * 4 + c
            off = dt * 64 + c * 16 * PV
            frags.append([f"ds_read_b64_tr_b16 {ar(VFR(i, 0), 2)}, {vr(VADDR)} offset:{off}",
                          f"ds_read_b64_tr_b16 {ar(VFR(i, 2), 2)}, {vr(VADDR)} offset:{off + 8 * PV}"])
    seq = [frags[0] + frags[1] + frags[2] + frags[3]]                     # four fragments ahead (ring of 6: a slot is rewritten two MFMAs after its reader)
    for i in range(16):
        dt, c = divmod(i, 4)
        later = min(3, 15 - i)                        # fragments issued after fragment i that may still be in flight
        grp = [f"s_waitcnt lgkmcnt({2 * later})",
               f"v_mfma_f32_32x32x16_bf16 {ar(O(dt), 16)}, {ar(VFR(i), 4)}, {vr(P(4 * c), 4)}, {ar(O(dt), 16)}"]
        if i + 4 < 16:
            grp += frags[i + 4]
        seq.append(grp)
    return seq


def mask_tile(slot):
    """diagonal tile: key index kj = key0 + (r & 3) + 8 (r >> 2) + 4 h (+ 32 for the second half) must be <= the lane's query row (QI holds qi - key0)"""
    o = []
    for r in range(16):
        kk = (r & 3) + 8 * (r >> 2)
        o += [f"v_add_u32 {vr(TA)}, {kk}, {vr(KJ)}", f"v_cmp_le_i32 vcc, {vr(TA)}, {vr(QI)}", "s_nop 1",
              f"v_cndmask_b32 {vr(S(slot, r))}, {vr(TE)}, {vr(S(slot, r))}, vcc",
              f"v_add_u32 {vr(TA)}, {kk + 32}, {vr(KJ)}", f"v_cmp_le_i32 vcc, {vr(TA)}, {vr(QI)}", "s_nop 1",
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


def build(simple=False):
    G = Gen()
    e = G.e
    # ============ common prologue =====================================================================================================
    ins = [f"s_mov_b32 {sr(d)}, %[{n}]" for d, n in (
        (S_Q0, "q0"), (S_Q1, "q1"), (S_K0, "k0"), (S_K1, "k1"), (S_V0, "v0"), (S_V1, "v1"), (S_O0, "o0"), (S_O1, "o1"), (S_QSS, "qss"), (S_KSS, "kss"),
        (S_VSS, "vss"), (S_OSS, "oss"), (S_S, "seq"), (S_R, "rr"), (S_HKV, "hkv"), (S_NS, "ns"), (S_NI, "ni"), (S_MR, "mr"), (S_MHKV, "mhkv"), (S_LDS, "lds"),
        (S_WAVE, "wave"), (S_BID, "bid"), (S_G, "grid"))]
    ins += [f"v_mov_b32 {vr(TA)}, %[scale]", f"v_mul_f32 {vr(TA)}, 0x3fb8aa3b, {vr(TA)}", "s_nop 0", f"v_readfirstlane_b32 {sr(S_SL2)}, {vr(TA)}",
            f"s_mov_b32 {sr(S_ROUND)}, 0", f"s_mov_b32 {sr(S_STG)}, {STG}",
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
              f"v_mul_lo_u32 {vr(TA)}, {vr(L31)}, {sr(S_OSS)}", f"v_lshl_add_u32 {vr(OOFF)}, {vr(HH)}, 3, {vr(TA)}",
              f"v_lshlrev_b32 {vr(KJ)}, 2, {vr(HH)}", f"v_mov_b32 {vr(TE)}, 0xff800000",
              f"v_accvgpr_write_b32 {ar(A_K0)}, {vr(TG)}", f"v_accvgpr_write_b32 {ar(A_V0)}, {vr(TF)}", f"v_accvgpr_write_b32 {ar(A_L31)}, {vr(L31)}",
              f"v_accvgpr_write_b32 {ar(A_QOFF)}, {vr(QOFF)}", f"v_accvgpr_write_b32 {ar(A_OOFF)}, {vr(OOFF)}"]:
        e(i)
    # KADDR / VADDR are rebuilt from the parked stage-0 addresses at every item start
    e("C_ITEM%=:")
    item_decode(G, loader=False)
    e("s_cbranch_scc1 C_DONE%=")                      # no item left
    # unit of this wave: u = 4 q + wave; slice = NS - 1 - u / R; head = g R + u % R; T_w = tiles of the slice
    for i in [f"s_lshl_b32 {sr(S_U)}, {sr(S_QD)}, 2", f"s_add_u32 {sr(S_U)}, {sr(S_U)}, {sr(S_WAVE)}",
              f"s_mul_hi_u32 {sr(S_TMP)}, {sr(S_U)}, {sr(S_MR)}",                    # u / R
              f"s_mul_i32 {sr(S_TMP2)}, {sr(S_TMP)}, {sr(S_R)}", f"s_sub_u32 {sr(S_TMP2)}, {sr(S_U)}, {sr(S_TMP2)}",       # u % R
              f"s_mul_i32 {sr(S_HEAD)}, {sr(S_GRP)}, {sr(S_R)}", f"s_add_u32 {sr(S_HEAD)}, {sr(S_HEAD)}, {sr(S_TMP2)}",
              f"s_sub_u32 {sr(S_SLICE)}, {sr(S_NS)}, 1", f"s_sub_i32 {sr(S_SLICE)}, {sr(S_SLICE)}, {sr(S_TMP)}",            # may go negative: no unit
              f"s_lshl_b32 {sr(S_Q0ROW)}, {sr(S_SLICE)}, 5",
              f"s_add_u32 {sr(S_TW)}, {sr(S_Q0ROW)}, 31", f"s_lshr_b32 {sr(S_TW)}, {sr(S_TW)}, 6", f"s_add_u32 {sr(S_TW)}, {sr(S_TW)}, 1",
              f"s_cmp_lt_i32 {sr(S_SLICE)}, 0", f"s_cselect_b32 {sr(S_TW)}, 0, {sr(S_TW)}",                               # T_w = 0: this wave only joins the barriers
              ]:
        e(i)
    e("s_barrier")                                     # B_start: every wave has left the previous item's LDS image
    e(f"s_cmp_eq_u32 {sr(S_TW)}, 0")
    e("s_cbranch_scc1 C_IDLE%=")
    # descriptors of this unit: Q / O rows of (head), base + head * 256 bytes; range = whole tensor rows (S - 1) * stride + 256
    for (lo, hi, ss, dst) in ((S_Q0, S_Q1, S_QSS, S_QR), (S_O0, S_O1, S_OSS, S_OR)):
        for i in [f"s_lshl_b32 {sr(S_TMP)}, {sr(S_HEAD)}, 8", f"s_add_u32 {sr(dst)}, {sr(lo)}, {sr(S_TMP)}", f"s_addc_u32 {sr(dst + 1)}, {sr(hi)}, 0",
                  f"s_and_b32 {sr(dst + 1)}, {sr(dst + 1)}, 0xffff", f"s_sub_u32 {sr(S_TMP)}, {sr(S_S)}, 1", f"s_mul_i32 {sr(S_TMP)}, {sr(S_TMP)}, {sr(ss)}",
                  f"s_add_u32 {sr(dst + 2)}, {sr(S_TMP)}, 256", f"s_mov_b32 {sr(dst + 3)}, 0x20000"]:
            e(i)
    # Q^T fragments: lane (q = l31, h) <- 16 bytes of row q0 + l31 at column 16 ks + 8 h (rows past the end read as zeros: range-checked)
    e(f"s_mul_i32 {sr(S_SOFF)}, {sr(S_Q0ROW)}, {sr(S_QSS)}")
    e(f"v_accvgpr_read_b32 {vr(QOFF)}, {ar(A_QOFF)}"); e(f"v_accvgpr_read_b32 {vr(L31)}, {ar(A_L31)}"); e("s_nop 1")
    for ks in range(8):
        e(f"buffer_load_dwordx4 {vr(Q(ks), 4)}, {vr(QOFF)}, {sr(S_QR, 4)}, {sr(S_SOFF)} offen offset:{ks * 32}")
    for r in range(64):
        e(f"v_accvgpr_write_b32 {ar(r)}, 0")
    for i in [f"v_mov_b32 {vr(MRUN)}, 0xff800000", f"v_mov_b32 {vr(LRUN)}, 0", f"v_accvgpr_read_b32 {vr(KADDR)}, {ar(A_K0)}", f"v_accvgpr_read_b32 {vr(VADDR)}, {ar(A_V0)}",
              f"s_mov_b32 {sr(S_STAGE)}, 0", f"s_mov_b32 {sr(S_T)}, 0",
              # QI = (q0 + l31) - key0 of the diagonal tile = rows relative to the last tile's first key
              f"s_sub_u32 {sr(S_TMP)}, {sr(S_TW)}, 1", f"s_lshl_b32 {sr(S_TMP)}, {sr(S_TMP)}, 6", f"s_sub_u32 {sr(S_TMP)}, {sr(S_Q0ROW)}, {sr(S_TMP)}",
              f"v_add_u32 {vr(QI)}, {sr(S_TMP)}, {vr(L31)}",
              "s_waitcnt vmcnt(0)"]:
        e(i)
    # ---- tile 0: B_0, K reads + QK^T, (mask if T_w == 1), first max: m_run = tmax
    e("s_barrier")
    seq = phase_a_mfma(0)
    for grp in seq:
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
        e("s_barrier")                                              # B_{t+1}: tile t + 1 has landed
        for i in advance_stage():
            e(i)
        e(f"v_add_u32 {vr(KADDR)}, {sr(S_DELTA)}, {vr(KADDR)}")     # K of tile t + 1; V of tile t is still at VADDR
        seqa = phase_a_mfma(nxt)
        fin = finish_softmax(slot)
        emit(G, seqa[0])
        groups = seqa[1:]
        placed = spread(fin, 0, len(groups) - 1)
        for gi, grp in enumerate(groups):
            emit(G, grp)
            for g, ins in placed:
                if (len(groups) - 1 if simple else g) == gi:
                    e(ins)
        e("s_nop 1")
        seqb = phase_b_mfma()
        st = (mask_tile(nxt) if masked else []) + start_softmax(nxt)
        emit(G, seqb[0])
        groups = seqb[1:]
        # the scores of tile t + 1 were written by the last QK^T MFMAs: first use >= 3 MFMAs later
        placed = spread(st, 3, len(groups) - 1)
        for gi, grp in enumerate(groups):
            emit(G, grp)
            for g, ins in placed:
                if (len(groups) - 1 if simple else g) == gi:
                    e(ins)
        e(f"v_add_u32 {vr(VADDR)}, {sr(S_DELTA)}, {vr(VADDR)}")     # V of tile t + 1
        lb, back = G.label("RS"), G.label("BK")
        e(f"s_cmp_lg_u64 {sr(S_ANY, 2)}, 0"); e(f"s_cbranch_scc1 {lb}"); e(f"{back}:")
        G.stubs.append((lb, back))
        e(f"s_add_u32 {sr(S_T)}, {sr(S_T)}, 1")

    def last_iter(slot):
        """the wave's last tile (scores in S[slot], masked, max / decision done): finish + PV, no next tile"""
        lb = G.label("NOB")
        e(f"s_mov_b32 {sr(S_CNT)}, {sr(S_TW)}")
        e(f"s_cmp_lt_u32 {sr(S_TW)}, {sr(S_TQ)}"); e(f"s_cbranch_scc0 {lb}")
        e("s_barrier"); e(f"s_add_u32 {sr(S_CNT)}, {sr(S_CNT)}, 1")      # B_{T_w} (a longer unit of the quad goes on): frees tile T_w - 2's stage only
        e(f"{lb}:")
        fin = finish_softmax(slot)
        for i in fin:
            e(i)
        e("s_nop 1")
        for grp in phase_b_mfma():
            emit(G, grp)

    G.stubs = []
    # ---- loop: full iterations while t + 2 < T_w (unmasked), then the masked one (t + 2 == T_w), then the last tile
    e("C_LOOP_E%=:")
    e(f"s_add_u32 {sr(S_TMP)}, {sr(S_T)}, 2"); e(f"s_cmp_lt_u32 {sr(S_TMP)}, {sr(S_TW)}"); e("s_cbranch_scc0 C_TAIL_E%=")
    full_iter(0, False)
    e(f"s_add_u32 {sr(S_TMP)}, {sr(S_T)}, 2"); e(f"s_cmp_lt_u32 {sr(S_TMP)}, {sr(S_TW)}"); e("s_cbranch_scc0 C_TAIL_O%=")
    full_iter(1, False)
    e("s_branch C_LOOP_E%=")
    for par, slot in (("E", 0), ("O", 1)):
        e(f"C_TAIL_{par}%=:")
        lab_last = f"C_LAST_{par}%="
        e(f"s_add_u32 {sr(S_TMP)}, {sr(S_T)}, 1"); e(f"s_cmp_lt_u32 {sr(S_TMP)}, {sr(S_TW)}"); e(f"s_cbranch_scc0 {lab_last}")
        full_iter(slot, True)                       # t + 2 == T_w: the next tile is the diagonal one
        e(f"s_branch C_LAST_{'O' if par == 'E' else 'E'}%=")
    for par, slot in (("E", 0), ("O", 1)):
        e(f"C_LAST_{par}%=:")
        last_iter(slot)
        e("s_branch C_STORE%=")
    # ---- normalise and store: d = 32 dt + (r & 3) + 8 (r >> 2) + 4 h
    e("C_STORE%=:")
    e("s_nop 7"); e("s_nop 7"); e("s_nop 7")
    d = [TA, TB, TC, TD, T0]
    for i in [f"v_mov_b32 {vr(T1)}, {vr(LRUN)}", "s_nop 1", f"v_permlane32_swap_b32 {vr(LRUN)}, {vr(T1)}", f"v_add_f32 {vr(LRUN)}, {vr(LRUN)}, {vr(T1)}",
              # (lanes 0-31 now hold lo + hi in LRUN?  swap: LRUN = [lo, lo'], T1 = [hi, hi'] where primed = the other half's value -> lo + hi in both)
              f"v_div_scale_f32 {vr(d[0])}, {sr(S_ANY, 2)}, {vr(LRUN)}, {vr(LRUN)}, 1.0", f"v_rcp_f32 {vr(d[1])}, {vr(d[0])}", "s_nop 0",
              f"v_fma_f32 {vr(d[2])}, -{vr(d[0])}, {vr(d[1])}, 1.0", f"v_fmac_f32 {vr(d[1])}, {vr(d[2])}, {vr(d[1])}",
              f"v_div_scale_f32 {vr(d[2])}, vcc, 1.0, {vr(LRUN)}, 1.0", f"v_mul_f32 {vr(d[3])}, {vr(d[2])}, {vr(d[1])}",
              f"v_fma_f32 {vr(d[4])}, -{vr(d[0])}, {vr(d[3])}, {vr(d[2])}", f"v_fmac_f32 {vr(d[3])}, {vr(d[4])}, {vr(d[1])}",
              f"v_fma_f32 {vr(d[0])}, -{vr(d[0])}, {vr(d[3])}, {vr(d[2])}", "s_nop 1",
              f"v_div_fmas_f32 {vr(d[0])}, {vr(d[0])}, {vr(d[1])}, {vr(d[3])}", f"v_div_fixup_f32 {vr(INV)}, {vr(d[0])}, {vr(LRUN)}, 1.0",
              f"v_cmp_lt_f32 vcc, 0, {vr(LRUN)}", "s_nop 1", f"v_cndmask_b32 {vr(INV)}, 0, {vr(INV)}, vcc",
              f"s_mul_i32 {sr(S_SOFF)}, {sr(S_Q0ROW)}, {sr(S_OSS)}", f"v_accvgpr_read_b32 {vr(S(0, 6))}, {ar(A_OOFF)}"]:
        e(i)
    for dt in range(4):
        for g4 in range(4):
            x = [S(0, i) for i in range(4)]
            for i in range(4):
                e(f"v_accvgpr_read_b32 {vr(x[i])}, {ar(O(dt, 4 * g4 + i))}")
            e("s_nop 0")
            for i in range(4):
                e(f"v_mul_f32 {vr(x[i])}, {vr(x[i])}, {vr(INV)}")
            e(f"v_cvt_pk_bf16_f32 {vr(S(0, 4))}, {vr(x[0])}, {vr(x[1])}")
            e(f"v_cvt_pk_bf16_f32 {vr(S(0, 5))}, {vr(x[2])}, {vr(x[3])}")
            e("s_nop 0")
            e(f"buffer_store_dwordx2 {vr(S(0, 4), 2)}, {vr(S(0, 6))}, {sr(S_OR, 4)}, {sr(S_SOFF)} offen offset:{dt * 64 + g4 * 16}")
            e("s_nop 1")
    # ---- the barriers of the tiles this wave does not have (T_w .. T_q - 1), then the next item
    e("s_branch C_IDLE_LOOP%=")
    e("C_IDLE%=:")
    e(f"s_mov_b32 {sr(S_CNT)}, 0")
    e("C_IDLE_LOOP%=:")
    # barriers executed so far inside the item (after B_start): B_0 .. B_{T_w - 1} (T_w of them; 0 when idle) -> T_q - T_w more
    e(f"s_cmp_lt_u32 {sr(S_CNT)}, {sr(S_TQ)}"); e("s_cbranch_scc0 C_NEXT%=")
    e("s_barrier"); e(f"s_add_u32 {sr(S_CNT)}, {sr(S_CNT)}, 1"); e("s_branch C_IDLE_LOOP%=")
    e("C_NEXT%=:")
    e(f"s_add_u32 {sr(S_ROUND)}, {sr(S_ROUND)}, 1")
    e("s_branch C_ITEM%=")
    for lb, back in G.stubs:
        e(f"{lb}:")
        for i in rescale_stub():
            e(i)
        e(f"s_branch {back}")
    e("C_DONE%=:")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_branch END%=")

    # ============ loader waves ========================================================================================================
    e("LOADER%=:")
    # slot j of loader i = global slot 4 j + i (i = wave - 4): piece p = slot (< 37) else padding piece slot - 37 (-> scratch)
    # per-lane source offset of a piece: padded chunk n = 64 p + lane; K (p < 17): row n / 17, c n % 17; V: n' = n - 1088, row n' / 20, c n' % 20; pad chunks fetch c = 0
    e(f"s_sub_u32 {sr(S_TMP3)}, {sr(S_WAVE)}, 4")
    for j in range(10):
        # p = 4 j + i (scalar, per wave); branchless: compute K-form and V-form, select by p < 17 (p >= 37 -> p - 37 < 17: K form)
        for i in [f"s_add_u32 {sr(S_TMP)}, {sr(S_TMP3)}, {4 * j}", f"s_sub_u32 {sr(S_TMP2)}, {sr(S_TMP)}, 37", f"s_cmp_ge_u32 {sr(S_TMP)}, 37",
                  f"s_cselect_b32 {sr(S_TMP)}, {sr(S_TMP2)}, {sr(S_TMP)}",                                 # piece index (padding pieces re-fetch K pieces 0..2)
                  f"s_lshl_b32 {sr(S_TMP2)}, {sr(S_TMP)}, 6", f"v_add_u32 {vr(10)}, {sr(S_TMP2)}, {vr(LANE)}",     # n = 64 p + lane
                  # K form
                  f"s_mov_b32 {sr(S_TMP2)}, 0xf0f0f10", f"v_mul_hi_u32 {vr(11)}, {vr(10)}, {sr(S_TMP2)}", f"v_mul_u32_u24 {vr(12)}, 17, {vr(11)}", f"v_sub_u32 {vr(12)}, {vr(10)}, {vr(12)}",     # row, c
                  f"v_cmp_gt_u32 vcc, 16, {vr(12)}", "s_nop 1", f"v_cndmask_b32 {vr(12)}, 0, {vr(12)}, vcc", f"v_lshlrev_b32 {vr(12)}, 4, {vr(12)}",
                  f"v_mul_lo_u32 {vr(13)}, {vr(11)}, {sr(S_KSS)}", f"v_add_u32 {vr(13)}, {vr(13)}, {vr(12)}",                                          # K offset
                  # V form
                  f"v_subrev_u32 {vr(14)}, 1088, {vr(10)}", f"s_mov_b32 {sr(S_TMP2)}, 0xccccccd", f"v_mul_hi_u32 {vr(11)}, {vr(14)}, {sr(S_TMP2)}", f"v_mul_u32_u24 {vr(12)}, 20, {vr(11)}",
                  f"v_sub_u32 {vr(12)}, {vr(14)}, {vr(12)}", f"v_cmp_gt_u32 vcc, 16, {vr(12)}", "s_nop 1", f"v_cndmask_b32 {vr(12)}, 0, {vr(12)}, vcc",
                  f"v_lshlrev_b32 {vr(12)}, 4, {vr(12)}", f"v_mul_lo_u32 {vr(14)}, {vr(11)}, {sr(S_VSS)}", f"v_add_u32 {vr(14)}, {vr(14)}, {vr(12)}",  # V offset
                  f"s_cmp_lt_u32 {sr(S_TMP)}, 17", f"s_cselect_b64 vcc, -1, 0", "s_nop 1", f"v_cndmask_b32 {vr(j)}, {vr(14)}, {vr(13)}, vcc"]:
            e(i)
    e("L_ITEM%=:")
    item_decode(G, loader=True)
    e("s_cbranch_scc1 L_DONE%=")
    # K / V descriptors of the item's kv head
    for (lo, hi, ss, dst) in ((S_K0, S_K1, S_KSS, S_KR), (S_V0, S_V1, S_VSS, S_VR)):
        for i in [f"s_lshl_b32 {sr(S_TMP)}, {sr(S_GRP)}, 8", f"s_add_u32 {sr(dst)}, {sr(lo)}, {sr(S_TMP)}", f"s_addc_u32 {sr(dst + 1)}, {sr(hi)}, 0",
                  f"s_and_b32 {sr(dst + 1)}, {sr(dst + 1)}, 0xffff", f"s_sub_u32 {sr(S_TMP)}, {sr(S_S)}, 1", f"s_mul_i32 {sr(S_TMP)}, {sr(S_TMP)}, {sr(ss)}",
                  f"s_add_u32 {sr(dst + 2)}, {sr(S_TMP)}, 256", f"s_mov_b32 {sr(dst + 3)}, 0x20000"]:
            e(i)
    e("s_barrier")                                     # B_start

    # generic "issue tile number S_TMP2 into ring stage (S_TMP2 % 4)": the piece's LDS destination = stage base + 1024 p (K, V) or scratch
    def issue():
        o = [f"s_and_b32 {sr(S_STAGE)}, {sr(S_TMP2)}, 3", f"s_mul_i32 {sr(S_STAGE)}, {sr(S_STAGE)}, {STG}", f"s_add_u32 {sr(S_STAGE)}, {sr(S_STAGE)}, {sr(S_LDS)}",
             f"s_lshl_b32 {sr(S_SOFF)}, {sr(S_TMP2)}, 6", f"s_mul_i32 {sr(S_DELTA)}, {sr(S_SOFF)}, {sr(S_KSS)}", f"s_mul_i32 {sr(S_SOFF)}, {sr(S_SOFF)}, {sr(S_VSS)}",
             f"s_sub_u32 {sr(S_U)}, {sr(S_WAVE)}, 4"]
        for j in range(10):
            # p = 4 j + i: K piece (p < 17), V piece (17 <= p < 37), padding (p >= 37)
            lk, lv, lp, ld = G.label("PK"), G.label("PV"), G.label("PP"), G.label("PD")
            o += [f"s_add_u32 {sr(S_TMP)}, {sr(S_U)}, {4 * j}", f"s_cmp_lt_u32 {sr(S_TMP)}, 17", f"s_cbranch_scc1 {lk}",
                  f"s_cmp_lt_u32 {sr(S_TMP)}, 37", f"s_cbranch_scc1 {lv}",
                  # padding piece -> scratch
                  f"s_sub_u32 {sr(S_TMP)}, {sr(S_TMP)}, 37", f"s_lshl_b32 {sr(S_TMP)}, {sr(S_TMP)}, 10", f"s_add_u32 m0, {sr(S_TMP)}, {sr(S_LDS)}", f"s_add_u32 m0, m0, {SCRATCH}",
                  "s_nop 0", f"buffer_load_dwordx4 {vr(j)}, {sr(S_KR, 4)}, {sr(S_DELTA)} offen lds", f"s_branch {ld}",
                  f"{lk}:", f"s_lshl_b32 {sr(S_TMP)}, {sr(S_TMP)}, 10", f"s_add_u32 m0, {sr(S_TMP)}, {sr(S_STAGE)}", "s_nop 0",
                  f"buffer_load_dwordx4 {vr(j)}, {sr(S_KR, 4)}, {sr(S_DELTA)} offen lds", f"s_branch {ld}",
                  f"{lv}:", f"s_lshl_b32 {sr(S_TMP)}, {sr(S_TMP)}, 10", f"s_add_u32 m0, {sr(S_TMP)}, {sr(S_STAGE)}", "s_nop 0",
                  f"buffer_load_dwordx4 {vr(j)}, {sr(S_VR, 4)}, {sr(S_SOFF)} offen lds",
                  f"{ld}:"]
        return o
    # tiles 0, 1 up front
    for t0 in range(2):
        lb = G.label("LSKIP")
        e(f"s_cmp_le_u32 {sr(S_TQ)}, {t0}"); e(f"s_cbranch_scc1 {lb}")
        e(f"s_mov_b32 {sr(S_TMP2)}, {t0}")
        for i in issue():
            e(i)
        e(f"{lb}:")
    e(f"s_mov_b32 {sr(S_T)}, 0")
    e("L_LOOP%=:")
    # wait for tile t: tile t + 1 (if it exists) stays in flight
    l2, l3 = G.label("LW2"), G.label("LWD")
    e(f"s_add_u32 {sr(S_TMP)}, {sr(S_T)}, 1"); e(f"s_cmp_lt_u32 {sr(S_TMP)}, {sr(S_TQ)}"); e(f"s_cbranch_scc0 {l2}")
    e("s_waitcnt vmcnt(10)"); e(f"s_branch {l3}")
    e(f"{l2}:")
    e("s_waitcnt vmcnt(0)")
    e(f"{l3}:")
    e("s_barrier")                                     # B_t: tile t is in; the compute waves are done with tile t - 2 (its stage = that of tile t + 2)
    lb = G.label("LNO")
    e(f"s_add_u32 {sr(S_TMP2)}, {sr(S_T)}, 2"); e(f"s_cmp_lt_u32 {sr(S_TMP2)}, {sr(S_TQ)}"); e(f"s_cbranch_scc0 {lb}")
    for i in issue():
        e(i)
    e(f"{lb}:")
    e(f"s_add_u32 {sr(S_T)}, {sr(S_T)}, 1")
    e(f"s_cmp_lt_u32 {sr(S_T)}, {sr(S_TQ)}"); e("s_cbranch_scc1 L_LOOP%=")
    e(f"s_add_u32 {sr(S_ROUND)}, {sr(S_ROUND)}, 1")
    e("s_branch L_ITEM%=")
    e("L_DONE%=:")
    e("s_waitcnt vmcnt(0)")
    e("END%=:")
    return G.lines


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
    clob = [f"v{i}" for i in range(128)] + [f"a{i}" for i in range(120)] + [f"s{i}" for i in range(38, 100)] + ["vcc", "scc", "memory"]
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_attn_c128.py%s -- do not edit.  %d instructions.\n" % (" --simple" if simple else "", len(lines)))
        f.write("#define UFV_ATTN_C128_ASM \\\n")
        for l in lines:
            f.write('    "%s\\n\\t" \\\n' % l)
        f.write('    ""\n')
        f.write("#define UFV_ATTN_C128_CLOBBERS " + ", ".join('"%s"' % c for c in clob) + "\n")
        f.write("#define UFV_ATTN_C128_LDS %d\n" % LDS_BYTES)
    print(out, len(lines), "lines")


if __name__ == "__main__":
    main()
