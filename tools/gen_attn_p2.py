#!/usr/bin/env python3
"""Generator of the third-generation ViT attention kernel body (head_dim 72, non-causal, S = 576): writes
ufvideo_amd/csrc/attn_vit_p2_asm.inc -- ONE inline-asm statement that owns the whole 512-register file of a wave.

Structure (cdna_hip_programming.md, 'Fused attention prefill', 4-wave one-wave-per-SIMD form, adapted to hd 72 / S 576):
  * block = 4 waves = 2 heads; wave w works on head w >> 1, query rows [288 (w & 1), +288) in 3 passes of 96 rows (3 units of 32);
    every SIMD holds exactly one wave and every wave does the same work: 3 passes x 9 key tiles x 3 units x 22 MFMA 32x32x16.
  * software pipeline at unit granularity.  Period (j, u) of key tile j issues the MFMAs  PV(item before) [12]  +  QK^T(item after)
    [10]  while the vector pipe runs the softmax of item (j, u) in the gaps: max -> (rare) rescale -> exp2 -> bf16 pack.
  * K/V tiles by LDS-DMA (buffer_load ... lds) into a 2-stage ring per head, one tile ahead, one s_barrier per tile; K fragments
    (ds_read_b128) and V^T fragments (ds_read_b64_tr_b16) land in AGPRs and are shared by the wave's 3 units.
  * Q of the next pass is DMA'd into a per-wave staging area during the pass, O leaves through the same area as whole rows.
  * numerics identical to attn_fwd_vit72<9> (attn_vit.inc): pre-scaled bf16 q, running max carried in the contraction padding, row
    sums out of the PV MFMAs, deferred rescale at 2^6 decided per 32-query unit -- the outputs are bit-identical.

The order of instructions inside each period is decided here (gap placement tables), not by hipcc.
"""
import os
import sys

HD = 72
PK = 144                  # K / V / Q / O row pitch in LDS (9 chunks of 16 B)
KT = 64 * PK              # 9216
VOFF = KT + 128
STG = VOFF + KT + 128     # 18688 per stage (K | pad | V | pad)
NPIECE = 9                # DMA pieces per wave per tile (a wave loads the K tile or the V tile of its head)
QPIECES = 14              # 96 rows x 144 B = 13.5 KiB -> 14 pieces
UNIT_BYTES = 32 * PK      # 4608
NT = 9                    # key tiles per pass (S = 576)
NPASS = 3
THR = 0x40C00000          # 6.0f

# ---- register map ------------------------------------------------------------------------------------------------------------
def S(u, r): return 32 * u + r                      # v: scores of unit u, r = 0..31 (0..15 keys 0-31 tile, 16..31 keys 32-63)
def P(u, k): return 96 + 16 * u + k                 # v: packed bf16 P, k = 0..15
def Q(u, ks, i=0): return 144 + 20 * u + 4 * ks + i  # v: Q fragments
def VD2(c, i=0): return 204 + 4 * c + i             # v: V^T fragments of d-tile 2 (ones row substituted)
VOFFR = lambda k: 220 + k                           # v220..228: DMA per-lane source offsets
KADDR, K4A0, K4A1, VADDR = 229, 230, 231, 232
T0, T1, TA, TB, TC, TD = 233, 234, 235, 236, 237, 238
MRUN = lambda u: 239 + u
ONES, LANE, QADDR, OWADDR, ORADDR = 242, 243, 244, 245, 246
VOFFO = lambda i: 247 + i                           # v247..251
TE, TF, TG, TH = 252, 253, 254, 255

def O(u, dt, r=0): return 48 * u + 16 * dt + r      # a: O^T accumulators
def KF(ks, half, i=0): return 144 + 8 * ks + 4 * half + i   # a: K fragments
def VF(dt, c, i=0): return 184 + 16 * dt + 4 * c + i        # a: V^T fragments of d-tiles 0, 1

# fixed SGPRs (clobbered): s40..s99
KVR, QR, ORS = 40, 44, 48
S_ONR, S_SS, S_OS, S_T64, S_SC, S_RING, S_DST, S_DDST, S_DRD, S_QST, S_QSOFF, S_OSOFF, S_PASS, S_TMP = range(52, 66)
S_HM, S_ONE, S_RET, S_T2, S_MAGIC, S_KONE, S_T3, S_T4, S_EXLO = 66, 68, 70, 72, 74, 75, 76, 77, 78
S_ORN = 80          # current num_records word of the O descriptor (0 in pass 0: stores dropped)
S_ODESC = 84        # s[84:87]: O descriptor actually used by the stores


class Gen:
    def __init__(self):
        self.lines = []
        self.vm_log = []        # VMEM ops issued, tags
        self.label_n = 0
        self.stubs = []

    def e(self, s):
        self.lines.append(s)

    def label(self, base):
        self.label_n += 1
        return f"{base}_{self.label_n}%="


def vr(n, cnt=1):
    return f"v{n}" if cnt == 1 else f"v[{n}:{n + cnt - 1}]"


def ar(n, cnt=1):
    return f"a{n}" if cnt == 1 else f"a[{n}:{n + cnt - 1}]"


def sr(n, cnt=1):
    return f"s{n}" if cnt == 1 else f"s[{n}:{n + cnt - 1}]"


# ---- instruction groups ------------------------------------------------------------------------------------------------------
def mfma_pv(u):
    """O[u][dt] += V^T(dt, c) * P[u](c), c = 0..3 in order per dt (the accumulation order of attn_vit.inc)"""
    out = []
    for dt in range(3):
        for c in range(4):
            a = ar(VF(dt, c), 4) if dt < 2 else vr(VD2(c), 4)
            out.append(f"v_mfma_f32_32x32x16_bf16 {ar(O(u, dt), 16)}, {a}, {vr(P(u, 4 * c), 4)}, {ar(O(u, dt), 16)}")
    return out


def mfma_qk(u):
    out = []
    for ks in range(5):
        for half in range(2):
            d = vr(S(u, 16 * half), 16)
            out.append(f"v_mfma_f32_32x32x16_bf16 {d}, {ar(KF(ks, half), 4)}, {vr(Q(u, ks), 4)}, {'0' if ks == 0 else d}")
    return out


def sm_max(u):
    """tile max of unit u in T0 (all lanes of a query agree), two interleaved chains"""
    s = lambda r: vr(S(u, r))
    o = [f"v_max_f32 {vr(T0)}, {s(0)}, {s(16)}", f"v_max_f32 {vr(T1)}, {s(1)}, {s(17)}"]
    for r in range(2, 16, 2):
        o.append(f"v_max3_f32 {vr(T0)}, {vr(T0)}, {s(r)}, {s(16 + r)}")
        o.append(f"v_max3_f32 {vr(T1)}, {vr(T1)}, {s(r + 1)}, {s(17 + r)}")
    # NOTE: max is exact and order-independent, so the two-chain form gives the bits of the single chain
    o += [f"v_max_f32 {vr(T0)}, {vr(T0)}, {vr(T1)}", f"v_mov_b32 {vr(T1)}, {vr(T0)}", "s_nop 1",
          f"v_permlane32_swap_b32 {vr(T0)}, {vr(T1)}", f"v_max_f32 {vr(T0)}, {vr(T0)}, {vr(T1)}"]
    return o


def sm_exp_cvt(u):
    """32 exp2 in place + 16 packs; pack k follows its two exps by >= 2 instructions"""
    ex = [f"v_exp_f32 {vr(S(u, r))}, {vr(S(u, r))}" for r in range(32)]
    cv = [f"v_cvt_pk_bf16_f32 {vr(P(u, k))}, {vr(S(u, 2 * k))}, {vr(S(u, 2 * k + 1))}" for k in range(16)]
    out = []
    ei = ci = 0
    # pattern: e e e e then (c e e) ... keeps a pack 3+ instructions behind its inputs
    out += ex[:4]; ei = 4
    while ci < 16:
        out.append(cv[ci]); ci += 1
        out += ex[ei:ei + 2]; ei = min(32, ei + 2)
    assert ei == 32
    return out


def rescale_math(u, first):
    """m_new = bf16(m_run + (first ? tmax : max(tmax, 0))); delta = m_new - m_run; returns instrs; leaves delta in TA, m_new in MRUN"""
    o = []
    if first:
        o += [f"v_cvt_pk_bf16_f32 {vr(TB)}, {vr(T0)}, {vr(T0)}", f"v_lshlrev_b32 {vr(TA)}, 16, {vr(TB)}",   # TA = m_new (m_run was 0) = delta
              f"v_mov_b32 {vr(MRUN(u))}, {vr(TA)}"]
    else:
        o += [f"v_max_f32 {vr(TB)}, 0, {vr(T0)}", f"v_add_f32 {vr(TB)}, {vr(MRUN(u))}, {vr(TB)}",
              f"v_cvt_pk_bf16_f32 {vr(TB)}, {vr(TB)}, {vr(TB)}", f"v_lshlrev_b32 {vr(TB)}, 16, {vr(TB)}",
              f"v_sub_f32 {vr(TA)}, {vr(TB)}, {vr(MRUN(u))}", f"v_mov_b32 {vr(MRUN(u))}, {vr(TB)}"]
    # q'[4][0] of the h = 1 lanes <- bf16(-m_new) (upper half: 0)
    o += [f"v_cvt_pk_bf16_f32 {vr(TB)}, -{vr(MRUN(u))}, 0", f"v_cndmask_b32 {vr(Q(u, 4))}, {vr(Q(u, 4))}, {vr(TB)}, {sr(S_HM, 2)}"]
    o += [f"v_sub_f32 {vr(S(u, r))}, {vr(S(u, r))}, {vr(TA)}" for r in range(32)]
    return o


def rescale_o(u):
    """O[u] *= exp2(-delta) (delta in TA); O rows live in AGPRs"""
    o = [f"v_exp_f32 {vr(TC)}, -{vr(TA)}", "s_nop 0"]
    for r in range(48):
        o += [f"v_accvgpr_read_b32 {vr(TD)}, {ar(O(u, 0, r))}", "s_nop 0", f"v_mul_f32 {vr(TD)}, {vr(TD)}, {vr(TC)}", "s_nop 0",
              f"v_accvgpr_write_b32 {ar(O(u, 0, r))}, {vr(TD)}"]
    return o


def k_reads():
    """K fragments of the tile whose (per-lane) addresses are in KADDR / K4A0 / K4A1"""
    o = []
    for ks in range(4):
        for half in range(2):
            o.append(f"ds_read_b128 {ar(KF(ks, half), 4)}, {vr(KADDR)} offset:{ks * 32 + half * 32 * PK}")
    o.append(f"ds_read_b128 {ar(KF(4, 0), 4)}, {vr(K4A0)}")
    o.append(f"ds_read_b128 {ar(KF(4, 1), 4)}, {vr(K4A1)}")
    return o


def v_reads():
    """V^T fragments (tr reads); d-tile 2 first (it needs the ones-row substitution afterwards)"""
    o = []
    for dt in (2, 0, 1):
        for c in range(4):
            for half in range(2):
                off = dt * 64 + c * 16 * PK + half * 8 * PK
                dst = vr(VD2(c, 2 * half), 2) if dt == 2 else ar(VF(dt, c, 2 * half), 2)
                o.append(f"ds_read_b64_tr_b16 {dst}, {vr(VADDR)} offset:{off}")
    return o


def vd2_ones():
    return [f"v_cndmask_b32 {vr(VD2(c, i))}, {vr(VD2(c, i))}, {vr(ONES)}, {sr(S_ONE, 2)}" for c in range(4) for i in range(4)]


def addr_flip():
    """move the K / V fragment read addresses to the other ring stage"""
    return [f"v_add_u32 {vr(KADDR)}, {sr(S_DRD)}, {vr(KADDR)}", f"v_add_u32 {vr(VADDR)}, {sr(S_DRD)}, {vr(VADDR)}",
            f"v_add_u32 {vr(TE)}, 128, {vr(KADDR)}", f"v_add_u32 {vr(TF)}, {128 + 32 * PK}, {vr(KADDR)}",
            f"s_sub_u32 {sr(S_DRD)}, 0, {sr(S_DRD)}",
            f"v_cndmask_b32 {vr(K4A0)}, {vr(TE)}, {vr(TG)}, {sr(S_HM, 2)}", f"v_cndmask_b32 {vr(K4A1)}, {vr(TF)}, {vr(TG)}, {sr(S_HM, 2)}"]
    # TG holds the KONE address (set in the prologue and never reused)


def dma_piece(rsrc, voff, soff, m0_expr):
    """one LDS-DMA piece; m0_expr = list of SALU instrs that leave the destination in m0"""
    return m0_expr + ["s_nop 0", f"buffer_load_dwordx4 {vr(voff)}, {sr(rsrc, 4)}, {soff} offen lds"]


def q_load(u, tmp_bank):
    """Q fragments of unit u from the staging area (rows 32u..32u+31) into Q(u, ks), pre-multiplied by scale*log2e and re-rounded to bf16;
    d >= 72 (ks = 4, h = 1 lanes) zeroed.  tmp_bank: two free VGPRs."""
    a, b = tmp_bank
    o = [f"ds_read_b128 {vr(Q(u, ks), 4)}, {vr(QADDR)} offset:{u * UNIT_BYTES + ks * 32}" for ks in range(5)]
    o.append("s_waitcnt lgkmcnt(0)")
    for ks in range(5):
        for i in range(4):
            r = vr(Q(u, ks, i))
            o += [f"v_lshlrev_b32 {vr(a)}, 16, {r}", f"v_and_b32 {vr(b)}, 0xffff0000, {r}", f"v_mul_f32 {vr(a)}, {sr(S_SC)}, {vr(a)}",
                  f"v_mul_f32 {vr(b)}, {sr(S_SC)}, {vr(b)}", "s_nop 0", f"v_cvt_pk_bf16_f32 {r}, {vr(a)}, {vr(b)}"]
    o += [f"v_cndmask_b32 {vr(Q(u, 4, i))}, {vr(Q(u, 4, i))}, 0, {sr(S_HM, 2)}" for i in range(4)]
    return o


def drain(u, bank):
    """O of unit u (of the pass that just ended) -> normalise -> bf16 -> staging rows -> global; zero the accumulators.
    bank: base of 32 free VGPRs.  The store offset (rows of the ended pass, unit u) is S_OSOFF + 32 u OS."""
    x = [bank + i for i in range(4)]
    y = [bank + 4, bank + 5]
    l, l2, inv = bank + 6, bank + 7, bank + 8
    d = [bank + 9 + i for i in range(6)]
    rows = [bank + 16 + 4 * i for i in range(4)]         # 4 x 4 regs for row chunks (5 chunks: the 5th reuses the first)
    o = [f"v_accvgpr_read_b32 {vr(l)}, {ar(O(u, 2, 4))}", "s_nop 0", f"v_mov_b32 {vr(l2)}, {vr(l)}", "s_nop 1",
         f"v_permlane32_swap_b32 {vr(l)}, {vr(l2)}",
         # inv = 1.0f / l, IEEE (the sequence hipcc emits for the division in attn_vit.inc)
         f"v_div_scale_f32 {vr(d[0])}, {sr(S_T2, 2)}, {vr(l)}, {vr(l)}, 1.0", f"v_rcp_f32 {vr(d[1])}, {vr(d[0])}", "s_nop 0",
         f"v_fma_f32 {vr(d[2])}, -{vr(d[0])}, {vr(d[1])}, 1.0", f"v_fmac_f32 {vr(d[1])}, {vr(d[2])}, {vr(d[1])}",
         f"v_div_scale_f32 {vr(d[2])}, vcc, 1.0, {vr(l)}, 1.0", f"v_mul_f32 {vr(d[3])}, {vr(d[2])}, {vr(d[1])}",
         f"v_fma_f32 {vr(d[4])}, -{vr(d[0])}, {vr(d[3])}, {vr(d[2])}", f"v_fmac_f32 {vr(d[3])}, {vr(d[4])}, {vr(d[1])}",
         f"v_fma_f32 {vr(d[0])}, -{vr(d[0])}, {vr(d[3])}, {vr(d[2])}", "s_nop 1",
         f"v_div_fmas_f32 {vr(d[0])}, {vr(d[0])}, {vr(d[1])}, {vr(d[3])}", f"v_div_fixup_f32 {vr(inv)}, {vr(d[0])}, {vr(l)}, 1.0",
         f"v_cmp_lt_f32 vcc, 0, {vr(l)}", "s_nop 1", f"v_cndmask_b32 {vr(inv)}, 0, {vr(inv)}, vcc"]
    groups = [(dt, g4) for dt in range(2) for g4 in range(4)] + [(2, 0)]
    for dt, g4 in groups:
        for i in range(4):
            o.append(f"v_accvgpr_read_b32 {vr(x[i])}, {ar(O(u, dt, 4 * g4 + i))}")
        o.append("s_nop 0")
        for i in range(4):
            o.append(f"v_mul_f32 {vr(x[i])}, {vr(inv)}, {vr(x[i])}")
        o += ["s_nop 0", f"v_cvt_pk_bf16_f32 {vr(y[0])}, {vr(x[0])}, {vr(x[1])}", f"v_cvt_pk_bf16_f32 {vr(y[1])}, {vr(x[2])}, {vr(x[3])}", "s_nop 0",
              f"ds_write_b64 {vr(OWADDR)}, {vr(y[0], 2)} offset:{u * UNIT_BYTES + dt * 64 + g4 * 16}"]
    o += [f"v_accvgpr_write_b32 {ar(O(u, 0, r))}, 0" for r in range(48)]
    o += ["s_waitcnt lgkmcnt(0)"]
    # 288 chunks of 16 B: lane c = 64 i + lane; i = 4 covers c = 256..287 (lanes 0..31)
    o += [f"s_mul_i32 {sr(S_T3)}, {sr(S_OS)}, {32 * u}", f"s_add_u32 {sr(S_T3)}, {sr(S_OSOFF)}, {sr(S_T3)}"]
    for i in range(4):
        o.append(f"ds_read_b128 {vr(rows[i], 4)}, {vr(ORADDR)} offset:{u * UNIT_BYTES + 1024 * i}")
    o.append("s_waitcnt lgkmcnt(0)")
    for i in range(4):
        o.append(("VMEM", f"buffer_store_dwordx4 {vr(rows[i], 4)}, {vr(VOFFO(i))}, {sr(S_ODESC, 4)}, {sr(S_T3)} offen"))
    o += ["s_nop 1", f"ds_read_b128 {vr(rows[0], 4)}, {vr(ORADDR)} offset:{u * UNIT_BYTES + 4096}", "s_waitcnt lgkmcnt(0)",
          f"s_mov_b64 exec, {sr(S_EXLO, 2)}",
          ("VMEM", f"buffer_store_dwordx4 {vr(rows[0], 4)}, {vr(VOFFO(4))}, {sr(S_ODESC, 4)}, {sr(S_T3)} offen"),
          "s_mov_b64 exec, -1", "s_nop 1"]
    return o


# ---- period scheduler: fillers are (gap, priority, [instrs]) --------------------------------------------------------------
def spread(instrs, g0, g1):
    """assign instruction k of a stream to gaps g0..g1 evenly (in order)"""
    n = len(instrs)
    out = []
    for k, ins in enumerate(instrs):
        g = g0 + (k * (g1 - g0 + 1)) // max(n, 1)
        out.append((g, ins))
    return out


class Period:
    def __init__(self, mfmas):
        self.mfmas = mfmas
        self.gaps = [[] for _ in range(len(mfmas) + 1)]      # gap 0 = before the first MFMA, gap g = after MFMA g - 1

    def put(self, placed):
        for g, ins in placed:
            self.gaps[min(max(g, 0), len(self.mfmas))].append(ins)

    def emit(self, G):
        for ins in self.gaps[0]:
            G.emit_ins(ins)
        for g, m in enumerate(self.mfmas):
            G.e(m)
            for ins in self.gaps[g + 1]:
                G.emit_ins(ins)


def emit_ins(G, ins):
    if isinstance(ins, tuple):
        kind, text = ins
        if kind == "VMEM":
            G.vm_log.append(text)
            G.e(text)
        elif kind == "VMWAIT":          # wait until the ops tagged `text` (and everything older) are done: count what was issued after them
            idx = max(i for i, t in enumerate(G.vm_log) if t.startswith("TAG:" + text))
            later = sum(1 for t in G.vm_log[idx + 1:] if not t.startswith("TAG:"))
            G.e(f"s_waitcnt vmcnt({later})")
        elif kind == "TAG":
            G.vm_log.append("TAG:" + text)
        elif kind == "RAW":
            G.e(text)
    else:
        G.e(ins)


Gen.emit_ins = emit_ins


def build(simple=False):
    """simple=True: every period's vector work is emitted AFTER its MFMAs (no interleave): the bring-up form"""
    G = Gen()
    e = G.e
    # ================= prologue ==================================================================================================
    ins = [
        f"s_mov_b32 {sr(KVR + 0)}, %[kv0]", f"s_mov_b32 {sr(KVR + 1)}, %[kv1]", f"s_mov_b32 {sr(KVR + 2)}, %[kv2]", f"s_mov_b32 {sr(KVR + 3)}, 0x20000",
        f"s_mov_b32 {sr(QR + 0)}, %[q0]", f"s_mov_b32 {sr(QR + 1)}, %[q1]", f"s_mov_b32 {sr(QR + 2)}, %[q2]", f"s_mov_b32 {sr(QR + 3)}, 0x20000",
        f"s_mov_b32 {sr(ORS + 0)}, %[o0]", f"s_mov_b32 {sr(ORS + 1)}, %[o1]", f"s_mov_b32 {sr(ORS + 2)}, %[o2]", f"s_mov_b32 {sr(ORS + 3)}, 0x20000",
        f"s_mov_b32 {sr(S_SS)}, %[ss]", f"s_mov_b32 {sr(S_OS)}, %[os]",
        # scale * log2(e) in fp32 (one rounding, as attn_vit.inc computes it); %[sc] holds the bits of `scale`
        f"v_mov_b32 {vr(TA)}, %[sc]", f"v_mul_f32 {vr(TA)}, 0x3fb8aa3b, {vr(TA)}", "s_nop 0", f"v_readfirstlane_b32 {sr(S_SC)}, {vr(TA)}", f"s_mov_b32 {sr(S_RING)}, %[ring]",
        f"s_mov_b32 {sr(S_DST)}, %[dst]", f"s_mov_b32 {sr(S_QST)}, %[qst]", f"s_mov_b32 {sr(S_QSOFF)}, %[qsoff]", f"s_mov_b32 {sr(S_OSOFF)}, %[osoff]",
        f"s_mov_b32 {sr(S_KONE)}, %[kone]",
        f"s_lshl_b32 {sr(S_T64)}, {sr(S_SS)}, 6", f"s_mov_b32 {sr(S_MAGIC)}, 0x1c71c71d",
        f"s_mov_b32 {sr(S_HM)}, 0", f"s_mov_b32 {sr(S_HM + 1)}, -1", f"s_mov_b32 {sr(S_ONE)}, 0x100", f"s_mov_b32 {sr(S_ONE + 1)}, 0x100",
        f"s_mov_b32 {sr(S_EXLO)}, -1", f"s_mov_b32 {sr(S_EXLO + 1)}, 0", f"s_mov_b32 {sr(S_PASS)}, 0",
        f"s_mov_b32 {sr(S_DDST)}, {STG}", f"s_mov_b32 {sr(S_DRD)}, {STG}",
        f"v_mbcnt_lo_u32_b32 {vr(LANE)}, -1, 0", f"v_mbcnt_hi_u32_b32 {vr(LANE)}, -1, {vr(LANE)}",
        f"v_mov_b32 {vr(ONES)}, 0x3f803f80",
    ]
    for k in range(9):       # DMA source offsets: chunk c = 64 k + lane -> row c / 9, column chunk c % 9
        ins += [f"v_add_u32 {vr(TA)}, {64 * k}, {vr(LANE)}", f"v_mul_hi_u32 {vr(TB)}, {vr(TA)}, {sr(S_MAGIC)}", f"v_mul_u32_u24 {vr(TC)}, 9, {vr(TB)}",
                f"v_sub_u32 {vr(TC)}, {vr(TA)}, {vr(TC)}", f"v_lshlrev_b32 {vr(TC)}, 4, {vr(TC)}", f"v_mul_lo_u32 {vr(TD)}, {vr(TB)}, {sr(S_SS)}",
                f"v_add_u32 {vr(VOFFR(k))}, {vr(TD)}, {vr(TC)}"]
        if k < 5:
            ins += [f"v_mul_lo_u32 {vr(TD)}, {vr(TB)}, {sr(S_OS)}", f"v_add_u32 {vr(VOFFO(k))}, {vr(TD)}, {vr(TC)}"]
    ins += [
        f"v_and_b32 {vr(TA)}, 31, {vr(LANE)}", f"v_lshrrev_b32 {vr(TB)}, 5, {vr(LANE)}",          # TA = l31, TB = h
        f"v_mul_u32_u24 {vr(TC)}, {PK}, {vr(TA)}",                                                 # TC = l31 * 144
        f"v_lshlrev_b32 {vr(TD)}, 4, {vr(TB)}",                                                    # TD = 16 h
        f"v_add3_u32 {vr(KADDR)}, {vr(TC)}, {vr(TD)}, {sr(S_RING)}",
        f"v_mov_b32 {vr(TG)}, {sr(S_KONE)}",
        f"v_add_u32 {vr(TE)}, 128, {vr(KADDR)}", f"v_add_u32 {vr(TF)}, {128 + 32 * PK}, {vr(KADDR)}",
        f"v_cndmask_b32 {vr(K4A0)}, {vr(TE)}, {vr(TG)}, {sr(S_HM, 2)}", f"v_cndmask_b32 {vr(K4A1)}, {vr(TF)}, {vr(TG)}, {sr(S_HM, 2)}",
        f"v_add3_u32 {vr(QADDR)}, {vr(TC)}, {vr(TD)}, {sr(S_QST)}",
        f"v_lshlrev_b32 {vr(TD)}, 3, {vr(TB)}", f"v_add3_u32 {vr(OWADDR)}, {vr(TC)}, {vr(TD)}, {sr(S_QST)}",
        f"v_lshlrev_b32 {vr(TD)}, 4, {vr(LANE)}", f"v_add_u32 {vr(ORADDR)}, {sr(S_QST)}, {vr(TD)}",
        # vaddr = ring + VOFF + (4 h + ((lane & 15) >> 2)) * 144 + (16 ((lane >> 4) & 1) + 4 (lane & 3)) * 2
        f"v_and_b32 {vr(TC)}, 15, {vr(LANE)}", f"v_lshrrev_b32 {vr(TC)}, 2, {vr(TC)}", f"v_lshl_add_u32 {vr(TC)}, {vr(TB)}, 2, {vr(TC)}",
        f"v_mul_u32_u24 {vr(TC)}, {PK}, {vr(TC)}",
        f"v_bfe_u32 {vr(TD)}, {vr(LANE)}, 4, 1", f"v_lshlrev_b32 {vr(TD)}, 5, {vr(TD)}",
        f"v_and_b32 {vr(TE)}, 3, {vr(LANE)}", f"v_lshl_add_u32 {vr(TD)}, {vr(TE)}, 3, {vr(TD)}",
        f"v_add3_u32 {vr(VADDR)}, {vr(TC)}, {vr(TD)}, {sr(S_RING)}", f"v_add_u32 {vr(VADDR)}, {VOFF}, {vr(VADDR)}",
    ]
    ins += [f"v_accvgpr_write_b32 {ar(r)}, 0" for r in range(256)]
    ins += [f"v_mov_b32 {vr(r)}, 0" for r in list(range(0, 144)) + list(range(204, 220)) + [MRUN(0), MRUN(1), MRUN(2)]]
    for i in ins:
        e(i)
    # Q of pass 0 -> staging; K/V tiles 0 and 1 -> ring stages 0 and 1
    for k in range(QPIECES):
        so = sr(S_QSOFF) if k < 9 else sr(S_T3)
        if k == 9:
            e(f"s_add_u32 {sr(S_T3)}, {sr(S_QSOFF)}, {sr(S_T64)}")
        for i in dma_piece(QR, VOFFR(k % 9), so, [f"s_add_u32 m0, {sr(S_QST)}, {1024 * k}"]):
            e(i)
    e(f"s_mul_i32 {sr(S_T4)}, {sr(S_SS)}, 96")
    e(f"s_add_u32 {sr(S_QSOFF)}, {sr(S_QSOFF)}, {sr(S_T4)}")            # -> rows of pass 1
    for t in range(2):
        for k in range(NPIECE):
            so = "0" if t == 0 else sr(S_T64)
            for i in dma_piece(KVR, VOFFR(k), so, [f"s_add_u32 m0, {sr(S_DST)}, {1024 * k + t * STG}"]):
                e(i)
    # next DMA (tile 2) goes to stage 0: S_DST stays; S_DDST = +STG is the toggle applied after each tile's DMA
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_barrier")
    for u in range(3):
        for i in q_load(u, (TA, TB)):
            e(i)
    for i in k_reads():
        e(i)
    e("s_waitcnt lgkmcnt(0)")
    for m in mfma_qk(0):
        e(m)
    # O descriptor of pass 0's (non-existent) predecessor: num_records = 0 -> every store is dropped
    e(f"s_mov_b32 {sr(S_ODESC)}, {sr(ORS)}"); e(f"s_mov_b32 {sr(S_ODESC + 1)}, {sr(ORS + 1)}"); e(f"s_mov_b32 {sr(S_ODESC + 2)}, 0"); e(f"s_mov_b32 {sr(S_ODESC + 3)}, {sr(ORS + 3)}")
    e("s_nop 7"); e("s_nop 7")

    # ================= pass loop =================================================================================================
    e("PASS_LOOP%=:")
    G.vm_log = ["TAG:kv_next"]           # at loop entry the pieces of the NEXT tile (tile 1) are already complete (prologue / previous pass)
    for j in range(NT):
        for u in range(3):
            prev_u = (u + 2) % 3
            next_u = (u + 1) % 3
            per = Period(mfma_pv(prev_u) + mfma_qk(next_u))
            free_bank = S((u + 2) % 3, 0)           # 32 VGPRs nobody owns during period (., u)
            stream_sm = []
            # ---- softmax of item (j, u)
            mx = sm_max(u)
            if j == 0:
                body = mx + rescale_math(u, True)
            else:
                stub, back = G.label("RS"), G.label("BK")
                body = mx + [f"v_cmp_lt_f32 vcc, 0x{THR:08x}, {vr(T0)}", "s_nop 1", f"s_cbranch_vccnz {stub}", f"{back}:"]
                G.stubs.append((stub, back, u))
            ex = sm_exp_cvt(u)
            if simple:
                per.put([(22, i) for i in body + ex])
            else:
                per.put(spread(body, 3, 8 if j else 11))
                per.put(spread(ex, 9 if j else 12, 22))
            # ---- LDS / DMA / address work
            if u == 0:
                # V^T fragments of tile j: after the 12 PV MFMAs of the previous tile's last unit have been issued
                vrd = v_reads()
                per.put([(22, i) for i in vrd] if simple else spread(vrd, 13, 20))
                per.put([(22, "s_waitcnt lgkmcnt(0)")])
                # drain of the previous pass's unit 0 happens in this period when j == 0 (below)
            if u == 1:
                per.put([(0, i) for i in vd2_ones()] if simple else spread(vd2_ones(), 1, 6))
            if u == 2:
                # barrier B(j+1): my pieces of tile j+1 are in, everybody is done with tile j's LDS image
                per.put([(0, ("VMWAIT", "kv_next")), (0, "s_barrier")])
                fl = addr_flip()
                per.put([(0, i) for i in fl])
                kr = k_reads()
                per.put([(1, i) for i in kr] if simple else spread(kr, 1, 6))
                per.put([(11, "s_waitcnt lgkmcnt(0)")])
                # DMA of tile j + 2 into the stage everybody just left
                j2 = (j + 2) % NT
                dm = [f"s_mul_i32 {sr(S_TMP)}, {sr(S_T64)}, {j2}"]
                for k in range(NPIECE):
                    dm += [f"s_add_u32 m0, {sr(S_DST)}, {1024 * k}", "s_nop 0",
                           ("VMEM", f"buffer_load_dwordx4 {vr(VOFFR(k))}, {sr(KVR, 4)}, {sr(S_TMP)} offen lds")]
                dm += [("TAG", "kv_next"), f"s_add_u32 {sr(S_DST)}, {sr(S_DST)}, {sr(S_DDST)}", f"s_sub_u32 {sr(S_DDST)}, 0, {sr(S_DDST)}"]
                per.put([(12, i) for i in dm] if simple else spread(dm, 7, 21))
            # ---- pass seams
            if j == 0:
                dr = drain(u, free_bank)
                per.put([(22, i) for i in dr])
                if u == 2:
                    # the O descriptor becomes real after pass 0's drains; the store rows advance by 96 per pass
                    per.put([(22, f"s_cmp_eq_u32 {sr(S_PASS)}, 0"), (22, f"s_cselect_b32 {sr(S_T3)}, 0, 1"),
                             (22, f"s_mul_i32 {sr(S_T4)}, {sr(S_OS)}, 96"), (22, f"s_mul_i32 {sr(S_T4)}, {sr(S_T4)}, {sr(S_T3)}"),
                             (22, f"s_add_u32 {sr(S_OSOFF)}, {sr(S_OSOFF)}, {sr(S_T4)}"), (22, f"s_mov_b32 {sr(S_ODESC + 2)}, {sr(ORS + 2)}")])
            if 2 <= j <= 8 and u == 1:
                # Q of the next pass: 2 pieces per tile (tiles 2..8 -> 14 pieces)
                qd = []
                for k in (2 * (j - 2), 2 * (j - 2) + 1):
                    if k == 9 or (k > 9 and k % 2 == 0):
                        qd.append(f"s_add_u32 {sr(S_T3)}, {sr(S_QSOFF)}, {sr(S_T64)}")
                    so = sr(S_QSOFF) if k < 9 else sr(S_T3)
                    qd += [f"s_add_u32 m0, {sr(S_QST)}, {1024 * k}", "s_nop 0", ("VMEM", f"buffer_load_dwordx4 {vr(VOFFR(k % 9))}, {sr(QR, 4)}, {so} offen lds")]
                per.put([(10, i) for i in qd] if simple else spread(qd, 8, 12))
            if j == 8 and u == 1:
                # everything DMA'd for the next pass's Q is in before its fragments are read (this wave's own pieces: vmcnt is enough)
                per.put([(22, "s_waitcnt vmcnt(0)")])
                per.put([(22, i) for i in q_load(0, (free_bank, free_bank + 1))])
                G.vm_log = [t for t in G.vm_log if t.startswith("TAG:")][-1:]      # everything older is complete
                G.vm_log = ["TAG:kv_next"] if not G.vm_log else G.vm_log
            if j == 8 and u == 2:
                per.put([(22, i) for i in q_load(1, (free_bank, free_bank + 1))])
                per.put([(22, f"s_mul_i32 {sr(S_T4)}, {sr(S_SS)}, 96"), (22, f"s_add_u32 {sr(S_QSOFF)}, {sr(S_QSOFF)}, {sr(S_T4)}")])
            if j == 0 and u == 0:
                per.put([(21, i) for i in q_load(2, (TE, TF))])      # before this period's drain (same staging rows are NOT shared: unit 2 rows vs unit 0 rows)
            per.emit(G)
    e(f"s_add_u32 {sr(S_PASS)}, {sr(S_PASS)}, 1")
    e(f"s_cmp_lt_u32 {sr(S_PASS)}, {NPASS}")
    e("s_cbranch_scc1 PASS_LOOP%=")
    # ================= epilogue: the last PV and the three drains ===============================================================
    for m in mfma_pv(2):
        e(m)
    e("s_nop 7"); e("s_nop 7")
    for u in range(3):
        for i in drain(u, S(1, 0) if u != 1 else S(0, 0)):
            G.emit_ins(i)
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_branch END%=")
    # ================= out-of-line rescale paths ================================================================================
    for stub, back, u in G.stubs:
        e(f"{stub}:")
        for i in rescale_math(u, False) + rescale_o(u):
            e(i)
        e(f"s_branch {back}")
    e("END%=:")
    return G.lines


def main():
    simple = "--simple" in sys.argv
    lines = build(simple)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "ufvideo_amd", "csrc", "attn_vit_p2_asm.inc")
    clob = [f"v{i}" for i in range(256)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(40, 100)] + ["vcc", "memory"]
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_attn_p2.py%s -- do not edit.  %d instructions.\n" % (" --simple" if simple else "", len(lines)))
        f.write("#define UFV_VIT_P2_ASM \\\n")
        for l in lines:
            f.write('    "%s\\n\\t" \\\n' % l)
        f.write('    ""\n')
        f.write("#define UFV_VIT_P2_CLOBBERS " + ", ".join('"%s"' % c for c in clob) + "\n")
    print(out, len(lines), "lines")


if __name__ == "__main__":
    main()
