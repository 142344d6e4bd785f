#!/usr/bin/env python3
"""Generator of the third-generation ViT attention kernel body (head_dim 72, non-causal; S = 576 = SigLIP so400m at 336 px, and with
`--seq 729` the checkpoint's own 384 px geometry): writes ufvideo_amd/csrc/attn_vit_p2_asm.inc (attn_vit_p2_s729_asm.inc) -- ONE inline-asm
statement that owns the whole 512-register file of a wave.

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

Other sequence lengths (`--seq S`; built: 729): NT = ceil(S / 64) key tiles per pass (a multiple of 3, so that the ring stage of a tile stays an
immediate), NPASS = ceil(S / 192) passes of 96 rows per wave (wave 1 starts at row 96 NPASS).  Rows past S are never special-cased: the buffer
descriptors return zeros for them (q, k, v) and drop their stores (o).  Keys past S sit in the LAST tile only: their scores are overwritten with
-inf in front of that tile's row maximum (attn_vit.inc masks the same scores the same way; exp2 gives them P = 0).  With S = 576 the generated
text is byte for byte what the generator wrote before it had the parameter.
"""
import os
import sys

HD = 72
PK = 144                  # K / V / Q / O row pitch in LDS (9 chunks of 16 B)
KT = 64 * PK              # 9216
VOFF = KT                 # V tile right behind the K tile
KONE_REL = 2 * KT         # 16 bytes {1.0, 0, ...} behind the V tile of EVERY stage (the K fragment of the contraction-padding lanes), + 48 B pad for the
STG = 2 * KT + 64         # transposed reads of d-tile 2 that run past a row: 18496 per stage
NSTAGE = 3                # tile j lives in stage j % 3 (NT tiles per pass, NT % 3 == 0: the same in every pass, so every LDS offset is an immediate)
NPIECE = 9                # DMA pieces per wave per tile (a wave loads the K tile or the V tile of its head)
QPIECES = 9               # staging holds the rows of units 0 and 1 (64 x 144 B = 9 pieces); unit 2's fragments come straight from global memory
UNIT_BYTES = 32 * PK      # 4608
def _seq_arg():
    for i, a in enumerate(sys.argv):
        if a == "--seq":
            return int(sys.argv[i + 1])
        if a.startswith("--seq="):
            return int(a[6:])
    return 576
SEQ = _seq_arg()
NT = (SEQ + 63) // 64     # key tiles per pass (9 at S = 576, 12 at S = 729)
NPASS = (SEQ + 191) // 192    # passes of 96 query rows per wave (3 / 4)
REM = SEQ - 64 * (NT - 1)     # keys of the last tile that exist (64: nothing to mask)
assert NT % 3 == 0 and NT >= 9, "the ring stage of tile j is the immediate j % 3 in every pass: NT must be a multiple of 3 (and the Q staging needs tiles 1..5)"
NEG_INF = 0xFF800000
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
VQOFF = TH                                           # per-lane source offset of the direct Q loads (row l31, 16 h bytes)

def O(u, dt, r=0): return 48 * u + 16 * dt + r      # a: O^T accumulators
def KF(ks, half, i=0): return 144 + 8 * ks + 4 * half + i   # a: K fragments
def VF(dt, c, i=0): return 184 + 16 * dt + 4 * c + i        # a: V^T fragments of d-tiles 0, 1

# fixed SGPRs (clobbered): s40..s99
KVR, QR, ORS = 40, 44, 48
S_ONR, S_SS, S_OS, S_T64, S_SC, S_RING, S_DST, S_DDST, S_DRD, S_QST, S_QSOFF, S_OSOFF, S_PASS, S_TMP = range(52, 66)
S_HM, S_ONE, S_RET, S_T2, S_MAGIC, S_KONE, S_T3, S_T4, S_EXLO = 66, 68, 70, 72, 74, 75, 76, 77, 78
S_Q2OFF = 80        # source offset of unit 2's rows of the next pass (direct Q loads)
S_ODESC = 84        # s[84:87]: O descriptor actually used by the stores


STAMPS = "--stamps" in sys.argv      # lab build: s_memtime at the top of every period -> [wave][128] dwords at %[stp]
def stamp():
    return ("GROUP", ["s_memtime s[88:89]", "s_waitcnt lgkmcnt(0)", "s_store_dword s88, s[90:91], 0", "s_add_u32 s90, s90, 4", "s_addc_u32 s91, s91, 0"])


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
def mfma_pv(u, zero_c=False):
    """O[u][dt] (+)= V^T(dt, c) * P[u](c), c = 0..3 in order per dt (the accumulation order of attn_vit.inc); zero_c: the sums start from 0"""
    out = []
    for dt in range(3):
        for c in range(4):
            a = ar(VF(dt, c), 4) if dt < 2 else vr(VD2(c), 4)
            cc = "0" if (zero_c and c == 0) else ar(O(u, dt), 16)
            out.append(f"v_mfma_f32_32x32x16_bf16 {ar(O(u, dt), 16)}, {a}, {vr(P(u, 4 * c), 4)}, {cc}")
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
    ch = [T0, T1, TE, TF] if "max4" in OPT else [T0, T1]          # independent chains (max is exact and order-independent: same bits as one chain)
    n = len(ch)
    o = [f"v_max_f32 {vr(ch[k])}, {s(k)}, {s(16 + k)}" for k in range(n)]
    for r in range(n, 16):
        o.append(f"v_max3_f32 {vr(ch[r % n])}, {vr(ch[r % n])}, {s(r)}, {s(16 + r)}")
    if n == 4:
        o += [f"v_max_f32 {vr(T0)}, {vr(T0)}, {vr(TE)}", f"v_max_f32 {vr(T1)}, {vr(T1)}, {vr(TF)}"]
    o += [f"v_max_f32 {vr(T0)}, {vr(T0)}, {vr(T1)}", f"v_mov_b32 {vr(T1)}, {vr(T0)}", "s_nop 1",
          f"v_permlane32_swap_b32 {vr(T0)}, {vr(T1)}", f"v_max_f32 {vr(T0)}, {vr(T0)}, {vr(T1)}"]
    return o


def sm_mask_tail(u):
    """last key tile of a sequence that is not a multiple of 64: the scores of keys >= REM become -inf (register r of half `hf` holds key
    32 hf + (r & 3) + 8 (r >> 2) + 4 h, h = lane >> 5: the rows of a 32x32 MFMA result).  T1 is free until sm_max writes it."""
    if REM == 64:
        return []
    o = [f"v_mov_b32 {vr(T1)}, 0x{NEG_INF:08x}"]
    for hf in range(2):
        for r in range(16):
            k0 = 32 * hf + (r & 3) + 8 * (r >> 2)             # the h = 0 lanes' key; the h = 1 lanes hold key k0 + 4
            if k0 >= REM:
                o.append(f"v_mov_b32 {vr(S(u, 16 * hf + r))}, {vr(T1)}")
            elif k0 + 4 >= REM:
                o.append(f"v_cndmask_b32 {vr(S(u, 16 * hf + r))}, {vr(S(u, 16 * hf + r))}, {vr(T1)}, {sr(S_HM, 2)}")
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


def rescale_math(u, first, last_tile=False):
    """m_new = bf16(m_run + (first ? tmax : max(tmax, 0))); delta = m_new - m_run; returns instrs; leaves delta in TA, m_new in MRUN.
    last_tile: the rescale of a pass's LAST key tile does not touch the Q fragments -- no QK^T of this pass is left to read -m from them, and unit 2's fragment
    registers already belong to the NEXT pass's raw rows by then (q2_fetch is issued in the same period, before the softmax can branch here: a v_cndmask that
    executes after such a load has returned leaves -m in place of 8 bytes of q.  Round 6: seen as intermittent wrong rows of unit 2 whenever the load beat the
    stub -- few blocks on the chip, L2-resident q, a late-tile spike in a unit-2 row group; the one-in-eight-suites mismatch of round 5 at S = 576.)"""
    o = []
    if first:
        o += [f"v_cvt_pk_bf16_f32 {vr(TB)}, {vr(T0)}, {vr(T0)}", f"v_lshlrev_b32 {vr(TA)}, 16, {vr(TB)}",   # TA = m_new (m_run was 0) = delta
              f"v_mov_b32 {vr(MRUN(u))}, {vr(TA)}"]
    else:
        o += [f"v_max_f32 {vr(TB)}, 0, {vr(T0)}", f"v_add_f32 {vr(TB)}, {vr(MRUN(u))}, {vr(TB)}",
              f"v_cvt_pk_bf16_f32 {vr(TB)}, {vr(TB)}, {vr(TB)}", f"v_lshlrev_b32 {vr(TB)}, 16, {vr(TB)}",
              f"v_sub_f32 {vr(TA)}, {vr(TB)}, {vr(MRUN(u))}", f"v_mov_b32 {vr(MRUN(u))}, {vr(TB)}"]
    # q'[4][0] of the h = 1 lanes <- bf16(-m_new) (upper half: 0)
    if not last_tile:
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


def k_reads(st):
    """K fragments of the tile in ring stage st (per-lane stage-0 addresses in KADDR / K4A0 / K4A1)"""
    o = []
    for ks in range(4):
        for half in range(2):
            o.append(f"ds_read_b128 {ar(KF(ks, half), 4)}, {vr(KADDR)} offset:{st * STG + ks * 32 + half * 32 * PK}")
    o.append(f"ds_read_b128 {ar(KF(4, 0), 4)}, {vr(K4A0)} offset:{st * STG}")
    o.append(f"ds_read_b128 {ar(KF(4, 1), 4)}, {vr(K4A1)} offset:{st * STG}")
    return o


def v_reads(st):
    """V^T fragments (tr reads) of the tile in ring stage st; d-tile 2 first (it needs the ones-row substitution afterwards)"""
    o = []
    for dt in (2, 0, 1):
        for c in range(4):
            for half in range(2):
                off = st * STG + dt * 64 + c * 16 * PK + half * 8 * PK
                dst = vr(VD2(c, 2 * half), 2) if dt == 2 else ar(VF(dt, c, 2 * half), 2)
                o.append(f"ds_read_b64_tr_b16 {dst}, {vr(VADDR)} offset:{off}")
    return o


def vd2_ones():
    return [f"v_cndmask_b32 {vr(VD2(c, i))}, {vr(VD2(c, i))}, {vr(ONES)}, {sr(S_ONE, 2)}" for c in range(4) for i in range(4)]


def dma_piece(rsrc, voff, soff, m0_expr):
    """one LDS-DMA piece; m0_expr = list of SALU instrs that leave the destination in m0"""
    return m0_expr + ["s_nop 0", f"buffer_load_dwordx4 {vr(voff)}, {sr(rsrc, 4)}, {soff} offen lds"]


def q_load(u, tmp_bank):
    """Q fragments of unit u from the staging area (rows 32u..32u+31) into Q(u, ks), pre-multiplied by scale*log2e and re-rounded to bf16;
    d >= 72 (ks = 4, h = 1 lanes) zeroed.  tmp_bank: two free VGPRs."""
    a, b = tmp_bank
    o = []
    if u < 2:
        o = [f"ds_read_b128 {vr(Q(u, ks), 4)}, {vr(QADDR)} offset:{u * UNIT_BYTES + ks * 32}" for ks in range(5)]
        o.append("s_waitcnt lgkmcnt(0)")
    for ks in range(5):
        for i in range(4):
            r = vr(Q(u, ks, i))
            o += [f"v_lshlrev_b32 {vr(a)}, 16, {r}", f"v_and_b32 {vr(b)}, 0xffff0000, {r}", f"v_mul_f32 {vr(a)}, {sr(S_SC)}, {vr(a)}",
                  f"v_mul_f32 {vr(b)}, {sr(S_SC)}, {vr(b)}", f"v_cvt_pk_bf16_f32 {r}, {vr(a)}, {vr(b)}"]
    o += [f"v_cndmask_b32 {vr(Q(u, 4, i))}, {vr(Q(u, 4, i))}, 0, {sr(S_HM, 2)}" for i in range(4)]
    return o


def q2_fetch():
    """unit 2's raw Q rows of the NEXT pass straight into its fragment registers: lane (l31, h) loads 16 bytes of row l31 at column chunk 2 ks + h"""
    return [("VMEM", f"buffer_load_dwordx4 {vr(Q(2, ks), 4)}, {vr(VQOFF)}, {sr(QR, 4)}, {sr(S_Q2OFF)} offen offset:{ks * 32}") for ks in range(5)] + [("TAG", "q2")]


def drain(u, bank, bank_b=None):
    """O of unit u (of the pass that just ended) -> normalise -> bf16 -> staging rows (part A) -> global (part B).  The accumulators are not
    cleared: the next pass's first PV starts from 0.  bank / bank_b: bases of 16 / 20 free VGPRs for the two parts.  The store offset (rows of
    the ended pass, unit u) is S_OSOFF + 32 u OS.  Returns (part A, part B)."""
    bank_b = bank if bank_b is None else bank_b
    x = [bank + i for i in range(4)]
    y = [bank + 4, bank + 5]
    l, l2, inv = bank + 6, bank + 7, bank + 8
    d = [bank + 9 + i for i in range(6)]
    rows = [bank_b + 4 * i for i in range(4)]         # 4 x 4 regs for row chunks (5 chunks: the 5th reuses the first)
    reg = (0 if u == 2 else u) * UNIT_BYTES             # the staging area has two 32-row regions; unit 2's rows leave through region 0 (free again by then)
    a = [f"v_accvgpr_read_b32 {vr(l)}, {ar(O(u, 2, 4))}", "s_nop 0", f"v_mov_b32 {vr(l2)}, {vr(l)}", "s_nop 1",
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
            a.append(f"v_accvgpr_read_b32 {vr(x[i])}, {ar(O(u, dt, 4 * g4 + i))}")
        for i in range(4):
            a.append(f"v_mul_f32 {vr(x[i])}, {vr(inv)}, {vr(x[i])}")
        a += [f"v_cvt_pk_bf16_f32 {vr(y[0])}, {vr(x[0])}, {vr(x[1])}", f"v_cvt_pk_bf16_f32 {vr(y[1])}, {vr(x[2])}, {vr(x[3])}", "s_nop 0",
              f"ds_write_b64 {vr(OWADDR)}, {vr(y[0], 2)} offset:{reg + dt * 64 + g4 * 16}"]
    # 288 chunks of 16 B: lane c = 64 i + lane; i = 4 covers c = 256..287 (lanes 0..31)
    b = ["s_waitcnt lgkmcnt(0)", f"s_mul_i32 {sr(S_T3)}, {sr(S_OS)}, {32 * u}", f"s_add_u32 {sr(S_T3)}, {sr(S_OSOFF)}, {sr(S_T3)}"]
    for i in range(4):
        b.append(f"ds_read_b128 {vr(rows[i], 4)}, {vr(ORADDR)} offset:{reg + 1024 * i}")
    b.append("s_waitcnt lgkmcnt(0)")
    for i in range(4):
        if "ostore" not in DROP:
            b.append(("VMEM", f"buffer_store_dwordx4 {vr(rows[i], 4)}, {vr(VOFFO(i))}, {sr(S_ODESC, 4)}, {sr(S_T3)} offen{STMOD}"))
    b += ["s_nop 1", f"ds_read_b128 {vr(rows[0], 4)}, {vr(ORADDR)} offset:{reg + 4096}", "s_waitcnt lgkmcnt(0)",
          ("GROUP", [f"s_mov_b64 exec, {sr(S_EXLO, 2)}"] +
                    ([] if "ostore" in DROP else [("VMEM", f"buffer_store_dwordx4 {vr(rows[0], 4)}, {vr(VOFFO(4))}, {sr(S_ODESC, 4)}, {sr(S_T3)} offen{STMOD}")]) +
                    ["s_mov_b64 exec, -1"]), "s_nop 1"]
    return a, b


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
            tags = [i for i, t in enumerate(G.vm_log) if t.startswith("TAG:" + text)]
            later = sum(1 for t in G.vm_log[tags[-1] + 1:] if not t.startswith("TAG:")) if tags else 0
            G.e(f"s_waitcnt vmcnt({min(later, 63)})")
        elif kind == "TAG":
            G.vm_log.append("TAG:" + text)
        elif kind == "GROUP":           # instructions that must stay adjacent (m0 set-up + its DMA)
            for t in text:
                emit_ins(G, t)
        elif kind == "RAW":
            G.e(text)
    else:
        G.e(ins)


Gen.emit_ins = emit_ins


STMOD = "".join(" " + m for m in ("nt", "sc0", "sc1") if ("st_" + m) in os.environ.get("UFV_P2_OPT", "").split(","))      # cache-policy bits of the O stores (experiments)
OPT = set(os.environ.get("UFV_P2_OPT", "").split(","))        # scheduling experiments (results stay correct)
DROP = set(os.environ.get("UFV_P2_DROP", "").split(","))      # timing experiments only (wrong results): dma, exp, max, seam, barrier, vread, kread


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
        "s_mov_b32 s90, %[stp0]", "s_mov_b32 s91, %[stp1]",
        f"v_mbcnt_lo_u32_b32 {vr(LANE)}, -1, 0", f"v_mbcnt_hi_u32_b32 {vr(LANE)}, -1, {vr(LANE)}",
        f"v_mov_b32 {vr(ONES)}, 0x3f803f80",
    ]
    ins += [f"v_and_b32 {vr(TA)}, 31, {vr(LANE)}", f"v_lshrrev_b32 {vr(TB)}, 5, {vr(LANE)}", f"v_mul_lo_u32 {vr(TC)}, {vr(TA)}, {sr(S_SS)}",
            f"v_lshl_add_u32 {vr(VQOFF)}, {vr(TB)}, 4, {vr(TC)}"]                     # direct Q loads: row l31, byte 16 h
    for k in range(9):       # DMA source offsets: chunk c = 64 k + lane -> row c / 9, column chunk c % 9
        ins += [f"v_add_u32 {vr(TA)}, {64 * k}, {vr(LANE)}", f"v_mul_hi_u32 {vr(TB)}, {vr(TA)}, {sr(S_MAGIC)}", f"v_mul_u32_u24 {vr(TC)}, 9, {vr(TB)}",
                f"v_sub_u32 {vr(TC)}, {vr(TA)}, {vr(TC)}", f"v_lshlrev_b32 {vr(TC)}, 4, {vr(TC)}", f"v_mul_lo_u32 {vr(TD)}, {vr(TB)}, {sr(S_SS)}",
                f"v_add_u32 {vr(VOFFR(k))}, {vr(TD)}, {vr(TC)}"]
        if k < 5:
            if "stlin" in OPT:      # timing experiment: every store instruction writes ONE contiguous KB (the output lands in the wrong place)
                ins += [f"v_lshlrev_b32 {vr(TD)}, 4, {vr(LANE)}", f"v_add_u32 {vr(VOFFO(k))}, {1024 * k}, {vr(TD)}"]
            else:
                ins += [f"v_mul_lo_u32 {vr(TD)}, {vr(TB)}, {sr(S_OS)}", f"v_add_u32 {vr(VOFFO(k))}, {vr(TD)}, {vr(TC)}"]
    for i in ins:
        e(i)
    if STAMPS:
        G.emit_ins(stamp())
    # first in: K/V tile 0, the Q rows of units 0 / 1 (staging) and of unit 2 (straight to its registers); then tile 1 and the first half of tile 2
    for k in range(NPIECE):
        for i in dma_piece(KVR, VOFFR(k), "0", [f"s_add_u32 m0, {sr(S_DST)}, {1024 * k}"]):
            e(i)
    # Only what the first QK^T needs goes out first (tile 0 + unit 0's rows: 14 operations, 14 MB chip-wide); the other 23 (unit 1 / 2 rows, tile 1, half of
    # tile 2) are issued behind the first MFMAs.  With all 37 up front every block's first bytes queue behind 41 MB of everybody's later ones: the kernel
    # started ~5 k cycles later (lab, same box: 83 -> 79 us).  UFV_P2_OPT=early23 restores the old order.
    LATE = "early23" not in OPT
    NFIRST = QPIECES if ("p18" in OPT or not LATE) else 5      # p18: unit 1's rows too go out up front (18 operations), so that nothing is waited for twice
    for k in range(NFIRST):
        for i in dma_piece(QR, VOFFR(k), sr(S_QSOFF), [f"s_add_u32 m0, {sr(S_QST)}, {1024 * k}"]):
            e(i)
    e(f"s_add_u32 {sr(S_Q2OFF)}, {sr(S_QSOFF)}, {sr(S_T64)}")

    def rest23():
        for k in range(NFIRST, QPIECES):
            for i in dma_piece(QR, VOFFR(k), sr(S_QSOFF), [f"s_add_u32 m0, {sr(S_QST)}, {1024 * k}"]):
                e(i)
        for i in q2_fetch():
            if i[0] == "VMEM":
                e(i[1])
        for k in range(NPIECE):
            for i in dma_piece(KVR, VOFFR(k), sr(S_T64), [f"s_add_u32 m0, {sr(S_DST)}, {1024 * k + STG}"]):
                e(i)
        e(f"s_lshl_b32 {sr(S_TMP)}, {sr(S_T64)}, 1")
        for k in range(5):
            for i in dma_piece(KVR, VOFFR(k), sr(S_TMP), [f"s_add_u32 m0, {sr(S_DST)}, {1024 * k + 2 * STG}"]):
                e(i)
    if not LATE:
        rest23()
    e(f"s_mul_i32 {sr(S_T4)}, {sr(S_SS)}, 96")
    if not LATE:
        e(f"s_add_u32 {sr(S_QSOFF)}, {sr(S_QSOFF)}, {sr(S_T4)}")            # -> rows of pass 1
        e(f"s_add_u32 {sr(S_Q2OFF)}, {sr(S_Q2OFF)}, {sr(S_T4)}")
    # the rest of the per-lane addresses is computed while the first tiles are in flight.  Nothing is zeroed: every accumulator chain of a
    # pass starts from the constant 0, and what the dummy PV / drain of pass 0 computes from uninitialised registers is never stored
    ins2 = [
        f"v_and_b32 {vr(TA)}, 31, {vr(LANE)}", f"v_lshrrev_b32 {vr(TB)}, 5, {vr(LANE)}",          # TA = l31, TB = h
        f"v_mul_u32_u24 {vr(TC)}, {PK}, {vr(TA)}",                                                 # TC = l31 * 144
        f"v_lshlrev_b32 {vr(TD)}, 4, {vr(TB)}",                                                    # TD = 16 h
        f"v_add3_u32 {vr(KADDR)}, {vr(TC)}, {vr(TD)}, {sr(S_RING)}",
        f"v_mov_b32 {vr(TG)}, {sr(S_RING)}", f"v_add_u32 {vr(TG)}, {KONE_REL}, {vr(TG)}",       # the {1, 0, ...} constant of stage 0 (the stage offset is an immediate)
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
    for i in ins2:
        e(i)
    # the first QK^T needs tile 0 and unit 0's Q rows (staging pieces 0..4): 14 of the 37 operations; the memory system serves a wave's requests
    # in order, so the rest (Q pieces 5..8, unit 2's rows, tile 1, half of tile 2 = 23 operations) stays in flight behind the first MFMAs
    e("s_waitcnt vmcnt(0) lgkmcnt(0)" if LATE else "s_waitcnt vmcnt(23) lgkmcnt(0)")
    e("s_barrier")
    for i in q_load(0, (TA, TB)):
        e(i)
    for i in k_reads(0):
        e(i)
    e("s_waitcnt lgkmcnt(0)")
    for m in mfma_qk(0):
        e(m)
    if LATE:
        rest23()
        e(f"s_add_u32 {sr(S_QSOFF)}, {sr(S_QSOFF)}, {sr(S_T4)}")            # -> rows of pass 1
        e(f"s_add_u32 {sr(S_Q2OFF)}, {sr(S_Q2OFF)}, {sr(S_T4)}")
    e("s_waitcnt vmcnt(19)")                  # unit 1's rows (pieces 5..8); unit 2's fragments are loaded in period (0, 0), as in every pass
    for i in q_load(1, (TA, TB)):
        e(i)
    # O descriptor of pass 0's (non-existent) predecessor: num_records = 0 -> every store is dropped
    e(f"s_mov_b32 {sr(S_ODESC)}, {sr(ORS)}"); e(f"s_mov_b32 {sr(S_ODESC + 1)}, {sr(ORS + 1)}"); e(f"s_mov_b32 {sr(S_ODESC + 2)}, 0"); e(f"s_mov_b32 {sr(S_ODESC + 3)}, {sr(ORS + 3)}")
    e("s_nop 7"); e("s_nop 7")
    if STAMPS:
        G.emit_ins(stamp())

    # ================= pass loop =================================================================================================
    def emit_pass(G, carry):
        """one pass = 3 NT periods; `carry`: fillers handed over by the previous pass's last period (the second half of tile 1's DMA pieces)"""
        e = G.e
        e("PASS_LOOP%=:")
        # what was issued before the top of a pass, oldest first: ... tile 1 | unit 2's Q rows | first half of tile 2
        G.vm_log = ["TAG:kv1"] + ["q2"] * 5 + ["TAG:q2"] + ["kv2a"] * 5
        for j in range(NT):
            for u in range(3):
                prev_u = (u + 2) % 3
                next_u = (u + 1) % 3
                pv_first_tile = ((j == 0 and u > 0) or (j == 1 and u == 0)) and "nozc" not in OPT     # the PV of a pass's first key tile starts its sums from 0
                per = Period(mfma_pv(prev_u, zero_c=pv_first_tile) + mfma_qk(next_u))
                if STAMPS:
                    per.put([(0, stamp())])
                per.put(carry)
                carry = []
                free_bank = S((u + 2) % 3, 0)           # 32 VGPRs nobody owns during period (., u)
                # ---- softmax of item (j, u)
                mx = (sm_mask_tail(u) if j == NT - 1 else []) + sm_max(u)
                if j == 0:
                    body = mx + rescale_math(u, True)
                else:
                    stub, back = G.label("RS"), G.label("BK")
                    body = mx + [f"v_cmp_lt_f32 vcc, 0x{THR:08x}, {vr(T0)}", "s_nop 1", f"s_cbranch_vccnz {stub}", f"{back}:"]
                    G.stubs.append((stub, back, u, j == NT - 1 and "lastq" not in OPT))      # UFV_P2_OPT=lastq: the pre-round-6 text (negative control of tools/isa_p2_audit.py)
                ex = sm_exp_cvt(u)
                if "exp" in DROP:
                    ex = []
                if "max" in DROP:
                    body = []
                    if j != 0:
                        G.stubs.pop()
                if simple:
                    per.put([(22, i) for i in body + ex])
                else:
                    mg = int(os.environ.get("UFV_P2_MAXGAP", "5"))
                    per.put(spread(body, 2, mg if j else 9))
                    per.put(spread(ex, (mg + 1) if j else 10, 22))
                # ---- LDS / DMA / address work
                if u == 0:
                    # V^T fragments of tile j: after the 12 PV MFMAs of the previous tile's last unit have been issued
                    vrd = [] if "vread" in DROP else v_reads(j % NSTAGE)
                    per.put([(22, i) for i in vrd] if simple else spread(vrd, 13, 20))
                    per.put([(22, "s_waitcnt lgkmcnt(0)")])
                if u == 1:
                    per.put([(0, i) for i in vd2_ones()] if simple else spread(vd2_ones(), 1, 6))
                if u == 2:
                    # barrier B(j+1): my pieces of tile j+1 are in, everybody is done with tile j's LDS image
                    per.put([(0, ("VMWAIT", f"kv{(j + 1) % NT}"))] + ([] if "barrier" in DROP else [(0, "s_barrier")]))
                    kr = [] if "kread" in DROP else k_reads((j + 1) % NSTAGE)
                    per.put([(1, i) for i in kr] if simple else spread(kr, 1, 5))
                    per.put([(11, "s_waitcnt lgkmcnt(0)")])
                    # DMA of tile j + 3 into the stage everybody just left (stage j % 3): 5 pieces in this period, 4 in the next
                    j3 = (j + 3) % NT
                    dm = [f"s_mul_i32 {sr(S_TMP)}, {sr(S_T64)}, {j3}"]
                    dm2 = []
                    for k in range(NPIECE):
                        pc = [f"s_add_u32 m0, {sr(S_DST)}, {(j % NSTAGE) * STG + 1024 * k}", "s_nop 0",
                              ("VMEM", f"buffer_load_dwordx4 {vr(VOFFR(k))}, {sr(KVR, 4)}, {sr(S_TMP)} offen lds")]
                        if "dma" in DROP:
                            pc = pc[:2]
                        (dm if k < 5 else dm2).append(("GROUP", pc))
                    dm2 += [("TAG", f"kv{j3}")]
                    if simple:
                        per.put([(12, i) for i in dm]); carry += [(1, i) for i in dm2]
                    else:
                        per.put(spread(dm, 6, 21)); carry += spread(dm2, 1, 12)
                # ---- pass seams
                if j == 0 and "seam" not in DROP:
                    dra, drb = drain(u, free_bank, S(u, 16))        # part B runs in the next period, whose free bank is S[u]; its part A takes S[u][0..15]
                    if "nozc" in OPT:
                        dra = dra + [f"v_accvgpr_write_b32 {ar(O(u, 0, r))}, 0" for r in range(48)]
                    per.put([(22, i) for i in dra] if simple or "dr22" in OPT else spread(dra, 1, 22))
                    carry += [(2, i) for i in drb] if simple else spread(drb, 2, 10)
                if j == 0 and u == 2:
                    # the O descriptor becomes real after pass 0's drains; the store rows advance by 96 per pass (applied after the last drain's stores: next period)
                    carry += [(11, ("GROUP", [f"s_cmp_eq_u32 {sr(S_PASS)}, 0", f"s_cselect_b32 {sr(S_T3)}, 0, 1",      # (SCC: nothing may come between)
                                              f"s_mul_i32 {sr(S_T4)}, {sr(S_OS)}, 96", f"s_mul_i32 {sr(S_T4)}, {sr(S_T4)}, {sr(S_T3)}",
                                              f"s_add_u32 {sr(S_OSOFF)}, {sr(S_OSOFF)}, {sr(S_T4)}", f"s_mov_b32 {sr(S_ODESC + 2)}, {sr(ORS + 2)}"]))]
                if 1 <= j <= 5 and u == 1:
                    # Q rows of units 0 / 1 of the next pass -> staging: 2 pieces per tile (tiles 1..5 -> 9 pieces); the O rows that left through
                    # the staging regions are out by then (region 0 is read last, in period (1, 0))
                    qd = []
                    for k in [k for k in (2 * (j - 1), 2 * (j - 1) + 1) if k < QPIECES]:
                        if "qdma" in DROP:
                            continue
                        qd.append(("GROUP", [f"s_add_u32 m0, {sr(S_QST)}, {1024 * k}", "s_nop 0",
                                             ("VMEM", f"buffer_load_dwordx4 {vr(VOFFR(k))}, {sr(QR, 4)}, {sr(S_QSOFF)} offen lds")]))
                    if j == 5:
                        qd.append(("TAG", "q_done"))
                    per.put([(10, i) for i in qd] if simple else spread(qd, 13, 18))
                if "seam" not in DROP:
                    if j == NT - 1 and u == 1:
                        per.put([(0, ("VMWAIT", "q_done"))])
                        ql = q_load(0, (free_bank, free_bank + 1))
                        per.put([(22, i) for i in ql] if simple else spread(ql, 1, 22))
                    if j == NT - 1 and u == 2:
                        ql = q_load(1, (free_bank, free_bank + 1))
                        per.put([(22, i) for i in ql] if simple else spread(ql, 1, 22))
                        # unit 2's last QK^T of the pass was issued in the period before: its registers take the next pass's raw rows
                        per.put([(1, i) for i in q2_fetch()] if simple else spread(q2_fetch(), 1, 4))
                    if j == 0 and u == 0:
                        per.put([(0, ("VMWAIT", "q2"))])
                        ql = q_load(2, (TC, TD))
                        per.put([(21, i) for i in ql] if simple else spread(ql, 1, 21))
                if j == NT - 1 and u == 2:
                    per.put([(22, f"s_mul_i32 {sr(S_T4)}, {sr(S_SS)}, 96"), (22, f"s_add_u32 {sr(S_QSOFF)}, {sr(S_QSOFF)}, {sr(S_T4)}"),
                             (22, f"s_add_u32 {sr(S_Q2OFF)}, {sr(S_Q2OFF)}, {sr(S_T4)}")])
                per.emit(G)
        return carry

    scratch = Gen()
    loop_carry = emit_pass(scratch, [])          # first run: only to learn what the last period hands over to period (0, 0)
    last = emit_pass(G, loop_carry)
    assert [str(x) for x in last] == [str(x) for x in loop_carry]
    e(f"s_add_u32 {sr(S_PASS)}, {sr(S_PASS)}, 1")
    e(f"s_cmp_lt_u32 {sr(S_PASS)}, {NPASS}")
    e("s_cbranch_scc1 PASS_LOOP%=")
    if STAMPS:
        G.emit_ins(stamp())
    # ================= epilogue: the last PV and the three drains ===============================================================
    for m in mfma_pv(2):
        e(m)
    e("s_nop 7"); e("s_nop 7")
    parts = [drain(u, S(u, 0), S(u, 16)) for u in range(3)]       # every bank is free now: unit u's drain works in S[u]
    # unit 2's rows leave through staging region 0, which unit 0's rows must have left first
    for seq in (parts[0][0], parts[1][0], parts[0][1], parts[2][0], parts[1][1], parts[2][1]):
        for i in seq:
            G.emit_ins(i)
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_branch END%=")
    # ================= out-of-line rescale paths ================================================================================
    for stub, back, u, last_tile in G.stubs:
        e(f"{stub}:")
        for i in rescale_math(u, False, last_tile) + rescale_o(u):
            e(i)
        e(f"s_branch {back}")
    e("END%=:")
    if STAMPS:
        G.emit_ins(stamp())
        e("s_dcache_wb")
    return G.lines


def main():
    simple = "--simple" in sys.argv
    lines = build(simple)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sfx = "" if SEQ == 576 else f"_s{SEQ}"              # file and macro names of the other sequence lengths carry the length
    out = os.environ.get("UFV_P2_OUT") or os.path.join(root, "ufvideo_amd", "csrc", f"attn_vit_p2{sfx}_asm.inc")
    clob = [f"v{i}" for i in range(256)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(40, 100)] + ["vcc", "memory"]
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_attn_p2.py%s%s -- do not edit.  %d instructions.\n" % ("" if SEQ == 576 else f" --seq {SEQ}", " --simple" if simple else "", len(lines)))
        if STAMPS:
            f.write("#define UFV_VIT_P2_STAMPS 1\n")
        f.write(f"#define UFV_VIT_P2{sfx.upper()}_ASM \\\n")
        for l in lines:
            f.write('    "%s\\n\\t" \\\n' % l)
        f.write('    ""\n')
        if SEQ == 576:
            f.write("#define UFV_VIT_P2_CLOBBERS " + ", ".join('"%s"' % c for c in clob) + "\n")
    print(out, len(lines), "lines")


if __name__ == "__main__":
    main()
