#!/usr/bin/env python3
"""Audit of the generated ViT attention kernel bodies (csrc/attn_vit_p2_asm.inc, attn_vit_p2_s729_asm.inc) -- CPU only, on the generated text.

The bodies are ONE asm statement each that owns a wave's whole register file; hipcc checks nothing inside.  This tool executes the text
symbolically -- prologue, the pass loop unrolled NPASS times, epilogue; at every `s_cbranch_vccnz RS_n` the out-of-line rescale path is run
on a copy of the state as well (both outcomes are possible at run time) -- and proves four things:

  1. IN-FLIGHT   no instruction reads or writes a VGPR that is the destination of a buffer_load still in flight.  The load queue is the
                 hardware's: loads return in order, `s_waitcnt vmcnt(N)` leaves the N youngest outstanding.  (Round 6: the rescale path of a
                 pass's last key tile wrote -m into unit 2's Q fragment registers while the NEXT pass's rows were in flight to them; when the
                 load won, 8 bytes of q were replaced -- the intermittent mismatch of round 5.  This check finds that in the old text.)
  2. TAINT       the kernel deliberately zeroes nothing (pass 0 runs a dummy PV and three dummy drains on never-written registers).  Every
                 value derived from a never-written register is tracked; none may reach a store whose descriptor is live, an address, a
                 branch condition, a scalar register or the staging rows a later q_load reads.  (A stale-register dependence -- results that
                 depend on what the previous kernel left in the register file -- would show up here.)
  3. RING        K / V tiles: a ds_read of ring stage s is preceded by a vmcnt wait that retires this wave's DMA pieces into s and then an
                 s_barrier (the partner wave's pieces: it runs the same text); a DMA into stage s is preceded by an s_barrier that follows
                 this wave's last read of s.  The Q staging rows likewise (wait only: they are per wave).
  4. COUNTS      every vmcnt immediate fits the 6-bit field and no wait asks for more than is outstanding (a wait that cannot bind is a
                 generator bug even when it is harmless).

usage: python tools/isa_p2_audit.py ufvideo_amd/csrc/attn_vit_p2_asm.inc   -> findings, exit status 1 if any
"""
import re
import sys

STG, UNIT_BYTES, PIECE = 2 * 64 * 144 + 64, 32 * 144, 1024
# register names of tools/gen_attn_p2.py's map (checked against the generator in tests/test_isa_p2_audit.py)
KADDR, K4A0, K4A1, VADDR, QADDR, OWADDR, ORADDR = 229, 230, 231, 232, 244, 245, 246
S_DST, S_QST, S_ODESC, S_ORS, S_PASS = 58, 61, 84, 48, 64


def parse(path_or_text):
    text = open(path_or_text).read() if "\n" not in path_or_text else path_or_text
    ins = []
    for m in re.finditer(r'^\s*"(.*?)\\n\\t"', text, re.M):
        s = m.group(1).strip()
        if s:
            ins.append(s)
    return ins


def expand(tok):
    """register token -> list of names ('v12', 'a3', 's40', 'vcc', 'exec', 'm0', 'scc'); literals -> []"""
    tok = tok.strip().lstrip("-")
    if tok.startswith("|") and tok.endswith("|"):
        tok = tok[1:-1]
    m = re.fullmatch(r"([vas])\[(\d+):(\d+)\]", tok)
    if m:
        return [f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    if re.fullmatch(r"[vas]\d+", tok):
        return [tok]
    if tok in ("vcc", "exec"):
        return [tok]
    if tok == "m0":
        return ["m0"]
    return []


def split_ops(rest):
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        if ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


def decode(s):
    """-> (opcode, dst regs, src regs, info dict)"""
    op, _, rest = s.partition(" ")
    mods = {}
    for k in ("offset",):
        m = re.search(r"\boffset:(\d+)", rest)
        if m:
            mods["offset"] = int(m.group(1))
    rest_clean = re.sub(r"\b(offen|lds|nt|sc0|sc1|offset:\d+)\b", "", rest).strip().rstrip(",")
    # buffer ops keep their operands separated by commas except the trailing modifiers
    rest_clean = re.sub(r"\s+", " ", rest_clean)
    ops = split_ops(rest_clean.replace(" offen", ""))
    ops = [o for o in ops if o]
    R = lambda i: expand(ops[i]) if i < len(ops) else []      # noqa: E731
    dst, src = [], []
    if op in ("s_nop", "s_barrier", "s_waitcnt", "s_dcache_wb"):
        pass
    elif op in ("s_branch",):
        pass
    elif op in ("s_cbranch_vccnz",):
        src = ["vcc"]
    elif op in ("s_cbranch_scc1", "s_cbranch_scc0"):
        src = ["scc"]
    elif op.startswith("s_cmp") or op.startswith("s_bitcmp"):
        dst = ["scc"]; src = R(0) + R(1)
    elif op == "s_cselect_b32":
        dst = R(0); src = R(1) + R(2) + ["scc"]
    elif op == "s_addc_u32":
        dst = R(0) + ["scc"]; src = R(1) + R(2) + ["scc"]
    elif op in ("s_add_u32", "s_sub_u32", "s_lshl_b32", "s_lshr_b32", "s_and_b32", "s_sub_i32"):
        dst = R(0) + ["scc"]; src = R(1) + R(2)
    elif op in ("s_mul_i32", "s_mul_hi_u32"):
        dst = R(0); src = R(1) + R(2)
    elif op in ("s_mov_b32", "s_mov_b64"):
        dst = R(0); src = R(1)
        if ops[0] == "exec":
            dst = ["exec"]
    elif op in ("s_memtime", "s_store_dword"):
        dst = R(0) if op == "s_memtime" else []; src = [] if op == "s_memtime" else R(0) + R(1)
    elif op.startswith("v_mfma"):
        dst = R(0); src = R(1) + R(2) + R(3)
    elif op == "v_permlane32_swap_b32":
        dst = R(0) + R(1); src = R(0) + R(1)
    elif op.startswith("v_cmp"):
        dst = R(0); src = R(1) + R(2)
    elif op == "v_cndmask_b32":
        dst = R(0); src = R(1) + R(2) + (R(3) if len(ops) > 3 else ["vcc"])
    elif op == "v_div_scale_f32":
        dst = R(0) + R(1); src = R(2) + R(3) + R(4)
    elif op == "v_div_fmas_f32":
        dst = R(0); src = R(1) + R(2) + R(3) + ["vcc"]
    elif op in ("v_fmac_f32",):
        dst = R(0); src = R(0) + R(1) + R(2)
    elif op == "v_readfirstlane_b32":
        dst = R(0); src = R(1)
    elif op in ("v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32"):
        dst = R(0); src = R(1) + R(2)
    elif op.startswith("ds_read"):
        dst = R(0); src = R(1); mods["lds_read"] = True
    elif op.startswith("ds_write"):
        src = R(0) + R(1); mods["lds_write"] = True; mods["addr"] = R(0); mods["data"] = R(1)
    elif op.startswith("buffer_load"):
        if " lds" in s:
            src = R(0) + R(1) + R(2) + ["m0"]; mods["dma"] = True; mods["addr"] = R(0) + R(1) + R(2)
        else:
            dst = R(0); src = R(1) + R(2) + R(3); mods["load"] = True; mods["addr"] = R(1) + R(2) + R(3)
    elif op.startswith("buffer_store"):
        src = R(0) + R(1) + R(2) + R(3); mods["store"] = True; mods["data"] = R(0); mods["addr"] = R(1) + R(2) + R(3); mods["rsrc"] = R(2)
    elif op.startswith("v_") or op.startswith("s_"):
        dst = R(0); src = [r for i in range(1, len(ops)) for r in R(i)]
    else:
        raise ValueError("unknown instruction: " + s)
    return op, dst, src, mods


class State:
    def __init__(self):
        self.written = set()          # registers written inside the block (or bound inputs)
        self.taint = set()            # registers holding a value derived from a never-written register
        self.queue = []               # outstanding VMEM ops, oldest first: ("load", regs) | ("dma", "ring", stage) | ("dma", "stage", slot) | ("store",)
        self.m0 = None                # ("ring", stage, byte) | ("stage", slot) | None
        self.odesc_live = False
        self.slot_taint = [False] * 9                   # staging rows, per 1 KiB piece: written with tainted data since the last DMA refill
        self.ring_ready = {}                             # stage -> "dma_pending" | "waited" | "ready"   (this wave's view)
        self.ring_read_since_barrier = set()             # stages read since the last barrier
        self.stage_pending = [False] * 9                 # staging pieces whose DMA has not been waited for
        self.barriers = 0

    def copy(self):
        c = State()
        c.written, c.taint, c.queue = set(self.written), set(self.taint), list(self.queue)
        c.m0, c.odesc_live, c.slot_taint = self.m0, self.odesc_live, list(self.slot_taint)
        c.ring_ready, c.ring_read_since_barrier = dict(self.ring_ready), set(self.ring_read_since_barrier)
        c.stage_pending, c.barriers = list(self.stage_pending), self.barriers
        return c


def audit(ins, verbose=False):
    labels = {s[:-1]: i for i, s in enumerate(ins) if s.endswith(":")}
    top = labels["PASS_LOOP%="]
    back = next(i for i, s in enumerate(ins) if s.startswith("s_cbranch_scc1 PASS_LOOP"))
    npass = int(re.search(r"s_cmp_lt_u32 s\d+, (\d+)", ins[back - 1]).group(1))
    end = next(i for i, s in enumerate(ins) if s.startswith("s_branch END"))
    findings = []
    stats = dict(instructions=0, stubs_walked=0, loads=0, dmas=0, stores_live=0, stores_dead=0, tainted_reads=0, passes=npass, max_outstanding=0)

    def flag(kind, where, s, why):
        findings.append(f"{kind}: [{where}] `{s}`: {why}")

    def flying(st):
        f = set()
        for e in st.queue:
            if e[0] == "load":
                f |= e[1]
        return f

    def step(st, s, where, in_stub=False):
        if s.endswith(":"):
            return
        op, dst, src, mods = decode(s)
        stats["instructions"] += 1
        # ---- 4. waits
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", s)
            if m:
                n = int(m.group(1))
                if n > 63:
                    flag("COUNTS", where, s, "vmcnt immediate does not fit 6 bits")
                if n < len(st.queue):
                    retired, st.queue = st.queue[:len(st.queue) - n], st.queue[len(st.queue) - n:]
                    for e in retired:
                        if e[0] == "dma" and e[1] == "ring":
                            if st.ring_ready.get(e[2]) == "dma_pending" and not any(q[0] == "dma" and q[1] == "ring" and q[2] == e[2] for q in st.queue):
                                st.ring_ready[e[2]] = "waited"
                        if e[0] == "dma" and e[1] == "stage":
                            st.stage_pending[e[2]] = any(q[0] == "dma" and q[1] == "stage" and q[2] == e[2] for q in st.queue)
            return
        if op == "s_barrier":
            st.barriers += 1
            for k, v in list(st.ring_ready.items()):
                if v == "waited":
                    st.ring_ready[k] = "ready"
            st.ring_read_since_barrier = set()
            return
        if op in ("s_nop", "s_branch", "s_dcache_wb"):
            return
        # ---- 1. in-flight
        fl = flying(st)
        hit = (set(dst) | set(src)) & fl
        if hit:
            flag("IN-FLIGHT", where, s, f"touches {sorted(hit)[:6]} while a buffer_load into them is outstanding ({len(st.queue)} VMEM operations in flight)")
        # ---- 2. taint
        unwritten = [r for r in src if r not in st.written and r[0] in "vas"]
        tainted_in = [r for r in src if r in st.taint] + unwritten
        if tainted_in:
            stats["tainted_reads"] += 1
        if op.startswith("s_cbranch") and tainted_in:
            flag("TAINT", where, s, f"branch condition derived from never-written registers {tainted_in[:4]}")
        if mods.get("addr") and any(r in st.taint or (r not in st.written and r[0] in "vas") for r in mods["addr"]):
            flag("TAINT", where, s, "address / descriptor operand derived from a never-written register")
        if dst and dst[0][0] == "s" and tainted_in and not op.startswith("s_cbranch"):
            flag("TAINT", where, s, f"scalar register computed from never-written registers {tainted_in[:4]}")
        if op in ("v_cmp_lt_f32",) and tainted_in and "vcc" in s:
            pass                                  # (flagged when the branch consumes it)
        # ---- memory effects
        if mods.get("store"):
            live = st.odesc_live
            stats["stores_live" if live else "stores_dead"] += 1
            data_t = [r for r in mods["data"] if r in st.taint or r not in st.written]
            if live and data_t:
                flag("TAINT", where, s, f"stores {data_t[:4]} (derived from never-written registers) through a LIVE descriptor")
            st.queue.append(("store",))
        elif mods.get("load"):
            stats["loads"] += 1
            st.queue.append(("load", set(dst)))
        elif mods.get("dma"):
            stats["dmas"] += 1
            if st.m0 is None:
                flag("RING", where, s, "LDS-DMA with an untracked m0")
            elif st.m0[0] == "ring":
                stage = st.m0[1]
                if stage in st.ring_read_since_barrier:
                    flag("RING", where, s, f"DMA into ring stage {stage} with no s_barrier since this wave last read that stage")
                st.ring_ready[stage] = "dma_pending"
                st.queue.append(("dma", "ring", stage))
            else:
                slot = st.m0[1]
                st.queue.append(("dma", "stage", slot))
                st.stage_pending[slot] = True
                st.slot_taint[slot] = False
        elif mods.get("lds_read"):
            a = src[0]
            off = mods.get("offset", 0)
            if a in (f"v{KADDR}", f"v{K4A0}", f"v{K4A1}", f"v{VADDR}"):
                stage = off // STG
                if st.ring_ready.get(stage) != "ready":
                    flag("RING", where, s, f"reads ring stage {stage} in state {st.ring_ready.get(stage)} (needs: vmcnt wait for its DMA pieces, then s_barrier)")
                st.ring_read_since_barrier.add(stage)
            elif a in (f"v{QADDR}", f"v{ORADDR}"):
                lo = off if a == f"v{ORADDR}" else (off // UNIT_BYTES) * UNIT_BYTES
                hi = lo + (PIECE if a == f"v{ORADDR}" else UNIT_BYTES)
                slots = range(lo // PIECE, min(9, (hi - 1) // PIECE + 1))
                if a == f"v{QADDR}":
                    if any(st.stage_pending[k] for k in slots):
                        flag("RING", where, s, f"q_load reads staging pieces {list(slots)} whose DMA has not been waited for")
                    if any(st.slot_taint[k] for k in slots):
                        flag("TAINT", where, s, "q_load reads staging rows that hold values derived from never-written registers")
                elif any(st.slot_taint[k] for k in slots):
                    tainted_in = tainted_in + ["lds"]
            else:
                flag("RING", where, s, f"ds_read through an address register the audit does not know ({a})")
        elif mods.get("lds_write"):
            if mods["addr"][0] != f"v{OWADDR}":
                flag("RING", where, s, "ds_write through an address register the audit does not know")
            off = mods.get("offset", 0)
            lo = (off // UNIT_BYTES) * UNIT_BYTES
            slots = range(lo // PIECE, min(9, (lo + UNIT_BYTES - 1) // PIECE + 1))
            if any(st.stage_pending[k] for k in slots):
                flag("RING", where, s, f"O rows written into staging pieces {list(slots)} with a Q DMA into them outstanding")
            if any(r in st.taint or r not in st.written for r in mods["data"]):
                for k in slots:
                    st.slot_taint[k] = True
        # ---- scalar bookkeeping the audit follows
        if op == "s_add_u32" and dst and dst[0] == "m0":
            m = re.fullmatch(r"s_add_u32 m0, s(\d+), (\d+)", s)
            if m and int(m.group(1)) == S_DST:
                st.m0 = ("ring", int(m.group(2)) // STG, int(m.group(2)))
            elif m and int(m.group(1)) == S_QST:
                st.m0 = ("stage", int(m.group(2)) // PIECE)
            else:
                st.m0 = None
        if op == "s_mov_b32" and dst == [f"s{S_ODESC + 2}"]:
            st.odesc_live = s.endswith(f"s{S_ORS + 2}")
        # ---- results
        for r in dst:
            st.written.add(r)
            if tainted_in and not (op.startswith("v_mfma") and False):
                st.taint.add(r)
            else:
                st.taint.discard(r)
        stats["max_outstanding"] = max(stats["max_outstanding"], len(st.queue))

    def run(st, lo, hi, where):
        i = lo
        while i < hi:
            s = ins[i]
            if s.startswith("s_cbranch_vccnz"):
                step(st, s, f"{where}:{i}")
                stub = labels[s.split()[1] + ""] if s.split()[1] in labels else labels[s.split()[1]]
                j = stub + 1
                sub = st.copy()
                while not ins[j].startswith("s_branch"):
                    step(sub, ins[j], f"{where}:{i} -> stub {s.split()[1]}:{j}", in_stub=True)
                    j += 1
                stats["stubs_walked"] += 1
                if sub.queue != st.queue:
                    flag("IN-FLIGHT", f"{where}:{i}", s, "the out-of-line path changes the VMEM queue")
                # both outcomes possible: whatever the stub may have tainted stays tainted; what it wrote counts as written only if the fall-through wrote it too
                st.taint |= sub.taint
                i += 1
                continue
            step(st, s, f"{where}:{i}")
            i += 1

    st = State()
    # bound inputs: the asm's "s" operands are substituted textually as SGPRs the compiler chose; the audit sees them as %[name]
    run(st, 0, top, "prologue")
    for p in range(npass):
        run(st, top + 1, back + 1, f"pass{p}")
    run(st, back + 1, end, "epilogue")
    if st.queue and ins[end - 1] != "s_waitcnt vmcnt(0) lgkmcnt(0)":
        flag("COUNTS", "end", ins[end - 1], "the block ends with VMEM operations outstanding")
    stats["stores_total"] = stats["stores_live"] + stats["stores_dead"]
    return findings, stats


def main():
    ins = parse(sys.argv[1])
    findings, stats = audit(ins)
    print(stats)
    for f in findings[:40]:
        print(f)
    print(f"{len(findings)} findings")
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
