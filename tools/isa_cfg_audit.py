#!/usr/bin/env python3
"""Control-flow-graph audit of a generated asm body (csrc/attn_c128_asm.inc, and the ViT bodies as a cross-check of tools/isa_p2_audit.py) -- CPU only.

A forward dataflow analysis over the block's own control-flow graph (labels, s_branch, s_cbranch_*), run to its fixed point:

  IN-FLIGHT   for every VGPR that is the destination of a buffer_load: the smallest number of VMEM operations issued after that load on ANY path to
              this point (loads return in order, so `s_waitcnt vmcnt(N)` has retired exactly the loads with at least N younger operations).  An
              instruction that reads or writes such a register while the load may still be outstanding is reported.  Joins take the minimum age
              (the path on which the wait covers least); ages saturate at 64 (the counter has 6 bits).
  (a load whose destination an OLDER load is still in flight to is not reported: in-order return makes the younger data final; its address operands are checked.)
  UNWRITTEN   registers that are read on some path before any instruction of the block has written them (joins: intersection of the written
              sets).  For a block that owns its registers this is either empty or a documented list (the ViT bodies' dummy first-pass work,
              which tools/isa_p2_audit.py proves harmless by taint tracking).

usage: python tools/isa_cfg_audit.py ufvideo_amd/csrc/attn_c128_asm.inc
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_p2_audit import decode, parse  # noqa: E402

CAP = 64


def build_cfg(ins):
    labels = {s[:-1]: i for i, s in enumerate(ins) if s.endswith(":")}
    succ = []
    for i, s in enumerate(ins):
        op = s.split()[0]
        if op == "s_branch":
            succ.append([labels[s.split()[1]]])
        elif op.startswith("s_cbranch"):
            succ.append([i + 1, labels[s.split()[1]]])
        elif i + 1 < len(ins):
            succ.append([i + 1])
        else:
            succ.append([])
    return labels, succ


def partition_registers(ins):
    """SGPRs the block uses as small state flags: written by `s_mov_b32 sX, <imm>` and tested by `s_cmp_eq_u32 sX, <imm>` + a conditional branch.  The analysis keeps
    one state per (instruction, values of these registers) instead of joining them -- a path that sets the flag to 0 and then takes the flag-is-1 branch does not exist."""
    movs = {m.group(1) for s in ins for m in [re.fullmatch(r"s_mov_b32 (s\d+), (\d+)", s)] if m}
    cmps = {m.group(1) for i, s in enumerate(ins) for m in [re.fullmatch(r"s_cmp_eq_u32 (s\d+), (\d+)", s)] if m and i + 1 < len(ins) and ins[i + 1].startswith("s_cbranch_scc")}
    return sorted(movs & cmps)


def audit(ins):
    labels, succ = build_cfg(ins)
    n = len(ins)
    dec = [None if s.endswith(":") else decode(s) for s in ins]
    part = partition_registers(ins)
    none = tuple([None] * len(part))
    # state per (instruction, partition values) on entry: written frozenset, ages dict reg -> age
    W, A = {(0, none): frozenset()}, {(0, none): {}}
    work = [(0, none)]
    inflight, unwritten = {}, {}
    while work:
        node = work.pop()
        i, pv = node
        w, a = W[node], dict(A[node])
        d = dec[i]
        nxt = list(succ[i])
        if d is not None:
            op, dst, src, mods = d
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", ins[i])
                if m:
                    k = int(m.group(1))
                    a = {r: g for r, g in a.items() if g < k}
            elif op not in ("s_nop", "s_barrier", "s_branch"):
                # a load into registers that an older load is still in flight to is NOT a hazard (loads return in order: the younger data lands last); its address operands are checked
                touched = set(src) if mods.get("load") else (set(dst) | set(src))
                hit = touched & set(a)
                if hit:
                    inflight[i] = sorted(hit)
                uw = [r for r in src if r[0] in "vas" and r[1:].isdigit() and r not in w]
                if uw:
                    unwritten.setdefault(i, set()).update(uw)
                if mods.get("load") or mods.get("dma") or mods.get("store"):
                    a = {r: g + 1 for r, g in a.items() if g + 1 < CAP}
                    if mods.get("load"):
                        for r in dst:
                            a[r] = 0
                w = w | frozenset(dst)
                # partition registers: constant moves set them, any other write forgets them
                for k, r in enumerate(part):
                    if r in dst:
                        m = re.fullmatch(r"s_mov_b32 s\d+, (\d+)", ins[i])
                        pv = pv[:k] + ((int(m.group(1)) if m else None),) + pv[k + 1:]
                if op.startswith("s_cbranch_scc") and i > 0:
                    m = re.fullmatch(r"s_cmp_eq_u32 (s\d+), (\d+)", ins[i - 1])
                    if m and m.group(1) in part and pv[part.index(m.group(1))] is not None:
                        scc = pv[part.index(m.group(1))] == int(m.group(2))
                        taken = scc if op == "s_cbranch_scc1" else not scc
                        nxt = [succ[i][1]] if taken else [succ[i][0]]
        for j in nxt:
            key = (j, pv)
            if key not in W:
                W[key], A[key] = w, a
                work.append(key)
                continue
            nw = W[key] & w
            na = dict(A[key])
            changed = nw != W[key]
            for r, g in a.items():
                if r not in na or g < na[r]:
                    na[r] = g; changed = True
            if changed:
                W[key], A[key] = nw, na
                work.append(key)
    reach = len({k[0] for k in W})
    return inflight, unwritten, dict(instructions=n, reachable=reach, branches=sum(1 for s in ins if s.startswith("s_cbranch")), labels=len(labels), partition_registers=part,
                                     states=len(W))


def main():
    ins = parse(sys.argv[1])
    inflight, unwritten, stats = audit(ins)
    print(stats)
    for i, regs in sorted(inflight.items())[:30]:
        print(f"IN-FLIGHT: [{i}] `{ins[i]}` touches {regs[:6]}")
    regs = sorted(set().union(*unwritten.values())) if unwritten else []
    print(f"UNWRITTEN: {len(unwritten)} instructions read {len(regs)} registers before any write: {regs[:24]}{' ...' if len(regs) > 24 else ''}")
    for i in sorted(unwritten)[:12]:
        print(f"   [{i}] `{ins[i]}`: {sorted(unwritten[i])[:6]}")
    return 1 if inflight else 0


if __name__ == "__main__":
    sys.exit(main())
