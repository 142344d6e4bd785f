"""CPU: proofs on the TEXT of the generated attention kernel bodies (tools/isa_p2_audit.py, tools/isa_cfg_audit.py).

The ViT bodies are one asm statement that owns a wave's 512 registers and zeroes nothing; the round-5 suite saw ONE unexplained bit mismatch of that kernel.  Round 6 found
the cause (a rescale path writing into registers a load for the next pass was in flight to) with the S = 729 body, where it was frequent, and these checks now hold the class
of bug down on the CPU: no instruction touches a register with a load in flight (both outcomes of every data-dependent branch walked), nothing derived from a never-written
register reaches a live store / address / branch, the K / V ring and the Q staging rows are only read after their DMA was waited for (+ barrier) and only overwritten after a
barrier behind their last read.  Every check is shown to BIND by a mutation of the text that it must report (a check that cannot fail proves nothing)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
CSRC = os.path.join(ROOT, "ufvideo_amd", "csrc")

import isa_cfg_audit  # noqa: E402
import isa_p2_audit  # noqa: E402


def body(name):
    return isa_p2_audit.parse(os.path.join(CSRC, name))


@pytest.mark.parametrize("inc,npass,live_stores", [("attn_vit_p2_asm.inc", 3, 45), ("attn_vit_p2_s729_asm.inc", 4, 60)])
def test_generated_vit_bodies_pass_the_audit(inc, npass, live_stores):
    ins = body(inc)
    findings, st = isa_p2_audit.audit(ins)
    assert findings == [], findings[:5]
    # the walk really covered the kernel: every pass, every out-of-line path in every pass, the dummy drains of pass 0 (15 stores through the dead descriptor) and the real ones
    assert st["passes"] == npass and st["stubs_walked"] == npass * sum(1 for s in ins if s.startswith("s_cbranch_vccnz"))
    assert st["stores_dead"] == 15 and st["stores_live"] == live_stores and st["tainted_reads"] > 100 and st["loads"] == 5 * (npass + 1) + 5 * 0
    inflight, unwritten, cst = isa_cfg_audit.audit(ins)                 # the path-insensitive fixed point over the block's CFG agrees
    assert inflight == {} and cst["reachable"] == cst["instructions"]
    # what the CFG analysis sees read-before-written is exactly the deliberate pass-0 work: accumulators, P fragments, V^T fragments, drain temporaries -- never an address, a
    # scalar or a softmax statistic
    regs = set().union(*unwritten.values())
    assert regs and all(r[0] in "va" for r in regs)
    addr = {f"v{n}" for n in (isa_p2_audit.KADDR, isa_p2_audit.K4A0, isa_p2_audit.K4A1, isa_p2_audit.VADDR, isa_p2_audit.QADDR, isa_p2_audit.OWADDR, isa_p2_audit.ORADDR)}
    assert not (regs & addr)


def test_the_audits_register_map_is_the_generators():
    sys.argv, saved = ["gen_attn_p2.py"], sys.argv
    try:
        import gen_attn_p2 as G
    finally:
        sys.argv = saved
    A = isa_p2_audit
    assert (A.KADDR, A.K4A0, A.K4A1, A.VADDR, A.QADDR, A.OWADDR, A.ORADDR) == (G.KADDR, G.K4A0, G.K4A1, G.VADDR, G.QADDR, G.OWADDR, G.ORADDR)
    assert (A.S_DST, A.S_QST, A.S_ODESC, A.S_ORS, A.S_PASS) == (G.S_DST, G.S_QST, G.S_ODESC, G.ORS, G.S_PASS)
    assert (A.STG, A.UNIT_BYTES) == (G.STG, G.UNIT_BYTES)


def test_in_flight_check_finds_the_round5_mismatch_in_the_old_text(tmp_path):
    """UFV_P2_OPT=lastq regenerates the body as it was before round 6: the last key tile's rescale path writes q'[4][0] of unit 2 while the next pass's rows are in flight to
    that register.  Both analyses must report exactly that instruction, in every pass but none other."""
    out = tmp_path / "old.inc"
    env = dict({k: v for k, v in os.environ.items() if not k.startswith("UFV_")}, UFV_P2_OPT="lastq", UFV_P2_OUT=str(out))
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_attn_p2.py")], check=True, env=env, stdout=subprocess.DEVNULL)
    ins = isa_p2_audit.parse(str(out))
    findings, _ = isa_p2_audit.audit(ins)
    assert len(findings) == 3 and all(f.startswith("IN-FLIGHT") and "v_cndmask_b32 v200, v200" in f for f in findings), findings[:4]
    inflight, _, _ = isa_cfg_audit.audit(ins)
    assert len(inflight) == 1 and list(inflight.values()) == [["v200"]]


def _mutate(ins, pred, repl, nth=0):
    hits = [i for i, s in enumerate(ins) if pred(i, s)]
    i = hits[nth]
    out = list(ins)
    if repl is None:
        del out[i]
    else:
        out[i] = repl(out[i])
    return out


def test_every_check_binds():
    ins = body("attn_vit_p2_asm.inc")
    top = ins.index("PASS_LOOP%=:")
    # TAINT: the first PV of a pass must start its sums from the constant 0 -- let one accumulate onto the never-written accumulators instead
    mut = _mutate(ins, lambda i, s: i > top and s.startswith("v_mfma_f32_32x32x16_bf16 a[0:15]") and s.endswith(", 0"), lambda s: s[:-1] + "a[0:15]")
    f, _ = isa_p2_audit.audit(mut)
    assert any(x.startswith("TAINT") and "LIVE descriptor" in x for x in f), f[:3]
    # RING: a tile's barrier removed -> its K fragments are read from a stage whose DMA was waited for but not published
    mut = _mutate(ins, lambda i, s: i > top and s == "s_barrier", None, nth=2)
    f, _ = isa_p2_audit.audit(mut)
    assert any(x.startswith("RING") for x in f), f[:3]
    # RING: the wait in front of a barrier lets the youngest pieces of the tile stay in flight
    bar = [i for i, s in enumerate(ins) if i > top and s == "s_barrier"][3]
    w = max(i for i in range(bar) if ins[i].startswith("s_waitcnt vmcnt("))
    n = int(re.search(r"vmcnt\((\d+)\)", ins[w]).group(1))
    mut = list(ins); mut[w] = f"s_waitcnt vmcnt({n + 9})"
    f, _ = isa_p2_audit.audit(mut)
    assert any(x.startswith("RING") and "reads ring stage" in x for x in f), f[:3]
    # RING (staging): the prologue's wait for unit 0's Q rows removed in front of the first q_load (inside the loop the wait for the next pass's rows is implied by the
    # tile waits that follow the Q pieces: loads retire in order)
    qa = f"v{isa_p2_audit.QADDR}"
    first_q = next(i for i, s in enumerate(ins) if s.startswith("ds_read_b128") and f", {qa} " in s)
    assert first_q < top
    w = max(i for i in range(first_q) if ins[i].startswith("s_waitcnt vmcnt("))
    mut = list(ins); mut[w] = "s_waitcnt vmcnt(63) lgkmcnt(0)"
    f, _ = isa_p2_audit.audit(mut)
    assert any("q_load reads staging pieces" in x for x in f), f[:3]
    # IN-FLIGHT: unit 2's direct Q loads consumed one wait too early
    assert any(s.startswith("buffer_load_dwordx4 v[184:187]") and " lds" not in s for s in ins[top:])          # issued in a pass's last period ...
    wj = next(i for i in range(top, len(ins)) if ins[i].startswith("s_waitcnt vmcnt("))                      # ... waited for at the top of the next pass
    mut = list(ins); mut[wj] = "s_waitcnt vmcnt(63)"
    f, _ = isa_p2_audit.audit(mut)
    assert any(x.startswith("IN-FLIGHT") for x in f), f[:3]
    # COUNTS
    mut = list(ins); mut[wj] = "s_waitcnt vmcnt(64)"
    f, _ = isa_p2_audit.audit(mut)
    assert any(x.startswith("COUNTS") for x in f)


def test_causal_hd128_body_has_no_load_hazard_and_reads_nothing_unwritten():
    """attn_c128_asm.inc (the decoder's prefill attention) has real control flow -- work items, even / odd tile loops, loader and compute waves: the fixed point over its CFG
    (one state per value of its flag registers, so that 'parity 0, then the odd-tile epilogue' is not a path) finds no instruction touching a register with a load in flight and
    NO register read before it is written: that kernel's results cannot depend on what an earlier kernel left in the register file."""
    ins = body("attn_c128_asm.inc")
    inflight, unwritten, st = isa_cfg_audit.audit(ins)
    assert st["reachable"] == st["instructions"] and "s86" in st["partition_registers"]
    assert inflight == {}, [(i, ins[i]) for i in list(inflight)[:4]]
    assert unwritten == {}, [(i, ins[i], sorted(r)) for i, r in list(unwritten.items())[:4]]
    # binds: drop the wait in front of the first MFMA that consumes the directly loaded Q rows
    q = next(i for i, s in enumerate(ins) if s.startswith("buffer_load_dwordx4 v[80:83]"))
    w = next(i for i in range(q, len(ins)) if ins[i].startswith("s_waitcnt vmcnt(0)"))
    mut = list(ins); mut[w] = "s_nop 0"
    assert isa_cfg_audit.audit(mut)[0]
