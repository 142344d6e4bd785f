"""CPU: the store counts the ping-pong GEMM's relaxed item-seam waits rest on (csrc/gemm256_kernel.h: NST / NSTW / NSTS), checked on the gfx950 ISA.

The K loop's first waits of a tile are `s_waitcnt vmcnt(L + NSTx)`: they leave the PREVIOUS tile's epilogue stores in flight, which is right only if that
epilogue issued exactly NSTx vector-memory stores behind the next tile's prologue DMA.  A toolchain that merged, split or predicated those stores would make
the first K-tile read LDS that has not landed -- wrong sums that the short-K GPU tests can miss (round 4 saw exactly that with another invariant).  Here
the translation units that instantiate the kernel are compiled to assembly and every path through every epilogue form (bracketed by UFV_EPI_MARK
comments) is walked and its vector-memory instructions counted (tools/isa_epilogue_stores.py).  A toolchain bump that changes a count fails HERE, on the CPU."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = "/opt/rocm/bin/hipcc"
SOURCES = ("gemm256.hip", "gemm256_b.hip", "gemm256_q.hip", "gemm256_r.hip", "gemm256_m.hip", "gemm256_m2.hip")       # 256x256 (+ SwiGLU), the named bf16 / e4m3 shapes, the fused QKV + RoPE form, the MX forms


@pytest.fixture(scope="module")
def listings(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    d = tmp_path_factory.mktemp("isa")
    procs = []
    for src in SOURCES:                                           # the compiles run side by side (about a minute of wall time)
        out = d / (src + ".s")
        procs.append((src, out, subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S",
                                                  "--cuda-device-only", "-o", str(out), os.path.join(ROOT, "ufvideo_amd", "csrc", src)],
                                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    res = {}
    for src, out, p in procs:
        _, err = p.communicate(timeout=900)
        assert p.returncode == 0, (src, err[-2000:])
        res[src] = out.read_text()
    return res


def test_every_epilogue_path_issues_the_store_count_the_relaxed_waits_assume(listings):
    import isa_epilogue_stores as I
    seen = {}
    for src, text in listings.items():
        bad, n, forms = I.check(text)
        assert not bad, (src, bad[:6])
        seen[src] = (n, forms)
    # the kernels and forms this guards exist (a renamed marker or kernel must not turn the check into a no-op)
    assert seen["gemm256.hip"][0] >= 60 and {"wide", "swiglu_wide", "resid", "plain"} <= seen["gemm256.hip"][1], seen
    assert seen["gemm256_b.hip"][0] >= 100 and {"wide", "resid", "plain"} <= seen["gemm256_b.hip"][1], seen
    assert seen["gemm256_q.hip"][0] >= 100 and {"wide", "plain"} <= seen["gemm256_q.hip"][1], seen
    assert seen["gemm256_r.hip"] == (2, {"rope"}), seen          # bf16 and e4m3 operands
    assert seen["gemm256_m.hip"][0] >= 60 and "plain" in seen["gemm256_m.hip"][1], seen          # block-scaled A operand: the per-row kernels' epilogues
    assert seen["gemm256_m2.hip"][0] >= 8 and {"wide_mx", "swiglu_mx"} <= seen["gemm256_m2.hip"][1], seen


def test_the_wide_epilogues_are_straight_line_code(listings):
    """round 5: a half past N is stored to a dropped offset instead of being skipped, so a wide / SwiGLU-wide epilogue has no branch between its stores --
    min == max on every path is what the path walk asserts; this pins the stronger property that made it so (BEGIN and END in one basic block; the erf
    form of GELU branches inside the activation and is left to the path walk)"""
    import isa_epilogue_stores as I
    n = other = 0
    for name, t, body in I.kernels(listings["gemm256_b.hip"]) + I.kernels(listings["gemm256.hip"]):
        if not I.relax_ok(t):
            continue
        for label, inss in I.blocks(body):
            begins = [i for i, x in enumerate(inss) if x in ("MARK BEGIN wide", "MARK BEGIN swiglu_wide")]
            for b in begins:
                rest = inss[b + 1:]
                end = next((i for i, x in enumerate(rest) if x.startswith("MARK")), None)
                if end is None:                      # an activation with branches of its own (erf): covered by the path walk above
                    other += 1
                    continue
                assert rest[end].startswith("MARK END"), (name, label)
                assert not any(x.startswith(("s_cbranch", "s_branch")) for x in rest[:end]), (name, label)
                n += 1
    assert n >= 40 and other <= n // 6, (n, other)


def test_the_check_sees_a_missing_and_an_extra_store(listings):
    """the checker itself: delete one buffer store inside a `wide` region / duplicate one inside a `resid` region -> reported"""
    import isa_epilogue_stores as I
    text = listings["gemm256_b.hip"]
    lines = text.split("\n")
    b = next(i for i, l in enumerate(lines) if "UFV_EPI_BEGIN wide" in l)
    s = next(i for i in range(b, len(lines)) if "buffer_store_dwordx4" in lines[i])
    bad, _, _ = I.check("\n".join(lines[:s] + lines[s + 1:]))
    assert bad and bad[0][1] == "wide" and bad[0][2] == "stores"
    b = next(i for i, l in enumerate(lines) if "UFV_EPI_BEGIN resid" in l)
    s = next(i for i in range(b, len(lines)) if re.match(r"\s*global_store_dwordx", lines[i]))
    bad, _, _ = I.check("\n".join(lines[:s] + [lines[s]] + lines[s:]))
    assert bad and bad[0][1] == "resid" and bad[0][2] == "stores"


def test_no_mfma_step_is_empty(listings):
    """Every MFMA step of the ping-pong K loop is bracketed by `s_setprio 1` ... `s_setprio 0` between two barriers: one wave group multiplies while the other
    loads.  hipcc once moved a step's MFMAs BEHIND its closing barrier (round 5, the block-scaled e4m3 kernels: a shift in front of the MFMAs was enough; the
    listing showed the two s_setprio back to back, the two groups then multiplied and waited in the same phases and the kernel ran 25 % slower -- every numerics
    test green).  The listing must hold no such empty step."""
    for src, text in listings.items():
        lines = [l.strip() for l in text.split("\n") if l.strip() and not l.strip().startswith(";")]
        empty = sum(1 for a, b in zip(lines, lines[1:]) if a.startswith("s_setprio 1") and b.startswith("s_setprio 0"))
        steps = sum(1 for a in lines if a.startswith("s_setprio 1"))
        assert steps > 0 and empty == 0, (src, steps, empty)
