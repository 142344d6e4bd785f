"""CPU: the decode kernels issue some loads by hand (inline asm) so that the compiler cannot wait for them too early.  What the compiler does not know about it may also
TOUCH too early: the register allocator once moved such a buffer to another register across a loop's back edge while its loads were in flight (a pipelined form
of the 4-output decode GEMV, round 4: wrong sums on the GPU; that form was dropped, LABNOTES.md).  This test compiles the source to gfx950 assembly and checks, with an in-order model of the load queue, that no instruction reads or
writes the destination of a hand-issued load before a wait covers it (tools/isa_inflight_check.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src,pattern,expect", [("attn.hip", "attn_decode_fused", 2)])
def test_no_instruction_touches_a_register_whose_hand_issued_load_is_in_flight(tmp_path, src, pattern, expect):
    import isa_inflight_check
    out = tmp_path / (src + ".s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", str(out),
                        os.path.join(ROOT, "ufvideo_amd", "csrc", src)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    bad, kernels = isa_inflight_check.check(out.read_text(), pattern)
    assert kernels == expect, kernels            # the kernels this guards exist (a renamed kernel must not turn the check into a no-op)
    assert not bad, bad[:10]
