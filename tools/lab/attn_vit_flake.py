"""Lab (GPU box): repeat the third-generation ViT attention kernel's bit-identity case (tests/test_kernels_gpu.py::test_vit72_third_generation_bit_identical_and_rescale_paths)
many times, and at the bench's size, counting launches whose output differs from the first launch of the same kernel (self-consistency) and from the second-generation kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufvideo_amd import ops  # noqa: E402


def g(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).cuda()


import ctypes
_P = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblds_poison.so"))
_SINK = None


def poison(pattern):
    global _SINK
    if _SINK is None:
        _SINK = torch.zeros(4, dtype=torch.int32, device="cuda")
    rc = _P.lds_poison(ctypes.c_uint(pattern), ctypes.c_void_p(_SINK.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc


def case(B, H, seed, reps):
    S, hd = 576, 72
    q, k, v = g(B, S, H, hd, seed=seed), g(B, S, H, hd, seed=seed + 100), g(B, S, H, hd, seed=seed + 200)
    for row, key, amp in ((5, 70, 3.0), (5, 300, 6.0), (40, 560, 5.0), (70, 130, 4.0), (100, 300, 5.0), (200, 450, 6.0), (250, 70, 4.0), (300, 520, 7.0), (500, 200, 5.0), (575, 575, 6.0)):
        k[:, key] = q[:, row] * amp
    k[:, :64] = -q[:, 150:214].abs().mean(dim=(1, 2), keepdim=True) * torch.sign(q[:, 150:151]) * 2.0
    q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    st = ((S * H * hd, H * hd),) * 3
    old0 = ops.attention(q, k, v, B, H, H, S, S, hd, *st, kernel=11)
    new0 = ops.attention(q, k, v, B, H, H, S, S, hd, *st, kernel=14)
    bad_new = bad_old = 0
    first = None
    for i in range(reps):
        new = ops.attention(q, k, v, B, H, H, S, S, hd, *st, kernel=14)
        old = ops.attention(q, k, v, B, H, H, S, S, hd, *st, kernel=11)
        if not torch.equal(new, new0):
            bad_new += 1
            if first is None:
                d = (new != new0).nonzero()
                first = ("new", i, int((new != new0).sum()), d[:4].tolist(), d[-1].tolist())
        if not torch.equal(old, old0):
            bad_old += 1
            if first is None:
                d = (old != old0).nonzero()
                first = ("old", i, int((old != old0).sum()), d[:4].tolist(), d[-1].tolist())
    pz = {}
    for name, pat in (("nan", 0x7FC07FC0), ("inf", 0x7F807F80), ("ones", 0xFFFFFFFF), ("big", 0x7F7F7F7F), ("zero", 0)):
        poison(pat)
        n14 = ops.attention(q, k, v, B, H, H, S, S, hd, *st, kernel=14)
        poison(pat)
        n11 = ops.attention(q, k, v, B, H, H, S, S, hd, *st, kernel=11)
        pz[name] = (int((n14 != new0).sum()), int((n11 != old0).sum()), bool(torch.isfinite(n14.float()).all()))
    print("  after an LDS poison (mismatches kernel 14, kernel 11, finite):", pz, flush=True)
    print(f"B {B} H {H}: new0 == old0 {torch.equal(new0, old0)}; of {reps} repeats: kernel 14 differs from its first run {bad_new} x, kernel 11 {bad_old} x; first: {first}", flush=True)


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    for B, H, seed in ((2, 4, 50), (1, 16, 51), (3, 2, 52), (32, 16, 53)):
        case(B, H, seed, reps if B < 32 else reps // 3)
