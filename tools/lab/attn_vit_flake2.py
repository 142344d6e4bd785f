"""Lab (GPU box): kernel 14 (third-generation ViT attention) against kernel 11 on FRESH data every iteration (new seeds, new buffers from the caching allocator, other kernels
in between), the way the test suite meets it; counts element mismatches.  usage: python3 tools/lab/attn_vit_flake2.py [iterations]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from ufvideo_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
hd, S = 72, 576
gen = torch.Generator(device="cuda").manual_seed(7)
for it in range(n):
    B, H = [(2, 4), (1, 16), (3, 2), (32, 16), (5, 6), (1, 2)][it % 6]
    q, k, v = [torch.randn(B, S, H, hd, device="cuda", generator=gen) for _ in range(3)]
    if it % 2:
        for row, key, amp in ((5, 70, 3.0), (5, 300, 6.0), (300, 520, 7.0), (575, 575, 6.0)):
            k[:, key] = q[:, row] * amp
    q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    st = ((S * H * hd, H * hd),) * 3
    new = ops.attention(q, k, v, B, H, H, S, S, hd, *st, kernel=14)
    junk = torch.randn(1 << (10 + it % 12), device="cuda").sin_()                       # something else on the stream, other allocations
    old = ops.attention(q, k, v, B, H, H, S, S, hd, *st, kernel=11)
    again = ops.attention(q, k, v, B, H, H, S, S, hd, *st, kernel=14)
    if not (torch.equal(new, old) and torch.equal(again, old)):
        bad += 1
        d = (new != old).nonzero()
        d2 = (again != old).nonzero()
        print("MISMATCH it", it, (B, H), "new vs old", int(d.shape[0]), d[:3].tolist(), "again vs old", int(d2.shape[0]), d2[:3].tolist(), flush=True)
    del junk
print(f"{n} iterations, {bad} with a mismatch")
import test_kernels_gpu as t  # noqa: E402
fails = 0
for i in range(40):
    try:
        t.test_vit72_third_generation_bit_identical_and_rescale_paths()
    except AssertionError as e:
        fails += 1
        print("TEST FAILED", i, str(e)[:200], flush=True)
print("test function x 40:", fails, "failures")
