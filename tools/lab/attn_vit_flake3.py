"""Lab (GPU box): kernel 14 under timing perturbation -- a second stream copies / multiplies large buffers while the attention kernel runs (its LDS-DMA returns late and out of
step); the result is compared with the quiet run and with kernel 11.  usage: python3 tools/lab/attn_vit_flake3.py [iterations]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufvideo_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
hd, S = 72, 576
gen = torch.Generator(device="cuda").manual_seed(11)
side = torch.cuda.Stream()
big = torch.randn(64 << 20, device="cuda")            # 256 MB
big2 = torch.empty_like(big)
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
bad = 0
for it in range(n):
    B, H = [(32, 16), (2, 4), (1, 16), (3, 2), (8, 16)][it % 5]
    q, k, v = [torch.randn(B, S, H, hd, device="cuda", generator=gen).to(torch.bfloat16) for _ in range(3)]
    st = ((S * H * hd, H * hd),) * 3
    quiet = ops.attention(q, k, v, B, H, H, S, S, hd, *st, kernel=14)
    old = ops.attention(q, k, v, B, H, H, S, S, hd, *st, kernel=11)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(3):
            if it % 2:
                big2.copy_(big)
            else:
                torch.mm(a, a)
    noisy = [ops.attention(q, k, v, B, H, H, S, S, hd, *st, kernel=14) for _ in range(4)]
    torch.cuda.synchronize()
    ok = torch.equal(quiet, old) and all(torch.equal(x, old) for x in noisy)
    if not ok:
        bad += 1
        print("MISMATCH it", it, (B, H), [int((x != old).sum()) for x in [quiet] + noisy], flush=True)
print(f"{n} iterations x 5 launches under a second stream's traffic: {bad} with a mismatch")
