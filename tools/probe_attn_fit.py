"""Diagnostic: ViT attention (hd 72, 32 frames x 16 heads x 576 queries) time vs number of 64-key tiles -> fixed cost + per-tile cost."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ufvideo_amd import ops


def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


T, H, S, hd = 32, 16, 576, 72
qkv = torch.randn(T * S, 3 * H * hd, device="cuda").to(torch.bfloat16)
o = torch.empty(T * S, H * hd, device="cuda", dtype=torch.bfloat16)
st = (S * 3 * H * hd, 3 * H * hd)
for kern in (1, 6):
    for Sk in (64, 128, 192, 320, 448, 576):
        ms = timeit(lambda: ops.attention(qkv, qkv[:, H * hd:], qkv[:, 2 * H * hd:], T, H, H, S, Sk, hd, st, st, st, out=o, kernel=kern))
        print(f"kernel={kern} Sk={Sk:4d} tiles={Sk // 64}: {ms * 1e3:7.1f} us", flush=True)
