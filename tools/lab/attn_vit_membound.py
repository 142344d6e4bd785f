"""Lab (GPU box): is attn_fwd_vit72_p2 paced by its K / V re-reads (3 passes per head: PMC FETCH_SIZE 440 MB against 127 MB of q, k, v)?  The same launch with the batch
stride of k / v (and q) set to 0: every frame then reads frame 0's rows, which stay in L2, and the instruction stream is unchanged."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufvideo_amd import ops  # noqa: E402


def timed(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B, H, S, hd = 32, 16, 576, 72
    dev = torch.device("cuda", 0)
    qkv = torch.randn(B * S, 3 * H * hd, device=dev).to(torch.bfloat16)
    ld = qkv.stride(0)
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:2 * H * hd], qkv[:, 2 * H * hd:]
    out = torch.empty(B * S, H * hd, device=dev, dtype=torch.bfloat16)
    full = (S * ld, ld)
    zero = (0, ld)
    for name, qs, ks in (("all frames distinct", full, full), ("k / v of frame 0 for every frame", full, zero), ("q, k, v of frame 0 for every frame", zero, zero)):
        t = timed(lambda: ops.attention(q, k, v, B, H, H, S, S, hd, qs, ks, ks, out=out))
        print(f"{name:40s} {t:7.1f} us")


def pair_major():
    """the same work with q / k / v stored [frame x head pair][token][2 heads x 72]: the rows a block streams are 288 contiguous bytes that no other block shares"""
    B, H, S, hd = 32, 16, 576, 72
    dev = torch.device("cuda", 0)
    q, k, v = [torch.randn(B * H // 2, S, 2 * hd, device=dev).to(torch.bfloat16) for _ in range(3)]
    out = torch.empty(B * H // 2 * S, 2 * hd, device=dev, dtype=torch.bfloat16)
    st = (S * 2 * hd, 2 * hd)
    t = timed(lambda: ops.attention(q, k, v, B * H // 2, 2, 2, S, S, hd, st, st, st, out=out))
    print(f"{'pair-major q / k / v / o':40s} {t:7.1f} us")
    out2 = torch.empty(B * S, H * hd, device=dev, dtype=torch.bfloat16)       # token-major o as the product needs it (one frame's worth of address pattern per 8 pairs is not expressible: pitch only)
    t = timed(lambda: ops.attention(q, k, v, B * H // 2, 2, 2, S, S, hd, st, st, st, out=out2.view(B * H // 2 * S, 2 * hd)))
    print(f"{'pair-major q / k / v, o other buffer':40s} {t:7.1f} us")


if __name__ == "__main__":
    pair_major()
    main()
