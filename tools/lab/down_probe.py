"""lab: the decoder's `down` / `o_proj` products under the kernel forms the library has (AUTO, stream-K, named ping-pong shapes, split-K)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufvideo_amd import ops, _lib

def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / n

PP = lambda shape: 4 | (shape << 8)
for name, (M, N, K) in (("down", (2399, 3584, 18944)), ("o_proj", (2399, 3584, 3584)), ("vit_fc2", (18432, 1152, 4352)), ("qkv", (2399, 4608, 3584))):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16) * 0.02
    x = torch.randn(M, N, device="cuda")
    out_bf16 = name == "qkv"
    ref = None
    row = []
    forms = [("auto", ops.GEMM_AUTO), ("streamk", ops.GEMM_STREAMK)] + [(str(s), PP(s)) for s in (1442, 1441, 1432, 1431, 1332, 1331, 1322)] + \
            [(f"{s}x{p}", PP(s + 10000 * p)) for s, p in ((1442, 2), (1442, 4), (1331, 2))]
    for label, kid in forms:
        try:
            if out_bf16:
                fn = lambda: ops.gemm(a, w, kernel=kid)
            else:
                y = x.clone()
                fn = lambda: ops.gemm(a, w, resid=y, out=y, kernel=kid)
            us = t(fn)
            row.append(f"{label} {us:6.1f}")
        except Exception as ex:
            row.append(f"{label}   n/a")
    print(f"{name:8s} M{M} N{N} K{K}  [{2 * M * N * K / 1e6:.0f} MF]: " + " | ".join(row), flush=True)
