"""lab: `down` (2399 x 3584 x 18944, fp32 + residual) on the 192x192 and 160x256 ping-pong shapes, interleaved repeats"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufvideo_amd import ops
def t(fn, n=40):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / n
PP = lambda shape: 4 | (shape << 8)
for M in (2399, 4703):
    N, K = 3584, 18944
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16) * 0.02
    y = torch.randn(M, N, device="cuda")
    for rep in range(3):
        row = []
        for s in (1331, 1322, 1332, 1431):
            row.append(f"{s} {t(lambda: ops.gemm(a, w, resid=y, out=y, kernel=PP(s))):6.1f}")
        print(f"M {M} rep {rep}: " + " | ".join(row), flush=True)
