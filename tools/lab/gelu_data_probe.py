"""lab: does the GELU-tanh epilogue's cost depend on the data?  (libm tanhf takes different branches by magnitude)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufvideo_amd import ops
def t(fn, n=30):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / n
M, N, K = 18432, 4352, 1152
w = torch.randn(N, K, device="cuda").to(torch.bfloat16) * 0.02
bias = torch.zeros(N, device="cuda")
for scale in (0.05, 0.2, 0.5, 1.0, 3.0):
    a = (torch.randn(M, K, device="cuda") * scale).to(torch.bfloat16)
    pre = (a[:256].float() @ w.float().t())
    row = [f"pre-activation std {float(pre.std()):.3f}"]
    for act in (None, "gelu_tanh", "gelu", "silu", "quick_gelu"):
        row.append(f"{act}: {t(lambda: ops.gemm(a, w, bias=bias, act=act)):6.1f}")
    print(" | ".join(row), flush=True)
