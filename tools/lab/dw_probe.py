"""lab: the connector's depthwise 3x3 + LayerNorm + SiLU at its two stage shapes"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufvideo_amd import ops
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / n
for F, H, W, C in ((32, 24, 24, 3584), (16, 12, 12, 3584)):
    x = torch.randn(F, H, W, C, device="cuda").to(torch.bfloat16)
    w9 = torch.randn(9, C, device="cuda") * 0.3
    lnw, lnb = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    us = t(lambda: ops.dwconv3x3_ln_silu(x, w9, lnw, lnb, F, H, W, C, 1e-5))
    print(f"dwconv_ln_silu F{F} {H}x{W} C{C}: {us:.1f} us  ({2 * x.numel() * 2 / us / 1e6:.2f} TB/s in + out)")
