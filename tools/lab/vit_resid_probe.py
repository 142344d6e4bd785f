"""lab: what the ViT's out_proj / fc2 would cost with a bf16 output instead of the fp32 in-place residual (upper bound of a bf16 residual stream's gain)"""
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
for name, (M, N, K) in (("vit_o", (18432, 1152, 1152)), ("vit_fc2", (18432, 1152, 4352))):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16) * 0.02
    bias = torch.randn(N, device="cuda")
    y = torch.randn(M, N, device="cuda")
    for rep in range(2):
        f32 = t(lambda: ops.gemm(a, w, bias=bias, resid=y, out=y))
        bf = t(lambda: ops.gemm(a, w, bias=bias))
        print(f"{name}: fp32 in-place residual {f32:6.1f} us | bf16 out, no residual {bf:6.1f} us", flush=True)
