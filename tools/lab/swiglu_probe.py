"""lab: the gate/up GEMM with the SwiGLU epilogue (dominant kernel), repeated"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufvideo_amd import ops
from ufvideo_amd.model.videorefer_qwen2 import pack_swiglu
def t(fn, n=30):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / n
M, I, K = 2399, 18944, 3584
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(2 * I, K, device="cuda") * 0.02).to(torch.bfloat16)
for rep in range(4):
    sw = t(lambda: ops.gemm(a, w, swiglu=True))
    plain = t(lambda: ops.gemm(a, w[:I]))
    print(f"rep {rep}: gate/up + SwiGLU {sw:6.1f} us ({2 * M * 2 * I * K / sw / 1e6:.0f} TF/s) | half the weights, plain bf16 out {plain:6.1f} us", flush=True)
