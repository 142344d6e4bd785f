"""Lab (GPU box): the e4m3 GEMMs of the fused W8A8 chain one at a time: per-row-scaled A (ufv_gemm_fp8) against block-scaled A (ufv_gemm_fp8_mx) at the decoder's
down_proj and the tower's fc2, and the bf16-output against the MX-emitting epilogue at gate/up and fc1.  UFV_LAB=1 UFV_LIBRARY=<variant .so> times another build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ufvideo_amd import ops
from ufvideo_amd.model.videorefer_qwen2 import pack_swiglu
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(0)


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, M, N, K in (("down", 2399, 3584, 18944), ("fc2", 18432, 1152, 4352)):
    a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    x = torch.randn(M, N, device=dev, generator=g)
    wq = ops.Fp8Weight(w)
    aq, sa = ops.quantize_fp8(a)
    am = ops.quantize_mx(a)
    t0 = timed(lambda: ops.gemm_fp8(aq, sa, wq, resid=x, out=x))
    t1 = timed(lambda: ops.gemm_fp8_mx(am, wq, resid=x, out=x))
    print(f"{name} {M}x{N}x{K}: per-row A {t0:.1f} us, block-scaled A {t1:.1f} us")
for name, M, N, K, sw in (("gate/up", 2399, 37888, 3584, True), ("fc1", 18432, 4352, 1152, False)):
    a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    wq = ops.Fp8Weight(w)
    aq, sa = ops.quantize_fp8(a)
    qa = ops.QAct(aq, sa)
    out = torch.empty(M, N // 2 if sw else N, device=dev, dtype=torch.bfloat16)
    t0 = timed(lambda: ops.gemm_fp8(aq, sa, wq, swiglu=sw, act=None if sw else "gelu_pytorch_tanh", out=out))
    t1 = timed(lambda: ops.gemm_fp8_mx(qa, wq, swiglu=sw, act=None if sw else "gelu_pytorch_tanh", mx_out=True))
    print(f"{name} {M}x{N}x{K}: bf16 out {t0:.1f} us, MX out {t1:.1f} us")
