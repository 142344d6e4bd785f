"""Diagnostic: stream-K split vs tile kernels on every GEMM shape of config #2 (bf16 out and fp32 residual epilogue)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ufvideo_amd import ops


def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device="cuda") * scale).to(torch.bfloat16)


shapes = {"vit_qkv": (18432, 3456, 1152, 0), "vit_o": (18432, 1152, 1152, 1), "vit_fc1": (18432, 4352, 1152, 0), "vit_fc2": (18432, 1152, 4352, 1),
          "proj_c1": (18432, 3584, 1152, 0), "proj_c3": (18432, 3584, 3584, 0),
          "llm_qkv": (2399, 4608, 3584, 0), "llm_o": (2399, 3584, 3584, 1), "llm_down": (2399, 3584, 18944, 1)}
for name, (M, N, K, r) in shapes.items():
    a, w = rnd(M, K), rnd(N, K, scale=0.02)
    bias = torch.zeros(N, device="cuda")
    res = torch.randn(M, N, device="cuda") if r else None
    ob = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    line = f"{name:9s}"
    for kn, kk in (("k128", ops.GEMM_FAST), ("k256", ops.GEMM_FAST256), ("sk", ops.GEMM_STREAMK)):
        if r:
            ms = timeit(lambda: ops.gemm(a, w, bias=bias, resid=res, out=res, kernel=kk))
        else:
            ms = timeit(lambda: ops.gemm(a, w, bias=bias, out=ob, kernel=kk))
        line += f" | {kn} {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF/s"
    print(line + ("  (fp32 resid)" if r else ""), flush=True)
