"""lab: the connector's squeeze-excite FCs (M = frames) on the GEMV kernel vs the 128-tile MFMA kernel"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufvideo_amd import ops, _lib

def t(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / n

for M in (8, 16, 32, 64):
    for N, K in ((896, 3584), (3584, 896), (288, 1152), (1152, 288)):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        res = {}
        for name, kid in (("auto", ops.GEMM_AUTO), ("gemv", ops.GEMM_GEMV), ("fast128", ops.GEMM_FAST)):
            try:
                o = ops.gemm(a, w, bias=bias, act="silu", kernel=kid)
                res[name] = (t(lambda: ops.gemm(a, w, bias=bias, act="silu", kernel=kid)), o)
            except Exception as ex:
                res[name] = (float("nan"), None)
        d = (res["gemv"][1].float() - res["fast128"][1].float()).abs().max().item() if res["fast128"][1] is not None else -1
        print(f"M {M:3d} N {N:5d} K {K:5d}: " + "  ".join(f"{k} {v[0]:6.1f} us" for k, v in res.items()) + f"   max|gemv - fast| {d:.3g}")
