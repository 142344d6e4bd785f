"""Diagnostic: bf16 vs fp8 (W8A8) GEMM at the config-#2 shapes, same process (clock state shared)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ufvideo_amd import ops

def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

shapes = [("vit_qkv", 18432, 3456, 1152, 0), ("vit_o", 18432, 1152, 1152, 0), ("vit_fc1", 18432, 4352, 1152, 0), ("vit_fc2", 18432, 1152, 4352, 0),
          ("proj_c3", 18432, 3584, 3584, 0), ("llm_qkv", 2399, 4608, 3584, 0), ("llm_o", 2399, 3584, 3584, 0), ("llm_gu", 2399, 37888, 3584, 1),
          ("llm_down", 2399, 3584, 18944, 0)]
tot_b = tot_q = 0.0
for name, M, N, K, sw in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    out = torch.empty(M, N // 2 if sw else N, device="cuda", dtype=torch.bfloat16)
    W8 = ops.Fp8Weight(w); aq, sa = ops.quantize_fp8(a)
    tb = timeit(lambda: ops.gemm(a, w, out=out, swiglu=bool(sw)))
    res = {}
    for kn, kern in (("auto", ops.GEMM_AUTO), ("k128", ops.GEMM_FAST), ("k256", ops.GEMM_FAST256)):
        res[kn] = timeit(lambda: ops.gemm_fp8(aq, sa, W8, out=out, swiglu=bool(sw), kernel=kern))
    tq = timeit(lambda: ops.quantize_fp8(a, out=aq, scale=sa))
    fl = 2.0 * M * N * K
    print(f"{name:9s} M={M:6d} N={N:6d} K={K:6d}: bf16 {tb*1e3:7.1f} us {fl/tb/1e9:7.1f} TF/s | fp8 auto {res['auto']*1e3:7.1f} us {fl/res['auto']/1e9:7.1f} TF/s"
          f" (k128 {res['k128']*1e3:6.1f}, k256 {res['k256']*1e3:6.1f}) | quantize A {tq*1e3:6.1f} us", flush=True)
