"""lab: the connector's second-stage GEMM shapes (M = 2304) under AUTO and the named ping-pong shapes"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufvideo_amd import ops, _lib
def t(fn, n=30):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / n
PP = lambda shape: 4 | (shape << 8)
lib = _lib.load()
for name, (M, N, K, f32) in (("s2 w1/w3", (2304, 3584, 3584, False)), ("sampler", (2304, 3584, 28672, False)), ("readout", (2304, 3584, 3584, True)), ("s1 w1 b0", (18432, 3584, 1152, False)),
                             ("s1 w3", (18432, 3584, 3584, False))):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16) * 0.02
    row = []
    for label, kid in [("auto", ops.GEMM_AUTO), ("fast128", ops.GEMM_FAST)] + [(str(s), PP(s)) for s in (1442, 1441, 1432, 1431, 1332, 1331, 1322)]:
        try:
            us = t(lambda: ops.gemm(a, w, kernel=kid, out_dtype=torch.float32 if f32 else torch.bfloat16))
            row.append(f"{label} {us:6.1f}")
        except Exception as ex:
            row.append(f"{label}   n/a")
    pick = lib.ufv_gemm_choice(M, N, K, int(f32), 0, 1) if hasattr(lib, "ufv_gemm_choice") else -1
    print(f"{name:9s} M{M} N{N} K{K} {'f32' if f32 else 'bf16'} [auto picks {pick}]: " + " | ".join(row), flush=True)
