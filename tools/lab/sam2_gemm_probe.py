"""lab: Hiera's short-K GEMM shapes under AUTO, the 128-wide kernel and the named ping-pong shapes"""
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
for name, (M, N, K, mode) in (("s3 fc1", (32768, 2304, 640, "gelu")), ("s3 fc2", (32768, 640, 2304, "res")), ("s3 qkv", (32768, 1920, 640, "")), ("s3 proj", (32768, 640, 640, "")),
                              ("vit fc1", (18432, 4352, 1152, "gelu_tanh")), ("readout", (2304, 3584, 3584, "gelu")), ("s4 fc1", (8192, 4608, 1152, "gelu")), ("s2 fc1", (131072, 1152, 384, "gelu")), ("s2 fc2", (131072, 384, 1152, "res")), ("s1 fc1", (524288, 640, 256, "gelu"))):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16) * 0.02
    bias = torch.randn(N, device="cuda")
    y = torch.randn(M, N, device="cuda") if mode == "res" else None
    row = []
    for label, kid in [("auto", ops.GEMM_AUTO), ("fast128", ops.GEMM_FAST)] + [(str(s), PP(s)) for s in (1442, 1441, 1432, 1431, 1332, 1331, 1322)]:
        try:
            if mode == "res":
                fn = lambda: ops.gemm(a, w, bias=bias, resid=y, out=y, kernel=kid)
            else:
                fn = lambda: ops.gemm(a, w, bias=bias, act=(mode if mode.startswith("gelu") else None), kernel=kid)
            row.append(f"{label} {t(fn):6.1f}")
        except Exception as ex:
            row.append(f"{label}   n/a")
    pick = lib.ufv_gemm_choice(M, N, K, int(mode == "res"), 0, int(not mode.startswith("gelu")))
    print(f"{name:8s} M{M} N{N} K{K} {mode:4s} [auto {pick}]: " + " | ".join(row), flush=True)
