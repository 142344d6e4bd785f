"""lab: A/B of two builds on the connector's elementwise kernels (usage: ab_conn.py <old.so> <new.so>)"""
import sys, os, subprocess
if len(sys.argv) == 3:
    for rep in range(2):
        for tag, so in (("old", sys.argv[1]), ("new", sys.argv[2])):
            subprocess.run([sys.executable, __file__, "--run", tag, so], check=True)
    sys.exit(0)
tag, so = sys.argv[2], sys.argv[3]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ufvideo_amd import _lib
_lib.LIB_PATH = so
from ufvideo_amd import ops
def t(fn, n=40):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / n
F, H, W, C = 32, 24, 24, 3584
M = F * H * W
x = torch.randn(M, C, device="cuda").to(torch.bfloat16)
x2 = torch.randn(M, C, device="cuda").to(torch.bfloat16)
w, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
w9 = torch.randn(9, C, device="cuda") * 0.3
row = [f"LN+SiLU {t(lambda: ops.layernorm(x, w, b, 1e-6, act='silu')):6.1f}",
       f"ln_add_silu {t(lambda: ops.ln_add_silu(x, w, b, x2, None, None, 1e-5)):6.1f}",
       f"dwconv_ln_silu {t(lambda: ops.dwconv3x3_ln_silu(x.view(F, H, W, C), w9, w, b, F, H, W, C, 1e-5)):6.1f}"]
print(tag, " | ".join(row), flush=True)
