"""lab: A/B of two builds of the library on the 192-row GEMM shapes, same process order interleaved (usage: ab_gemm.py <old.so> <new.so>)"""
import sys, os, subprocess
if len(sys.argv) == 3:
    for rep in range(3):
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
row = []
for name, (M, N, K) in (("down", (2399, 3584, 18944)), ("o_proj", (2399, 3584, 3584)), ("vit_o", (18432, 1152, 1152)), ("vit_fc2", (18432, 1152, 4352)), ("llm_qkv", (2399, 4608, 3584))):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16) * 0.02
    if name == "llm_qkv":
        us = t(lambda: ops.gemm(a, w))
    else:
        y = torch.randn(M, N, device="cuda")
        us = t(lambda: ops.gemm(a, w, resid=y, out=y))
    row.append(f"{name} {us:6.1f}")
print(tag, " | ".join(row), flush=True)
