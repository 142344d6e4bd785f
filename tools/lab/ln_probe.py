"""lab: LayerNorm on the ViT stream shape vs a plain device copy of the same bytes (what the box's HBM gives a read + write stream)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufvideo_amd import ops

def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / n

M, D = 18432, 1152
DT = torch.float32 if "--f32" in sys.argv else torch.bfloat16
x = torch.randn(M, D, device="cuda").to(DT)
y = torch.empty(M, D, device='cuda', dtype=torch.bfloat16)
w = torch.randn(D, device="cuda"); b = torch.randn(D, device="cuda")
# rotate over several buffers so that the stream comes from HBM, not from the 256 MB Infinity Cache
xs = [x.clone() for _ in range(8)]; ys = [torch.empty(M, D, device='cuda', dtype=torch.bfloat16) for _ in range(8)]
i = [0]
def ln():
    i[0] = (i[0] + 1) % 8
    ops.layernorm(xs[i[0]], w, b, 1e-6, out=ys[i[0]])
def cp():
    i[0] = (i[0] + 1) % 8
    ys[i[0]].copy_(xs[i[0]])
def ln_hot():
    ops.layernorm(x, w, b, 1e-6, out=y)
def cp_hot():
    y.copy_(x)
gb = M * D * (2 + x.element_size()) / 1e3
for name, fn in (("layernorm (8 rotating buffers)", ln), ("torch copy (8 rotating buffers)", cp), ("layernorm (one buffer)", ln_hot), ("torch copy (one buffer)", cp_hot)):
    us = t(fn)
    print(f"{name:36s} {us:7.1f} us   {gb / us / 1e3:5.2f} TB/s")
