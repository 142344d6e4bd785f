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
for n in (4096, 8192):
    a = (torch.rand(n, n, device="cuda") * 2 - 1).to(torch.bfloat16); w = (torch.rand(n, n, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
    for name, k in (("k128", ops.GEMM_FAST), ("k256", ops.GEMM_FAST256)):
        ms = timeit(lambda: ops.gemm(a, w, out=out, kernel=k))
        print(f"{n}^3 {name}: {ms:.3f} ms  {2*n**3/ms/1e9:.0f} TF/s")
