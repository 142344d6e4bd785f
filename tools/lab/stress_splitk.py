import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ufvideo_amd import ops
torch.manual_seed(0)
M, N, K = 2399, 3584, 18944
a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
r = torch.randn(M, N, device="cuda")
ref = ops.gemm(a, w, resid=r, out_dtype=torch.float32)
bad = 0
for i in range(1500):
    x = ops.gemm(a, w, resid=r, out_dtype=torch.float32)
    if i % 50 == 0 and not torch.equal(x, ref): bad += 1
# interleave with other shapes (different tile counts -> flag reuse across shapes)
a2 = torch.randn(300, K, device="cuda").to(torch.bfloat16); r2 = torch.randn(300, N, device="cuda")
ref2 = ops.gemm(a2, w, resid=r2, out_dtype=torch.float32)
for i in range(300):
    x = ops.gemm(a, w, resid=r, out_dtype=torch.float32); y = ops.gemm(a2, w, resid=r2, out_dtype=torch.float32)
    if i % 25 == 0 and (not torch.equal(x, ref) or not torch.equal(y, ref2)): bad += 1
torch.cuda.synchronize()
print("stress done, mismatches:", bad)
