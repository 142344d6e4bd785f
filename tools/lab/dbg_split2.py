import sys, os; sys.path.insert(0, "/root/repo")
import torch
from ufvideo_amd import ops
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
PP = lambda shape: 4 | (shape << 8)
M, N, K = 2399, 3584, 18944
a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
x = torch.randn(M, N, device="cuda"); y = torch.empty_like(x)
for rep in range(2):
    print("auto in place      %.1f" % timeit(lambda: ops.gemm(a, w, resid=x, out=x)))
    print("auto out of place  %.1f" % timeit(lambda: ops.gemm(a, w, resid=x, out=y)))
    print("1441/4 in place    %.1f" % timeit(lambda: ops.gemm(a, w, resid=x, out=x, kernel=PP(41441))))
    print("1441/4 out of place %.1f" % timeit(lambda: ops.gemm(a, w, resid=x, out=y, kernel=PP(41441))))
    os.environ["UFV_GEMM_NO_SPLITK"] = "1"
    print("unsplit in place   %.1f" % timeit(lambda: ops.gemm(a, w, resid=x, out=x)))
    del os.environ["UFV_GEMM_NO_SPLITK"]
