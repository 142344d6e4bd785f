import sys; sys.path.insert(0, "/root/repo")
import torch
from ufvideo_amd import ops
PP = lambda shape: 4 | (shape << 8)
torch.manual_seed(0)
for (M, N, K, shape, parts) in [(256, 256, 256, 1442, 2), (256, 256, 512, 1442, 4), (512, 512, 256, 1442, 2), (2399, 3584, 1024, 1441, 4), (2399, 3584, 18944, 1441, 4)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    res0 = torch.randn(M, N, device="cuda")
    ref = ops.gemm(a, w, resid=res0, out_dtype=torch.float32, kernel=ops.GEMM_FAST)
    x = res0.clone()
    ops.gemm(a, w, resid=x, out=x, kernel=PP(shape + 10000 * parts))
    torch.cuda.synchronize()
    d = (x - ref).abs()
    bad = (d > 1e-3).nonzero()
    print(M, N, K, shape, parts, "max err", float(d.max()), "bad", bad.shape[0], "of", M * N, bad[:5].tolist() if bad.shape[0] else "")
    if bad.shape[0]:
        i, j = bad[0].tolist()
        print("   x", float(x[i, j]), "ref", float(ref[i, j]), "res0", float(res0[i, j]), "prod", float(ref[i, j] - res0[i, j]))
