"""Diagnostic: tile kernels vs the stream-K split at the config-#2 GEMM shapes (bf16 and fp8), same process."""
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

shapes = [("vit_qkv", 18432, 3456, 1152, 0), ("vit_o", 18432, 1152, 1152, 1), ("vit_fc1", 18432, 4352, 1152, 0), ("vit_fc2", 18432, 1152, 4352, 1),
          ("proj_c1", 18432, 3584, 1152, 0), ("proj_c3", 18432, 3584, 3584, 0), ("proj_c3d", 2304, 3584, 28672, 0), ("proj_ro", 2304, 3584, 3584, 0),
          ("llm_qkv", 2399, 4608, 3584, 0), ("llm_o", 2399, 3584, 3584, 1), ("llm_down", 2399, 3584, 18944, 1)]
for name, M, N, K, f32res in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    if f32res:
        out = torch.randn(M, N, device="cuda"); kw = dict(resid=out, out=out)
    else:
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); kw = dict(out=out)
    r = {}
    for kn, kern in (("auto", ops.GEMM_AUTO), ("k128", ops.GEMM_FAST), ("k256", ops.GEMM_FAST256), ("sk", ops.GEMM_STREAMK)):
        r[kn] = timeit(lambda: ops.gemm(a, w, kernel=kern, **kw))
    W8 = ops.Fp8Weight(w); aq, sa = ops.quantize_fp8(a)
    q = {}
    for kn, kern in (("auto", ops.GEMM_AUTO), ("k128", ops.GEMM_FAST), ("k256", ops.GEMM_FAST256), ("sk", ops.GEMM_STREAMK)):
        q[kn] = timeit(lambda: ops.gemm_fp8(aq, sa, W8, kernel=kern, **kw))
    fl = 2.0 * M * N * K
    best = min(r, key=r.get)
    print(f"{name:9s} M={M:6d} N={N:6d} K={K:6d} {'f32+res' if f32res else 'bf16   '}: bf16 auto {r['auto']*1e3:6.1f} k128 {r['k128']*1e3:6.1f} k256 {r['k256']*1e3:6.1f} sk {r['sk']*1e3:6.1f} us"
          f" (best {best} {fl/r[best]/1e9:6.0f} TF/s) | fp8 auto {q['auto']*1e3:6.1f} k128 {q['k128']*1e3:6.1f} k256 {q['k256']*1e3:6.1f} sk {q['sk']*1e3:6.1f} us", flush=True)
