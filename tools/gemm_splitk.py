"""Diagnostic: aligned split-K of the ping-pong GEMM (UFV_GEMM_PP(shape + 10000 * parts)) on the residual GEMMs with few tiles; checks each
result against the unsplit kernel (same products, a different summation tree: <= 1e-6 relative) and repeated launches bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ufvideo_amd import ops, _lib

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

PP = lambda shape: 4 | (shape << 8)
cases = [("llm_down", 2399, 3584, 18944), ("llm_o", 2399, 3584, 3584), ("down_64f", 4703, 3584, 18944), ("vit_fc2", 18432, 1152, 4352),
         ("down_1200", 1200, 3584, 18944), ("down_300", 300, 3584, 18944)]
combos = [(1432, 3), (1442, 5), (1441, 4), (1432, 4), (1432, 5), (1322, 8), (1322, 6), (1331, 5)]
for name, M, N, K in cases:
    torch.manual_seed(0)
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    res0 = torch.randn(M, N, device="cuda")
    ref = ops.gemm(a, w, resid=res0, out_dtype=torch.float32, kernel=ops.GEMM_FAST)
    line = f"{name:9s} M={M:6d} N={N:6d} K={K:6d}:"
    prev = _lib.load().ufv_gemm_set_splitk(0)
    t0 = timeit(lambda: ops.gemm(a, w, resid=res0, out=torch.empty_like(res0)))
    _lib.load().ufv_gemm_set_splitk(prev)
    xa = res0.clone(); ops.gemm(a, w, resid=xa, out=xa)
    assert float((xa - ref).abs().max() / ref.abs().max()) < 4e-6
    t = timeit(lambda: ops.gemm(a, w, resid=res0, out=torch.empty_like(res0)))
    line += f" unsplit {t0*1e3:6.1f} auto {t*1e3:6.1f}"
    best = ("auto", t)
    for shape, parts in combos:
        bn = 256 if shape % 10 == 2 else 192
        if N % bn not in (0, 128): continue
        nk = K // 64
        if (parts - 1) * -(-nk // parts) >= nk: continue           # an empty last K range: refused by the launcher
        kern = PP(shape + 10000 * parts)
        outs = []
        for _ in range(3):
            x = res0.clone()
            ops.gemm(a, w, resid=x, out=x, kernel=kern)            # in place, as the layers call it
            outs.append(x)
        err = float((outs[0] - ref).abs().max() / ref.abs().max())
        assert err < 4e-6, (name, shape, parts, err)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (name, shape, parts, "not deterministic")
        x = res0.clone()
        t = timeit(lambda: ops.gemm(a, w, resid=x, out=x, kernel=kern))
        line += f" {shape}/{parts} {t*1e3:6.1f}"
        if t < best[1]: best = (f"{shape}/{parts}", t)
        if M < 256: break
    print(line + f" us | best {best[0]} {2.0*M*N*K/best[1]/1e9:6.0f} TF/s", flush=True)
