"""lab: Hiera-L stage-3 GEMM shapes (K = 576 or 2304, widths 1728 / 576 / 2304 unpadded since round 6) on every ping-pong tile shape that takes them: which shape should AUTO pick for short-K products?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufvideo_amd import ops, _lib
def pp(code): return 4 | (code << 8)
cases = [(39200, 1728, 576, False, None), (32768, 1728, 576, False, None), (39200, 576, 576, False, None), (32768, 2304, 576, False, "gelu"), (32768, 576, 2304, True, None)]
for M, N, K, f32, act in cases:
    a = (torch.randn(M, K, device="cuda")).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16) for _ in range(4)]
    bias = torch.zeros(N, device="cuda")
    x = torch.zeros(M, N, device="cuda") if f32 else None
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    res = {}
    for code in (0, 1441, 1431, 1331, 1442, 1432, 1332, 1322):
        kern = ops.GEMM_AUTO if code == 0 else pp(code)
        def run(i):
            if f32: ops.gemm(a, ws[i % 4], bias=bias, resid=x, out=x, kernel=kern)
            else: ops.gemm(a, ws[i % 4], bias=bias, act=act, out=out, kernel=kern)
        try:
            for i in range(4): run(i)
        except _lib.UfvError as e:
            continue
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(20): run(i)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        res[code] = best
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} {'f32+res' if f32 else 'bf16'} {act or ''}: " + "  ".join(f"{'AUTO' if c == 0 else c}: {t:.1f} us ({fl / t / 1e6:.0f})" for c, t in res.items()), flush=True)
