"""Calibration of the GEMM kernel-choice model (csrc/gemm.hip: choose_kernel): per tile shape, the time of one K-tile step and of one
tile seam (prologue + epilogue), from exactly-tiled problems of three full rounds at two K depths.  Prints a table to paste."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ufvideo_amd import ops

def timeit(fn, iters=12, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3      # us

PP = lambda shape: 4 | (shape << 8)
DIMS = {442: (256, 256), 322: (160, 256), 332: (192, 256), 432: (224, 256), 321: (160, 192), 331: (192, 192), 431: (224, 192), 441: (256, 192)}
K1, K2 = 1152, 4352
rows = []
cands = [(f"pp{1000 + s}", PP(1000 + s), DIMS[s], 256) for s in DIMS] + [("pp442", PP(442), (256, 256), 256)]
cands += [(f"k128_mt{mt}", None, (32 * mt, 128), 512) for mt in (4, 5, 6)]
for name, kern, (bm, bn), slots in cands:
    # 3 full rounds: tiles_m x tiles_n = 3 * slots
    tn = 8 if slots == 256 else 16
    tm = 3 * slots // tn
    M, N = bm * tm, bn * tn
    res = {}
    for f32 in (0, 1):
        for K in (K1, K2):
            a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
            bias = torch.randn(N, device="cuda")
            if f32:
                r0 = torch.randn(M, N, device="cuda"); out = torch.empty(M, N, device="cuda"); kw = dict(resid=r0, out=out)
            else:
                out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); kw = dict(out=out)
            if kern is None:
                # the 128-wide kernel picks its own row count from (M, N): exact tiling makes it pick bm
                t = timeit(lambda: ops.gemm(a, w, bias=bias, kernel=ops.GEMM_FAST, **kw))
            else:
                t = timeit(lambda: ops.gemm(a, w, bias=bias, kernel=kern, **kw))
            res[(f32, K)] = t / 3.0                    # per round
            del a, w, out
    for f32 in (0, 1):
        tk = (res[(f32, K2)] - res[(f32, K1)]) / ((K2 - K1) / 64)
        te = res[(f32, K1)] - tk * K1 / 64
        tf = 2.0 * M * N * K2 / (3 * res[(f32, K2)]) / 1e6
        print(f"{name:10s} {bm}x{bn} {'f32+res' if f32 else 'bf16   '}: K-tile {tk:6.3f} us  seam {te:6.2f} us   ({tf:5.0f} TF/s at K={K2})", flush=True)
