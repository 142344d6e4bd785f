"""Diagnostic: the ping-pong GEMM at every tile shape (UFV_GEMM_PP) against the 128-wide and 256x256 kernels, at the config-#2 shapes;
checks each result against torch.matmul first.  usage: python tools/gemm_shapes.py [fp8]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ufvideo_amd import ops

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

PP = lambda shape: 4 | (shape << 8)
SHAPES = [1442, 1432, 1332, 1322, 1441, 1431, 1331]
DIMS = {442: (256, 256), 322: (160, 256), 332: (192, 256), 432: (224, 256), 321: (160, 192), 331: (192, 192), 431: (224, 192), 441: (256, 192)}
cases = [("vit_qkv", 18432, 3456, 1152, 0), ("vit_o", 18432, 1152, 1152, 1), ("vit_fc1", 18432, 4352, 1152, 0), ("vit_fc2", 18432, 1152, 4352, 1),
         ("proj_c1", 18432, 3584, 1152, 0), ("proj_c3", 18432, 3584, 3584, 0), ("proj_ro", 2304, 3584, 3584, 0),
         ("llm_qkv", 2399, 4608, 3584, 0), ("llm_o", 2399, 3584, 3584, 1), ("llm_down", 2399, 3584, 18944, 1), ("lm_64f", 4703, 3584, 3584, 1)]
only = [a for a in sys.argv[1:] if a != "fp8"]
for name, M, N, K, f32res in cases:
    if only and name not in only:
        continue
    torch.manual_seed(0)
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    ref = a.float() @ w.float().T + bias
    res0 = torch.randn(M, N, device="cuda")
    if f32res:
        ref = ref + res0
    line = f"{name:9s} M={M:6d} N={N:6d} K={K:6d} {'f32+res' if f32res else 'bf16   '}:"
    fl = 2.0 * M * N * K
    best = None
    for kn, kern in [("auto", ops.GEMM_AUTO), ("k128", ops.GEMM_FAST)] + [(str(s), PP(s)) for s in SHAPES]:
        if kn.isdigit():
            bn = DIMS[int(kn) % 1000][1]
            if N % bn not in (0, 128):
                continue
        if f32res:
            out = torch.empty(M, N, device="cuda"); kw = dict(resid=res0, out=out)
        else:
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); kw = dict(out=out)
        ops.gemm(a, w, bias=bias, kernel=kern, **kw)
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        assert err < (1e-5 if f32res else 6e-3), (name, kn, err)
        t = timeit(lambda: ops.gemm(a, w, bias=bias, kernel=kern, **kw))
        line += f" {kn} {t*1e3:6.1f}"
        if best is None or t < best[1]:
            best = (kn, t)
    print(line + f" us | best {best[0]} {fl/best[1]/1e9:6.0f} TF/s", flush=True)

# gate/up with the SwiGLU epilogue (weight rows interleaved [16 gate | 16 up]): 256x256 tile, four vs two phases per K-tile
if not only or "gateup" in only:
    M, N, K = 2399, 37888, 3584
    torch.manual_seed(0)
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    outs = {}
    line = f"gate/up   M={M:6d} N={N:6d} K={K:6d} swiglu :"
    for kn, kern in (("auto", ops.GEMM_AUTO), ("1442", PP(1442))):
        out = torch.empty(M, N // 2, device="cuda", dtype=torch.bfloat16)
        ops.gemm(a, w, swiglu=True, out=out, kernel=kern)
        outs[kn] = out.clone()
        t = timeit(lambda: ops.gemm(a, w, swiglu=True, out=out, kernel=kern))
        line += f" {kn} {t*1e3:6.1f} us {2.0*M*N*K/t/1e9:6.0f} TF/s"
    assert torch.equal(outs["auto"], outs["1442"])
    print(line, flush=True)
