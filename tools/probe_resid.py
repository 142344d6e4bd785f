"""Diagnostic: fp32-residual GEMM epilogues vs plain bf16 outputs, and ViT attention efficiency vs sequence length."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ufvideo_amd import ops


def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device="cuda") * scale).to(torch.bfloat16)


dev = "cuda"
shapes = {"vit_o": (18432, 1152, 1152), "vit_fc2": (18432, 1152, 4352), "llm_o": (2399, 3584, 3584), "llm_down": (2399, 3584, 18944)}
for name, (M, N, K) in shapes.items():
    a, w = rnd(M, K), rnd(N, K, scale=0.02)
    bias = torch.zeros(N, device=dev)
    res = torch.randn(M, N, device=dev)
    ob = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for kn, kk in (("k128", ops.GEMM_FAST), ("k256", ops.GEMM_FAST256)):
        ms_r = timeit(lambda: ops.gemm(a, w, bias=bias, resid=res, out=res, kernel=kk))
        ms_b = timeit(lambda: ops.gemm(a, w, bias=bias, out=ob, kernel=kk))
        fl = 2.0 * M * N * K / 1e9
        print(f"{name:9s} {kn}: resid-f32 {ms_r*1e3:7.1f} us {fl/ms_r:7.1f} TF/s | bf16-out {ms_b*1e3:7.1f} us {fl/ms_b:7.1f} TF/s", flush=True)

H, hd = 16, 72
for S, T in ((576, 32), (1152, 16), (2304, 8), (4608, 4)):
    qkv = rnd(T * S, 3 * H * hd)
    o = torch.empty(T * S, H * hd, device=dev, dtype=torch.bfloat16)
    st = (S * 3 * H * hd, 3 * H * hd)
    for kern in (1, 4, 6):
        try:
            ms = timeit(lambda: ops.attention(qkv, qkv[:, H * hd:], qkv[:, 2 * H * hd:], T, H, H, S, S, hd, st, st, st, out=o, kernel=kern))
            print(f"attn hd72 S={S:5d} T={T:3d} kernel={kern}: {ms*1e3:8.1f} us  {4.0*T*H*S*S*hd/ms/1e9:7.1f} TF/s useful", flush=True)
        except Exception as ex:
            print("attn", S, kern, "failed", ex)
