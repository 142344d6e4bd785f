"""Diagnostic micro-benchmarks of the hot kernels at the BASELINE config-#2 shapes (not bench.py)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ufvideo_amd import ops

dev = "cuda"

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(torch.bfloat16)

res = {}
shapes = {  # name: (M, N, K, swiglu)
    "vit_qkv": (18432, 3456, 1152, 0), "vit_o": (18432, 1152, 1152, 0), "vit_fc1": (18432, 4352, 1152, 0),
    "vit_fc2": (18432, 1152, 4352, 0), "proj_c1": (18432, 3584, 1152, 0), "proj_c3": (18432, 3584, 3584, 0),
    "proj_c3d": (2304, 3584, 28672, 0), "llm_qkv": (2399, 4608, 3584, 0), "llm_o": (2399, 3584, 3584, 0),
    "llm_gu": (2399, 37888, 3584, 1), "llm_down": (2399, 3584, 18944, 0),
}
only = sys.argv[1:] 
for name, (M, N, K, sw) in shapes.items():
    if only and name not in only and "gemm" not in only: continue
    a, w = rnd(M, K), rnd(N, K, scale=0.02)
    out = torch.empty(M, N // 2 if sw else N, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: ops.gemm(a, w, out=out, swiglu=bool(sw), kernel=ops.GEMM_FAST))
    ms2 = timeit(lambda: ops.gemm(a, w, out=out, swiglu=bool(sw), kernel=ops.GEMM_FAST256))
    ms3 = timeit(lambda: ops.gemm(a, w, out=out, swiglu=bool(sw)))
    tf = 2.0 * M * N * K / ms / 1e9; tf2 = 2.0 * M * N * K / ms2 / 1e9
    res[name] = dict(ms=round(ms, 4), tflops=round(tf, 1), ms256=round(ms2, 4), tflops256=round(tf2, 1))
    print(f"{name:10s} M={M:6d} N={N:6d} K={K:6d}  k128 {ms:8.3f} ms {tf:7.1f} TF/s | k256 {ms2:8.3f} ms {tf2:7.1f} TF/s | auto {ms3:8.3f} ms", flush=True)

if not only or "attn" in only:
    # ViT attention: 32 frames x 16 heads x 576 x 72
    T, H, S, hd = 32, 16, 576, 72
    qkv = rnd(T * S, 3 * H * hd)
    o = torch.empty(T * S, H * hd, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.attention(qkv, qkv[:, H * hd:], qkv[:, 2 * H * hd:], T, H, H, S, S, hd, (S * 3 * H * hd, 3 * H * hd),
                              (S * 3 * H * hd, 3 * H * hd), (S * 3 * H * hd, 3 * H * hd), out=o, kernel=1)
    ms = timeit(f); tf = 4.0 * T * H * S * S * hd / ms / 1e9
    res["vit_attn"] = dict(ms=round(ms, 4), tflops=round(tf, 1)); print(f"vit_attn  {ms:8.3f} ms  {tf:7.1f} TF/s (useful)")
    f3 = lambda: ops.attention(qkv, qkv[:, H * hd:], qkv[:, 2 * H * hd:], T, H, H, S, S, hd, (S * 3 * H * hd, 3 * H * hd),
                               (S * 3 * H * hd, 3 * H * hd), (S * 3 * H * hd, 3 * H * hd), out=o, kernel=4)
    ms = timeit(f3); print(f"vit_attn (6-wave dma variant) {ms:8.3f} ms  {4.0 * T * H * S * S * hd / ms / 1e9:7.1f} TF/s")
    f6 = lambda: ops.attention(qkv, qkv[:, H * hd:], qkv[:, 2 * H * hd:], T, H, H, S, S, hd, (S * 3 * H * hd, 3 * H * hd),
                               (S * 3 * H * hd, 3 * H * hd), (S * 3 * H * hd, 3 * H * hd), out=o, kernel=6)
    ms = timeit(f6); print(f"vit_attn (9-wave lockstep, no ping-pong) {ms:8.3f} ms  {4.0 * T * H * S * S * hd / ms / 1e9:7.1f} TF/s")
    S, Hq, Hkv, hd = 2399, 28, 4, 128
    q = rnd(S, Hq * hd); kv = rnd(S, 2 * Hkv * hd)
    o = torch.empty(S, Hq * hd, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.attention(q, kv, kv[:, Hkv * hd:], 1, Hq, Hkv, S, S, hd, (0, Hq * hd), (0, 2 * Hkv * hd), (0, 2 * Hkv * hd),
                              causal=True, out=o, kernel=1)
    ms = timeit(f); tf = 2.0 * Hq * S * S * hd / ms / 1e9
    res["llm_attn"] = dict(ms=round(ms, 4), tflops=round(tf, 1)); print(f"llm_attn  {ms:8.3f} ms  {tf:7.1f} TF/s (causal useful)")

if not only or "mem" in only:
    x = torch.randn(18432, 1152, device=dev); w = torch.ones(1152, device=dev); b = torch.zeros(1152, device=dev)
    ms = timeit(lambda: ops.layernorm(x, w, b, 1e-6)); print(f"layernorm 18432x1152 f32->bf16 {ms*1e3:7.1f} us  {(x.numel()*6)/ms/1e6:7.1f} GB/s")
    xb = rnd(18432, 3584); w = torch.ones(3584, device=dev); b = torch.zeros(3584, device=dev)
    ms = timeit(lambda: ops.layernorm(xb, w, b, 1e-5, act="silu")); print(f"ln+silu 18432x3584 bf16 {ms*1e3:7.1f} us  {(xb.numel()*4)/ms/1e6:7.1f} GB/s")
    w9 = torch.randn(9, 3584, device=dev)
    ms = timeit(lambda: ops.dwconv3x3_ln_silu(xb, w9, w, b, 32, 24, 24, 3584, 1e-5)); print(f"dwconv+ln+silu {ms*1e3:7.1f} us  {(xb.numel()*4)/ms/1e6:7.1f} GB/s(min)")
    ms = timeit(lambda: ops.colmean(xb, 32, 576)); print(f"colmean {ms*1e3:7.1f} us")
    gate = rnd(32, 3584)
    ms = timeit(lambda: ops.scale_channels(xb, gate, 32, 576)); print(f"scale_channels {ms*1e3:7.1f} us  {(xb.numel()*4)/ms/1e6:7.1f} GB/s")
    ms = timeit(lambda: ops.ln_add_silu(xb, w, b, xb, None, None, 1e-5)); print(f"ln_add_silu {ms*1e3:7.1f} us  {(xb.numel()*6)/ms/1e6:7.1f} GB/s")
    xs = rnd(32, 24, 24, 3584)
    ms = timeit(lambda: ops.conv3d_gather(xs, 32, 24, 24, 3584, (2, 2, 2), 0)); print(f"conv3d_gather {ms*1e3:7.1f} us")
    # decode-like gemv
    a1 = rnd(1, 3584); wv = rnd(37888, 3584, scale=0.02)
    ms = timeit(lambda: ops.gemm(a1, wv, swiglu=True, kernel=ops.GEMM_GEMV)); print(f"gemv gate/up {ms*1e3:7.1f} us  {wv.numel()*2/ms/1e6:7.1f} GB/s")
    wl = rnd(151748, 3584, scale=0.02)
    ms = timeit(lambda: ops.gemm(a1, wl, out_dtype=torch.float32, kernel=ops.GEMM_GEMV)); print(f"gemv lm_head {ms*1e3:7.1f} us  {wl.numel()*2/ms/1e6:7.1f} GB/s")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bench_kernels.json", "w"), indent=1)
