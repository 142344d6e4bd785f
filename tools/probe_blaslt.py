"""Calibration only (not product): what the vendor library reaches on the same NT shapes, bf16 out, random data."""
import torch

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

shapes = {"vit_qkv": (18432, 3456, 1152), "vit_o": (18432, 1152, 1152), "vit_fc1": (18432, 4352, 1152), "vit_fc2": (18432, 1152, 4352),
          "llm_qkv": (2399, 4608, 3584), "llm_o": (2399, 3584, 3584), "llm_gu": (2399, 37888, 3584), "llm_down": (2399, 3584, 18944)}
for name, (M, N, K) in shapes.items():
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ms = timeit(lambda: torch.mm(a, w.t(), out=out))
    print(f"{name:9s} hipBLASLt/rocBLAS via torch.mm: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
