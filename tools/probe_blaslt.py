"""Calibration only (not product): the vendor library (hipBLASLt / rocBLAS through torch.mm, bf16 out, no epilogue) against ufv_gemm AUTO on the
eight config-#2 NT shapes, same process, same box, random data, alternating runs.  ufv runs the PRODUCT epilogue of each shape (bf16 out for
qkv / fc1, SwiGLU for gate/up, fp32 output + in-place residual for o / fc2 / down), i.e. more work than the vendor call it is compared with.
Writes gpurun_out/<tag>/gemm_vs_vendor.json (tag = argv[1], default r03)."""
import json, os, sys
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


shapes = {"vit_qkv": (18432, 3456, 1152, "bf16"), "vit_o": (18432, 1152, 1152, "res"), "vit_fc1": (18432, 4352, 1152, "bf16"), "vit_fc2": (18432, 1152, 4352, "res"),
          "llm_qkv": (2399, 4608, 3584, "bf16"), "llm_o": (2399, 3584, 3584, "res"), "llm_gu": (2399, 37888, 3584, "swiglu"), "llm_down": (2399, 3584, 18944, "res")}
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
rows = {}
for name, (M, N, K, kind) in shapes.items():
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(M, N, device="cuda") if kind == "res" else None
    if kind == "bf16":
        ours = lambda: ops.gemm(a, w, out=out)
    elif kind == "swiglu":
        o2 = torch.empty(M, N // 2, device="cuda", dtype=torch.bfloat16)
        ours = lambda: ops.gemm(a, w, out=o2, swiglu=True)
    else:
        ours = lambda: ops.gemm(a, w, resid=res, out=res)
    vend = lambda: torch.mm(a, w.t(), out=out)
    tv, to = [], []
    for _ in range(3):
        tv.append(timeit(vend)); to.append(timeit(ours))
    fl = 2.0 * M * N * K
    v, o = min(tv), min(to)
    rows[name] = {"M": M, "N": N, "K": K, "epilogue_ours": kind, "vendor_us": round(v * 1e3, 1), "ours_us": round(o * 1e3, 1),
                  "vendor_tflops": round(fl / v / 1e9, 1), "ours_tflops": round(fl / o / 1e9, 1), "ours_over_vendor": round(v / o, 3)}
    print(f"{name:9s} vendor {v*1e3:7.1f} us {fl/v/1e9:7.1f} TF/s | ufv AUTO ({kind:6s}) {o*1e3:7.1f} us {fl/o/1e9:7.1f} TF/s | x{v/o:.3f}", flush=True)
outdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", tag)
os.makedirs(outdir, exist_ok=True)
json.dump(rows, open(os.path.join(outdir, "gemm_vs_vendor.json"), "w"), indent=1)
