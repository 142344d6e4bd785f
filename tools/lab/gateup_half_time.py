"""gate/up GEMM (SwiGLU epilogue) with and without the last round of half-tile items (csrc/gemm256_kernel.h HALF; UFV_GEMM_HALF=1 is read per call): M = 2399 / 2799 / 4703,
28 distinct weight matrices in rotation as in the decoder.  usage: python tools/lab/gateup_half_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ufvideo_amd import ops
N2, K = 37888, 3584
ws = [(torch.randn(N2, K, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(8)]
for M in [int(x) for x in (sys.argv[1:] or "2399 2799 4703 1200".split())]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N2 // 2, device="cuda", dtype=torch.bfloat16)
    res = {}
    for rep in range(2):
        for mode in ("half", "whole"):
            if mode == "half":
                os.environ["UFV_GEMM_HALF"] = "1"
            else:
                os.environ.pop("UFV_GEMM_HALF", None)
            for i in range(8):
                ops.gemm(a, ws[i % 8], swiglu=True, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(56):
                ops.gemm(a, ws[i % 8], swiglu=True, out=out)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) / 56 * 1e3)
    os.environ.pop("UFV_GEMM_HALF", None)
    fl = 2.0 * M * N2 * K
    h, w = min(res["half"]), min(res["whole"])
    print(f"M={M}: half-item round {h:.1f} us ({fl / h / 1e6:.0f} TF/s, {fl / h / 1e6 / 2500:.3f})   whole tiles {w:.1f} us ({fl / w / 1e6:.0f} TF/s, {fl / w / 1e6 / 2500:.3f})   {100 * (h / w - 1):+.1f} %", flush=True)
