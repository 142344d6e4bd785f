"""Lab: the connector's elementwise kernels (LayerNorm + SiLU, depthwise 3x3 + LayerNorm + SiLU, LayerNorm + add + SiLU) at the config-#2 shapes:
timing and a digest of the output bytes.  Run twice -- plain (round-4 kernels) and with UFV_OLD_ELEMENTWISE=1 (round-3 kernels) -- and compare the digests:
the new forms keep every summation order, so the outputs must be bit-identical.  usage: python tools/lab/ab_elementwise.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ufvideo_amd import ops


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def digest(t):
    return hashlib.sha256(t.contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:16]


tag = "old" if os.environ.get("UFV_OLD_ELEMENTWISE") else "new"
g = torch.Generator(device="cuda").manual_seed(0)
for (F, H, W, C) in ((32, 24, 24, 3584), (16, 12, 12, 3584), (3, 5, 7, 264)):
    M = F * H * W
    x = torch.randn(M, C, device="cuda", generator=g).to(torch.bfloat16)
    x2 = torch.randn(M, C, device="cuda", generator=g).to(torch.bfloat16)
    w = 1 + 0.1 * torch.randn(C, device="cuda", generator=g); b = 0.1 * torch.randn(C, device="cuda", generator=g)
    w2 = 1 + 0.1 * torch.randn(C, device="cuda", generator=g); b2 = 0.1 * torch.randn(C, device="cuda", generator=g)
    w9 = (0.2 * torch.randn(9, C, device="cuda", generator=g)).to(torch.bfloat16).float().contiguous()
    y1 = ops.layernorm(x, w, b, 1e-5, act="silu")
    y2 = ops.dwconv3x3_ln_silu(x, w9, w, b, F, H, W, C, 1e-5)
    y3 = ops.ln_add_silu(x, w, b, x2, w2, b2, 1e-5)
    y4 = ops.ln_add_silu(x, w, b, x2, None, None, 1e-5)
    t1 = timeit(lambda: ops.layernorm(x, w, b, 1e-5, act="silu"))
    t2 = timeit(lambda: ops.dwconv3x3_ln_silu(x, w9, w, b, F, H, W, C, 1e-5))
    t3 = timeit(lambda: ops.ln_add_silu(x, w, b, x2, w2, b2, 1e-5))
    t4 = timeit(lambda: ops.ln_add_silu(x, w, b, x2, None, None, 1e-5))
    mb = M * C * 2 / 1e6
    print(f"{tag} F={F} H={H} W={W} C={C}: ln_silu {t1:7.1f} us ({2 * mb / t1 / 1e6 * 1e6 / 1e3:5.2f} TB/s) {digest(y1)} | dwconv {t2:7.1f} us ({2 * mb / t2:5.2f} TB/s) {digest(y2)} | "
          f"ln_add_silu(ds) {t3:7.1f} us ({3 * mb / t3:5.2f} TB/s) {digest(y3)} | ln_add_silu(id) {t4:7.1f} us ({3 * mb / t4:5.2f} TB/s) {digest(y4)}", flush=True)
