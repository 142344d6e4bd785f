import os, sys, torch
sys.path.insert(0, os.getcwd())
from ufvideo_amd import ops
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for dt in (torch.bfloat16, torch.float32):
    x = torch.randn(32, 3, 336, 336, device="cuda").to(dt)
    y = ops.patchify(x, 14, 640)
    ref = x.to(torch.bfloat16).view(32, 3, 24, 14, 24, 14).permute(0, 2, 4, 1, 3, 5).reshape(32 * 576, 588)
    print(dt, "exact", torch.equal(y[:, :588], ref), "pad zero", float(y[:, 588:].abs().max()), "%.1f us" % timed(lambda: ops.patchify(x, 14, 640)))
