"""lab: the GEMM calls of one Hiera-L + FPN forward (8 x 1024^2) with shapes, times and rates"""
import sys, os, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufvideo_amd import ops
from ufvideo_amd.model.sam2 import SAM2
sam = SAM2(device="cuda")
base = sam.sam2_model
x = torch.randn(8, 3, 1024, 1024, device="cuda", dtype=torch.bfloat16)
base.forward_image_tokens(x); torch.cuda.synchronize()
log = collections.OrderedDict()
orig = ops.gemm
def gemm(a, w, *args, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(a, w, *args, **kw); e1.record()
    M, K = a.shape; N = w.shape[0]
    key = (M, N, K, "f32" if out.dtype == torch.float32 else "bf16", "res" if kw.get("resid") is not None else "", kw.get("act") or "", "bias" if kw.get("bias") is not None else "")
    log.setdefault(key, []).append((e0, e1))
    return out
ops.gemm = gemm
base.forward_image_tokens(x); torch.cuda.synchronize()
tot = 0
print("     M      N      K  out  flags             calls   us/call   TF/s   ms")
for k, ev in sorted(log.items(), key=lambda kv: -sum(a.elapsed_time(b) for a, b in kv[1])):
    us = [a.elapsed_time(b) * 1e3 for a, b in ev]
    M, N, K = k[:3]
    ms = sum(us) / 1e3; tot += ms
    print(f"{M:7d} {N:6d} {K:6d} {k[3]:4s} {' '.join(x for x in k[4:] if x):16s} {len(us):5d} {sum(us)/len(us):9.1f} {2*M*N*K/(sum(us)/len(us))/1e6:7.0f} {ms:7.2f}")
print("total ms", round(tot, 2))
