"""lab: every ufv_gemm call of one bench step with its shape and time (an event pair around each call), grouped by shape"""
import sys, os, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ufvideo_amd import ops
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model = bench.build_model(dev, 32)
video, ids, am = bench.synthetic_inputs(dev, 32)
from ufvideo_amd.model import KVCache
cfg = model.config
cache = KVCache(cfg.num_hidden_layers, 2304 + 128, 2 * cfg.num_key_value_heads * cfg.head_dim, dev)
for _ in range(2):
    bench.one_step(model, video, ids, am, cache)
torch.cuda.synchronize()
log = collections.OrderedDict()
orig = ops.gemm
def gemm(a, w, *args, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(a, w, *args, **kw); e1.record()
    M, K = a.shape; N = w.shape[0]
    key = (M, N, K, "f32" if out.dtype == torch.float32 else "bf16", "res" if kw.get("resid") is not None else "", "swiglu" if kw.get("swiglu") else "", kw.get("act") or "")
    log.setdefault(key, []).append((e0, e1))
    return out
ops.gemm = gemm
import ufvideo_amd.model.projector as P, ufvideo_amd.model.encoder as E, ufvideo_amd.model.videorefer_qwen2 as Q
bench.one_step(model, video, ids, am, cache)
torch.cuda.synchronize()
tot = 0
print("   M      N      K   out  flags            calls   us/call   TF/s   ms/step")
for k, ev in sorted(log.items(), key=lambda kv: -sum(a.elapsed_time(b) for a, b in kv[1])):
    us = [a.elapsed_time(b) * 1e3 for a, b in ev]
    M, N, K = k[:3]
    ms = sum(us) / 1e3; tot += ms
    print(f"{M:6d} {N:6d} {K:6d}  {k[3]:4s} {' '.join(x for x in k[4:] if x):14s} {len(us):5d} {sum(us)/len(us):9.1f} {2*M*N*K/(sum(us)/len(us))/1e6:7.0f} {ms:8.3f}")
print("total ms", tot)
