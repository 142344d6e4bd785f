"""lab: the attention calls of one Hiera-L forward (8 x 1024^2) with shapes and times"""
import sys, os, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufvideo_amd import ops
from ufvideo_amd.model.sam2 import SAM2
sam = SAM2(device="cuda")
base = sam.sam2_model
x = torch.randn(8, 3, 1024, 1024, device="cuda", dtype=torch.bfloat16)
base.forward_image_tokens(x); torch.cuda.synchronize()
log = collections.OrderedDict()
orig = ops.attention
def attention(q, k, v, B, Hq, Hkv, Sq, Sk, hd, qs, ks, vs, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(q, k, v, B, Hq, Hkv, Sq, Sk, hd, qs, ks, vs, **kw); e1.record()
    key = (B, Hq, Hkv, Sq, Sk, hd, qs, ks, vs, q.data_ptr() % 16, k.data_ptr() % 16, v.data_ptr() % 16)
    log.setdefault(key, []).append((e0, e1))
    return out
ops.attention = attention
import ufvideo_amd.model.sam2 as S2
base.forward_image_tokens(x); torch.cuda.synchronize()
for k, ev in sorted(log.items(), key=lambda kv: -sum(a.elapsed_time(b) for a, b in kv[1])):
    us = [a.elapsed_time(b) * 1e3 for a, b in ev]
    print(f"B {k[0]:6d} H {k[1]}/{k[2]} Sq {k[3]:5d} Sk {k[4]:5d} hd {k[5]} strides {k[6]} {k[7]} {k[8]} align {k[9:]}: {len(us)} calls, {sum(us)/len(us):8.1f} us each, {sum(us)/1e3:.2f} ms")
