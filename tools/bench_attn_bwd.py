"""Time the two attention-backward paths (materialised per-group pipeline vs fused flash-style kernels) at the decoder's shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ufvideo_amd import ops

S, H, KV, hd = int(os.environ.get("S", 2399)), 28, 4, 128
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(S, H * hd, device=dev, generator=g).bfloat16()
Sp = ops.round_up(S, 128)
kv = torch.zeros(Sp, 2 * KV * hd, device=dev, dtype=torch.bfloat16)
kv[:S] = torch.randn(S, 2 * KV * hd, device=dev, generator=g).bfloat16()
dO = torch.randn(S, H * hd, device=dev, generator=g).bfloat16()
o = torch.empty(S, H * hd, device=dev, dtype=torch.bfloat16)
lse = torch.empty(H, S, device=dev, dtype=torch.float32)
d1 = torch.zeros(S, (H + 2 * KV) * hd, device=dev, dtype=torch.bfloat16)
d2 = torch.zeros_like(d1)


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

fwd = lambda: ops.attention_causal_lse(q, kv, kv[:, KV * hd:], o, lse, S, H, KV, hd)
mat = lambda: ops.attention_bwd(q, kv, kv[:, KV * hd:], dO, d1, d1[:, H * hd:], d1[:, (H + KV) * hd:], S, H, KV, hd)
fus = lambda: ops.attention_bwd_fused(q, kv, kv[:, KV * hd:], o, dO, lse, d2, d2[:, H * hd:], d2[:, (H + KV) * hd:], S, H, KV, hd)
t_f, t_m, t_u = timeit(fwd), timeit(mat), timeit(fus)
fl = 2.0 * S * S / 2 * hd * H
print(f"S={S} fwd {t_f:.3f} ms ({2 * fl / t_f / 1e9:.0f} TF/s causal)  materialised bwd {t_m:.3f} ms  fused bwd {t_u:.3f} ms "
      f"({5 * fl / t_u / 1e9:.0f} TF/s causal, 5 products)")
e = (d1.float() - d2.float()).norm() / d1.float().norm()
print("rel diff fused vs materialised", float(e))
