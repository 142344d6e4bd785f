import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ufvideo_amd import ops
T, H, S, hd = 32, 16, 576, 72
qkv = (torch.randn(T * S, 3 * H * hd, device="cuda")).to(torch.bfloat16)
o = torch.empty(T * S, H * hd, device="cuda", dtype=torch.bfloat16)
st = (S * 3 * H * hd, 3 * H * hd)
for _ in range(5):
    ops.attention(qkv, qkv[:, H * hd:], qkv[:, 2 * H * hd:], T, H, H, S, S, hd, st, st, st, out=o, kernel=1)
torch.cuda.synchronize()
