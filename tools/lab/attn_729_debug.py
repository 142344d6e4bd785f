import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ufvideo_amd import ops
hd, S = 72, 729
for B, H in ((2, 4), (1, 16), (32, 16)):
    torch.manual_seed(B)
    x = torch.randn(B * S, 3 * H * hd, device="cuda").to(torch.bfloat16)
    st = ((S * 3 * H * hd, 3 * H * hd),) * 3
    outs = []
    for kern in (14, 11, 0, 14, 0, 11, 14, 0):
        outs.append((kern, ops.attention(x, x[:, H * hd:], x[:, 2 * H * hd:], B, H, H, S, S, hd, *st, kernel=kern).clone()))
    ref = outs[1][1]
    for kern, o in outs:
        d = (o != ref).nonzero()
        rows = sorted(set((d[:, 0] % S).tolist()))
        print(B, H, kern, "diff elems", d.shape[0], "rows", rows[:8], rows[-4:], "heads", sorted(set((d[:, 1] // hd).tolist()))[:8], "frames", sorted(set((d[:, 0] // S).tolist()))[:8], flush=True)
