"""ViT attention launch time at the two SigLIP so400m sequence lengths (576 = 336 px, 729 = 384 px): the generated kernel (id 14), the second generation (id 11),
32 frames x 16 heads, q / k / v as column views of the fused projection output (the tower's layout), rotating buffers.  usage: python tools/lab/attn_729_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ufvideo_amd import ops

T, H, hd = 32, 16, 72
res = {}
for S in (576, 729):
    bufs = [torch.randn(T * S, 3 * H * hd, device="cuda").to(torch.bfloat16) for _ in range(4)]
    o = torch.empty(T * S, H * hd, device="cuda", dtype=torch.bfloat16)
    st = (S * 3 * H * hd, 3 * H * hd)
    for kern in (14, 11):
        def run(i):
            x = bufs[i % 4]
            ops.attention(x, x[:, H * hd:], x[:, 2 * H * hd:], T, H, H, S, S, hd, st, st, st, out=o, kernel=kern)
        for i in range(10):
            run(i)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(50):
                run(i)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
        gf = 4.0 * S * S * hd * T * H / 1e9
        res[(S, kern)] = best
        print(f"S={S} kernel={kern}: {best:.1f} us  {gf / best * 1e3:.0f} TF/s useful ({gf / best * 1e3 / 2500:.3f} of 2.5 PF)", flush=True)
print(f"ratio 729/576 generated kernel: {res[(729, 14)] / res[(576, 14)]:.3f} (work ratio {(729 / 576) ** 2:.3f})")
