import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ufvideo_amd.model import build_sam2_image_encoder
enc = build_sam2_image_encoder(device="cuda")
for F in (1, 4):
    x = torch.randn(F, 3, 1024, 1024, device="cuda", dtype=torch.bfloat16)
    for _ in range(2): out = enc(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): out = enc(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"Hiera-L + FPN, {F} x 1024^2: {dt*1e3:.1f} ms  ({dt/F*1e3:.1f} ms/frame, ~{1.8e12*F/dt/1e12:.0f} TF/s at 1.8 TF/frame)", [tuple(f.shape) for f in out["backbone_fpn"]])
