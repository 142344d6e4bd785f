"""lab: kernel-time vs wall-time of the Hiera-L + FPN trunk at 8 x 1024^2 (run under rocprofv3 --kernel-trace --stats)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ufvideo_amd.model.sam2 import SAM2
sam = SAM2(device="cuda")
base = sam.sam2_model
F = 8
x = torch.randn(F, 3, 1024, 1024, device="cuda", dtype=torch.bfloat16)
for _ in range(2): base.forward_image_tokens(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 5
for _ in range(N): base.forward_image_tokens(x)
torch.cuda.synchronize()
print(f"wall {(time.perf_counter() - t0) / N * 1e3:.2f} ms per call of {F} frames")
