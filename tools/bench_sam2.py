"""Diagnostic timing of the SAM2 segmentation path at full size (Hiera-L + FPN + language-prompted heads), config #5b."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ufvideo_amd.model.sam2 import SAM2


def timed(fn, n=3, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


sam = SAM2(device="cuda")
base = sam.sam2_model
if "--fp8" in sys.argv:                       # config #5: W8A8 e4m3 GEMMs in the Hiera-L trunk and the FPN laterals
    from ufvideo_amd.model._params import set_gemm_dtype
    set_gemm_dtype(sam, "fp8")
    print("fp8 trunk")
for F in (1, 4, 8):
    x = torch.randn(F, 3, 1024, 1024, device="cuda", dtype=torch.bfloat16)
    dt, feats = timed(lambda: base.forward_image_tokens(x))
    print(f"Hiera-L + FPN, {F} x 1024^2: {dt*1e3:.1f} ms ({dt/F*1e3:.2f} ms/frame, ~{1.8*F/dt:.0f} TF/s at 1.8 TF/frame)")
    lang = torch.randn(F, 1, 256, device="cuda")
    dh, out = timed(lambda: base.forward_sam_heads_tokens(feats, F, lang))
    print(f"  SAM heads (prompt tokens -> two-way transformer -> masks 256^2 -> 1024^2): {dh*1e3:.2f} ms ({dh/F*1e3:.2f} ms/frame)",
          tuple(out["high_res_masks"].shape))
    state = sam.get_sam2_embeddings(x)
    emb = [lang[f] for f in range(F)]
    t0 = time.perf_counter(); m = sam.language_embd_inference(state, emb); torch.cuda.synchronize(); cold = time.perf_counter() - t0
    t0 = time.perf_counter(); m = sam.language_embd_inference(state, emb); torch.cuda.synchronize(); warm = time.perf_counter() - t0
    print(f"  language_embd_inference: first [SEG] {cold*1e3:.1f} ms, further [SEG]s on the cached features {warm*1e3:.1f} ms", tuple(m.shape))
