"""Generate tests/golden/geom_odd.npz by RUNNING THE REFERENCE in the build container: the two properties of the released checkpoint's tower geometry
(siglip-so400m-patch14-384: 384 = 27 x 14 + 6) at tiny dimensions --
  * an image size that is NO multiple of the patch: HF's stride-14 convolution without padding drops the remainder (76 = 5 x 14 + 6 -> 5 x 5 = 25 tokens; the same remainder
    as 384), through the reference's own `SiglipVisionTower` (ufvideo/model/encoder.py:96-146);
  * an ODD token grid through the STC-v35 sampler (Conv3d kernel = stride = (2, 2, 2), padding 0: 5 -> 2, as 27 -> 13) and its readout, through the reference's own
    `STCConnectorV35(depth=0)` (ufvideo/model/projector.py:189-238; depth 0: timm's RegStage is absent offline and stays unpinned).
Test infrastructure only.  Usage (build container, /root/reference present):  python oracle/gen_fixtures_geom.py
The oracle is asserted equal to the reference while the fixture is written; tests/test_oracle_golden.py re-checks it on the CPU, tests/test_model_gpu.py runs the HIP path against it."""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import refshim  # noqa: E402
from oracle import ref_cpu as O  # noqa: E402

refshim.install()
import transformers  # noqa: E402
from transformers import SiglipVisionConfig, SiglipVisionModel  # noqa: E402
sys.modules["transformers"].TRANSFORMERS_CACHE = "/tmp/_no_cache"
import ufvideo.model.projector as RP  # noqa: E402
import ufvideo.model.encoder as RE  # noqa: E402

VIT = dict(hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=2, image_size=76, patch_size=14)


def close(a, b, tol=2e-5, what=""):
    err = (a.float() - b.float()).abs().max().item()
    ref = b.float().abs().max().item() + 1e-12
    assert err / ref < tol, f"{what}: oracle vs reference mismatch {err} / {ref}"
    return err / ref


def main():
    torch.set_grad_enabled(False)
    work = tempfile.mkdtemp(prefix="ufv_fx_geom_")
    os.chdir(work)                                   # the reference hard-codes a cwd-relative tower path (encoder.py:108)
    print("transformers", transformers.__version__, "torch", torch.__version__, "cwd", work)
    torch.manual_seed(5)
    m = SiglipVisionModel(SiglipVisionConfig(**VIT)).eval()
    for p_ in m.parameters():
        if p_.ndim == 1:
            p_.add_(torch.randn_like(p_) * 0.05)
    path = os.path.join(work, "siglip-so400m-patch14-384")
    m.save_pretrained(path)
    with open(os.path.join(path, "preprocessor_config.json"), "w") as f:
        json.dump({"image_processor_type": "SiglipImageProcessor", "size": {"height": 76, "width": 76}, "image_mean": [0.5, 0.5, 0.5], "image_std": [0.5, 0.5, 0.5],
                   "do_resize": True, "do_rescale": True, "do_normalize": True, "resample": 3, "rescale_factor": 1 / 255}, f)

    class A:
        mm_vision_select_layer = -2
        mm_vision_select_feature = "patch"
    tower = RE.SiglipVisionTower("siglip", A(), delay_load=False)
    x = torch.randn(4, 3, 76, 76)
    y = tower(x)                                     # [4, 25, 32]: the 6 remainder pixels of every row / column belong to no patch
    assert tuple(y.shape) == (4, 25, 32), y.shape
    x_cut = x.clone(); x_cut[:, :, 70:, :] = 7.0; x_cut[:, :, :, 70:] = -7.0
    assert torch.equal(tower(x_cut), y)              # ... and do not reach the output at all
    sd = dict(tower.vision_tower.state_dict())
    pre = "vision_model." if any(k.startswith("vision_model.") for k in sd) else ""
    print("   tower 76 px rel err", close(O.siglip_tower(sd, VIT, x, prefix=pre, select_layer=-2), y, what="siglip tower at 76 px"))

    class Cfg:
        mm_hidden_size = 32
        hidden_size = 32
    torch.manual_seed(6)
    proj = RP.STCConnectorV35(Cfg(), depth=0).eval()
    z = proj(y[None])                                # [1, (4 / 2) * 2 * 2, 32]: the odd 5 x 5 grid floors to 2 x 2
    assert tuple(z.shape) == (1, 8, 32), z.shape
    psd = dict(proj.state_dict())
    print("   v35(depth 0) on the 5 x 5 grid rel err", close(O.stc_connector(psd, y[None], downsample=(2, 2, 2), padding=0, depth=0), z, what="v35 on an odd grid"))
    arrs = dict(x=x, y=y, z=z, prefix=np.frombuffer(pre.encode(), dtype=np.uint8))
    arrs.update({"w::" + k: v.detach().float().cpu().numpy() for k, v in sd.items()})
    arrs.update({"w::proj." + k: v.detach().float().cpu().numpy() for k, v in psd.items()})
    conv = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in arrs.items()}
    out = os.path.join(OUT, "geom_odd.npz")
    np.savez_compressed(out, **conv)
    print(f"  wrote geom_odd.npz ({os.path.getsize(out) / 1024:.1f} KiB)")


if __name__ == "__main__":
    main()
