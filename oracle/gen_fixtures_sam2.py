"""Golden vectors for the SAM2 image encoder (Hiera trunk + FPN neck), produced by RUNNING the reference's own
classes in the build container.  Test infrastructure only.  Usage: python oracle/gen_fixtures_sam2.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import refshim, ref_cpu as O  # noqa: E402

refshim.install()
import ufvideo.model.sam2 as RS  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
TINY = dict(embed_dim=16, num_heads=1, stages=(1, 2, 3, 1), global_att_blocks=(4,), window_spec=(8, 4, 8, 4),
            window_pos_embed_bkg_spatial_size=(7, 7))


def close(a, b, tol=3e-5, what=""):
    err = (a.float() - b.float()).abs().max().item() / (b.float().abs().max().item() + 1e-12)
    assert err < tol, f"{what}: oracle vs reference {err}"
    return err


def main():
    torch.manual_seed(31)
    torch.set_grad_enabled(False)
    trunk = RS.Hiera(**TINY).eval()
    neck = RS.FpnNeck(position_encoding=RS.PositionEmbeddingSine(num_pos_feats=32), d_model=32,
                      backbone_channel_list=trunk.channel_list, fpn_top_down_levels=[2, 3], fpn_interp_model="nearest").eval()
    enc = RS.ImageEncoder(trunk=trunk, neck=neck, scalp=1).eval()
    for p_ in enc.parameters():
        if p_.ndim == 1 or p_.shape[0] == 1:
            p_.add_(torch.randn_like(p_) * 0.05)
    x = torch.randn(2, 3, 128, 128)
    outs = trunk(x)
    sd = dict(enc.state_dict())
    o = O.hiera_forward(sd, TINY, x, prefix="trunk.")
    for a, b in zip(o, outs):
        print("  hiera stage rel err", close(a, b, what="hiera stage"), tuple(b.shape))
    full = enc(x)
    cfgo = dict(TINY, d_model=32)
    fo = O.sam2_image_encoder(sd, cfgo, x)
    for a, b in zip(fo["backbone_fpn"], full["backbone_fpn"]):
        print("  fpn rel err", close(a, b, what="fpn"), tuple(b.shape))
    for a, b in zip(fo["vision_pos_enc"], full["vision_pos_enc"]):
        close(a, b, what="pos enc")
    blocks, ends = O.hiera_schedule(TINY)
    assert [b["window"] for b in blocks] == [blk.window_size for blk in trunk.blocks]
    assert [b["heads"] for b in blocks] == [blk.attn.num_heads for blk in trunk.blocks]
    arrs = {"x": x.numpy(), "pos0": full["vision_pos_enc"][0].numpy()}
    for i, t_ in enumerate(outs):
        arrs[f"stage{i}"] = t_.numpy()
    for i, t_ in enumerate(full["backbone_fpn"]):
        arrs[f"fpn{i}"] = t_.numpy()
    for k, v in sd.items():
        arrs["w::" + k] = v.float().numpy()
    np.savez_compressed(os.path.join(OUT, "sam2_encoder_tiny.npz"), **arrs)
    print("wrote sam2_encoder_tiny.npz", os.path.getsize(os.path.join(OUT, "sam2_encoder_tiny.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
