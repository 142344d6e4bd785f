"""GPU: SAM2 image encoder (Hiera trunk + FPN neck) through the HIP path vs (a) golden vectors produced by the
reference's own Hiera / FpnNeck / ImageEncoder classes (tiny config), (b) the CPU oracle at Hiera-L dimensions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden, t, rel_err  # noqa: E402
from oracle import ref_cpu as O  # noqa: E402
from ufvideo_amd import ops  # noqa: E402
from ufvideo_amd.model import sam2 as S  # noqa: E402

DEV = "cuda"
TINY = dict(embed_dim=16, num_heads=1, stages=(1, 2, 3, 1), global_att_blocks=(4,), window_spec=(8, 4, 8, 4),
            window_pos_embed_bkg_spatial_size=(7, 7))


def _sub(w, pre):
    return {k[len(pre):]: v for k, v in w.items() if k.startswith(pre)}


def test_sam2_support_kernels():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 3, 20, 24, generator=g).to(DEV)
    cols, (Ho, Wo) = ops.im2col(x, 7, 4, 3, 192)
    ref = torch.nn.functional.unfold(x, 7, padding=3, stride=4).transpose(1, 2).reshape(-1, 147)
    assert (Ho, Wo) == (5, 6) and torch.equal(cols[:, :147].float(), ref.to(torch.bfloat16).float()) and cols[:, 147:].abs().sum() == 0
    a = torch.randn(3 * 4 * 6, 10, generator=g).to(DEV)
    p = ops.maxpool2x2(a, 3, 4, 6, 10)
    refp = torch.nn.functional.max_pool2d(a.view(3, 4, 6, 10).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).reshape(-1, 10)
    assert torch.equal(p, refp)
    ab = a.to(torch.bfloat16)
    assert torch.equal(ops.maxpool2x2(ab, 3, 4, 6, 10).float(), torch.nn.functional.max_pool2d(ab.float().view(3, 4, 6, 10).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).reshape(-1, 10))
    dst = torch.randn(7, 10, generator=g).to(DEV); d0 = dst.clone()
    idx = torch.tensor([3, -1, 0, 6], device=DEV)
    src = torch.randn(4, 10, generator=g).to(DEV)
    ops.add_rows(src, dst, idx)
    exp = d0.clone(); exp[3] += src[0]; exp[0] += src[2]; exp[6] += src[3]
    assert torch.allclose(dst, exp)
    lat = torch.randn(2 * 4 * 6, 8, generator=g).to(DEV); prev = torch.randn(2 * 2 * 3, 8, generator=g).to(DEV)
    refu = lat.view(2, 4, 6, 8) + torch.nn.functional.interpolate(prev.view(2, 2, 3, 8).permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest").permute(0, 2, 3, 1)
    assert torch.allclose(ops.upsample2x_add(lat.clone(), prev, 2, 4, 6, 8).view(2, 4, 6, 8), refu)
    # gather with -1 = zero row
    out = torch.full((3, 10), 7.0, device=DEV)
    ops.gather_rows(src, torch.tensor([2, -1, 0], device=DEV), out, None)
    assert torch.equal(out[1], torch.zeros(10, device=DEV)) and torch.equal(out[0], src[2])


def test_window_index_matches_reference_partition():
    idx, (Hp, Wp) = S.window_index(2, 6, 10, 4, "cpu")
    x = torch.arange(2 * 6 * 10, dtype=torch.float32).view(2, 6, 10, 1)
    w, pad = O._window_partition(x + 1, 4)                     # +1 so that zero padding is distinguishable
    got = torch.where(idx >= 0, idx.float() + 1, torch.zeros(()))
    assert pad == (Hp, Wp) == (8, 12) and torch.equal(got, w.reshape(-1))


def test_hiera_fpn_tiny_vs_reference_golden():
    a, w = load_golden("sam2_encoder_tiny")
    trunk = S.Hiera(**TINY)
    trunk.load_state_dict(_sub(w, "trunk."))
    neck = S.FpnNeck(S.PositionEmbeddingSine(32), d_model=32, backbone_channel_list=trunk.channel_list, fpn_top_down_levels=[2, 3],
                     fpn_interp_model="nearest")
    neck.load_state_dict(_sub(w, "neck."))
    enc = S.ImageEncoder(trunk, neck, scalp=1).to(DEV)
    x = t(a["x"]).to(DEV)
    feats = enc.trunk(x)
    assert [b["window"] for b in trunk.schedule] == [b["window"] for b in O.hiera_schedule(TINY)[0]]
    for i, f in enumerate(feats):
        assert f.shape == a[f"stage{i}"].shape and rel_err(f.cpu(), t(a[f"stage{i}"])) < 3e-2, i
    out = enc(x)
    assert len(out["backbone_fpn"]) == 3
    for i, f in enumerate(out["backbone_fpn"]):
        assert rel_err(f.cpu(), t(a[f"fpn{i}"])) < 3e-2, i
    assert rel_err(out["vision_pos_enc"][0].cpu(), t(a["pos0"])) < 1e-5
    assert out["vision_features"].shape == a["fpn2"].shape


def test_hiera_l_one_frame_vs_oracle():
    """Hiera-L dims (144/288/576/1152, hd 72, windows 8/4/16/8 + 3 global blocks) on one 512x512 frame."""
    cfg = dict(embed_dim=144, num_heads=2, stages=(2, 6, 36, 4), global_att_blocks=(23, 33, 43), window_spec=(8, 4, 16, 8),
               window_pos_embed_bkg_spatial_size=(7, 7))
    sd = O.make_hiera_weights(cfg, seed=20, prefix="")
    trunk = S.Hiera(**cfg)
    trunk.load_state_dict(sd); trunk = trunk.to(DEV)
    x = torch.randn(1, 3, 512, 512, generator=torch.Generator().manual_seed(3))
    feats = trunk(x.to(DEV))
    ref = O.hiera_forward(sd, cfg, x)
    assert [tuple(f.shape) for f in feats] == [tuple(r.shape) for r in ref] == [(1, 144, 128, 128), (1, 288, 64, 64), (1, 576, 32, 32), (1, 1152, 16, 16)]
    for f, r in zip(feats, ref):
        assert rel_err(f.cpu(), r) < 4e-2
