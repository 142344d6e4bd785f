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
    # runs of blocks with one window size keep the token stream in window order (no per-block partition / un-partition): every token goes through the same
    # arithmetic, so the features must be BIT-identical to the per-block form, here with 2 frames and a grid (48 x 40 at stage 1) whose windows still divide it
    x2 = torch.randn(2, 3, 384, 320, generator=torch.Generator().manual_seed(4)).to(DEV)
    kept = trunk(x2)
    trunk.keep_window_order = False
    per_block = trunk(x2)
    trunk.keep_window_order = True
    for a_, b_ in zip(kept, per_block):
        assert torch.equal(a_, b_)


# ---- SAM heads with the language token -----------------------------------------------------------------------------------
HEAD_SEEDS = dict(trunk=40, neck=41, heads=42)


def _tiny_sam2(image_size=128):
    trunk = S.Hiera(**TINY)
    neck = S.FpnNeck(S.PositionEmbeddingSine(256), d_model=256, backbone_channel_list=trunk.channel_list, fpn_top_down_levels=[2, 3],
                     fpn_interp_model="nearest")
    m = S.SAM2(image_encoder=S.ImageEncoder(trunk, neck, scalp=1), image_size=image_size)
    cfg = dict(TINY, d_model=256)
    sd = {}
    sd.update(O.make_hiera_weights(cfg, seed=HEAD_SEEDS["trunk"], prefix="image_encoder.trunk."))
    sd.update(O.make_fpn_weights([128, 64, 32, 16], 256, seed=HEAD_SEEDS["neck"], prefix="image_encoder.neck."))
    sd.update(O.make_sam_head_weights(256, seed=HEAD_SEEDS["heads"]))
    missing, unexpected = m.sam2_model.load_state_dict(sd, strict=False)
    assert not unexpected and all("mask_downscaling" in k for k in missing), (missing, unexpected)
    return m.to(DEV), sd, cfg


def test_sam_head_kernels():
    g = torch.Generator().manual_seed(5)
    a = torch.randn(37, 24, generator=g).to(DEV); b = torch.randn(5, 24, generator=g).to(DEV)
    exp = a + b[torch.arange(37, device=DEV) % 5]
    assert torch.equal(ops.add_bcast(a, b, out_dtype=torch.float32), exp)
    assert torch.equal(ops.add_bcast(a.to(torch.bfloat16), None, out_dtype=torch.float32), a.to(torch.bfloat16).float())
    assert torch.equal(ops.add_bcast(a, b), exp.to(torch.bfloat16))
    # bilinear resize == F.interpolate(align_corners=False), up and down, with and without plane selection
    x = torch.randn(3, 4, 9, 13, generator=g).to(DEV)
    for size in [(36, 52), (20, 7), (9, 13), (4, 5)]:
        ref = torch.nn.functional.interpolate(x, size=size, mode="bilinear", align_corners=False)
        assert torch.allclose(ops.resize_bilinear(x, size), ref, atol=1e-5), size
    sel = torch.tensor([2, 0, 1], device=DEV, dtype=torch.int32)
    got = ops.resize_bilinear(x, (18, 26), sel=sel, sel_off=1)
    ref = torch.nn.functional.interpolate(x, size=(18, 26), mode="bilinear", align_corners=False)[torch.arange(3), sel.long() + 1][:, None]
    assert torch.allclose(got, ref, atol=1e-5)
    s = torch.tensor([[0.1, 0.7, 0.7], [0.9, 0.2, 0.3], [0.0, 0.0, 0.0]], device=DEV)
    assert ops.argmax_rows(s).tolist() == [1, 0, 0]
    # fused pixel-shuffle + skip + GELU + hypernetwork dot
    B, h, w = 2, 6, 5
    up2 = torch.randn(B * h * w, 128, generator=g).to(DEV).to(torch.bfloat16)
    s0 = torch.randn(B * 4 * h * w, 128, generator=g).to(DEV).to(torch.bfloat16)
    hyper = torch.randn(B, 4, 32, generator=g).to(DEV)
    u = up2.float().view(B, h, w, 2, 2, 32).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * h, 2 * w, 32)
    act = torch.nn.functional.gelu(u + s0[:, :32].float().view(B, 2 * h, 2 * w, 32))
    ref = torch.einsum("bic,byxc->biyx", hyper, act)
    assert rel_err(ops.sam_mask_head(up2, s0, hyper, B, h, w).cpu(), ref.cpu()) < 1e-5
    # few-keys attention (image -> token direction) vs softmax reference, and vs the generic kernel
    Bq, H, Sq, Sk, hd = 2, 8, 200, 9, 16
    q = torch.randn(Bq * Sq, H * hd, generator=g).to(DEV).to(torch.bfloat16)
    k = torch.randn(Bq * Sk, H * hd, generator=g).to(DEV).to(torch.bfloat16)
    v = torch.randn(Bq * Sk, H * hd, generator=g).to(DEV).to(torch.bfloat16)
    args = (q, k, v, Bq, H, H, Sq, Sk, hd, (Sq * H * hd, H * hd), (Sk * H * hd, H * hd), (Sk * H * hd, H * hd))
    o5, o2 = ops.attention(*args, kernel=5), ops.attention(*args, kernel=2)
    sp = lambda t, n: t.float().view(Bq, n, H, hd).transpose(1, 2)
    ref = (torch.softmax(sp(q, Sq) @ sp(k, Sk).transpose(-1, -2) * hd ** -0.5, -1) @ sp(v, Sk)).transpose(1, 2).reshape(Bq * Sq, H * hd)
    assert rel_err(o5.float().cpu(), ref.cpu()) < 1e-2 and rel_err(o2.float().cpu(), ref.cpu()) < 1e-2
    assert torch.equal(ops.attention(*args), o5)                              # auto picks the few-keys kernel here


def test_sam_heads_vs_reference_golden():
    """SAM2Base.track_step(is_init_cond_frame=True, language_embd=...) of the reference (oracle/gen_fixtures_sam2_heads.py)."""
    a, _ = load_golden("sam2_heads_tiny")
    assert a["seeds"].tolist()[:3] == [HEAD_SEEDS[k] for k in ("trunk", "neck", "heads")]
    m, sd, cfg = _tiny_sam2()
    x, lang = t(a["x"]).to(DEV), t(a["lang"]).to(DEV)
    feats = m.sam2_model.forward_image_tokens(x.to(torch.bfloat16))
    out = m.sam2_model.forward_sam_heads_tokens(feats, 2, lang)
    ref = O.sam2_language_masks(sd, cfg, x.cpu().to(torch.bfloat16).float(), lang.cpu())
    assert rel_err(out["ious"].cpu(), t(a["ious"])) < 2e-2
    assert rel_err(out["object_score_logits"].cpu(), t(a["obj"])) < 3e-2
    assert rel_err(out["low_res_multimasks"].cpu(), t(a["multimasks"])) < 4e-2
    assert out["best"].tolist() == ref["best"].tolist() == t(a["ious"]).argmax(-1).tolist()
    assert rel_err(out["low_res_masks"].cpu(), t(a["pred_masks"])) < 4e-2
    assert rel_err(out["high_res_masks"].cpu(), t(a["high_res"])) < 4e-2
    # wrapper: state + per-frame language embeddings -> video-res logits, as `language_embd_inference` returns them
    state = m.get_sam2_embeddings(x)
    masks = m.language_embd_inference(state, [lang[0], lang[1]])
    assert masks.shape == (2, 1, 128, 128) and rel_err(masks.cpu(), t(a["video_res"])) < 4e-2
    # binarised masks agree with the reference's except for pixels whose logit is within the bf16 noise of 0
    ref_bin, got_bin = t(a["video_res"]) > 0, masks.cpu() > 0
    unsure = t(a["video_res"]).abs() < 0.05 * t(a["video_res"]).abs().max()
    assert ((ref_bin != got_bin) & ~unsure).sum() == 0
    # two objects per frame share the cached encoder features
    two = m.language_embd_inference(state, [torch.stack([lang[0, 0], lang[1, 0]]), torch.stack([lang[1, 0], lang[0, 0]])])
    assert two.shape == (4, 1, 128, 128) and torch.equal(two[0::2], masks) and len(state["cached_features"]) == 1
