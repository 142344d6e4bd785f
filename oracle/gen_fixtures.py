"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

Test infrastructure only.  Usage (build container, /root/reference present):

    python oracle/gen_fixtures.py

Every fixture stores inputs, the (tiny) weights and the outputs the reference produced, so
the tests never need the reference again.  The reference's source never enters the repo.
Cross-checks against oracle/ref_cpu.py are asserted here as the fixtures are written.
"""
import os
import sys
import json
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
from transformers import SiglipVisionConfig, SiglipVisionModel, CLIPVisionConfig, CLIPVisionModel  # noqa: E402
sys.modules["transformers"].TRANSFORMERS_CACHE = "/tmp/_no_cache"   # the lazy module may have been re-created
import ufvideo.constants as RC  # noqa: E402
import ufvideo.mm_utils as RMU  # noqa: E402
import ufvideo.model.layer as RL  # noqa: E402
import ufvideo.model.projector as RP  # noqa: E402
import ufvideo.model.encoder as RE  # noqa: E402
import ufvideo.model.videorefer_arch as RA  # noqa: E402
import ufvideo.model.videorefer_qwen2 as RQ  # noqa: E402


def npz(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = v
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)")


def sd_np(sd, prefix=""):
    return {("w::" + prefix + k): v.detach().float().cpu().numpy() for k, v in sd.items()}


def close(a, b, tol=2e-5, what=""):
    err = (a.float() - b.float()).abs().max().item()
    ref = b.float().abs().max().item() + 1e-12
    assert err / ref < tol, f"{what}: oracle vs reference mismatch {err} / {ref}"
    return err / ref


class CharTok:
    """char-code tokenizer stand-in (each character -> ord)."""
    bos_token_id = None

    def __call__(self, text, add_special_tokens=False):
        class R:
            pass
        r = R(); r.input_ids = [ord(c) for c in text]
        return r

    def convert_tokens_to_ids(self, toks):
        return [self.region_id for _ in toks]


def fx_int_helpers():
    print("[int helpers]")
    cases = [(100, 32), (17, 4), (10, 16), (2, 8), (33, 32), (1000, 64), (5, 5)]
    out = {}
    for d, n in cases:
        out[f"uniform_{d}_{n}"] = RMU.frame_sample(d, mode="uniform", num_frames=n)
        assert (O.frame_sample(d, "uniform", n) == out[f"uniform_{d}_{n}"]).all()
    for d, fps in [(100, 25), (7, 30), (300, 24.0), (64, 1)]:
        out[f"fps_{d}_{fps}"] = RMU.frame_sample(d, mode="fps", fps=fps)
        assert (O.frame_sample(d, "fps", fps=fps) == out[f"fps_{d}_{fps}"]).all()
    tok = CharTok()
    prompts = ["ab<video>\ncd<video>e", "<video>\nhello", "no tags at all", "<image><image>", "x<video>"]
    for i, pr in enumerate(prompts):
        mt = "<image>" if "<image>" in pr else "<video>"
        ids = RMU.tokenizer_multimodal_token(pr, tok, mt, return_tensors="pt").numpy()
        out[f"tok_{i}"] = ids
        out[f"tokprompt_{i}"] = np.frombuffer((mt + "|" + pr).encode(), dtype=np.uint8)
        assert O.tokenizer_multimodal_token(pr, lambda s: [ord(c) for c in s], mt) == ids.tolist()
    ids = RMU.tokenizer_multimodal_token("plain", tok, "", return_tensors="pt").numpy()
    out["tok_plain"] = ids
    out["model_name_a"] = np.frombuffer(RMU.get_model_name_from_path("/a/b/UFVideo-7B/").encode(), dtype=np.uint8)
    out["model_name_b"] = np.frombuffer(RMU.get_model_name_from_path("/a/run1/checkpoint-300").encode(), dtype=np.uint8)
    # expand2square
    from PIL import Image
    rng = np.random.default_rng(5)
    for i, (h, w) in enumerate([(6, 10), (9, 4), (5, 5)]):
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        sq = np.array(RMU.expand2square(Image.fromarray(im), (127, 127, 127)))
        out[f"sq_in_{i}"] = im; out[f"sq_out_{i}"] = sq
    x = torch.arange(3 * 4 * 4, dtype=torch.float32).view(3, 4, 4)
    out["sam_pre_in"] = x; out["sam_pre_out"] = RMU.sam_preprocess(x)
    npz("int_helpers", **out)


def fx_processor():
    print("[image processor]")
    from transformers import SiglipImageProcessor
    proc = SiglipImageProcessor(size={"height": 56, "width": 56}, image_mean=[0.5] * 3, image_std=[0.5] * 3)
    from PIL import Image
    rng = np.random.default_rng(7)
    imgs = [rng.integers(0, 256, (80, 100, 3), dtype=np.uint8), rng.integers(0, 256, (56, 56, 3), dtype=np.uint8),
            rng.integers(0, 256, (30, 41, 3), dtype=np.uint8)]
    outs = {}
    for i, im in enumerate(imgs):
        pv = proc.preprocess([Image.fromarray(im)], return_tensors="pt")["pixel_values"]
        outs[f"in_{i}"] = im; outs[f"out_{i}"] = pv[0]
    # the no-resize case pins the arithmetic tail exactly
    close(O.siglip_preprocess(imgs[1][None])[0], outs["out_1"], 1e-6, "siglip_preprocess")
    npz("processor", **outs)


TINY_VIT = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4, image_size=56,
                patch_size=14)


def fx_towers(workdir):
    print("[vision towers]")
    torch.manual_seed(0)
    cfg = SiglipVisionConfig(**TINY_VIT)
    m = SiglipVisionModel(cfg).eval()
    # make biases / LN affine non-trivial
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.ndim == 1:
                p_.add_(torch.randn_like(p_) * 0.05)
    path = os.path.join(workdir, "siglip-so400m-patch14-384")
    m.save_pretrained(path)
    with open(os.path.join(path, "preprocessor_config.json"), "w") as f:
        json.dump({"image_processor_type": "SiglipImageProcessor", "size": {"height": 56, "width": 56},
                   "image_mean": [0.5, 0.5, 0.5], "image_std": [0.5, 0.5, 0.5], "do_resize": True,
                   "do_rescale": True, "do_normalize": True, "resample": 3, "rescale_factor": 1 / 255}, f)

    class A:
        mm_vision_select_layer = -2
        mm_vision_select_feature = "patch"
    tower = RE.SiglipVisionTower("siglip", A(), delay_load=False)
    x = torch.randn(3, 3, 56, 56)
    y = tower(x)
    sd = {k: v for k, v in tower.vision_tower.state_dict().items()}
    pre = "vision_model." if any(k.startswith("vision_model.") for k in sd) else ""
    y_o = O.siglip_tower(sd, TINY_VIT, x, prefix=pre, select_layer=-2)
    print("   siglip tiny rel err", close(y_o, y, what="siglip tower"))
    hs_all = tower.vision_tower(x, output_hidden_states=True).hidden_states
    hs_o = O.siglip_tower(sd, TINY_VIT, x, prefix=pre, return_all=True)
    for a, b in zip(hs_o, hs_all):
        close(a, b, what="siglip hidden states")
    npz("siglip_tiny", x=x, y=y, hs1=hs_all[1], hs3=hs_all[3], prefix=np.frombuffer(pre.encode(), dtype=np.uint8),
        **sd_np(sd))

    # CLIP tiny (encoder.py:12-93)
    ccfg = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4, image_size=56,
                patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5)
    cm = CLIPVisionModel(CLIPVisionConfig(**ccfg)).eval()
    with torch.no_grad():
        for p_ in cm.parameters():
            if p_.ndim == 1:
                p_.add_(torch.randn_like(p_) * 0.05)
    cpath = os.path.join(workdir, "clip-tiny")
    cm.save_pretrained(cpath)
    with open(os.path.join(cpath, "preprocessor_config.json"), "w") as f:
        json.dump({"image_processor_type": "CLIPImageProcessor", "size": {"shortest_edge": 56},
                   "crop_size": {"height": 56, "width": 56}}, f)
    ctower = RE.CLIPVisionTower(cpath, A(), delay_load=False)
    yc = ctower(x)
    csd = dict(ctower.vision_tower.state_dict())
    cpre = "vision_model." if any(k.startswith("vision_model.") for k in csd) else ""
    yc_o = O.clip_tower(csd, ccfg, x, prefix=cpre)
    print("   clip tiny rel err", close(yc_o, yc, what="clip tower"))
    npz("clip_tiny", x=x, y=yc, prefix=np.frombuffer(cpre.encode(), dtype=np.uint8), **sd_np(csd))

    # one full-dimension SigLIP-so400m layer, seeded weights (too large to store): store slices + stats
    FULL = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=1, num_attention_heads=16, image_size=336,
                patch_size=14, hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6)
    fcfg = SiglipVisionConfig(**FULL)
    fm = SiglipVisionModel(fcfg).eval()
    fsd_o = O.make_siglip_weights(FULL, seed=11)
    tgt = fm.state_dict()
    pre_f = "vision_model." if any(k.startswith("vision_model.") for k in tgt) else ""
    fm.load_state_dict({**tgt, **{pre_f + k: v for k, v in fsd_o.items()}})
    gx = torch.Generator().manual_seed(12)
    xin = torch.randn(1, 576, 1152, generator=gx)
    with torch.no_grad():
        yfull = fm.vision_model.encoder.layers[0](xin, None) if hasattr(fm, "vision_model") else fm.encoder.layers[0](xin, None)
    if isinstance(yfull, tuple):
        yfull = yfull[0]
    yo = O.vit_encoder_layer(fsd_o, "encoder.layers.0.", xin, 16, 1e-6, "gelu_pytorch_tanh")
    print("   siglip full-dim layer rel err", close(yo, yfull, what="full layer"))
    npz("siglip_fulldim_layer", y_first4=yfull[0, :4], y_last4=yfull[0, -4:], y_absmean=yfull.abs().mean(),
        y_absmax=yfull.abs().max(), seed_w=np.int64(11), seed_x=np.int64(12))
    return path, sd, pre


def fx_projector():
    print("[projector]")

    class Cfg:
        mm_hidden_size = 16
        hidden_size = 32
    torch.manual_seed(3)
    out = {}
    # spatial_conv: depth 0, downsample (1,2,2), padding 1   (projector.py:241-244)
    m = RP.SpatialConv(Cfg()).eval()
    # NB: with depth=0 the reference feeds mm_hidden_size channels into Conv3d(hidden, hidden):
    # only runnable when mm_hidden_size == hidden_size.
    class Cfg2:
        mm_hidden_size = 32
        hidden_size = 32
    m = RP.SpatialConv(Cfg2()).eval()
    x = torch.randn(1, 4, 16, 32)
    y = m(x)
    sd = dict(m.state_dict())
    yo = O.stc_connector(sd, x, downsample=(1, 2, 2), padding=1, depth=0)
    print("   spatial_conv rel err", close(yo, y, what="spatial_conv"), tuple(y.shape))
    out.update({"sc_x": x, "sc_y": y}); out.update({k.replace("w::", "w::sc."): v for k, v in sd_np(sd).items()})
    # v35 sampler + readout with depth 0 (pins Conv3d k2 s2 p0 + SiLU + readout GELU(erf))
    m2 = RP.STCConnectorV35(Cfg2(), depth=0).eval()
    x2 = torch.randn(2, 4, 36, 32)
    y2 = m2(x2)
    sd2 = dict(m2.state_dict())
    yo2 = O.stc_connector(sd2, x2, downsample=(2, 2, 2), padding=0, depth=0)
    print("   v35(depth0) rel err", close(yo2, y2, what="v35 depth0"), tuple(y2.shape))
    out.update({"v35_x": x2, "v35_y": y2}); out.update({k.replace("w::", "w::v35."): v for k, v in sd_np(sd2).items()})
    # base STC (padding 1, (2,2,2)) depth 0
    m3 = RP.STCConnector(Cfg2(), depth=0).eval()
    y3 = m3(x2)
    sd3 = dict(m3.state_dict())
    yo3 = O.stc_connector(sd3, x2, downsample=(2, 2, 2), padding=1, depth=0)
    print("   stc(depth0,pad1) rel err", close(yo3, y3, what="stc depth0"), tuple(y3.shape))
    out.update({"stc_x": x2, "stc_y": y3}); out.update({k.replace("w::", "w::stc."): v for k, v in sd_np(sd3).items()})
    # AvgPool3d samplers: spatial_pool (1,2,2) and stp_connector (2,2,2), depth 0 (projector.py:218-222,247-250)
    m5 = RP.SpatialPool(Cfg2()).eval()
    y5 = m5(x)
    sd5 = dict(m5.state_dict())
    yo5 = O.stc_connector(sd5, x, downsample=(1, 2, 2), depth=0, avgpool=True)
    print("   spatial_pool rel err", close(yo5, y5, what="spatial_pool"), tuple(y5.shape))
    out.update({"sp_x": x, "sp_y": y5}); out.update({k.replace("w::", "w::sp."): v for k, v in sd_np(sd5).items()})
    m6 = RP.STPConnector(Cfg2(), depth=0).eval()
    x6 = torch.randn(2, 5, 49, 32)                          # odd sizes: AvgPool3d floors (5,7,7) -> (2,3,3)
    y6 = m6(x6)
    sd6 = dict(m6.state_dict())
    yo6 = O.stc_connector(sd6, x6, downsample=(2, 2, 2), depth=0, avgpool=True)
    print("   stp(depth0) rel err", close(yo6, y6, what="stp depth0"), tuple(y6.shape))
    out.update({"stp_x": x6, "stp_y": y6}); out.update({k.replace("w::", "w::stp."): v for k, v in sd_np(sd6).items()})
    # mlp2x_gelu
    class Cfg3:
        mm_hidden_size = 16
        hidden_size = 32
        mm_projector_type = "mlp2x_gelu"
    m4 = RP.build_vision_projector(Cfg3()).eval()
    x4 = torch.randn(2, 9, 16)
    out.update({"mlp_x": x4, "mlp_y": m4(x4)}); out.update({k.replace("w::", "w::mlp."): v for k, v in sd_np(dict(m4.state_dict())).items()})
    npz("projector", **out)


def fx_region():
    print("[region encoder]")

    class Cfg:
        mm_hidden_size = 16
        hidden_size = 24
    torch.manual_seed(4)
    out = {}
    ex = RL.MaskExtractor("square", Cfg()).eval()
    sd = dict(ex.state_dict())
    feats = torch.randn(7, 16, 16)                                  # 7 frames, 4x4 patches, C=16
    g = torch.Generator().manual_seed(5)
    masks = [(torch.rand(3, 20, 28, generator=g) > 0.5).float(), (torch.rand(6, 9, 9, generator=g) > 0.6).float()]
    # sample 0: two objects over frames [0],[1,2]; sample 1: one object over 6 frames (-> token_merge to 4)
    ann = [[[0], [1, 2]], [[1, 2, 3, 4, 5, 6]]]
    y, nums = ex(feats, masks, None, ann, None)
    yo, nums_o = O.mask_extractor(sd, feats, masks, ann)
    assert nums == nums_o, (nums, nums_o)
    print("   mask_extractor rel err", close(yo, y, what="mask_extractor"), nums)
    out.update(dict(feats=feats, mask0=masks[0], mask1=masks[1], y=y, nums=np.array(nums)))
    out.update(sd_np(sd))
    # pad aspect
    exp = RL.MaskExtractor("pad", Cfg()).eval()
    exp.load_state_dict(sd)
    y2, nums2 = exp(feats, masks, None, ann, None)
    yo2, _ = O.mask_extractor(sd, feats, masks, ann, image_aspect_ratio="pad")
    close(yo2, y2, what="mask_extractor pad")
    out["y_pad"] = y2
    # token_merge direct
    xm = torch.randn(1, 9, 16, generator=g)
    for r in (1, 3, 5):
        tm = RL.token_merge(xm, r)
        close(O.token_merge(xm, r), tm, what="token_merge")
        out[f"tm_{r}"] = tm
    out["tm_x"] = xm
    npz("region", **out)


TINY_LLM = dict(vocab_size=300, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, max_position_embeddings=512, rope_theta=10000.0, rms_norm_eps=1e-6)


def build_ref_model(workdir, projector="spatial_conv"):
    cfg = RQ.VideoReferQwen2Config(**{k: v for k, v in TINY_LLM.items() if k != "rope_theta"})
    try:
        cfg.rope_theta = TINY_LLM["rope_theta"]
    except Exception:
        pass
    for k, v in dict(mm_vision_tower="siglip", mm_vision_select_layer=-2, mm_vision_select_feature="patch",
                     mm_projector_type=projector, mm_hidden_size=64, mm_region_encoder_type="pooling",
                     image_aspect_ratio="square", train_mask_decoder=False, sam_pretrained=None, sam_out_dim=256,
                     num_frames=4, seg_token_id=299).items():
        setattr(cfg, k, v)
    torch.manual_seed(21)
    model = RQ.VideoReferQwen2ForCausalLM(cfg).eval()
    model.get_vision_tower().load_model()
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if "vision_tower" in n_:
                continue
            if p_.ndim == 1:
                p_.add_(torch.randn_like(p_) * 0.05)
            else:
                p_.mul_(3.0)          # make logits less flat so argmax is robust
    # transformers >= 5 installs its hidden-state capture hooks lazily and, when the LLM's first forward comes AFTER the
    # tower's, a second time on the nested tower -- `hidden_states[-2]` of the tower then silently means a different layer.
    # (4.46.3, the reference's pin, has no such hooks.)  One LLM forward first installs them exactly once, top-down.
    RQ.Qwen2ForCausalLM.forward(model, inputs_embeds=torch.zeros(1, 2, cfg.hidden_size), output_hidden_states=True)
    tok = CharTok(); tok.region_id = 290
    for m_ in model.modules():
        m_.tokenizer = tok
    return model, cfg, tok


GEN_SCALE = 3.75        # fx_model: decoder-matrix scale of the second pair of greedy sequences (searched on the oracle: 8 + 6 distinct ids, widest top-1 margins)


def fx_model(workdir):
    """End-to-end on a tiny model through the REFERENCE's own classes: splice, forward, generate."""
    print("[model: splice / forward / generate]")
    model, cfg, tok = build_ref_model(workdir)
    sd = {k: v for k, v in model.state_dict().items()}
    vt_pre = "model.vision_tower.vision_tower."
    if any(k.startswith(vt_pre + "vision_model.") for k in sd):
        vt_pre += "vision_model."
    out = {}
    g = torch.Generator().manual_seed(22)
    video = torch.randn(4, 3, 56, 56, generator=g)
    frame = torch.randn(2, 3, 56, 56, generator=g)
    mask = (torch.rand(2, 40, 40, generator=g) > 0.5).float()
    R = tok.region_id
    # ---- splice cases (a7) -------------------------------------------------------------
    cases = {
        "vid_region": dict(ids=[[5, 6, -201, 7, 8, R, 9, R, 10]], images=[(video, "video")], frame=[frame], masks=[mask],
                           ann=[[[0], [1]]], fn=[2]),
        "vid_only": dict(ids=[[5, 6, -201, 7, 8, 9]], images=[(video, "video")], frame=None, masks=None, ann=None, fn=None),
        "img_only": dict(ids=[[-200, 7, 8]], images=[(video[:1], "image")], frame=None, masks=None, ann=None, fn=None),
        "batch_pad": dict(ids=[[5, -201, 7, 8, 9, 11], [1, 2, 3, -201, 4, 0]], images=[(video, "video"), (video.flip(0), "video")],
                          frame=None, masks=None, ann=None, fn=None),
        "vid_noregion_frame": dict(ids=[[5, -201, 7, 8]], images=[(video, "video")], frame=[frame[:1]], masks=[mask[:1]],
                                   ann=[[[0]]], fn=[1]),
        "vid_trailing": dict(ids=[[5, 9, -201]], images=[(video, "video")], frame=None, masks=None, ann=None, fn=None),
    }
    with torch.no_grad():
        mmf = model.encode_images_or_videos([(video, "video")])
        out["mm_features"] = mmf
        tower_out = model.get_vision_tower()(video)
        out["tower_out"] = tower_out
        for name, c in cases.items():
            ids = torch.tensor(c["ids"], dtype=torch.long)
            am = torch.ones_like(ids)
            if name == "batch_pad":
                am[1, -1] = 0
            for with_labels in (False, True):
                labels = ids.clone() if with_labels else None
                if with_labels:
                    labels[labels < 0] = RC.IGNORE_INDEX
                r = model.prepare_inputs_labels_for_multimodal(ids, am, None, labels, c["images"], c["masks"], c["frame"],
                                                               c["ann"], c["fn"])
                _, am2, _, emb, lab2, mark = r
                tag = f"{name}_{'lab' if with_labels else 'nolab'}"
                out[f"sp_{tag}_am"] = am2; out[f"sp_{tag}_emb"] = emb; out[f"sp_{tag}_mark"] = np.array(mark)
                if lab2 is not None:
                    out[f"sp_{tag}_labels"] = lab2
                # cross-check oracle splice (encoders via oracle too)
                feats_o = []
                for (d, modal) in c["images"]:
                    d_ = d.expand(4, -1, -1, -1) if modal == "image" else d
                    f_ = O.siglip_tower(sd, TINY_VIT, d_, prefix=vt_pre)
                    feats_o.append(O.stc_connector(sd, f_[None], prefix="model.mm_projector.", downsample=(1, 2, 2),
                                                   padding=1, depth=0)[0])
                feats_o = torch.stack(feats_o)
                if c["frame"] is not None:
                    ff = O.siglip_tower(sd, TINY_VIT, torch.cat(c["frame"]), prefix=vt_pre)
                    mf, nums = O.mask_extractor(sd, ff, c["masks"], c["ann"], prefix="model.region_encoder.")
                else:
                    mf, nums = [], []
                am_o, emb_o, lab_o, mark_o = O.splice(sd["model.embed_tokens.weight"].float(), ids, am, labels, feats_o, mf,
                                                      nums, R, c["frame"] is not None)
                assert mark_o == mark, (tag, mark_o, mark)
                assert torch.equal(am_o, am2), tag
                if lab2 is not None:
                    assert torch.equal(lab_o, lab2), tag
                close(emb_o, emb, what="splice embeds " + tag)
            out[f"sp_{name}_ids"] = np.array(c["ids"]); out[f"sp_{name}_am_in"] = am
        print("   splice cases ok:", list(cases))
        # ---- LLM forward (a8) through the reference class ------------------------------------
        c = cases["vid_region"]
        ids = torch.tensor(c["ids"], dtype=torch.long); am = torch.ones_like(ids)
        sam = torch.zeros(1, 4, 3, 8, 8)
        fo = model(input_ids=ids, attention_mask=am, images=c["images"], masks=c["masks"], frame=c["frame"],
                   ann_indices=c["ann"], frame_nums=c["fn"], images_sam=sam, inference=True, output_hidden_states=True,
                   use_cache=True, return_dict=True)
        emb = out["sp_vid_region_nolab_emb"]; am2 = out["sp_vid_region_nolab_am"]
        oo = O.qwen2_forward(sd, TINY_LLM, emb, am2)
        print("   qwen2 tiny logits rel err", close(oo["logits"], fo.logits, what="qwen2 logits"))
        for a, b in zip(oo["hidden_states"], fo.hidden_states):
            close(a, b, what="qwen2 hidden")
        out["fw_logits"] = fo.logits; out["fw_hidden_last"] = fo.hidden_states[-1]; out["fw_hidden_1"] = fo.hidden_states[1]
        pk = fo.past_key_values
        try:
            k0, v0 = pk.layers[0].keys, pk.layers[0].values
        except Exception:
            k0, v0 = pk[0][0], pk[0][1]
        close(oo["past"][0][0], k0, what="kv k"); close(oo["past"][0][1], v0, what="kv v")
        out["fw_k0"] = k0; out["fw_v0"] = v0
        # ---- generate (a9), QA branch -----------------------------------------------------------
        gen = model.generate(ids, attention_mask=am, images=c["images"], masks=c["masks"], frame=c["frame"],
                             ann_indices=c["ann"], frame_nums=c["fn"], images_sam=sam, offset=[0, 1], masks_list=None,
                             label_list=torch.zeros(56, 56), do_sample=False, max_new_tokens=8, use_cache=True,
                             pad_token_id=0, eos_token_id=298)
        toks = gen["output"]
        toks_o, _ = O.greedy_generate(sd, TINY_LLM, emb, am2, 8, eos_token_ids=(298,))
        print("   generate tokens ref", toks.tolist(), "oracle", toks_o.tolist())
        assert toks.tolist() == toks_o.tolist()
        out["gen_tokens"] = toks; out["gen_pred_masks_len"] = np.int64(len(gen["pred_masks"]))
        # second prompt, video only
        c2 = cases["vid_only"]; ids2 = torch.tensor(c2["ids"], dtype=torch.long)
        gen2 = model.generate(ids2, attention_mask=torch.ones_like(ids2), images=c2["images"], images_sam=sam, offset=[0, 1],
                              label_list=torch.zeros(56, 56), do_sample=False, max_new_tokens=6, use_cache=True,
                              pad_token_id=0, eos_token_id=298)
        toks2_o, _ = O.greedy_generate(sd, TINY_LLM, out["sp_vid_only_nolab_emb"], out["sp_vid_only_nolab_am"], 6, (298,))
        assert gen2["output"].tolist() == toks2_o.tolist(), (gen2["output"].tolist(), toks2_o.tolist())
        out["gen2_tokens"] = gen2["output"]
        # ---- generate again with the decoder's matrices x GEN_SCALE.  With the weights above both greedy sequences are ONE repeated id
        # ([235] x 8, [1] x 6), which a broken KV cache or a stuck position counter would also emit.  Scaled, the reference walks through
        # 8 and 6 DIFFERENT ids (top-1 margins >= 3 % of the largest logit at every step, so the sequence does not hinge on a rounding).
        scaled = [p_ for n_, p_ in model.named_parameters() if n_.startswith("model.layers.") and p_.ndim == 2]
        saved = [p_.detach().clone() for p_ in scaled]
        for p_ in scaled:
            p_.mul_(GEN_SCALE)
        gs = model.generate(ids, attention_mask=am, images=c["images"], masks=c["masks"], frame=c["frame"],
                            ann_indices=c["ann"], frame_nums=c["fn"], images_sam=sam, offset=[0, 1], masks_list=None,
                            label_list=torch.zeros(56, 56), do_sample=False, max_new_tokens=8, use_cache=True,
                            pad_token_id=0, eos_token_id=298)["output"]
        gs2 = model.generate(ids2, attention_mask=torch.ones_like(ids2), images=c2["images"], images_sam=sam, offset=[0, 1],
                             label_list=torch.zeros(56, 56), do_sample=False, max_new_tokens=6, use_cache=True,
                             pad_token_id=0, eos_token_id=298)["output"]
        for p_, s_ in zip(scaled, saved):
            p_.copy_(s_)                   # (x 3.75 then / 3.75 is not the identity in fp32)
        sds = {k: (v * GEN_SCALE if (k.startswith("model.layers.") and v.ndim == 2) else v) for k, v in sd.items()}      # (sd aliases the parameters: built after they are restored)
        gs_o, _ = O.greedy_generate(sds, TINY_LLM, emb, am2, 8, eos_token_ids=(298,))
        gs2_o, _ = O.greedy_generate(sds, TINY_LLM, out["sp_vid_only_nolab_emb"], out["sp_vid_only_nolab_am"], 6, (298,))
        print("   generate (decoder matrices x %g) ref" % GEN_SCALE, gs.tolist(), gs2.tolist())
        assert gs.tolist() == gs_o.tolist() and gs2.tolist() == gs2_o.tolist(), (gs_o.tolist(), gs2_o.tolist())
        assert len(set(gs[0].tolist())) == 8 and len(set(gs2[0].tolist())) == 6, "the scaled sequences must not repeat an id"
        out["gens_scale"] = np.float64(GEN_SCALE); out["gens_tokens"] = gs; out["gens2_tokens"] = gs2
        # text_hidden_fcs (a10)
        hfc = model.get_model().text_hidden_fcs[0](fo.hidden_states[-1])
        close(O.text_hidden_fcs(sd, fo.hidden_states[-1]), hfc, what="text_hidden_fcs")
        out["fcs_out"] = hfc
    out["video"] = video; out["frame"] = frame; out["mask"] = mask
    out["region_id"] = np.int64(R)
    keep = {k: v for k, v in sd.items() if "mask_encoder" not in k}
    out.update(sd_np(keep))
    npz("model_tiny", **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_grad_enabled(False)
    work = tempfile.mkdtemp(prefix="ufv_fx_")
    os.chdir(work)                       # the reference hard-codes a cwd-relative tower path (encoder.py:108)
    print("transformers", transformers.__version__, "torch", torch.__version__, "cwd", work)
    fx_int_helpers()
    fx_processor()
    fx_towers(work)
    fx_projector()
    fx_region()
    fx_model(work)
    print("done")


if __name__ == "__main__":
    main()
