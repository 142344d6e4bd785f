"""CPU: the oracle (oracle/ref_cpu.py) against golden vectors produced by running the reference
(oracle/gen_fixtures.py, build container).  These pin the oracle; the -m gpu tests then use the
oracle as the checker for the HIP path."""
import numpy as np
import torch

from conftest import load_golden, t, rel_err, scaled_decoder
from oracle import ref_cpu as O

TINY_VIT = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4, image_size=56,
                patch_size=14)
TINY_LLM = dict(vocab_size=300, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, max_position_embeddings=512, rope_theta=10000.0, rms_norm_eps=1e-6)
TOL = 2e-5


def test_frame_sample_and_tokenizer_kats():
    a, _ = load_golden("int_helpers")
    for k in a:
        if k.startswith("uniform_"):
            _, d, n = k.split("_")
            assert (O.frame_sample(int(d), "uniform", int(n)) == a[k]).all(), k
        elif k.startswith("fps_"):
            _, d, fps = k.split("_")
            assert (O.frame_sample(int(d), "fps", fps=float(fps)) == a[k]).all(), k
    # SURVEY §8c known answers
    assert O.frame_sample(17, "uniform", 4).tolist() == [2, 6, 10, 14]
    assert O.frame_sample(100, "fps", fps=25).tolist() == [12, 37, 62, 87]
    for i in range(5):
        mt, prompt = bytes(a[f"tokprompt_{i}"]).decode().split("|", 1)
        assert O.tokenizer_multimodal_token(prompt, lambda s: [ord(c) for c in s], mt) == a[f"tok_{i}"].tolist()
    assert O.tokenizer_multimodal_token("ab<video>\ncd<video>e", lambda s: [ord(c) for c in s], "<video>") == \
        [97, 98, -201, 10, 99, 100, -201, 101]


def test_siglip_preprocess_tail():
    a, _ = load_golden("processor")
    assert rel_err(O.siglip_preprocess(a["in_1"][None])[0], t(a["out_1"])) < 1e-6


def test_siglip_tiny_tower():
    a, w = load_golden("siglip_tiny")
    pre = bytes(a["prefix"]).decode()
    x = t(a["x"])
    assert rel_err(O.siglip_tower(w, TINY_VIT, x, prefix=pre), t(a["y"])) < TOL
    hs = O.siglip_tower(w, TINY_VIT, x, prefix=pre, return_all=True)
    assert rel_err(hs[1], t(a["hs1"])) < TOL and rel_err(hs[3], t(a["hs3"])) < TOL
    assert torch.equal(hs[2], O.siglip_tower(w, TINY_VIT, x, prefix=pre))      # select_layer=-2 of 3 layers


def _sub0(w, pre):
    return {k[len(pre):]: v for k, v in w.items() if k.startswith(pre)}


GEOM_VIT = dict(hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=2, image_size=76, patch_size=14)


def test_geometry_of_the_released_checkpoint_remainder_pixels_and_odd_grid_vs_reference_golden():
    """oracle/gen_fixtures_geom.py ran the REFERENCE's own SiglipVisionTower at 76 px (76 = 5 x 14 + 6: the remainder of 384 = 27 x 14 + 6) and its STCConnectorV35 (depth 0) on the
    odd 5 x 5 grid (floors to 2 x 2, as 27 -> 13): the stride-14 convolution drops the 6 remainder pixels and the sampler drops the odd row / column, in the oracle as in the reference."""
    a, w = load_golden("geom_odd")
    pre = bytes(a["prefix"]).decode()
    x = t(a["x"])
    y = O.siglip_tower(w, GEOM_VIT, x, prefix=pre)
    assert y.shape == (4, 25, 32) and rel_err(y, t(a["y"])) < TOL
    x2 = x.clone(); x2[:, :, 70:, :] = 9.0; x2[:, :, :, 70:] = -9.0                # the remainder pixels reach nothing
    assert torch.equal(O.siglip_tower(w, GEOM_VIT, x2, prefix=pre), y)
    z = O.stc_connector(_sub0(w, "proj."), t(a["y"])[None], downsample=(2, 2, 2), padding=0, depth=0)
    assert z.shape == (1, 8, 32) and rel_err(z, t(a["z"])) < TOL


def test_clip_tiny_tower():
    a, w = load_golden("clip_tiny")
    cfg = dict(TINY_VIT, hidden_act="quick_gelu", layer_norm_eps=1e-5)
    assert rel_err(O.clip_tower(w, cfg, t(a["x"]), prefix=bytes(a["prefix"]).decode()), t(a["y"])) < TOL


def test_siglip_fulldim_layer_seeded():
    a, _ = load_golden("siglip_fulldim_layer")
    cfg = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=1, num_attention_heads=16, image_size=336,
               patch_size=14)
    sd = O.make_siglip_weights(cfg, seed=int(a["seed_w"]))
    x = torch.randn(1, 576, 1152, generator=torch.Generator().manual_seed(int(a["seed_x"])))
    y = O.vit_encoder_layer(sd, "encoder.layers.0.", x, 16, 1e-6, "gelu_pytorch_tanh")
    assert rel_err(y[0, :4], t(a["y_first4"])) < TOL and rel_err(y[0, -4:], t(a["y_last4"])) < TOL
    assert abs(y.abs().mean().item() - float(a["y_absmean"])) < 1e-5


def _sub(w, pre):
    return {k[len(pre):]: v for k, v in w.items() if k.startswith(pre)}


def test_projector_variants():
    a, w = load_golden("projector")
    assert rel_err(O.stc_connector(_sub(w, "sc."), t(a["sc_x"]), downsample=(1, 2, 2), padding=1, depth=0), t(a["sc_y"])) < TOL
    assert rel_err(O.stc_connector(_sub(w, "v35."), t(a["v35_x"]), downsample=(2, 2, 2), padding=0, depth=0), t(a["v35_y"])) < TOL
    assert rel_err(O.stc_connector(_sub(w, "stc."), t(a["stc_x"]), downsample=(2, 2, 2), padding=1, depth=0), t(a["stc_y"])) < TOL
    assert a["sc_y"].shape == (1, 54, 32)        # SURVEY §2.3-C token-count probe
    for pre, ds in (("sp", (1, 2, 2)), ("stp", (2, 2, 2))):          # AvgPool3d samplers: spatial_pool / stp_connector (odd sizes floor)
        y = O.stc_connector(_sub(w, pre + "."), t(a[pre + "_x"]), downsample=ds, depth=0, avgpool=True)
        assert y.shape == a[pre + "_y"].shape and rel_err(y, t(a[pre + "_y"])) < TOL, pre


def test_regstage_shapes_and_token_rule():
    # RegStage is parity-unpinned (timm absent offline): check structure + the v35 token rule (T/2)*floor(g/2)^2
    sd = O.make_stc_weights(16, 32, seed=3)
    y = O.stc_connector(sd, torch.randn(1, 4, 16, 16), depth=4)
    assert y.shape == (1, 2 * 2 * 2, 32) and torch.isfinite(y).all()


def test_region_encoder():
    a, w = load_golden("region")
    masks = [t(a["mask0"]), t(a["mask1"])]
    ann = [[[0], [1, 2]], [[1, 2, 3, 4, 5, 6]]]
    y, nums = O.mask_extractor(w, t(a["feats"]), masks, ann)
    assert nums == a["nums"].tolist()
    assert rel_err(y, t(a["y"])) < TOL
    y2, _ = O.mask_extractor(w, t(a["feats"]), masks, ann, image_aspect_ratio="pad")
    assert rel_err(y2, t(a["y_pad"])) < TOL
    for r in (1, 3, 5):
        assert rel_err(O.token_merge(t(a["tm_x"]), r), t(a[f"tm_{r}"])) < TOL


CASES = {
    "vid_region": dict(frame=True, ann=[[[0], [1]]]),
    "vid_only": dict(frame=False), "img_only": dict(frame=False), "batch_pad": dict(frame=False),
    "vid_noregion_frame": dict(frame=True, ann=[[[0]]]), "vid_trailing": dict(frame=False),
}


def _encode(w, a, name):
    vt = "model.vision_tower.vision_tower."
    if any(k.startswith(vt + "vision_model.") for k in w):
        vt += "vision_model."
    video = t(a["video"])
    imgs = {"vid_region": [video], "vid_only": [video], "img_only": [video[:1].expand(4, -1, -1, -1)],
            "batch_pad": [video, video.flip(0)], "vid_noregion_frame": [video], "vid_trailing": [video]}[name]
    feats = torch.stack([O.stc_connector(w, O.siglip_tower(w, TINY_VIT, v, prefix=vt)[None], prefix="model.mm_projector.",
                                         downsample=(1, 2, 2), padding=1, depth=0)[0] for v in imgs])
    c = CASES[name]
    if c["frame"]:
        fr = t(a["frame"]) if name == "vid_region" else t(a["frame"])[:1]
        mk = t(a["mask"]) if name == "vid_region" else t(a["mask"])[:1]
        mf, nums = O.mask_extractor(w, O.siglip_tower(w, TINY_VIT, fr, prefix=vt), [mk], c["ann"], prefix="model.region_encoder.")
    else:
        mf, nums = [], []
    return feats, mf, nums


def test_splice_all_cases_bit_exact_indices():
    a, w = load_golden("model_tiny")
    R = int(a["region_id"])
    table = w["model.embed_tokens.weight"]
    for name, c in CASES.items():
        feats, mf, nums = _encode(w, a, name)
        ids = t(a[f"sp_{name}_ids"]); am = t(a[f"sp_{name}_am_in"])
        for lab in (False, True):
            labels = None
            if lab:
                labels = ids.clone(); labels[labels < 0] = -100
            tag = f"{name}_{'lab' if lab else 'nolab'}"
            am_o, emb_o, lab_o, mark_o = O.splice(table, ids, am, labels, feats, mf, nums, R, c["frame"])
            assert np.array_equal(np.array(mark_o), a[f"sp_{tag}_mark"]), tag
            assert np.array_equal(am_o.numpy(), a[f"sp_{tag}_am"]), tag
            if lab:
                assert np.array_equal(lab_o.numpy(), a[f"sp_{tag}_labels"]), tag
            assert rel_err(emb_o, t(a[f"sp_{tag}_emb"])) < TOL, tag


def test_qwen2_forward_kv_and_generate():
    a, w = load_golden("model_tiny")
    emb = t(a["sp_vid_region_nolab_emb"]); am = t(a["sp_vid_region_nolab_am"])
    o = O.qwen2_forward(w, TINY_LLM, emb, am)
    assert rel_err(o["logits"], t(a["fw_logits"])) < TOL
    assert rel_err(o["hidden_states"][-1], t(a["fw_hidden_last"])) < TOL
    assert rel_err(o["hidden_states"][1], t(a["fw_hidden_1"])) < TOL
    assert rel_err(o["past"][0][0], t(a["fw_k0"])) < TOL and rel_err(o["past"][0][1], t(a["fw_v0"])) < TOL
    toks, hid = O.greedy_generate(w, TINY_LLM, emb, am, 8, eos_token_ids=(298,))
    assert toks.tolist() == a["gen_tokens"].tolist()
    toks2, _ = O.greedy_generate(w, TINY_LLM, t(a["sp_vid_only_nolab_emb"]), t(a["sp_vid_only_nolab_am"]), 6, (298,))
    assert toks2.tolist() == a["gen2_tokens"].tolist()
    assert rel_err(O.text_hidden_fcs(w, t(a["fw_hidden_last"])), t(a["fcs_out"])) < TOL


def test_generate_distinct_tokens_golden():
    """The reference's greedy sequences with the decoder matrices x 3.75 (oracle/gen_fixtures.py GEN_SCALE): 8 and 6 DIFFERENT ids, so
    a stuck position counter or a broken KV cache cannot reproduce them (the unscaled goldens are one id repeated).  fp32 oracle and
    its bf16 mirror both walk the reference's sequence; the top-1 margins are >= 2.5 % of the largest logit at every step."""
    a, w = load_golden("model_tiny")
    ws = scaled_decoder(w, a["gens_scale"])
    g1, g2 = a["gens_tokens"].tolist(), a["gens2_tokens"].tolist()
    assert len(set(g1[0])) == 8 and len(set(g2[0])) == 6
    for emb, am, n, gold in ((t(a["sp_vid_region_nolab_emb"]), t(a["sp_vid_region_nolab_am"]), 8, g1),
                             (t(a["sp_vid_only_nolab_emb"]), t(a["sp_vid_only_nolab_am"]), 6, g2)):
        toks, _ = O.greedy_generate(ws, TINY_LLM, emb, am, n, eos_token_ids=(298,))
        assert toks.tolist() == gold
        with O.bf16_mirror():
            toks_m, _ = O.greedy_generate(ws, TINY_LLM, emb, am, n, eos_token_ids=(298,))
        assert toks_m.tolist() == gold
        full = torch.cat([emb, ws["model.embed_tokens.weight"].float()[toks[0]][None]], 1)
        lg = O.qwen2_forward(ws, TINY_LLM, full, torch.ones(1, full.shape[1], dtype=torch.long))["logits"][0, emb.shape[1] - 1:-1]
        top2 = lg.topk(2, -1).values
        assert ((top2[:, 0] - top2[:, 1]).min() / lg.abs().max()).item() > 0.025


def test_kv_cache_decode_equals_full_forward():
    sd = O.make_qwen2_weights(TINY_LLM, seed=9)
    x = torch.randn(1, 11, 64, generator=torch.Generator().manual_seed(1))
    full = O.qwen2_forward(sd, TINY_LLM, x)
    pre = O.qwen2_forward(sd, TINY_LLM, x[:, :10])
    step = O.qwen2_forward(sd, TINY_LLM, x[:, 10:], past=pre["past"])
    assert rel_err(step["logits"][0, -1], full["logits"][0, -1]) < 1e-5


def test_sam2_image_encoder_tiny():
    """Hiera trunk + FPN neck restatement vs the reference's own classes (oracle/gen_fixtures_sam2.py)."""
    a, w = load_golden("sam2_encoder_tiny")
    cfg = dict(embed_dim=16, num_heads=1, stages=(1, 2, 3, 1), global_att_blocks=(4,), window_spec=(8, 4, 8, 4),
               window_pos_embed_bkg_spatial_size=(7, 7), d_model=32)
    x = t(a["x"])
    stages = O.hiera_forward(w, cfg, x, prefix="trunk.")
    for i, s in enumerate(stages):
        assert rel_err(s, t(a[f"stage{i}"])) < 3e-5
    out = O.sam2_image_encoder(w, cfg, x)
    for i, f in enumerate(out["backbone_fpn"]):
        assert rel_err(f, t(a[f"fpn{i}"])) < 3e-5
    assert rel_err(out["vision_pos_enc"][0], t(a["pos0"])) < 1e-6
    blocks, ends = O.hiera_schedule(dict(embed_dim=144, num_heads=2, stages=(2, 6, 36, 4), global_att_blocks=(23, 33, 43),
                                         window_spec=(8, 4, 16, 8)))
    assert ends == [1, 7, 43, 47] and [b["q_stride"] for b in blocks].count(2) == 3
    assert {b["dim_out"] // b["heads"] for b in blocks} == {72}                   # head_dim 72 at every stage
    assert [blocks[i]["window"] for i in (0, 2, 8, 23, 44)] == [8, 8, 4, 0, 16]


def test_sam2_heads_language_prompt():
    """SAM heads restatement vs the reference's SAM2Base.track_step with a language token (gen_fixtures_sam2_heads.py).
    Weights are regenerated from the seeds stored with the vectors."""
    a, _ = load_golden("sam2_heads_tiny")
    s_trunk, s_neck, s_heads = a["seeds"].tolist()[:3]
    cfg = dict(embed_dim=16, num_heads=1, stages=(1, 2, 3, 1), global_att_blocks=(4,), window_spec=(8, 4, 8, 4),
               window_pos_embed_bkg_spatial_size=(7, 7), d_model=256)
    sd = {}
    sd.update(O.make_hiera_weights(cfg, seed=s_trunk, prefix="image_encoder.trunk."))
    sd.update(O.make_fpn_weights([128, 64, 32, 16], 256, seed=s_neck, prefix="image_encoder.neck."))
    sd.update(O.make_sam_head_weights(256, seed=s_heads))
    out = O.sam2_language_masks(sd, cfg, t(a["x"]), t(a["lang"]))
    assert rel_err(out["low_res_multimasks"], t(a["multimasks"])) < 1e-4
    assert rel_err(out["ious"], t(a["ious"])) < 1e-4 and rel_err(out["object_score_logits"], t(a["obj"])) < 1e-4
    assert rel_err(out["low_res_masks"], t(a["pred_masks"])) < 1e-4 and rel_err(out["high_res_masks"], t(a["high_res"])) < 1e-4
    assert rel_err(out["video_res_masks"], t(a["video_res"])) < 1e-4
    assert out["best"].tolist() == t(a["ious"]).argmax(-1).tolist()


def _seg_setup():
    a, _ = load_golden("seg_tiny")
    m, w = load_golden("model_tiny")
    cfg = dict(embed_dim=16, num_heads=1, stages=(1, 2, 3, 1), global_att_blocks=(4,), window_spec=(8, 4, 8, 4),
               window_pos_embed_bkg_spatial_size=(7, 7), d_model=256)
    s_trunk, s_neck, s_heads = a["sam_seeds"].tolist()
    sam = {}
    sam.update(O.make_hiera_weights(cfg, seed=s_trunk, prefix="image_encoder.trunk."))
    sam.update(O.make_fpn_weights([128, 64, 32, 16], 256, seed=s_neck, prefix="image_encoder.neck."))
    sam.update(O.make_sam_head_weights(256, seed=s_heads))
    return a, m, w, cfg, sam


def test_seg_branches_of_generate():
    """[SEG] -> masks glue of generate() vs the reference's own generate() (oracle/gen_fixtures_seg.py)."""
    a, m, w, cfg, sam = _seg_setup()
    images_sam = t(a["images_sam"])[0]
    # generated [SEG]: step 0 contributes every prompt position, later steps one state each
    toks, hid = O.greedy_generate(w, TINY_LLM, t(m["sp_vid_only_nolab_emb"]), t(m["sp_vid_only_nolab_am"]), 6, (298,))
    assert toks.tolist() == m["gen2_tokens"].tolist()
    masks, logits = O.seg_masks_generated(w, toks, hid, int(a["gen_seg_id"]), sam, cfg, images_sam, (40, 50))
    S = t(m["sp_vid_only_nolab_emb"]).shape[1]
    assert len(masks) == a["gen_masks"].shape[0] == S + 4
    for j, i in enumerate(a["gen_logits_idx"].tolist()):
        assert rel_err(logits[i], t(a["gen_logits"][j])) < 2e-4
    got = torch.stack(masks).numpy()
    assert (got != a["gen_masks"]).mean() < 1e-4
    # [SEG] in the prompt: the LLM forward over the re-spliced prompt, then the same glue
    tab, e1 = w["model.embed_tokens.weight"].float(), t(m["sp_vid_only_nolab_emb"])[0]
    e2 = torch.cat([e1[:-2], tab[299][None], tab[9][None], tab[299][None]], 0)[None]
    hl = O.qwen2_forward(w, TINY_LLM, e2, torch.ones(1, e2.shape[1], dtype=torch.long), None)["hidden_states"][-1]
    assert rel_err(hl, t(a["prompt_hidden_last"])) < 1e-5
    pm, pl = O.seg_masks_prompt(w, a["prompt_ids"], [S - 3 + 0, 4], t(a["prompt_hidden_last"]), 299, sam, cfg, images_sam, (33, 47))
    assert rel_err(pl, t(a["prompt_logits"])) < 2e-4 and (pm.numpy() != a["prompt_masks"][0]).mean() < 1e-4


def test_training_losses_vs_reference_forward():
    """CE + mask BCE + DICE restatement vs the reference's forward(inference=False) (oracle/gen_fixtures_train.py)."""
    a, _ = load_golden("train_tiny")
    _, m, w, cfg, sam = _seg_setup()
    tab, e1 = w["model.embed_tokens.weight"].float(), t(m["sp_vid_only_nolab_emb"])[0]
    mm = e1[2:-3]                                               # the video tokens of the vid_only splice: [5, 6, <video>, 7, 8, 9]
    images_sam = t(a["images_sam"])[0]
    for name in ("two_obj", "one_obj", "no_seg"):
        ids, labels = t(a[name + "_ids"])[0], t(a[name + "_labels"])
        k = ids.tolist().index(-201)
        emb = torch.cat([tab[ids[:k]], mm, tab[ids[k + 1:]]], 0)[None]
        lab = torch.cat([labels[0, :k], torch.full((mm.shape[0],), -100), labels[0, k + 1:]])[None]
        gt = t(a[name + "_gt"])
        r = O.training_losses(w, TINY_LLM, emb, torch.ones(1, emb.shape[1], dtype=torch.long), lab, 299, sam, cfg, images_sam, [gt],
                              [torch.zeros(gt.shape[1:]) if gt.shape[0] else torch.zeros(1, 1)], tuple(a["loss_weights"].tolist()))
        got = torch.stack([r[k_].float() for k_ in ("loss", "ce_loss", "mask_bce_loss", "mask_dice_loss", "mask_loss")])
        assert torch.allclose(got, t(a[name + "_losses"]).float(), rtol=2e-5, atol=1e-6), (name, got, a[name + "_losses"])


def spliced_embed_ids(ids, S):
    """vocabulary row of every spliced position, -1 for the inserted visual tokens (one <video> sentinel expands to S - len + 1)"""
    ids = [int(i) for i in ids]
    n_vis = S - (len(ids) - 1)
    out = []
    for i in ids:
        out += [-1] * n_vis if i < 0 else [i]
    return torch.tensor(out, dtype=torch.long)


def test_decoder_train_grads_vs_reference_backward():
    """oracle autograd + AdamW restatement == the reference's own ce_loss.backward() / clip / AdamW.step() (train_grad_tiny)."""
    a, _ = load_golden("train_grad_tiny")
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_grad_tiny.npz"))
    _, w = load_golden("model_tiny")
    loss, grads, de = O.decoder_train_grads(w, TINY_LLM, t(a["inputs_embeds"]), t(a["labels"]))
    assert abs(float(loss) - float(a["ce_loss"])) < 1e-5
    ref_g = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("g::")}
    assert set(ref_g) == set(grads)
    # embed_tokens enters through the splice (videorefer_arch.py:239-370): its gradient = rows of d_inputs_embeds scattered by id
    eids = spliced_embed_ids(a["ids"][0], de.shape[1])
    ge = torch.zeros_like(grads["model.embed_tokens.weight"])
    ge.index_add_(0, eids[eids >= 0], de[0][eids >= 0])
    grads["model.embed_tokens.weight"] = ge
    for k, g in ref_g.items():
        assert rel_err(grads[k], g) < 1e-4, k
    assert rel_err(de, t(a["d_inputs_embeds"])) < 1e-4
    lr, wd, b1, b2, eps, clip = a["hyper"].tolist()
    new, norm = O.adamw_first_step({k: w[k] for k in ref_g}, ref_g, lr, wd, (b1, b2), eps, clip)
    assert abs(float(norm) - float(a["grad_norm"])) < 1e-3 * float(a["grad_norm"])
    for k in ref_g:
        assert torch.allclose(new[k], torch.from_numpy(z["p1::" + k]), atol=2e-6, rtol=1e-5), k


def hd128_golden():
    """train_grad_hd128.npz (reference's own backward on a 2-layer decoder with 2 / 1 heads of 128): arrays, llm dict and the
    weights, regenerated from the stored seed exactly as the generating script built them (bf16-rounded, matrices x 3)"""
    import os
    a, _ = load_golden("train_grad_hd128")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_grad_hd128.npz"))
    V, D, I, L, H, KV = (int(x) for x in a["llm"])
    llm = dict(vocab_size=V, hidden_size=D, intermediate_size=I, num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=KV,
               rope_theta=10000.0, rms_norm_eps=1e-6)
    sc = float(a["weight_scale"])
    w = {k: (v * (sc if v.ndim == 2 else 1.0)).to(torch.bfloat16).to(torch.float32) for k, v in O.make_qwen2_weights(llm, seed=int(a["seed"])).items()}
    ref_g = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("g::")}
    return a, llm, w, ref_g


def test_decoder_train_grads_hd128_vs_reference_backward():
    """the same pin at head_dim 128 (the shape the fused attention-backward kernels serve): oracle autograd == the reference
    class's own loss.backward()"""
    a, llm, w, ref_g = hd128_golden()
    loss, grads, de = O.decoder_train_grads(w, llm, t(a["inputs_embeds"]), t(a["labels"]))
    assert abs(float(loss) - float(a["loss"])) < 1e-5
    assert rel_err(de, t(a["d_inputs_embeds"])) < 1e-4
    assert len(ref_g) == 13
    for k, g in ref_g.items():
        assert rel_err(grads[k], g) < 1e-4, k


def test_sampling_distribution_vs_transformers_warpers():
    """the oracle's temperature / top-k / top-p restatement == transformers' own LogitsWarpers (the chain HF generate() builds
    from the kwargs the reference forwards)"""
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    g = torch.Generator().manual_seed(9)
    for V, T, k, p in ((1000, 0.2, 50, 0.9), (5000, 0.7, 0, 0.8), (300, 1.3, 20, 1.0), (300, 0.2, 0, 0.9)):
        lg = torch.randn(1, V, generator=g) * 3
        x = TemperatureLogitsWarper(T)(None, lg.clone())
        if k:
            x = TopKLogitsWarper(k)(None, x)
        if p < 1.0:
            x = TopPLogitsWarper(p)(None, x)
        ref = x.softmax(-1)[0]
        got = O.sampling_distribution(lg[0], T, k, p)
        assert torch.equal(got > 0, ref > 0) and torch.allclose(got, ref, atol=1e-7)


def test_bf16_mirror_is_the_same_graph_with_bf16_storage_points():
    """`O.bf16_mirror()` (the checker for the north star's 1e-3 bf16 tolerance) pinned on CPU against the fp32 restatement and
    the reference's goldens: same graph, values rounded where the HIP path stores bf16 -- so it stays within bf16 noise of the
    reference, what it stores is exactly bf16-representable, and leaving the context restores the fp32 graph bit for bit."""
    def is_bf16(x):
        return torch.equal(x, x.to(torch.bfloat16).float())

    a, w = load_golden("siglip_tiny")
    pre = bytes(a["prefix"]).decode()
    x = t(a["x"])
    y32 = O.siglip_tower(w, TINY_VIT, x, prefix=pre)
    with O.bf16_mirror():
        y16 = O.siglip_tower(w, TINY_VIT, x, prefix=pre)
        assert O._MIRROR
    assert not O._MIRROR and torch.equal(O.siglip_tower(w, TINY_VIT, x, prefix=pre), y32)
    e = rel_err(y16, t(a["y"]))
    assert 1e-4 < e < 1e-2, e                                   # bf16 storage is visible, and stays bf16-sized
    # decoder: logits / hidden within bf16 noise of the reference, KV cache entries and greedy tokens as the reference's
    a, w = load_golden("model_tiny")
    emb = t(a["sp_vid_region_nolab_emb"]); am = t(a["sp_vid_region_nolab_am"])
    with O.bf16_mirror():
        o = O.qwen2_forward(w, TINY_LLM, emb, am)
        toks, _ = O.greedy_generate(w, TINY_LLM, emb, am, 8, eos_token_ids=(298,))
        fcs = O.text_hidden_fcs(w, t(a["fw_hidden_last"]))
    assert 1e-4 < rel_err(o["logits"], t(a["fw_logits"])) < 2e-2
    assert rel_err(o["hidden_states"][-1], t(a["fw_hidden_last"])) < 2e-2
    assert is_bf16(o["past"][0][0]) and is_bf16(o["past"][1][1])            # what the KV cache holds
    assert not is_bf16(o["hidden_states"][1])                              # the residual stream stays fp32
    assert toks.tolist() == a["gen_tokens"].tolist()
    assert rel_err(fcs, t(a["fcs_out"])) < 1e-2
    # connector (RegStage blocks included) and region encoder
    sd = O.make_stc_weights(64, 128, seed=5)
    xs = torch.randn(1, 4, 36, 64, generator=torch.Generator().manual_seed(6))
    with O.bf16_mirror():
        ym = O.stc_connector(sd, xs)
    assert 1e-4 < rel_err(ym, O.stc_connector(sd, xs)) < 2e-2
    a, w = load_golden("region")
    masks = [t(a["mask0"]), t(a["mask1"])]
    ann = [[[0], [1, 2]], [[1, 2, 3, 4, 5, 6]]]
    with O.bf16_mirror():
        y, nums = O.mask_extractor(w, t(a["feats"]), masks, ann)
    assert nums == a["nums"].tolist() and rel_err(y, t(a["y"])) < 1e-2


def test_bf16_chain_noise_floor():
    """Why a chain of bf16 stages cannot be compared at 1e-3 in the max norm: the SAME correct bf16 algorithm (the mirror) run
    on inputs that differ by 1e-7 relative -- less than any two fp32 accumulation orders differ -- ends up apart by the order of
    one bf16 ulp after three small decoder layers (after ONE at the 7B dimensions, where more elements sit near a rounding
    boundary: 3.4e-3 max / 2.5e-3 rms), because every storage point turns a discrepancy d into one-ulp flips with
    probability d / ulp (rms sqrt(d * ulp): 1e-7 -> 2e-5 -> 3e-4 -> 1e-3 ...).  The fp32 graph is stable under the same
    perturbation.  tests/test_parity_bf16_gpu.py reports HIP-vs-mirror next to this floor; tests/test_kernel_rounding_gpu.py makes
    the per-kernel statement that does not suffer from it."""
    cfg = dict(vocab_size=64, hidden_size=512, intermediate_size=1536, num_hidden_layers=3, num_attention_heads=4,
               num_key_value_heads=2, rope_theta=1e6, rms_norm_eps=1e-6)
    sd = O.make_qwen2_weights(cfg, seed=12, std=0.06)     # branch outputs comparable to the residual stream, as at 7B dims
    g = torch.Generator().manual_seed(14)
    x = torch.randn(1, 160, 512, generator=g) * 0.5
    xp = x * (1 + 1e-7 * torch.randn(x.shape, generator=g))
    with O.bf16_mirror():
        a = O.qwen2_forward(sd, cfg, x)["hidden_states"][-1]
        b = O.qwen2_forward(sd, cfg, xp)["hidden_states"][-1]
    f = O.qwen2_forward(sd, cfg, x)["hidden_states"][-1]
    fp = O.qwen2_forward(sd, cfg, xp)["hidden_states"][-1]
    assert rel_err(fp, f) < 1e-5                       # fp32: the perturbation stays a perturbation
    floor = rel_err(b, a)
    assert 1e-3 < floor < 3e-2, floor                  # bf16 storage: amplified to the ulp scale
    assert floor > 0.1 * rel_err(a, f)                 # ... i.e. a sizeable part of the whole bf16-vs-fp32 distance


def test_mask_loss_grads_vs_reference_backward():
    """The rest of row a12: loss = ce + bce_w BCE + dice_w DICE of the reference's forward(inference=False) with
    train_mask_decoder semantics, `loss.backward()` (oracle/gen_fixtures_seg_grad.py).  torch autograd over the oracle's
    restatement must reproduce: the losses, d(loss)/d(inputs_embeds), the gradients of text_hidden_fcs and of every decoder
    parameter in full, and norm / sum / a strided sample of the gradient of every sam_mask_decoder parameter; the IoU and
    object-score heads receive no gradient (the mask is PICKED by arg-max IoU).  This pins the oracle as the checker of the HIP
    mask-loss backward (tests/test_seg_train_gpu.py)."""
    a, _ = load_golden("seg_grad_tiny")
    _, m, w, cfg, sam = _seg_setup()
    assert a["sam_seeds"].tolist() == t(load_golden("seg_tiny")[0]["sam_seeds"]).tolist()
    tab, e1 = w["model.embed_tokens.weight"].float(), t(m["sp_vid_only_nolab_emb"])[0]
    mm = e1[2:-3]
    images_sam = t(a["images_sam"])[0]
    for name in ("two_obj", "one_obj", "blob"):          # "blob": the structured ground truth of round 3 (two discs, an empty and a full mask)
        ws = {k: v.float().clone().requires_grad_(k.startswith(("model.layers.", "model.norm.", "lm_head.", "model.text_hidden_fcs.", "model.embed_tokens.")))
              for k, v in w.items()}
        ss = {k: v.float().clone().requires_grad_(k.startswith("sam_mask_decoder.")) for k, v in sam.items()}
        ids, labels = t(a[name + "_ids"])[0], t(a[name + "_labels"])
        k = ids.tolist().index(-201)
        emb = torch.cat([tab[ids[:k]], mm, tab[ids[k + 1:]]], 0)[None].clone().requires_grad_(True)
        lab = torch.cat([labels[0, :k], torch.full((mm.shape[0],), -100), labels[0, k + 1:]])[None]
        gt = t(a[name + "_gt"])
        r = O.training_losses(ws, TINY_LLM, emb, torch.ones(1, emb.shape[1], dtype=torch.long), lab, 299, ss, cfg, images_sam, [gt],
                              [torch.zeros(gt.shape[1:])], tuple(a["loss_weights"].tolist()))
        got = torch.stack([r[k_].detach().float() for k_ in ("loss", "ce_loss", "mask_bce_loss", "mask_dice_loss", "mask_loss")])
        assert torch.allclose(got, t(a[name + "_losses"]).float(), rtol=2e-5, atol=1e-6), name
        r["loss"].backward()
        assert rel_err(emb.grad, t(a[name + "_d_inputs_embeds"])) < 2e-4, name
        n_full = n_samp = 0
        for key in a:
            if key.startswith(name + "_g::"):
                pn = key[len(name) + 4:]
                if pn == "model.embed_tokens.weight":                     # our embeds are a leaf (gathered rows): compare through d_inputs_embeds above
                    continue
                assert ws[pn].grad is not None, pn
                assert rel_err(ws[pn].grad, t(a[key])) < 3e-4, (name, pn, rel_err(ws[pn].grad, t(a[key])))
                n_full += 1
            elif key.startswith(name + "_gs::"):
                pn = key[len(name) + 5:]
                g_ = ss[pn].grad
                assert g_ is not None, pn
                ref = t(a[key]).float()
                f = g_.reshape(-1)
                samp = f[torch.linspace(0, f.numel() - 1, min(97, f.numel())).long()]
                scale = float(ref[0]) + 1e-12
                # (1e-8 floor: e.g. the key biases of an attention have an analytically zero gradient -- softmax is shift-invariant)
                assert abs(float(g_.norm()) - float(ref[0])) < 3e-4 * scale + 1e-8, (name, pn)
                assert abs(float(g_.sum()) - float(ref[1])) < 3e-4 * scale * max(1.0, f.numel() ** 0.5) + 1e-8, (name, pn)
                assert float((samp - ref[2:]).abs().max()) < 3e-4 * float(g_.abs().max()) + 1e-8, (name, pn)
                n_samp += 1
            elif key.startswith(name + "_nograd::"):
                pn = key[len(name) + 9:]
                assert ss[pn].grad is None or float(ss[pn].grad.abs().max()) == 0.0, pn
        assert n_full >= 20 and n_samp >= 100, (n_full, n_samp)


def test_mask_loss_grad_bf16_storage_noise():
    """Why tests/test_seg_train_gpu.py compares the mask-branch gradients at 15 %, not at the 1e-2 of a forward pass: the fixture's ground
    truth is a random binary mask, so every gradient behind the resizes and the mask product is the remainder of a cancelling sum.  Measured
    here with the oracle's autograd: the fp32 graph against the SAME graph with bf16 weights and with op outputs (and hence, through
    autograd's cast rule, the gradients at the same points) rounded to bf16 -- the storage the HIP path and the reference's own bf16
    training use.  The HIP path measures the same per-tensor figures (text_hidden_fcs.0.0.weight: 9 % / 15 % here, 8 % / 15 % on the GPU).
    A 2x error of one ROW of that gradient is typical: the row is rank-1 in one element of a 256-term cancelling sum."""
    a, _ = load_golden("seg_grad_tiny")
    _, m, w, cfg, sam = _seg_setup()
    tab, e1 = w["model.embed_tokens.weight"].float(), t(m["sp_vid_only_nolab_emb"])[0]
    mm = e1[2:-3]
    real_f = O.F

    class RoundedOps:
        def __getattr__(self, n):
            f = getattr(real_f, n)
            if n in ("linear", "layer_norm", "gelu", "relu", "conv_transpose2d", "conv2d", "silu"):
                return lambda *a_, **k_: f(*a_, **k_).to(torch.bfloat16).float()
            return f

    def grads(rounded, name="two_obj"):
        ws = {k: v.float().clone().requires_grad_(k.startswith("model.text_hidden_fcs.")) for k, v in w.items()}
        ss = {k: v.float().clone().requires_grad_(k.startswith("sam_mask_decoder.")) for k, v in sam.items()}
        ids, labels = t(a[name + "_ids"])[0], t(a[name + "_labels"])
        k = ids.tolist().index(-201)
        emb = torch.cat([tab[ids[:k]], mm, tab[ids[k + 1:]]], 0)[None].clone()
        lab = torch.cat([labels[0, :k], torch.full((mm.shape[0],), -100), labels[0, k + 1:]])[None]
        gt = t(a[name + "_gt"])
        O.F = RoundedOps() if rounded else real_f
        O._MIRROR = bool(rounded)                          # ... and the weights are the bf16 copies the kernels read
        try:
            r = O.training_losses(ws, TINY_LLM, emb, torch.ones(1, emb.shape[1], dtype=torch.long), lab, 299, ss, cfg, t(a["images_sam"])[0], [gt],
                                  [torch.zeros(gt.shape[1:])], tuple(a["loss_weights"].tolist()))
        finally:
            O.F, O._MIRROR = real_f, False
        r["mask_loss"].backward()
        return {k: v.grad for k, v in {**ss, **ws}.items() if v.grad is not None and float(v.grad.norm()) > 1e-8}
    # "blob" (round 3, VERDICT r2 next #4a): a STRUCTURED ground truth (two discs, an empty and a full mask) was expected to remove the
    # amplification.  It does not: median 7 % (random masks: 10-11 %), text_hidden_fcs.0.0 12 % -- the cancellation sits in the CHANNEL
    # contraction of the hyper-network product ([pixels x 32] . [32]) and in the token-side MLPs, not in the spatial sign pattern of
    # d(loss)/d(logit).  So the GPU test holds the blob case to 20 % per tensor (the measured 9-15 % x 1.5), not to 5 %; what a 5 % bound
    # was meant to catch -- a wrong bilinear tap, a mis-scaled DICE term -- is caught by the per-kernel tests at 1e-5 (test_seg_train_gpu.py).
    for name, fcs0 in (("two_obj", 0.094), ("one_obj", 0.157), ("blob", 0.118)):
        g0, g1 = grads(False, name), grads(True, name)
        errs = {k: rel_err(g1[k], g0[k]) for k in g0}
        med = sorted(errs.values())[len(errs) // 2]
        assert 0.03 < med < 0.3 and max(errs.values()) > 0.15, (med, max(errs.values()))   # token-side MLPs / query projections: 20 % and more
        assert errs["sam_mask_decoder.output_upscaling.3.weight"] < 0.03           # the pixel path is benign: ~1 %
        assert abs(errs["model.text_hidden_fcs.0.0.weight"] - fcs0) < 0.4 * fcs0, errs["model.text_hidden_fcs.0.0.weight"]
        # noise adds in quadrature, so the NORMS stay put: that is the sharp check the GPU test keeps (4 % + a floor)
        for k in g0:
            if float(g0[k].norm()) > 0.1 * max(float(v.norm()) for v in g0.values()):
                assert abs(float(g1[k].norm()) / float(g0[k].norm()) - 1) < 0.05, k
