"""Golden vectors for the SAM2 heads driven the way the reference's inference drives them
(SAM2Base.forward_image -> _prepare_backbone_features -> track_step(is_init_cond_frame=True, language_embd=...)),
produced by RUNNING the reference classes in the build container with seeded weights.  Test infrastructure only."""
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

TINY = dict(embed_dim=16, num_heads=1, stages=(1, 2, 3, 1), global_att_blocks=(4,), window_spec=(8, 4, 8, 4),
            window_pos_embed_bkg_spatial_size=(7, 7), d_model=256)
SEEDS = dict(trunk=40, neck=41, heads=42, x=43, lang=44)


def build_reference(image_size=128, cls=None):
    trunk = RS.Hiera(**{k: v for k, v in TINY.items() if k != "d_model"})
    neck = RS.FpnNeck(position_encoding=RS.PositionEmbeddingSine(num_pos_feats=256), d_model=256, backbone_channel_list=trunk.channel_list,
                      fpn_top_down_levels=[2, 3], fpn_interp_model="nearest")
    enc = RS.ImageEncoder(trunk=trunk, neck=neck, scalp=1)
    w = RS.SAM2.__new__(RS.SAM2); torch.nn.Module.__init__(w)
    m = (cls or RS.SAM2Base)(image_encoder=enc, memory_attention=RS.SAM2.build_memory_attention(w), memory_encoder=RS.SAM2.build_memory_encoder(w),
                    num_maskmem=7, image_size=image_size, sigmoid_scale_for_mem_enc=20.0, sigmoid_bias_for_mem_enc=-10.0,
                    use_mask_input_as_output_without_sam=True, directly_add_no_mem_embed=True, use_high_res_features_in_sam=True,
                    multimask_output_in_sam=True, iou_prediction_use_sigmoid=True, use_obj_ptrs_in_encoder=True, add_tpos_enc_to_obj_ptrs=False,
                    only_obj_ptrs_in_the_past_for_eval=True, pred_obj_scores=True, pred_obj_scores_mlp=True, fixed_no_obj_ptr=True,
                    multimask_output_for_tracking=True, use_multimask_token_for_obj_ptr=True, multimask_min_pt_num=0, multimask_max_pt_num=1,
                    use_mlp_for_obj_ptr_proj=True, compile_image_encoder=False,
                    sam_mask_decoder_extra_args={"dynamic_multimask_via_stability": True, "dynamic_multimask_stability_delta": 0.05,
                                                 "dynamic_multimask_stability_thresh": 0.98}).eval()
    return m


def seeded_weights():
    sd = {}
    sd.update(O.make_hiera_weights(TINY, seed=SEEDS["trunk"], prefix="image_encoder.trunk."))
    sd.update(O.make_fpn_weights([128, 64, 32, 16], 256, seed=SEEDS["neck"], prefix="image_encoder.neck."))
    sd.update(O.make_sam_head_weights(256, seed=SEEDS["heads"]))
    return sd


def main():
    torch.set_grad_enabled(False)
    m = build_reference()
    sd = seeded_weights()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert not [k for k in missing if k.startswith(("image_encoder", "sam_mask_decoder", "sam_prompt_encoder", "no_mem_embed"))
                and "mask_downscaling" not in k], missing
    x = torch.randn(2, 3, 128, 128, generator=torch.Generator().manual_seed(SEEDS["x"]))
    lang = torch.randn(2, 1, 256, generator=torch.Generator().manual_seed(SEEDS["lang"]))
    bo = m.forward_image(x)
    _, vf, vp, fs = m._prepare_backbone_features(bo)
    out = m.track_step(frame_idx=0, is_init_cond_frame=True, current_vision_feats=vf, current_vision_pos_embeds=vp, feat_sizes=fs,
                       point_inputs=None, mask_inputs=None, output_dict={"cond_frame_outputs": {}, "non_cond_frame_outputs": {}}, num_frames=2,
                       run_mem_encoder=False, language_embd=lang)
    hi = [f.permute(1, 2, 0).view(f.size(1), f.size(2), *s) for f, s in zip(vf[:-1], fs[:-1])]
    pix = (vf[-1] + m.no_mem_embed).permute(1, 2, 0).view(2, 256, *fs[-1])
    heads = m._forward_sam_heads(backbone_features=pix, high_res_features=hi, multimask_output=True, language_embd=lang)
    video = torch.nn.functional.interpolate(out["pred_masks"], size=(128, 128), mode="bilinear", align_corners=False)
    o = O.sam2_language_masks(sd, TINY, x, lang)
    rel = lambda a, b: ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()
    print("low_res_masks rel", rel(o["low_res_masks"], out["pred_masks"]), "multimasks", rel(o["low_res_multimasks"], heads[0]),
          "ious", rel(o["ious"], heads[2]), "obj", rel(o["object_score_logits"], heads[6]), "video", rel(o["video_res_masks"], video))
    assert rel(o["low_res_masks"], out["pred_masks"]) < 1e-4 and rel(o["ious"], heads[2]) < 1e-4 and rel(o["video_res_masks"], video) < 1e-4
    assert rel(o["high_res_masks"], out["pred_masks_high_res"]) < 1e-4
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sam2_heads_tiny.npz"), x=x.numpy(), lang=lang.numpy(),
                        pred_masks=out["pred_masks"].numpy(), high_res=out["pred_masks_high_res"].numpy(), multimasks=heads[0].numpy(),
                        ious=heads[2].numpy(), obj=heads[6].numpy(), video_res=video.numpy(),
                        seeds=np.array([SEEDS[k] for k in ("trunk", "neck", "heads", "x", "lang")]))
    print("wrote sam2_heads_tiny.npz")


if __name__ == "__main__":
    main()
