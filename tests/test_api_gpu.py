"""GPU: the reference's top-level call surface end to end -- a tiny checkpoint directory (config.json + safetensors +
tokenizer files) -> `model_init` -> `mm_infer` for the QA branch and the [SEG] branch, checked against the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden, t  # noqa: E402
from oracle import ref_cpu as O  # noqa: E402

TINY_LLM = dict(vocab_size=300, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, rope_theta=10000.0, rms_norm_eps=1e-6)
TINY_VIT = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4, image_size=56, patch_size=14)
SAM_TINY = dict(embed_dim=16, num_heads=1, stages=(1, 2, 3, 1), global_att_blocks=(4,), window_spec=(8, 4, 8, 4),
                window_pos_embed_bkg_spatial_size=(7, 7), image_size=128)


def _write_checkpoint(path, weights, sam_seeds):
    from safetensors.torch import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    cfg = dict(TINY_LLM, model_type="videorefer_qwen2", mm_vision_tower="siglip", mm_vision_select_layer=-2,
               mm_vision_select_feature="patch", mm_projector_type="spatial_conv", mm_hidden_size=64, mm_region_encoder_type="pooling",
               image_aspect_ratio="square", num_frames=4, seg_token_id=299, vision_config=TINY_VIT, sam2_trunk=SAM_TINY,
               eos_token_id=281, pad_token_id=282)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    scfg = dict({k: v for k, v in SAM_TINY.items() if k != "image_size"}, d_model=256)
    sam = {}
    sam.update(O.make_hiera_weights(scfg, seed=sam_seeds[0], prefix="image_encoder.trunk."))
    sam.update(O.make_fpn_weights([128, 64, 32, 16], 256, seed=sam_seeds[1], prefix="image_encoder.neck."))
    sam.update(O.make_sam_head_weights(256, seed=sam_seeds[2]))
    sd = {k: v.contiguous() for k, v in weights.items()}
    sd.update({"model.mask_encoder.sam2_model." + k: v.contiguous() for k, v in sam.items()})
    save_file(sd, os.path.join(path, "model.safetensors"))
    vocab = {f"t{i}": i for i in range(280)}
    vocab.update({"<unk>": 280, "<eos>": 281, "<pad>": 282, "user": 283, "assistant": 284, "[SEG]": 299})
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", eos_token="<eos>", pad_token="<pad>")
    fast.chat_template = ("{% for m in messages %}{{ m['role'] }} {{ m['content'] }} {% endfor %}"
                          "{% if add_generation_prompt %}assistant {% endif %}")
    fast.save_pretrained(path)
    return sam, scfg


def test_model_init_and_mm_infer_end_to_end(tmp_path):
    from ufvideo_amd import model_init, mm_infer
    a, w = load_golden("model_tiny")
    sg, _ = load_golden("seg_tiny")
    sam_sd, sam_cfg = _write_checkpoint(str(tmp_path), w, sg["sam_seeds"].tolist())
    model, processor, tokenizer = model_init(str(tmp_path))
    assert processor is model.get_vision_tower().image_processor and model.get_model().mask_encoder is not None
    for m in model.modules():
        m.tokenizer = tokenizer
    video = t(a["video"])
    sam_frames = t(sg["images_sam"])
    # ---- QA branch: greedy text, checked against the oracle's greedy loop over the same spliced prompt
    text, out = mm_infer(video, "t5 t6", model, tokenizer, modal="video", images_sam=sam_frames.cuda(), offset=[0, 1],
                         label_list=torch.zeros(40, 50), max_new_tokens=6)
    ids = tokenizer("user").input_ids + [-201] + tokenizer("t5 t6 assistant").input_ids
    tab = w["model.embed_tokens.weight"].float()
    mmf = model.encode_images_or_videos([(video.cuda(), "video")])[0].float().cpu()          # tower/projector parity is covered elsewhere
    k = ids.index(-201)
    emb = torch.cat([tab[torch.tensor(ids[:k])], mmf, tab[torch.tensor(ids[k + 1:])]], 0)[None]
    toks, hid = O.greedy_generate(w, TINY_LLM, emb, torch.ones(1, emb.shape[1], dtype=torch.long), 6, eos_token_ids=(281,),
                                  stop_fn=None)
    assert out["output"][0].tolist() == toks[0].tolist()
    assert text == tokenizer.batch_decode(toks, skip_special_tokens=True)[0].strip() and out["pred_masks"] == []
    # ---- [SEG] in the instruction: masks for the 2 SAM frames at the label size, vs the oracle glue on the same hidden states
    res = mm_infer(video, "t5 [SEG] t6", model, tokenizer, modal="video", images_sam=sam_frames.cuda(), offset=[0, 1],
                   label_list=[torch.zeros(33, 47)], seg=True, max_new_tokens=2)
    pm = res["pred_masks"][0]
    assert pm.shape == (2, 33, 47) and pm.dtype == torch.bool and res["gt_masks"] is None
    ids2 = tokenizer("user").input_ids + [-201] + tokenizer("t5 [SEG] t6 assistant").input_ids
    k = ids2.index(-201)
    emb2 = torch.cat([tab[torch.tensor(ids2[:k])], mmf, tab[torch.tensor(ids2[k + 1:])]], 0)[None]
    hl = O.qwen2_forward(w, TINY_LLM, emb2, torch.ones(1, emb2.shape[1], dtype=torch.long), None)["hidden_states"][-1]
    ids2_t = torch.tensor([ids2])
    ref_mask, ref_logits = O.seg_masks_prompt(w, ids2_t, [emb2.shape[1] - (len(ids2) - k - 1), len(ids2) - k - 1], hl, 299, sam_sd, sam_cfg,
                                              sam_frames[0], (33, 47))
    big = torch.nn.functional.interpolate(ref_logits[:, None], size=(33, 47), mode="bilinear", align_corners=False)[:, 0]
    sure = big.abs() > 0.06 * ref_logits.abs().max()
    assert ((pm.cpu() != ref_mask) & sure).sum() == 0 and sure.float().mean() > 0.5
