"""Golden vectors for the decoder TRAINING STEP: the reference's own `forward(inference=False)`
(videorefer_qwen2.py:198-352) on the tiny model of model_tiny.npz, `ce_loss.backward()` through torch autograd, then
`clip_grad_norm_` + `torch.optim.AdamW.step()` (what HF Trainer / DeepSpeed apply, train.py:749) -- RUN in the build
container.  Saved: the spliced inputs_embeds / labels the reference fed its decoder, the loss, the gradient of every decoder
parameter and of inputs_embeds, and the parameters after one optimizer step.  Test infrastructure only."""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_fixtures as GF  # noqa: E402
import gen_fixtures_seg as GS  # noqa: E402

LR, WD, BETAS, EPS, CLIP = 1e-3, 0.01, (0.9, 0.999), 1e-8, 1.0


def main():
    work = tempfile.mkdtemp(prefix="ufv_fx_traingrad_")
    os.chdir(work)
    with torch.no_grad():
        tower = GF.SiglipVisionModel(GF.SiglipVisionConfig(**GF.TINY_VIT))
    tpath = os.path.join(work, "siglip-so400m-patch14-384")
    tower.save_pretrained(tpath)
    with open(os.path.join(tpath, "preprocessor_config.json"), "w") as f:
        json.dump({"image_processor_type": "SiglipImageProcessor", "size": {"height": 56, "width": 56}, "image_mean": [0.5, 0.5, 0.5],
                   "image_std": [0.5, 0.5, 0.5], "do_resize": True, "do_rescale": True, "do_normalize": True, "resample": 3,
                   "rescale_factor": 1 / 255}, f)
    with torch.no_grad():
        model, cfg, tok = GF.build_ref_model(work)
    z = np.load(os.path.join(GF.OUT, "model_tiny.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    res = model.load_state_dict(sd, strict=False)
    assert all("mask_encoder" in k for k in res.missing_keys) and not res.unexpected_keys
    wrap, _, _ = GS.tiny_sam_wrapper()
    model.get_model().mask_encoder = wrap
    model.config.seg_token_id = 299
    model.config.ce_loss_weight, model.config.bce_loss_weight, model.config.dice_loss_weight = 1.0, 2.0, 0.5
    model.train(False)
    # trainable set of this slice: the decoder (layers, norm, embed_tokens) and lm_head
    names = []
    for n, p in model.named_parameters():
        dec = n.startswith("model.layers.") or n in ("model.norm.weight", "model.embed_tokens.weight", "lm_head.weight")
        p.requires_grad_(dec)
        if dec:
            names.append(n)
    video = torch.from_numpy(z["video"])
    g = torch.Generator().manual_seed(71)
    sam = torch.randn(1, 4, 3, 128, 128, generator=g)
    ids = torch.tensor([[5, 6, -201, 7, 8, 9, 12, 13, 14, 9, 7]], dtype=torch.long)
    labels = ids.clone(); labels[labels < 0] = -100; labels[:, :2] = -100
    cap = {}
    orig = model.prepare_inputs_labels_for_multimodal

    def spy(*a, **k):
        out = orig(*a, **k)
        emb = out[3]
        emb.retain_grad()
        cap["embeds"], cap["labels"], cap["mask"] = emb, out[4], out[1]
        return out
    model.prepare_inputs_labels_for_multimodal = spy
    torch.set_grad_enabled(True)
    r = model(input_ids=ids, attention_mask=torch.ones_like(ids), labels=labels, images=[(video, "video")], images_sam=sam,
              offset=torch.tensor([0, 1]), masks_list=[torch.zeros(0, 20, 30)], label_list=[torch.zeros(20, 30)], inference=False)
    ce = r["ce_loss"]
    ce.backward()
    out = {"ids": ids, "labels_in": labels, "inputs_embeds": cap["embeds"].detach(), "labels": cap["labels"].detach(),
           "d_inputs_embeds": cap["embeds"].grad.detach(), "ce_loss": np.array(float(ce)),
           "hyper": np.array([LR, WD, BETAS[0], BETAS[1], EPS, CLIP])}
    params = dict(model.named_parameters())
    for n in names:
        out["g::" + n] = params[n].grad.detach().clone()
    # one optimizer step the way HF Trainer does it: clip to max_grad_norm, AdamW with no decay on biases / norm weights
    tr = [params[n] for n in names]
    norm = torch.nn.utils.clip_grad_norm_(tr, CLIP)
    out["grad_norm"] = np.array(float(norm))
    decay = [params[n] for n in names if params[n].ndim >= 2]
    nodecay = [params[n] for n in names if params[n].ndim < 2]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": WD}, {"params": nodecay, "weight_decay": 0.0}], lr=LR, betas=BETAS, eps=EPS)
    opt.step()
    for n in names:
        out["p1::" + n] = params[n].detach().clone()
    print("ce", float(ce), "grad norm", float(norm), "S", cap["embeds"].shape)
    GF.npz("train_grad_tiny", **out)


if __name__ == "__main__":
    main()
