"""Golden vectors for the MASK-LOSS BACKWARD (rest): the reference's own `forward(inference=False)`
(videorefer_qwen2.py:198-352) on the tiny model with the seeded tiny SAM2 in place of SAM2-L, `train_mask_decoder=True`
semantics (sam_mask_decoder and text_hidden_fcs trainable, videorefer_arch.py:124-149), loss = ce + bce_w * BCE + dice_w * DICE,
`loss.backward()` through torch autograd -- RUN in the build container.  Saved: inputs, the five loss values, the [SEG] embeddings
and their gradient, d(loss)/d(last hidden state), d(loss)/d(inputs_embeds), the FULL gradients of text_hidden_fcs and of the
decoder, and for every sam_mask_decoder parameter (4.2 M values, regenerated from a stored seed on the test side) its gradient's
L2 norm, sum and a strided sample of 97 elements.  Test infrastructure only."""
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


def sample(t, n=97):
    f = t.detach().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return f[idx].clone()


def main():
    work = tempfile.mkdtemp(prefix="ufv_fx_seggrad_")
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
    dec_names, fcs_names, sam_names = [], [], []
    for n, p in model.named_parameters():
        dec = n.startswith("model.layers.") or n in ("model.norm.weight", "model.embed_tokens.weight", "lm_head.weight")
        fcs = n.startswith("model.text_hidden_fcs.")
        samd = n.startswith("model.mask_encoder.sam2_model.sam_mask_decoder.")
        p.requires_grad_(dec or fcs or samd)
        (dec_names if dec else fcs_names if fcs else sam_names if samd else []).append(n)
    video = torch.from_numpy(z["video"])
    g = torch.Generator().manual_seed(81)
    T = 4
    sam = torch.randn(1, T, 3, 128, 128, generator=g)
    cap = {}
    orig = model.prepare_inputs_labels_for_multimodal

    def spy(*a, **k):
        out = orig(*a, **k)
        out[3].retain_grad()
        cap["embeds"] = out[3]
        return out
    model.prepare_inputs_labels_for_multimodal = spy
    fcs_mod = model.get_model().text_hidden_fcs[0]

    def fcs_hook(mod, inp, outp):
        inp[0].retain_grad(); outp.retain_grad()
        cap["hidden"], cap["fcs_out"] = inp[0], outp
    fcs_mod.register_forward_hook(fcs_hook)
    out = {"images_sam": sam, "loss_weights": np.array([1.0, 2.0, 0.5]), "sam_seeds": np.array([GS.GH.SEEDS[k] for k in ("trunk", "neck", "heads")])}
    # "blob" (round 3): a STRUCTURED ground truth -- two discs, one empty and one full mask -- so that d(loss)/d(logit) has one sign over whole
    # regions and the gradients behind the resizes are sums that add up instead of remainders of cancelling ones (the random masks of the first
    # two cases amplify bf16 storage noise to 10-30 % per tensor); the test holds this case to 5 % per tensor and 2 % on norms.
    cases = {"two_obj": dict(ids=[[5, 6, -201, 7, 299, 9, 299, 11]], n_obj=2, hw=(40, 50)),
             "one_obj": dict(ids=[[5, 6, -201, 7, 8, 299, 11]], n_obj=1, hw=(33, 47)),
             "blob": dict(ids=[[5, 6, -201, 7, 8, 299, 11]], n_obj=1, hw=(40, 50), structured=True)}
    torch.set_grad_enabled(True)
    params = dict(model.named_parameters())
    for name, c in cases.items():
        model.zero_grad(set_to_none=True)
        ids = torch.tensor(c["ids"], dtype=torch.long)
        labels = ids.clone(); labels[labels < 0] = -100; labels[:, :2] = -100
        if c.get("structured"):
            yy, xx = torch.meshgrid(torch.arange(c["hw"][0]), torch.arange(c["hw"][1]), indexing="ij")
            disc = lambda cy, cx, r: (((yy - cy) ** 2 + (xx - cx) ** 2) < r * r).float()      # noqa: E731
            gt = torch.stack([disc(18, 22, 13), disc(24, 30, 9), torch.zeros(*c["hw"]), torch.ones(*c["hw"])])
            assert gt.shape[0] == T * c["n_obj"]
        else:
            gt = (torch.rand(T * c["n_obj"], *c["hw"], generator=g) > 0.5).float()
        r = model(input_ids=ids, attention_mask=torch.ones_like(ids), labels=labels, images=[(video, "video")], images_sam=sam,
                  offset=torch.tensor([0, 1]), masks_list=[gt], label_list=[torch.zeros(*c["hw"])], inference=False)
        r["loss"].backward()
        print(name, {k: float(v) for k, v in r.items()})
        out[name + "_ids"] = ids; out[name + "_labels"] = labels; out[name + "_gt"] = gt
        out[name + "_losses"] = np.array([float(r[k]) for k in ("loss", "ce_loss", "mask_bce_loss", "mask_dice_loss", "mask_loss")])
        out[name + "_d_inputs_embeds"] = cap["embeds"].grad.detach().clone()
        out[name + "_hidden_last"] = cap["hidden"].detach().clone()
        out[name + "_d_hidden_fcs"] = cap["hidden"].grad.detach().clone()          # d(loss)/d(last hidden): CE (lm_head reads the same tensor) + the mask terms through text_hidden_fcs
        out[name + "_fcs_out"] = cap["fcs_out"].detach().clone()
        out[name + "_d_fcs_out"] = cap["fcs_out"].grad.detach().clone()
        for n in fcs_names + dec_names:
            out[f"{name}_g::{n}"] = params[n].grad.detach().clone()
        for n in sam_names:
            gr = params[n].grad
            short = n[len("model.mask_encoder.sam2_model."):]
            if gr is None:                                   # not on the loss path (IoU / object-score heads: the mask is PICKED by arg-max IoU)
                out[f"{name}_nograd::{short}"] = np.array(1)
                continue
            out[f"{name}_gs::{short}"] = np.concatenate([np.array([float(gr.norm()), float(gr.sum())], dtype=np.float32), sample(gr).numpy()])
        print("   |d fcs_out|", float(cap["fcs_out"].grad.norm()), " |d hidden (fcs path)|", float(cap["hidden"].grad.norm()),
              " sam decoder params with grad:", len(sam_names))
    GF.npz("seg_grad_tiny", **out)


if __name__ == "__main__":
    main()
