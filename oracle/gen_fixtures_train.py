"""Golden vectors for the training-loss forward `VideoReferQwen2ForCausalLM.forward(inference=False)`
(videorefer_qwen2.py:198-352: CE + mask BCE + DICE through `get_sam2_embeddings_train` / `inject_language_embd_train`),
produced by RUNNING the reference in the build container: tiny LLM/tower/projector from model_tiny.npz, the seeded tiny
SAM2 of gen_fixtures_sam2_heads.py in place of SAM2-L.  Forward values only (no backward).  Test infrastructure only."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_fixtures as GF  # noqa: E402
import gen_fixtures_seg as GS  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    work = tempfile.mkdtemp(prefix="ufv_fx_train_")
    os.chdir(work)
    import json
    tower = GF.SiglipVisionModel(GF.SiglipVisionConfig(**GF.TINY_VIT))
    tpath = os.path.join(work, "siglip-so400m-patch14-384")
    tower.save_pretrained(tpath)
    with open(os.path.join(tpath, "preprocessor_config.json"), "w") as f:
        json.dump({"image_processor_type": "SiglipImageProcessor", "size": {"height": 56, "width": 56}, "image_mean": [0.5, 0.5, 0.5],
                   "image_std": [0.5, 0.5, 0.5], "do_resize": True, "do_rescale": True, "do_normalize": True, "resample": 3,
                   "rescale_factor": 1 / 255}, f)
    model, cfg, tok = GF.build_ref_model(work)
    z = np.load(os.path.join(GF.OUT, "model_tiny.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    res = model.load_state_dict(sd, strict=False)
    assert all("mask_encoder" in k for k in res.missing_keys) and not res.unexpected_keys
    wrap, sam_sd, _ = GS.tiny_sam_wrapper()
    model.get_model().mask_encoder = wrap
    model.config.seg_token_id = 299
    model.config.ce_loss_weight, model.config.bce_loss_weight, model.config.dice_loss_weight = 1.0, 2.0, 0.5
    video = torch.from_numpy(z["video"])
    g = torch.Generator().manual_seed(61)
    T = 4                                                    # the reference hard-codes num_frames_sam = 4 (videorefer_qwen2.py:157)
    sam = torch.randn(1, T, 3, 128, 128, generator=g)
    out = {"images_sam": sam}
    cases = {
        "two_obj": dict(ids=[[5, 6, -201, 7, 299, 9, 299, 11]], n_obj=2, hw=(40, 50)),
        "one_obj": dict(ids=[[5, 6, -201, 7, 8, 299, 11]], n_obj=1, hw=(33, 47)),
        "no_seg": dict(ids=[[5, 6, -201, 7, 8, 9]], n_obj=0, hw=(20, 30)),
    }
    for name, c in cases.items():
        ids = torch.tensor(c["ids"], dtype=torch.long)
        labels = ids.clone(); labels[labels < 0] = -100; labels[:, :2] = -100
        gt = (torch.rand(T * c["n_obj"], *c["hw"], generator=g) > 0.5).float()
        r = model(input_ids=ids, attention_mask=torch.ones_like(ids), labels=labels, images=[(video, "video")], images_sam=sam,
                  offset=torch.tensor([0, 1]), masks_list=[gt], label_list=[torch.zeros(*c["hw"])], inference=False)
        print(name, {k: float(v) for k, v in r.items()})
        out[name + "_ids"] = ids; out[name + "_labels"] = labels; out[name + "_gt"] = gt
        out[name + "_losses"] = np.array([float(r[k]) for k in ("loss", "ce_loss", "mask_bce_loss", "mask_dice_loss", "mask_loss")])
    out["loss_weights"] = np.array([1.0, 2.0, 0.5])
    out["sam_seeds"] = np.array([GS.GH.SEEDS[k] for k in ("trunk", "neck", "heads")])
    GF.npz("train_tiny", **out)


if __name__ == "__main__":
    main()
