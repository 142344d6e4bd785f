"""Golden vectors for the [SEG] -> mask branches of VideoReferQwen2ForCausalLM.generate (videorefer_qwen2.py:428-458 and
:461-518), produced by RUNNING the reference's own generate() in the build container.  The LLM / tower / projector are the
tiny reference model of gen_fixtures.py (weights taken from tests/golden/model_tiny.npz); its SAM2-L is swapped for the
tiny seeded SAM2VideoPredictor of gen_fixtures_sam2_heads.py so the vectors stay small.  CPU-only container: Tensor.cuda()
is made a no-op and the predictor state is kept on the CPU.  Test infrastructure only."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_fixtures as GF  # noqa: E402
import gen_fixtures_sam2_heads as GH  # noqa: E402
from oracle import ref_cpu as O  # noqa: E402
import ufvideo.model.sam2 as RS  # noqa: E402


def tiny_sam_wrapper():
    pred = GH.build_reference(128, cls=RS.SAM2VideoPredictor)
    sd = GH.seeded_weights()
    missing, unexpected = pred.load_state_dict(sd, strict=False)
    assert not unexpected
    w = RS.SAM2.__new__(RS.SAM2)
    torch.nn.Module.__init__(w)
    w.sam2_model, w.hidden_dim = pred, pred.hidden_dim
    w.img_mean, w.img_std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    init = pred.init_state

    def init_cpu(images):
        st = init(images)
        st["device"] = st["storage_device"] = torch.device("cpu")
        return st
    pred.init_state = init_cpu
    logits_log = []
    infer = w.language_embd_inference

    def infer_logged(state, emb):
        out = infer(state, emb)
        logits_log.append(out.float().clone())
        return out
    w.language_embd_inference = infer_logged
    return w, sd, logits_log


def main():
    torch.set_grad_enabled(False)
    torch.Tensor.cuda = lambda self, *a, **k: self
    work = tempfile.mkdtemp(prefix="ufv_fx_seg_")
    os.chdir(work)
    # the reference resolves its tower from a cwd-relative directory (encoder.py:108); weights are overwritten below
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
    wrap, sam_sd, logits_log = tiny_sam_wrapper()
    model.get_model().mask_encoder = wrap
    video = torch.from_numpy(z["video"])
    sam = torch.randn(1, 2, 3, 128, 128, generator=torch.Generator().manual_seed(51))
    out = {"images_sam": sam}
    ids = torch.tensor([[5, 6, -201, 7, 8, 9]], dtype=torch.long)
    base = torch.from_numpy(z["gen2_tokens"])[0].tolist()
    print("tokens of the QA run:", base)

    def gen(seg_id, inp=ids, label=torch.zeros(40, 50), n=6):
        model.config.seg_token_id = seg_id
        logits_log.clear()
        return model.generate(inp, attention_mask=torch.ones_like(inp), images=[(video, "video")], images_sam=sam, offset=[0, 1],
                              label_list=label, do_sample=False, max_new_tokens=n, use_cache=True, pad_token_id=0, eos_token_id=298)

    # (1) generated [SEG]: the tiny model emits the same token at every step, so with that id as [SEG] every step o with
    # token o+1 == [SEG] contributes -- step 0 with the hidden states of the WHOLE prompt (HF's hidden_states[0][-1] is
    # [1, S, D]), later steps with one state each.
    seg = base[1]
    g = gen(seg)
    assert g["output"][0].tolist() == base
    hits = [o for o in range(len(base) - 1) if base[o + 1] == seg]
    print("gen: seg id", seg, "hits", hits, "n masks", len(g["pred_masks"]), "mask shape", tuple(g["pred_masks"][0].shape))
    keep = [0, 30, 58, 59, len(g["pred_masks"]) - 1]
    out["gen_seg_id"] = np.int64(seg)
    out["gen_masks"] = torch.stack(g["pred_masks"]).numpy()                               # [n, T, 40, 50] bool
    out["gen_logits_idx"] = np.array(keep)
    out["gen_logits"] = torch.stack([logits_log[i][:, 0] for i in keep]).numpy()          # video-res logits of a few of them
    # (2) [SEG] in the prompt's trailing text
    ids2 = torch.tensor([[5, 6, -201, 7, 299, 9, 299]], dtype=torch.long)
    g = gen(299, ids2, [torch.zeros(33, 47)])
    print("prompt branch: n masks", len(g["pred_masks"]), tuple(g["pred_masks"][0].shape), "logits calls", len(logits_log))
    out["prompt_ids"] = ids2
    out["prompt_masks"] = torch.stack(g["pred_masks"]).numpy()
    out["prompt_logits"] = torch.stack([l[:, 0] for l in logits_log]).numpy()
    out["prompt_hidden_last"] = g["output"].hidden_states[-1]
    out["sam_seeds"] = np.array([GH.SEEDS[k] for k in ("trunk", "neck", "heads")])
    GF.npz("seg_tiny", **out)


if __name__ == "__main__":
    main()
