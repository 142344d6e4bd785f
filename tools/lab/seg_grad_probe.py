"""lab: per-parameter error of SegHeadGrad against the reference-backward golden (prints, asserts nothing)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from conftest import load_golden, t, rel_err
from test_seg_train_gpu import _seg_model
from ufvideo_amd.train_seg import SegHeadGrad
name = sys.argv[1] if len(sys.argv) > 1 else "two_obj"
a, m, arrs, w = _seg_model()
seg = SegHeadGrad(m)
leaves = {k: v.detach().float().clone().requires_grad_(True) for k, v in SegHeadGrad.trainable(m)}
labels = t(a[name + "_labels"])[0]; ids = t(a[name + "_ids"])[0].tolist(); hidden = t(a[name + "_hidden_last"])[0]
S = hidden.shape[0]; k = ids.index(-201)
lab = torch.cat([labels[:k], torch.full((S - len(ids) + 1,), -100), labels[k + 1:]])
shifted = torch.cat([lab[1:], torch.full((1,), -100)])
rows = torch.nonzero(shifted == 299).reshape(-1)
gt = t(a[name + "_gt"])
hid = hidden[rows].to("cuda").requires_grad_(True)
_, w_bce, w_dice = a["loss_weights"].tolist()
bce, dice = seg.forward_backward(leaves, hid, t(a["images_sam"])[0].to("cuda"), gt, tuple(gt.shape[1:]), w_bce, w_dice, gt.shape[0])
print("losses", float(bce), float(dice), a[name + "_losses"])
for key in a:
    if key.startswith(name + "_g::model.text_hidden_fcs."):
        pn = key[len(name) + 4 + len("model."):]
        print(f"{rel_err(leaves[pn].grad.cpu(), t(a[key])):.4f}  {pn}")
rowsum = []
for key in a:
    if key.startswith(name + "_gs::"):
        pn = "mask_encoder.sam2_model." + key[len(name) + 5:]
        ref = t(a[key]).float(); g = leaves[pn].grad
        if g is None:
            print("NONE", pn); continue
        g = g.float().cpu(); f = g.reshape(-1)
        samp = f[torch.linspace(0, f.numel() - 1, min(97, f.numel())).long()]
        print(f"norm {float(g.norm()):.3e} ref {float(ref[0]):.3e}  sum {float(g.sum()):.3e} ref {float(ref[1]):.3e}  samp_err {float((samp-ref[2:]).norm()/(ref[2:].norm()+1e-12)):.4f}  {pn[len('mask_encoder.sam2_model.sam_mask_decoder.'):]}")
