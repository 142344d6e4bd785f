"""lab: the mask decoder's gradient tensors against the reference-backward golden -- per-tensor norm deviation (the quantity test_seg_train_gpu held at 4 %), the margin each
tensor leaves to that bound, and the robust aggregate (cosine / norm of the concatenated samples).  Three cases, three repeats each (prints, asserts nothing)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from conftest import load_golden, t, rel_err
from test_seg_train_gpu import _seg_model
from ufvideo_amd.train_seg import SegHeadGrad
for name in ("two_obj", "one_obj", "blob"):
    a, m, arrs, w = _seg_model()
    seg = SegHeadGrad(m)
    labels = t(a[name + "_labels"])[0]; ids = t(a[name + "_ids"])[0].tolist(); hidden = t(a[name + "_hidden_last"])[0]
    S = hidden.shape[0]; k = ids.index(-201)
    lab = torch.cat([labels[:k], torch.full((S - len(ids) + 1,), -100), labels[k + 1:]])
    shifted = torch.cat([lab[1:], torch.full((1,), -100)])
    rows = torch.nonzero(shifted == 299).reshape(-1)
    gt = t(a[name + "_gt"])
    _, w_bce, w_dice = a["loss_weights"].tolist()
    for rep in range(2):
        leaves = {k: v.detach().float().clone().requires_grad_(True) for k, v in SegHeadGrad.trainable(m)}
        hid = hidden[rows].to("cuda").requires_grad_(True)
        seg.forward_backward(leaves, hid, t(a["images_sam"])[0].to("cuda"), gt, tuple(gt.shape[1:]), w_bce, w_dice, gt.shape[0])
        scale = max(float(a[k][0]) for k in a if k.startswith(name + "_gs::"))
        rowsn, allg, allr, n2g, n2r = [], [], [], 0.0, 0.0
        for key in a:
            if key.startswith(name + "_gs::"):
                pn = "mask_encoder.sam2_model." + key[len(name) + 5:]
                ref = t(a[key]).float(); g = leaves[pn].grad.float().cpu(); f = g.reshape(-1)
                samp = f[torch.linspace(0, f.numel() - 1, min(97, f.numel())).long()]
                dev = abs(float(g.norm()) - float(ref[0]))
                bound = 0.04 * float(ref[0]) + 2e-3 * scale
                rowsn.append((dev / bound, dev / (float(ref[0]) + 1e-30), float(ref[0]) / scale, float((samp - ref[2:]).norm() / (ref[2:].norm() + 1e-20)), pn[len('mask_encoder.sam2_model.sam_mask_decoder.'):]))
                allg.append(samp); allr.append(ref[2:]); n2g += float(g.norm()) ** 2; n2r += float(ref[0]) ** 2
        rowsn.sort(reverse=True)
        G, R = torch.cat(allg).double(), torch.cat(allr).double()
        cos = float(torch.dot(G, R) / (G.norm() * R.norm()))
        print(f"{name} rep {rep}: tensors {len(rowsn)}; worst use of the 4 % norm bound: " + "; ".join(f"{r[0]:.2f} (dev {r[1]*100:.1f} %, |g|/max {r[2]:.3f}, samp err {r[3]:.2f}) {r[4]}" for r in rowsn[:4]))
        print(f"   aggregate: 1 - cos(samples) {1 - cos:.2e}; sample norm ratio {float(G.norm() / R.norm()):.4f}; whole-gradient norm ratio {(n2g / n2r) ** 0.5:.4f}; tensors over 0.8 of the bound: {sum(1 for r in rowsn if r[0] > 0.8)}", flush=True)
