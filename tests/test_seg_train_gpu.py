"""GPU: the mask-loss backward (rest).
  * the new kernels of csrc/seg_train.hip against torch autograd of the same op;
  * `SegHeadGrad` (text_hidden_fcs -> two-way transformer -> upscaling -> mask -> resizes -> BCE + DICE, forward + backward) on
    the seeded tiny SAM2 against the REFERENCE's own backward (tests/golden/seg_grad_tiny.npz, oracle/gen_fixtures_seg_grad.py):
    loss values, d(loss)/d(hidden state), every gradient of text_hidden_fcs, norm / sample of every sam_mask_decoder gradient;
  * `DecoderTrainer(train_seg_head=True).train_step` on the same sample: total loss and the decoder's gradients (CE + mask terms)
    vs the reference.
Tolerances: bf16 activations between kernels, as in the rest of the training path (tests/test_train_gpu.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from conftest import load_golden, t, rel_err  # noqa: E402
from oracle import ref_cpu as O  # noqa: E402
from ufvideo_amd import ops  # noqa: E402
from test_model_gpu import tiny_model, SAM_TINY  # noqa: E402

DEV = "cuda"
# Gradient tolerance of the mask branch.  The ground truth of the fixture is a RANDOM binary mask, so d(loss)/d(logit) is sign-random per
# pixel and every gradient behind the two bilinear resizes and the [pixels x channels] mask product is the small remainder of a heavily
# cancelling sum: bf16 storage of activations and gradients between the kernels (what the reference's own bf16 training does as well)
# shows up amplified.  tests/test_oracle_golden.py::test_mask_loss_grad_bf16_storage_noise measures that amplification on the CPU with
# the oracle's autograd (fp32 graph vs the same graph with bf16 weights and every op output and gradient rounded to bf16): 1 % (upscaling
# path) ... 30 % (token-side MLP, query projections) per tensor, text_hidden_fcs 9 % / 15 % -- the figures the HIP path measures on the
# same tensors (1 % ... 25 %, text_hidden_fcs 8 % / 15 %).  Noise adds in quadrature, so gradient NORMS are compared at 4 %.  Tensors with
# a tiny gradient are judged against the decoder's largest gradient norm.
GRAD_TOL = 0.3
# The third fixture case ("blob", round 3) has a STRUCTURED ground truth (two discs, an empty and a full mask): d(loss)/d(logit) keeps its sign
# over whole regions and |d fcs_out| is 10x the random-mask cases'.  Measured on MI355X: 9-15 % per tensor, the same as the random masks, and the
# oracle's own fp32-vs-bf16-storage experiment gives the same figures for it (7 % median, text_hidden_fcs.0.0 12 %; HIP 12.4 %): the amplification
# comes from the channel contraction of the hyper-network product and the token-side MLPs, not from the spatial sign pattern.  The case is held to
# 20 % per tensor / 4 % on norms; the sharp checks (bilinear taps, BCE / DICE scaling: 1e-5 vs autograd) are the per-kernel tests above.
TOL = {"two_obj": (GRAD_TOL, 0.04), "one_obj": (GRAD_TOL, 0.04), "blob": (0.2, 0.04)}


def bfr(x):
    return x.to(torch.bfloat16).float()


@pytest.mark.parametrize("B,H,Nq,Nk,hd", [(3, 8, 9, 9, 32), (2, 8, 9, 200, 16), (2, 8, 200, 9, 16), (1, 2, 70, 70, 32), (4, 8, 9, 4096, 16)])
def test_small_attention_fwd_bwd_vs_autograd(B, H, Nq, Nk, hd):
    g = torch.Generator().manual_seed(Nq * 7 + Nk)
    q, k, v = (bfr(torch.randn(B * n, H * hd, generator=g)) for n in (Nq, Nk, Nk))
    dO = bfr(torch.randn(B * Nq, H * hd, generator=g))
    qa, ka, va = (x.clone().requires_grad_(True) for x in (q, k, v))
    sp = lambda x, n: x.view(B, n, H, hd).transpose(1, 2)      # noqa: E731
    att = torch.softmax(sp(qa, Nq) @ sp(ka, Nk).transpose(-1, -2) * hd ** -0.5, -1)
    ref = (att @ sp(va, Nk)).transpose(1, 2).reshape(B * Nq, H * hd)
    ref.backward(dO)
    dev = lambda x: x.to(DEV).to(torch.bfloat16)               # noqa: E731
    o, lse = ops.small_attn_fwd(dev(q), dev(k), dev(v), B, H, Nq, Nk, hd)
    assert rel_err(o.float().cpu(), ref.detach()) < 5e-3
    s = sp(q, Nq) @ sp(k, Nk).transpose(-1, -2) * hd ** -0.5
    assert rel_err(lse.cpu(), torch.logsumexp(s, -1)) < 1e-5
    dq, dk, dv = ops.small_attn_bwd(dev(q), dev(k), dev(v), o, dev(dO), lse, B, H, Nq, Nk, hd)
    for got, want, nm in ((dq, qa.grad, "dq"), (dk, ka.grad, "dk"), (dv, va.grad, "dv")):
        assert rel_err(got.float().cpu(), want) < 1.5e-2, (nm, rel_err(got.float().cpu(), want))


def test_mask_dot_resize_and_loss_gradients_vs_autograd():
    g = torch.Generator().manual_seed(3)
    B, P, C = 3, 1000, 32
    up = bfr(torch.randn(B * P, C, generator=g)); h = torch.randn(B, C, generator=g); dm = torch.randn(B, P, generator=g)
    ua, ha = up.clone().requires_grad_(True), h.clone().requires_grad_(True)
    ref = torch.einsum("bpc,bc->bp", ua.view(B, P, C), ha)
    ref.backward(dm)
    upd = up.to(DEV).to(torch.bfloat16)
    out = ops.mask_dot_fwd(upd, h.to(DEV), B, P)
    assert rel_err(out.cpu(), ref.detach()) < 1e-5
    dup, dh = ops.mask_dot_bwd(upd, h.to(DEV), dm.to(DEV), B, P)
    assert rel_err(dup.float().cpu(), ua.grad) < 5e-3 and rel_err(dh.cpu(), ha.grad) < 1e-5
    # bilinear backward == autograd of F.interpolate, up- and down-scaling, odd sizes, size-1 borders
    for (hi, wi), (ho, wo) in (((16, 16), (64, 64)), ((32, 32), (20, 27)), ((128, 128), (40, 50)), ((9, 13), (9, 13)), ((5, 7), (33, 2)), ((64, 64), (1024, 1024))):
        x = torch.randn(2, 1, hi, wi, generator=g, requires_grad=True)
        y = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=False)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        got = ops.resize_bilinear_bwd(dy.to(DEV), (hi, wi))
        assert rel_err(got.cpu(), x.grad) < 1e-5, ((hi, wi), (ho, wo), rel_err(got.cpu(), x.grad))
    # BCE-with-logits + DICE exactly as the reference combines them (videorefer_qwen2.py:34-77, 319-338)
    n, hh, ww = 4, 33, 47
    x = (torch.randn(n, hh, ww, generator=g) * 3).requires_grad_(True)
    tg = (torch.rand(n, hh, ww, generator=g) > 0.5).float()
    w_bce, w_dice, total = 2.0, 0.5, 6
    loss = (w_bce * O.sigmoid_ce_loss(x, tg, n) * n + w_dice * O.dice_loss(x, tg, n) * n) / (total + 1e-8)
    loss.backward()
    xd, td = x.detach().to(DEV).contiguous(), tg.to(DEV)
    sums = ops.mask_loss_sums(xd, td)
    k = n / (n + 1e-8) / (total + 1e-8)
    num = 2.0 * sums[:, 1] / 1000.0 + 1e-6
    den = sums[:, 2] / 1000.0 + sums[:, 3] / 1000.0 + 1e-6
    coef = torch.stack([-w_dice * k * 2.0 / (1000.0 * den), w_dice * k * num / (1000.0 * den * den)], 1).contiguous()
    dx = ops.mask_loss_bwd(xd, td, coef, w_bce * k / (hh * ww))
    assert rel_err(dx.cpu(), x.grad) < 1e-5
    val = w_bce * k * (sums[:, 0] / (hh * ww)).sum() + w_dice * k * (1.0 - num / den).sum()
    assert abs(float(val) - float(loss.detach())) < 1e-5 * abs(float(loss.detach()))


def _seg_model():
    a, _ = load_golden("seg_grad_tiny")
    m, arrs, w = tiny_model(sam2_trunk=dict(SAM_TINY, image_size=128), sam_seeds=a["sam_seeds"].tolist())
    m.config.seg_token_id = 299
    m.config.ce_loss_weight, m.config.bce_loss_weight, m.config.dice_loss_weight = a["loss_weights"].tolist()
    return a, m, arrs, w


def _check_sam_grads(a, name, leaves, tol, ntol=0.04):
    """Per tensor: 97 sampled elements within `tol` (the bf16-storage noise class documented above), the norm within `ntol` for the tensors that carry the gradient
    (|g| >= a quarter of the decoder's largest: measured <= 2.5 %) and within 3 x ntol for the small ones -- layer 0's q / k projections sit at 9 % of the largest norm and
    4-6 % off (tools/lab/seg_grad_norms.py; round 5 had ONE red run of this test at 7.1 % against the then-uniform 4 % + slack: a rounding moved by the small-M kernel
    of that hour across a bound the tensor had always used 90-100 % of -- a fragile bound, not a wrong gradient).  The sharp statement is the AGGREGATE one: the
    concatenated samples of all 119 tensors point where the reference's do (1 - cos <= 1e-2; measured 1.4e-3 .. 3.3e-3) and the whole gradient's norm is within 2 %
    (measured 0.2-0.7 %)."""
    n_checked = 0
    scale = max(float(a[k][0]) for k in a if k.startswith(name + "_gs::"))           # the largest gradient norm of the decoder
    allg, allr, n2g, n2r = [], [], 0.0, 0.0
    for key in a:
        if key.startswith(name + "_gs::"):
            pn = "mask_encoder.sam2_model." + key[len(name) + 5:]
            ref = t(a[key]).float()
            g = leaves[pn].grad
            assert g is not None, pn
            g = g.float().cpu()
            f = g.reshape(-1)
            samp = f[torch.linspace(0, f.numel() - 1, min(97, f.numel())).long()]
            nt = ntol if float(ref[0]) >= 0.25 * scale else 3 * ntol
            assert abs(float(g.norm()) - float(ref[0])) < nt * float(ref[0]) + 2e-3 * scale, (pn, float(g.norm()), float(ref[0]))
            err = float((samp - ref[2:]).norm())
            assert err < tol * float(ref[2:].norm()) + 2e-3 * scale * (samp.numel() / f.numel()) ** 0.5, (pn, err, float(ref[2:].norm()))
            allg.append(samp.double()); allr.append(ref[2:].double()); n2g += float(g.norm()) ** 2; n2r += float(ref[0]) ** 2
            n_checked += 1
        elif key.startswith(name + "_nograd::"):
            pn = "mask_encoder.sam2_model." + key[len(name) + 9:]
            assert leaves[pn].grad is None, pn
    assert n_checked >= 100
    G, R = torch.cat(allg), torch.cat(allr)
    cos = float(torch.dot(G, R) / (G.norm() * R.norm()))
    assert 1.0 - cos <= 1e-2 and abs((n2g / n2r) ** 0.5 - 1.0) <= 2e-2, (name, 1.0 - cos, (n2g / n2r) ** 0.5)


@pytest.mark.parametrize("name", ["two_obj", "one_obj", "blob"])
def test_seg_head_forward_backward_vs_reference_backward(name):
    tol, ntol = TOL[name]
    from ufvideo_amd.train_seg import SegHeadGrad
    a, m, arrs, w = _seg_model()
    seg = SegHeadGrad(m)
    leaves = {k: v.detach().float().clone().requires_grad_(True) for k, v in SegHeadGrad.trainable(m)}
    labels = t(a[name + "_labels"])[0]
    ids = t(a[name + "_ids"])[0].tolist()
    hidden = t(a[name + "_hidden_last"])[0]                                          # the reference's own last hidden states [S, D]
    S = hidden.shape[0]
    k = ids.index(-201)
    lab = torch.cat([labels[:k], torch.full((S - len(ids) + 1,), -100), labels[k + 1:]])
    shifted = torch.cat([lab[1:], torch.full((1,), -100)])
    rows = torch.nonzero(shifted == 299).reshape(-1)
    gt = t(a[name + "_gt"])
    hid = hidden[rows].to(DEV).requires_grad_(True)
    w_ce, w_bce, w_dice = a["loss_weights"].tolist()
    bce, dice = seg.forward_backward(leaves, hid, t(a["images_sam"])[0].to(DEV), gt, tuple(gt.shape[1:]), w_bce, w_dice, gt.shape[0])
    ref = a[name + "_losses"]
    assert abs(float(bce) - ref[2]) < 2e-2 * ref[2] and abs(float(dice) - ref[3]) < 2e-2 * ref[3], (float(bce), float(dice), ref)
    # d(mask loss)/d(hidden): the golden `_d_hidden_fcs` also holds the CE term (HF hands the SAME tensor to lm_head), so the mask part is
    # the reference's d(loss)/d([SEG] embedding) pulled back through text_hidden_fcs (Linear-ReLU-Linear) in fp32
    W0, b0, W2 = (w["model.text_hidden_fcs.0." + k].float() for k in ("0.weight", "0.bias", "2.weight"))
    want = ((t(a[name + "_d_fcs_out"])[0][rows] @ W2) * ((hidden[rows] @ W0.T + b0) > 0).float()) @ W0
    worst = {"d_hidden": rel_err(hid.grad.cpu(), want)}
    for key in a:
        if key.startswith(name + "_g::model.text_hidden_fcs."):
            pn = key[len(name) + 4 + len("model."):]
            worst[pn] = rel_err(leaves[pn].grad.cpu(), t(a[key]))
    print("seg grads", name, {k: round(v, 4) for k, v in worst.items()})
    assert max(worst.values()) < tol, worst
    _check_sam_grads(a, name, leaves, tol, ntol)


def test_train_step_with_seg_head_vs_reference_backward():
    """the whole step: causal-LM + BCE + DICE, decoder + text_hidden_fcs + mask decoder trained; total loss and the decoder's
    gradients (which now carry the mask terms through the [SEG] positions) against the reference's loss.backward()"""
    from test_train_gpu import spliced_embed_ids  # noqa: F401
    from ufvideo_amd.train import DecoderTrainer
    a, m, arrs, w = _seg_model()
    tr = DecoderTrainer(m, lr=1e-4, weight_decay=0.0, max_grad_norm=0.0, train_seg_head=True)
    name = "two_obj"
    ids, labels, gt = t(a[name + "_ids"]).to(DEV), t(a[name + "_labels"]).to(DEV), t(a[name + "_gt"])
    video = t(arrs["video"]).to(DEV)
    steps = []
    orig_step = tr.step
    tr.step = lambda: steps.append(1)                                                # keep the gradients in the buckets: compare them, no update
    r = tr.train_step(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids), images=[(video, "video")],
                      images_sam=t(a["images_sam"]).to(DEV), offset=[0, 1], masks_list=[gt], label_list=[torch.zeros(*gt.shape[1:])])
    ref = a[name + "_losses"]
    got = [float(r[k]) for k in ("loss", "ce_loss", "mask_bce_loss", "mask_dice_loss", "mask_loss")]
    assert np.allclose(got, ref, rtol=2e-2), (got, ref.tolist())
    cfg = m.config
    H, KV, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    worst = 0.0
    for i, b in enumerate(tr.layers):
        p = f"model.layers.{i}."
        worst = max(worst, rel_err(b.view(b.g, "wo").cpu(), t(a[f"{name}_g::{p}self_attn.o_proj.weight"])),
                    rel_err(b.view(b.g, "wd").cpu(), t(a[f"{name}_g::{p}mlp.down_proj.weight"])),
                    rel_err(b.view(b.g, "wqkv").cpu()[:H * hd], t(a[f"{name}_g::{p}self_attn.q_proj.weight"])))
    worst = max(worst, rel_err(tr.small.view(tr.small.g, "norm").cpu(), t(a[f"{name}_g::model.norm.weight"])))
    assert worst < 6e-2, worst
    pb = tr.proj_bucket
    for key in a:
        if key.startswith(name + "_g::model.text_hidden_fcs."):
            pn = key[len(name) + 4 + len("model."):]
            assert rel_err(pb.view(pb.g, pn).cpu(), t(a[key])) < GRAD_TOL, pn
    gsum = {k: type("G", (), {"grad": pb.view(pb.g, k)})() for k in tr.proj_params if k.startswith("mask_encoder.")}
    for key in a:
        if key.startswith(name + "_nograd::"):
            pn = "mask_encoder.sam2_model." + key[len(name) + 9:]
            assert float(gsum[pn].grad.abs().max()) == 0.0, pn
            gsum[pn].grad = None
    _check_sam_grads(a, name, gsum, GRAD_TOL)
    # and a real step: the update reaches the model (generate() reads the trainer's buffers / re-packed adapters), the loss goes down
    tr.step = orig_step
    kw = dict(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids), images=[(video, "video")], images_sam=t(a["images_sam"]).to(DEV),
              offset=[0, 1], masks_list=[gt], label_list=[torch.zeros(*gt.shape[1:])])
    tr.lr = tr.base_lr = 3e-4
    l0 = tr.train_step(**kw)
    for _ in range(12):
        l1 = tr.train_step(**kw)
    print("seg train: loss", float(l0["loss"]), "->", float(l1["loss"]), " mask", float(l0["mask_loss"]), "->", float(l1["mask_loss"]))
    assert float(l1["loss"]) < float(l0["loss"]) and float(l1["mask_loss"]) < float(l0["mask_loss"])


def test_seg_head_at_sam2_l_dimensions():
    """The mask branch at the reference's real geometry: SAM2-L (Hiera-L trunk, 64x64 feature grid, 256-d two-way decoder), 4 SAM
    frames 1024x1024, two [SEG] objects, 360x480 labels, hidden size 3584.  No fp32 oracle runs this size in test time: the training
    forward must reproduce the loss of the INFERENCE path (`language_embd_inference`, pinned against the oracle in
    tests/test_configs_gpu.py) on the same inputs, every trained parameter receives a finite gradient, the IoU / object-score heads none."""
    import time
    import bench
    from ufvideo_amd.model import VideoReferQwen2Config, VideoReferQwen2ForCausalLM, QWEN2_7B
    from ufvideo_amd.train_seg import SegHeadGrad
    dev = torch.device("cuda", 0)
    cfg = VideoReferQwen2Config(**dict(QWEN2_7B, num_hidden_layers=1), mm_vision_tower="siglip-so400m-patch14-384",
                                mm_vision_select_layer=-2, mm_vision_select_feature="patch", mm_projector_type="stc_connector_v35",
                                mm_hidden_size=1152, mm_region_encoder_type="pooling", image_aspect_ratio="square",
                                train_mask_decoder=True, sam_pretrained=None, sam_out_dim=256, num_frames=32, seg_token_id=151747,
                                sam2_trunk="hiera_l", vision_config=dict(bench.VISION, num_hidden_layers=1))
    model = VideoReferQwen2ForCausalLM(cfg, device=dev, seed=0)
    model.get_model().initialize_sam_modules(cfg, device=dev)          # train_mask_decoder=True: built by the training script (ref train.py), as there
    seg = SegHeadGrad(model)
    leaves = {k: v.detach().float().clone().requires_grad_(True) for k, v in SegHeadGrad.trainable(model)}
    g = torch.Generator().manual_seed(5)
    T, n_obj, hw = 4, 2, (360, 480)
    sam = torch.randn(1, T, 3, 1024, 1024, generator=g).to(dev)
    hid = torch.randn(n_obj, cfg.hidden_size, generator=g).to(dev).requires_grad_(True)
    yy, xx = torch.meshgrid(torch.arange(hw[0]), torch.arange(hw[1]), indexing="ij")
    gt = torch.stack([((yy - 180 - 20 * i) ** 2 + (xx - 200 - 30 * i) ** 2 < (60 + 10 * i) ** 2).float() for i in range(T * n_obj)])
    w_bce, w_dice = 2.0, 0.5
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bce, dice = seg.forward_backward(leaves, hid, sam[0], gt, hw, w_bce, w_dice, T * n_obj)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for p in leaves.values():
        p.grad = None
    hid.grad = None
    bce, dice = seg.forward_backward(leaves, hid, sam[0], gt, hw, w_bce, w_dice, T * n_obj)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"SAM2-L mask branch forward+backward, 4 x 1024^2 frames, 2 objects: first {1e3 * (t1 - t0):.0f} ms, then {1e3 * (t2 - t1):.0f} ms")
    with torch.no_grad():
        emb = model.get_model().text_hidden_fcs[0](hid.detach())
        enc = model.get_model().mask_encoder
        logits = enc.language_embd_inference(model._sam_state(sam), [emb] * T)
        logits = ops.resize_bilinear(logits.contiguous(), hw)[:, 0].float().cpu()
    n = T * n_obj
    want_bce = w_bce * float(O.sigmoid_ce_loss(logits, gt, n))
    want_dice = w_dice * float(O.dice_loss(logits, gt, n))
    assert abs(float(bce) - want_bce) < 2e-2 * want_bce and abs(float(dice) - want_dice) < 2e-2 * want_dice, (float(bce), want_bce, float(dice), want_dice)
    assert torch.isfinite(hid.grad).all() and float(hid.grad.abs().max()) > 0
    n_grad = 0
    for k, p in leaves.items():
        if "iou_prediction_head" in k or "pred_obj_score_head" in k:
            assert p.grad is None, k
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        n_grad += int(float(p.grad.abs().max()) > 0)
    assert n_grad >= len(leaves) - 12 - 18 - 8                 # three of the four hyper-networks are not picked; key biases are analytically zero


def test_autograd_path_with_seg_targets_fills_grad_of_text_hidden_fcs_and_mask_decoder():
    """`model(**batch)["loss"].backward()` (the HF Trainer's call, model/videorefer_qwen2.py _TrainingLoss) on a batch WITH [SEG] targets: the gradient
    engine behind the autograd node has no fp32 masters (optimizer_states=False) and runs inside an autograd.Function.forward (grad mode off), while the
    mask branch differentiates through torch autograd -- the engine builds its leaves from the working weights and switches grad mode on locally.  Checked
    against the reference's own loss.backward() (golden seg_grad_tiny `two_obj`): total loss, text_hidden_fcs gradients, mask-decoder gradient norms."""
    a, m, arrs, w = _seg_model()
    name = "two_obj"
    ids, labels, gt = t(a[name + "_ids"]).to(DEV), t(a[name + "_labels"]).to(DEV), t(a[name + "_gt"])
    video = t(arrs["video"]).to(DEV)
    batch = dict(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids), images=[(video, "video")], images_sam=t(a["images_sam"]).to(DEV),
                 offset=[0, 1], masks_list=[gt], label_list=[torch.zeros(*gt.shape[1:])])
    own = dict(m.named_parameters())
    names = [n for n in own if n.startswith(("model.layers.", "model.norm.", "lm_head.", "model.embed_tokens.", "model.text_hidden_fcs.",
                                             "model.mask_encoder.sam2_model.sam_mask_decoder."))]
    for n in names:
        own[n].requires_grad_(True)
    out = m(**batch)
    ref = a[name + "_losses"]
    got = [float(out[k]) for k in ("loss", "ce_loss", "mask_bce_loss", "mask_dice_loss", "mask_loss")]
    assert np.allclose(got, ref, rtol=2e-2), (got, ref.tolist())
    assert out["loss"].requires_grad
    out["loss"].backward()
    n_fcs = 0
    for key in a:
        if key.startswith(name + "_g::model.text_hidden_fcs."):
            pn = key[len(name) + 4:]
            assert own[pn].grad is not None, pn
            assert rel_err(own[pn].grad.float().cpu(), t(a[key])) < GRAD_TOL, pn
            n_fcs += 1
    assert n_fcs == 4
    gsum = {k[len("model."):]: type("G", (), {"grad": p.grad})() for k, p in own.items() if k.startswith("model.mask_encoder.")}
    for key in a:                       # heads the loss does not reach: the engine exports an all-zero gradient where torch leaves None
        if key.startswith(name + "_nograd::"):
            pn = "mask_encoder.sam2_model." + key[len(name) + 9:]
            assert gsum[pn].grad is None or float(gsum[pn].grad.abs().max()) == 0.0, pn
            gsum[pn].grad = None
    _check_sam_grads(a, name, gsum, GRAD_TOL + 0.05, 0.06)                       # (.grad takes the parameter's dtype: one more rounding than the fp32 buckets)
    k = "model.layers.1.mlp.down_proj.weight"
    assert rel_err(own[k].grad.float().cpu(), t(a[f"{name}_g::{k}"])) < 8e-2
    m.release_grad_engine()
