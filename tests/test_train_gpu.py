"""GPU: the decoder training step (csrc/train.hip + ufvideo_amd/train.py) against torch autograd of the CPU oracle and the
golden vectors captured from the reference's own backward / AdamW step (oracle/gen_fixtures_train_grad.py)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, t, rel_err
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu

from ufvideo_amd import ops  # noqa: E402
from ufvideo_amd.train import DecoderTrainer  # noqa: E402
from test_model_gpu import tiny_model, TINY_LLM, DEV  # noqa: E402
from test_oracle_golden import spliced_embed_ids, hd128_golden  # noqa: E402


def bfr(x):
    return x.to(torch.bfloat16).float()


# ---- kernels alone vs torch ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("R,C", [(5, 7), (64, 64), (130, 200), (2399, 136)])
def test_transpose_zero_padded(R, C):
    g = torch.Generator().manual_seed(R * 1000 + C)
    x = torch.randn(R, C + 3, generator=g).to(torch.bfloat16).to(DEV)[:, :C]          # row-strided view
    out = torch.full((C, ops.round_up(R, 128)), 7.0, device=DEV, dtype=torch.bfloat16)
    ops.transpose(x, out=out)
    assert torch.equal(out[:, :R].cpu(), x.cpu().t()) and (out[:, R:] == 0).all()


@pytest.mark.parametrize("M,D", [(9, 64), (300, 1024), (1030, 1028), (5, 4096), (2399, 3584)])
def test_rmsnorm_bwd_vs_autograd(M, D):
    g = torch.Generator().manual_seed(M + D)
    x = torch.randn(M, D, generator=g) * 2
    w = 1 + 0.1 * torch.randn(D, generator=g)
    dy = torch.randn(M, D, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    O.rmsnorm(xr, wr, 1e-6).backward(dy)
    dx0 = torch.randn(M, D, generator=g)
    dx = dx0.clone().to(DEV)
    dw = torch.zeros(D, device=DEV)
    ops.rmsnorm_bwd(x.to(DEV), w.to(DEV), dy.to(DEV), dx, dw, 1e-6, accumulate=True, dw_accumulate=False)
    assert rel_err(dx.cpu() - dx0, xr.grad) < 1e-5 and rel_err(dw.cpu(), wr.grad) < 1e-4
    dw2 = dw.clone()
    ops.rmsnorm_bwd(x.to(DEV), w.to(DEV), dy.to(DEV), dx, dw2, 1e-6, accumulate=False, dw_accumulate=True)
    assert rel_err(dx.cpu(), xr.grad) < 1e-5 and rel_err(dw2.cpu(), 2 * wr.grad) < 1e-4


def test_swiglu_fwd_bwd_vs_autograd():
    from ufvideo_amd.model.videorefer_qwen2 import pack_swiglu
    g = torch.Generator().manual_seed(3)
    M, I = 37, 160
    gate, up = bfr(torch.randn(M, I, generator=g) * 2), bfr(torch.randn(M, I, generator=g))
    dact = bfr(torch.randn(M, I, generator=g))
    gu = pack_swiglu(gate.t().contiguous(), up.t().contiguous()).t().contiguous().to(torch.bfloat16).to(DEV)   # rows interleaved -> columns
    act = ops.swiglu(gu)
    gr, ur = gate.clone().requires_grad_(True), up.clone().requires_grad_(True)
    ref = F.silu(gr) * ur
    assert rel_err(act.float().cpu(), ref.detach()) < 1e-2
    ref.backward(dact)
    dgu = ops.swiglu_bwd(gu, dact.to(torch.bfloat16).to(DEV)).float().cpu()
    dg = dgu.view(M, I // 16, 2, 16)[:, :, 0].reshape(M, I)
    du = dgu.view(M, I // 16, 2, 16)[:, :, 1].reshape(M, I)
    assert rel_err(dg, gr.grad) < 1e-2 and rel_err(du, ur.grad) < 1e-2


def test_rope_rows_backward_is_the_transposed_rotation():
    g = torch.Generator().manual_seed(4)
    S, H, hd = 19, 3, 16
    inv = (1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))).to(DEV)
    x = bfr(torch.randn(S, 8 + H * hd, generator=g))
    dy = bfr(torch.randn(S, H * hd, generator=g))
    cos, sin = O.rope_cos_sin(torch.arange(S), hd, 10000.0)
    xr = x[:, 8:].reshape(S, H, hd).clone().requires_grad_(True)
    y = xr * cos[:, None] + O.rotate_half(xr) * sin[:, None]
    fwd = x.to(torch.bfloat16).to(DEV)
    ops.rope_rows(fwd, 8, H, hd, inv, 0, backward=False)
    assert rel_err(fwd[:, 8:].float().cpu(), y.detach().reshape(S, H * hd)) < 1e-2 and torch.equal(fwd[:, :8].float().cpu(), x[:, :8])
    y.backward(dy.view(S, H, hd))
    buf = torch.cat([x[:, :8], dy], 1).to(torch.bfloat16).to(DEV)
    ops.rope_rows(buf, 8, H, hd, inv, 0, backward=True)
    assert rel_err(buf[:, 8:].float().cpu(), xr.grad.reshape(S, H * hd)) < 1e-2


def test_cross_entropy_bwd_colsum_scatter_sumsq():
    g = torch.Generator().manual_seed(5)
    M, V, Vp = 41, 1000, 1024
    lg = torch.randn(M, Vp, generator=g) * 3
    lab = torch.randint(0, V, (M,), generator=g); lab[::4] = -100
    lr_ = lg[:, :V].clone().requires_grad_(True)
    loss = F.cross_entropy(lr_, lab, ignore_index=-100, reduction="sum") * 0.05
    loss.backward()
    rows, dl = ops.cross_entropy_bwd(lg.to(DEV), lab.to(DEV), V, 0.05)
    assert abs(float(rows.sum()) * 0.05 - float(loss)) < 1e-4 * abs(float(loss))
    assert rel_err(dl[:, :V].float().cpu(), lr_.grad) < 1e-2 and (dl[:, V:] == 0).all() and (dl[::4] == 0).all()
    x = torch.randn(777, 300, generator=g).to(torch.bfloat16)
    out = torch.ones(300, device=DEV)
    ops.colsum(x.to(DEV), out, accumulate=True)
    assert torch.allclose(out.cpu(), 1 + x.float().sum(0), atol=1e-3)
    src = torch.randn(50, 64, generator=g)
    idx = torch.randint(-1, 9, (50,), generator=g)
    dst = torch.zeros(9, 64, device=DEV)
    ops.scatter_add_rows(src.to(DEV), idx.to(DEV), dst)
    ref = torch.zeros(9, 64).index_add_(0, idx[idx >= 0], src[idx >= 0])
    assert torch.allclose(dst.cpu(), ref, atol=1e-5)
    big = torch.randn(1_000_003, generator=g)
    assert abs(float(ops.sumsq(big.to(DEV)).sum()) - float((big.double() ** 2).sum())) < 1e-3 * big.numel()


def test_adamw_matches_torch_over_steps():
    g = torch.Generator().manual_seed(6)
    n = 10_007
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pb = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    gs = torch.tensor([0.5], device=DEV)
    for step in range(1, 5):
        grad = torch.randn(n, generator=g)
        ref.grad = grad * 0.5
        opt.step()
        ops.adamw(p, grad.to(DEV), m, v, pb, 3e-3, 0.9, 0.95, 1e-8, 0.05, step, gscale=gs)
        assert torch.allclose(p.cpu(), ref.detach(), atol=1e-6, rtol=1e-5), step
    assert torch.equal(pb.cpu(), p.cpu().to(torch.bfloat16))


def _attn_ref(q, k, v, H, KV, hd):
    """eager causal GQA attention of the oracle's decoder layer on [S, H*hd] / [S, KV*hd] rows (fp32, autograd)"""
    S = q.shape[0]
    rep = H // KV
    qq = q.view(S, H, hd).transpose(0, 1)
    kk = k.view(S, KV, hd).transpose(0, 1).repeat_interleave(rep, 0)
    vv = v.view(S, KV, hd).transpose(0, 1).repeat_interleave(rep, 0)
    att = qq @ kk.transpose(1, 2) * hd ** -0.5
    att = att.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf")).softmax(-1)
    return (att @ vv).transpose(0, 1).reshape(S, H * hd)


@pytest.mark.parametrize("S,H,KV,hd", [(37, 4, 2, 16), (300, 4, 2, 128), (515, 7, 1, 128)])
def test_attention_bwd_vs_autograd(S, H, KV, hd):
    """the tiny case runs on the generic GEMM, the hd=128 cases on the MFMA tile kernels (N = 128, K = padded key count)"""
    g = torch.Generator().manual_seed(S)
    Sp = ops.round_up(S, 128)
    q, k, v = (bfr(torch.randn(S, n * hd, generator=g)) for n in (H, KV, KV))
    dO = bfr(torch.randn(S, H * hd, generator=g))
    qr, kr, vr = (x.clone().requires_grad_(True) for x in (q, k, v))
    _attn_ref(qr, kr, vr, H, KV, hd).backward(dO)
    kv = torch.full((Sp, 2 * KV * hd), float("nan"), dtype=torch.bfloat16)         # pad rows must not leak
    kv[:S, :KV * hd] = k.to(torch.bfloat16); kv[:S, KV * hd:] = v.to(torch.bfloat16)
    kv = kv.to(DEV)
    dqkv = torch.zeros(S, (H + 2 * KV) * hd, device=DEV, dtype=torch.bfloat16)
    ops.attention_bwd(q.to(torch.bfloat16).to(DEV), kv, kv[:, KV * hd:], dO.to(torch.bfloat16).to(DEV), dqkv, dqkv[:, H * hd:],
                      dqkv[:, (H + KV) * hd:], S, H, KV, hd)
    d = dqkv.float().cpu()
    assert rel_err(d[:, :H * hd], qr.grad) < 2e-2
    assert rel_err(d[:, H * hd:(H + KV) * hd], kr.grad) < 2e-2
    assert rel_err(d[:, (H + KV) * hd:], vr.grad) < 2e-2


@pytest.mark.parametrize("S,H,KV", [(37, 4, 2), (128, 2, 2), (192, 2, 1), (300, 4, 2), (515, 7, 1), (1000, 14, 2)])
def test_attention_bwd_fused_vs_autograd(S, H, KV):
    """flash-style backward (hd 128): forward kernel's lse + output, then dQ / dK / dV without any S x S buffer"""
    hd = 128
    g = torch.Generator().manual_seed(S)
    q, k, v = (bfr(torch.randn(S, n * hd, generator=g)) for n in (H, KV, KV))
    dO = bfr(torch.randn(S, H * hd, generator=g))
    qr, kr, vr = (x.clone().requires_grad_(True) for x in (q, k, v))
    ref = _attn_ref(qr, kr, vr, H, KV, hd)
    ref.backward(dO)
    kv = torch.cat([k, v], 1).to(torch.bfloat16).to(DEV)
    qd, dod = q.to(torch.bfloat16).to(DEV), dO.to(torch.bfloat16).to(DEV)
    o = torch.empty(S, H * hd, device=DEV, dtype=torch.bfloat16)
    lse = torch.full((H, S), float("nan"), device=DEV, dtype=torch.float32)
    ops.attention_causal_lse(qd, kv, kv[:, KV * hd:], o, lse, S, H, KV, hd)
    assert rel_err(o.float().cpu(), ref.detach()) < 1e-2
    # lse (log2 domain) against the eager scores
    sc = torch.einsum("shd,thd->hst", q.view(S, H, hd), k.view(S, KV, hd).repeat_interleave(H // KV, 1)) * hd ** -0.5
    sc = sc.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
    assert (lse.cpu() - torch.logsumexp(sc, -1) / math.log(2.0)).abs().max() < 2e-2
    o2 = ops.attention(qd, kv, kv[:, KV * hd:], 1, H, KV, S, S, hd, (0, qd.stride(0)), (0, kv.stride(0)), (0, kv.stride(0)), causal=True)
    assert torch.equal(o, o2)                                                        # same kernel, same bits as the inference path
    dqkv = torch.full((S, (H + 2 * KV) * hd), float("nan"), device=DEV, dtype=torch.bfloat16)
    ops.attention_bwd_fused(qd, kv, kv[:, KV * hd:], o, dod, lse, dqkv, dqkv[:, H * hd:], dqkv[:, (H + KV) * hd:], S, H, KV, hd)
    d = dqkv.float().cpu()
    assert rel_err(d[:, :H * hd], qr.grad) < 2e-2
    assert rel_err(d[:, H * hd:(H + KV) * hd], kr.grad) < 2e-2
    assert rel_err(d[:, (H + KV) * hd:], vr.grad) < 2e-2
    # and against the materialised kernel path
    Sp = ops.round_up(S, 128)
    kvp = torch.zeros(Sp, 2 * KV * hd, device=DEV, dtype=torch.bfloat16); kvp[:S] = kv
    d2 = torch.zeros_like(dqkv)
    ops.attention_bwd(qd, kvp, kvp[:, KV * hd:], dod, d2, d2[:, H * hd:], d2[:, (H + KV) * hd:], S, H, KV, hd)
    assert rel_err(d, d2.float().cpu()) < 1.5e-2


def test_attention_bwd_fused_rejects_what_it_does_not_serve():
    """head_dim other than 128 and misaligned rows are errors of the C entry (UfvError with a message), never a silent fallback"""
    from ufvideo_amd import _lib
    S, H, KV, hd = 40, 2, 1, 64
    q = torch.zeros(S, H * hd, device=DEV, dtype=torch.bfloat16)
    kv = torch.zeros(S, 2 * KV * hd, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(H, S, device=DEV)
    d = torch.zeros(S, (H + 2 * KV) * hd, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(_lib.UfvError, match="head_dim"):
        ops.attention_bwd_fused(q, kv, kv[:, KV * hd:], q, q, lse, d, d[:, H * hd:], d[:, (H + KV) * hd:], S, H, KV, hd)
    with pytest.raises(_lib.UfvError):
        ops.attention_causal_lse(q, kv, kv[:, KV * hd:], q.clone(), lse, S, H, KV, hd)
    hd = 128
    q = torch.zeros(S, H * hd + 4, device=DEV, dtype=torch.bfloat16)[:, 4:]           # rows start 8 bytes off a 16-byte boundary
    kv = torch.zeros(S, 2 * KV * hd, device=DEV, dtype=torch.bfloat16)
    d = torch.zeros(S, (H + 2 * KV) * hd, device=DEV, dtype=torch.bfloat16)
    o = torch.zeros(S, H * hd, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(_lib.UfvError, match="aligned"):
        ops.attention_bwd_fused(q, kv, kv[:, KV * hd:], o, o, lse, d, d[:, H * hd:], d[:, (H + KV) * hd:], S, H, KV, hd)


# ---- the whole step on the tiny model vs the reference's own backward / optimizer step ------------------------------------------

def _shift(labels):
    return torch.cat([labels[1:], torch.full((1,), -100, dtype=labels.dtype)])


def test_tiny_model_step_vs_reference_golden():
    a, _ = load_golden("train_grad_tiny")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_grad_tiny.npz"))
    m, _, w = tiny_model()
    lr, wd, b1, b2, eps, clip = a["hyper"].tolist()
    tr = DecoderTrainer(m, lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, max_grad_norm=clip)
    emb = t(a["inputs_embeds"])[0].to(DEV)
    labels = t(a["labels"])[0]
    eids = spliced_embed_ids(a["ids"][0], emb.shape[0])
    tr.zero_grad()
    loss, dx = tr.forward_backward(emb, _shift(labels), embed_ids=eids)
    assert abs(float(loss) - float(a["ce_loss"])) < 2e-2 * float(a["ce_loss"])
    assert rel_err(dx.cpu(), t(a["d_inputs_embeds"])[0]) < 4e-2
    # gradients under the reference's parameter names
    cfg = m.config
    H, KV, hd, I = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.intermediate_size
    ref = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("g::")}
    got = {}
    for i, b in enumerate(tr.layers):
        p = f"model.layers.{i}."
        gq = b.view(b.g, "wqkv").cpu(); bq = tr.small.view(tr.small.g, f"bqkv.{i}").cpu()
        for nm, lo, hi in (("q_proj", 0, H * hd), ("k_proj", H * hd, (H + KV) * hd), ("v_proj", (H + KV) * hd, (H + 2 * KV) * hd)):
            got[p + f"self_attn.{nm}.weight"] = gq[lo:hi]; got[p + f"self_attn.{nm}.bias"] = bq[lo:hi]
        got[p + "self_attn.o_proj.weight"] = b.view(b.g, "wo").cpu()
        gu = b.view(b.g, "wgu").cpu().view(I // 16, 2, 16, -1)
        got[p + "mlp.gate_proj.weight"] = gu[:, 0].reshape(I, -1); got[p + "mlp.up_proj.weight"] = gu[:, 1].reshape(I, -1)
        got[p + "mlp.down_proj.weight"] = b.view(b.g, "wd").cpu()
        got[p + "input_layernorm.weight"] = tr.small.view(tr.small.g, f"ln1.{i}").cpu()
        got[p + "post_attention_layernorm.weight"] = tr.small.view(tr.small.g, f"ln2.{i}").cpu()
    got["model.norm.weight"] = tr.small.view(tr.small.g, "norm").cpu()
    got["lm_head.weight"] = tr.head.view(tr.head.g, "lm_head")[:cfg.vocab_size].cpu()
    got["model.embed_tokens.weight"] = tr.head.view(tr.head.g, "embed").cpu()
    assert set(got) == set(ref)
    worst = max((rel_err(got[k], ref[k]), k) for k in ref)
    assert worst[0] < 5e-2, worst
    assert (tr.head.view(tr.head.g, "lm_head")[cfg.vocab_size:] == 0).all()
    # optimizer step: clip + AdamW on fp32 masters; the FIRST Adam step moves every element by ~lr * sign(g), so elements whose
    # bf16-path gradient has the other sign (|g| ~ 0) land 2*lr away: require 97 % of the elements within 10 % of lr
    tr.step()
    assert abs(float(tr.last_grad_norm) - float(a["grad_norm"])) < 3e-2 * float(a["grad_norm"])
    sd = tr.export_state_dict()
    ok = tot = 0
    for k in ref:
        p1 = torch.from_numpy(z["p1::" + k])
        mine = sd[k].float().cpu()
        assert mine.shape == p1.shape, k
        if k in ("lm_head.weight", "model.embed_tokens.weight") or "layers" in k and p1.ndim == 2:
            ok += int(((mine - p1).abs() <= 0.1 * lr + 2 ** -8 * p1.abs()).sum()); tot += p1.numel()
    assert ok / tot > 0.97, ok / tot
    # the model itself now runs on the updated weights: a second forward gives a lower loss on the same sample
    tr.zero_grad()
    loss2, _ = tr.forward_backward(emb, _shift(labels), embed_ids=eids)
    assert float(loss2) < float(loss)


@pytest.mark.parametrize("attn", ["fused", "materialised"])
def test_hd128_step_vs_reference_golden(attn, monkeypatch):
    """train_grad_hd128: the reference class's own loss.backward() on a decoder with heads of 128 -- the shape served by the fused
    attention backward (csrc/attn_bwd.hip; S = 200 spans two query / key blocks) and, for comparison, the per-group GEMM pipeline"""
    from ufvideo_amd.model import VideoReferQwen2Config, VideoReferQwen2ForCausalLM
    monkeypatch.setenv("UFV_TRAIN_ATTN", attn)
    a, llm, w, ref_g = hd128_golden()
    m = VideoReferQwen2ForCausalLM(VideoReferQwen2Config(**llm, sam2_trunk=None))
    m.load_state_dict(w, strict=False)
    m = m.to(DEV)
    tr = DecoderTrainer(m, train_embed=False)
    assert tr.fused_attn_bwd == (attn == "fused")
    tr.zero_grad()
    loss, dx = tr.forward_backward(t(a["inputs_embeds"])[0].to(DEV), _shift(t(a["labels"])[0]))
    assert abs(float(loss) - float(a["loss"])) < 2e-2 * float(a["loss"])
    assert rel_err(dx.cpu(), t(a["d_inputs_embeds"])[0]) < 4e-2
    H, KV, hd = llm["num_attention_heads"], llm["num_key_value_heads"], 128
    got = {"model.norm.weight": tr.small.view(tr.small.g, "norm")}
    for i, b in enumerate(tr.layers):
        p = f"model.layers.{i}."
        bq = tr.small.view(tr.small.g, f"bqkv.{i}")
        got[p + "self_attn.q_proj.bias"], got[p + "self_attn.k_proj.bias"], got[p + "self_attn.v_proj.bias"] = \
            bq[:H * hd], bq[H * hd:(H + KV) * hd], bq[(H + KV) * hd:]
        got[p + "input_layernorm.weight"] = tr.small.view(tr.small.g, f"ln1.{i}")
        got[p + "post_attention_layernorm.weight"] = tr.small.view(tr.small.g, f"ln2.{i}")
        gq = b.view(b.g, "wqkv")
        got[p + "self_attn.k_proj.weight"], got[p + "self_attn.v_proj.weight"] = gq[H * hd:(H + KV) * hd], gq[(H + KV) * hd:]
    assert set(ref_g) <= set(got)
    worst = max((rel_err(got[k].float().cpu(), g), k) for k, g in ref_g.items())
    assert worst[0] < 5e-2, worst


def test_lora_q_v_adapters_gradients_step_and_merge():
    """lora=dict(r, alpha): the reference's --lora_enable stage (train.py:829-845; targets = q_proj / v_proj, videorefer_trainer.py:75-90).  Checker: the full
    decoder backward of this repo (pinned on the reference's own gradients by the tests above) on the MERGED weights W + s B A: loss and d(inputs) must agree,
    and by the chain rule dB = s dW A^T, dA = s B^T dW for the adapted projections.  Then: rank padding stays zero, AdamW moves only the adapters, the loss
    falls, the adapters export under peft's names, and detach() merges them into the model's weights."""
    from ufvideo_amd.model import VideoReferQwen2Config, VideoReferQwen2ForCausalLM
    a, llm, w, _ = hd128_golden()
    D, L, H, KV, hd = llm["hidden_size"], llm["num_hidden_layers"], llm["num_attention_heads"], llm["num_key_value_heads"], 128
    r, alpha = 8, 16
    s = alpha / r
    g = torch.Generator().manual_seed(5)
    init = {}
    for i in range(L):
        for nm, out in (("q_proj", H * hd), ("v_proj", KV * hd)):
            init[f"model.layers.{i}.self_attn.{nm}.lora_A.weight"] = bfr(torch.randn(r, D, generator=g) * 0.05)
            init[f"model.layers.{i}.self_attn.{nm}.lora_B.weight"] = bfr(torch.randn(out, r, generator=g) * 0.05)

    def build(weights):
        m = VideoReferQwen2ForCausalLM(VideoReferQwen2Config(**llm, sam2_trunk=None))
        m.load_state_dict(weights, strict=False)
        return m.to(DEV)
    emb, labels = t(a["inputs_embeds"])[0].to(DEV), _shift(t(a["labels"])[0])
    tr = DecoderTrainer(build(w), lr=2e-3, lora=dict(r=r, alpha=alpha, init=init), max_grad_norm=1.0)
    assert not tr.train_decoder and tr.buckets() == [tr.lora_bucket]
    tr.zero_grad()
    loss, dx = tr.forward_backward(emb, labels)
    # the same model with the adapters merged, full backward
    w2 = dict(w)
    for i in range(L):
        for nm in ("q_proj", "v_proj"):
            k = f"model.layers.{i}.self_attn.{nm}."
            w2[k + "weight"] = w[k + "weight"] + s * init[k + "lora_B.weight"] @ init[k + "lora_A.weight"]
    tr2 = DecoderTrainer(build(w2), train_embed=False)
    tr2.zero_grad()
    loss2, dx2 = tr2.forward_backward(emb, labels)
    assert abs(float(loss) - float(loss2)) < 2e-2 * float(loss2)
    assert rel_err(dx.cpu(), dx2.cpu()) < 4e-2
    lb, Rp = tr.lora_bucket, tr.Rp
    for i in range(L):
        gq = tr2.layers[i].view(tr2.layers[i].g, "wqkv").cpu()
        acat = lb.view(lb.g, f"Acat.{i}").cpu()
        for j, (nm, lo, hi) in enumerate((("q_proj", 0, H * hd), ("v_proj", (H + KV) * hd, (H + 2 * KV) * hd))):
            k = f"model.layers.{i}.self_attn.{nm}."
            A, B, dWe = init[k + "lora_A.weight"], init[k + "lora_B.weight"], gq[lo:hi]
            gb = lb.view(lb.g, f"B{nm[0]}.{i}").cpu()
            assert rel_err(gb[:, :r], s * dWe @ A.t()) < 5e-2, (i, nm)
            assert rel_err(acat[j * Rp:j * Rp + r], s * B.t() @ dWe) < 5e-2, (i, nm)
            assert (gb[:, r:] == 0).all() and (acat[j * Rp + r:(j + 1) * Rp] == 0).all()          # rank padding: exact zeros
    base_before = [b.w.clone() for b in tr.layers]
    losses = [float(loss)]
    for _ in range(4):
        tr.step()
        tr.zero_grad()
        losses.append(float(tr.forward_backward(emb, labels)[0]))
    assert losses[-1] < losses[0], losses
    assert all(torch.equal(b.w, w0) for b, w0 in zip(tr.layers, base_before))                     # the base weights are frozen
    # a sync_to_model() BETWEEN steps (a periodic checkpoint) writes the merged view into the parameters and changes nothing of the run:
    # next loss, gradients, masters and moments bit-identical to the same step without it, adapters still exported un-merged
    snap = (lb.w.clone(), lb.master.clone(), lb.m.clone(), lb.v.clone())
    tr.zero_grad(); l_a, _ = tr.forward_backward(emb, labels); g_a = lb.g.clone() if hasattr(lb, "g") else None
    tr.sync_to_model()
    assert all(torch.equal(a_, b_) for a_, b_ in zip(snap, (lb.w, lb.master, lb.m, lb.v)))
    assert all(torch.equal(b.w, w0) for b, w0 in zip(tr.layers, base_before))
    tr.zero_grad(); l_b, _ = tr.forward_backward(emb, labels)
    assert float(l_a) == float(l_b)
    if g_a is not None:
        assert torch.equal(g_a, lb.g)
    own_mid = dict(tr.model.named_parameters())
    sd_mid = tr.export_lora_state_dict()
    assert float(sd_mid["base_model.model.model.layers.0.self_attn.q_proj.lora_B.weight"].abs().max()) > 0          # not merged away
    k0 = "model.layers.0.self_attn.q_proj.weight"
    want0 = w[k0] + s * sd_mid["base_model.model.model.layers.0.self_attn.q_proj.lora_B.weight"].cpu() @ sd_mid["base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight"].cpu()
    assert rel_err(own_mid[k0].float().cpu(), want0) < 1e-2
    tr.zero_grad()
    sd = tr.export_lora_state_dict()
    assert len(sd) == 4 * L and sd["base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight"].shape == (r, D) and \
        sd["base_model.model.model.layers.1.self_attn.v_proj.lora_B.weight"].shape == (KV * hd, r)
    assert (lb.view(lb.w, "Bq.0")[:, r:] == 0).all() and (lb.view(lb.w, "Acat.0")[r:Rp] == 0).all()
    m = tr.model
    tr.detach()
    own = dict(m.named_parameters())
    for i in range(L):
        for nm in ("q_proj", "v_proj"):
            p = f"base_model.model.model.layers.{i}.self_attn.{nm}."
            k = f"model.layers.{i}.self_attn.{nm}.weight"
            want = w[k] + s * sd[p + "lora_B.weight"].cpu() @ sd[p + "lora_A.weight"].cpu()
            assert rel_err(own[k].float().cpu(), want) < 1e-2, k
    # inference on the merged model sees the trained adapters: the loss of a plain forward pass equals the trainer's last one
    tr3 = DecoderTrainer(m, train_embed=False)
    l3, _ = tr3.forward_backward(emb, labels)
    assert abs(float(l3) - losses[-1]) < 3e-2 * losses[-1]


def test_export_state_dict_roundtrip_before_any_step():
    m, _, w = tiny_model()
    tr = DecoderTrainer(m, train_embed=False)
    sd = tr.export_state_dict()
    for k, v in sd.items():
        assert torch.equal(v.float().cpu(), w[k].to(torch.bfloat16).float()), k
    assert "model.embed_tokens.weight" not in sd


@pytest.mark.parametrize("attn", ["fused", "materialised"])
def test_fulldim_layer_grads_vs_oracle_autograd(attn, monkeypatch):
    """One decoder layer at the 7B dims (D 3584, 28/4 heads x 128, d_ff 18944), S = 256, vocab 1024: every GEMM of the backward
    runs on the MFMA tile kernels (N % 128 == 0, K % 64 == 0), with either attention backward (flash-style kernels / per-group GEMM
    pipeline).  Checker: torch autograd over the oracle on the host."""
    monkeypatch.setenv("UFV_TRAIN_ATTN", attn)
    from ufvideo_amd.model import VideoReferQwen2Config, VideoReferQwen2ForCausalLM
    llm = dict(vocab_size=1024, hidden_size=3584, intermediate_size=18944, num_hidden_layers=1, num_attention_heads=28,
               num_key_value_heads=4, rope_theta=1e6, rms_norm_eps=1e-6)
    cfg = VideoReferQwen2Config(**llm, sam2_trunk=None)          # decoder only: no tower / projector / SAM2
    w = O.make_qwen2_weights(llm, seed=11, std=0.02)
    w = {k: bfr(v) for k, v in w.items()}
    m = VideoReferQwen2ForCausalLM(cfg)
    m.load_state_dict(w, strict=False)
    m = m.to(DEV)
    tr = DecoderTrainer(m, train_embed=False)
    assert tr.fused_attn_bwd == (attn == "fused")
    g = torch.Generator().manual_seed(12)
    S = 256
    emb = torch.randn(S, 3584, generator=g)
    labels = torch.randint(0, 1024, (S,), generator=g); labels[:5] = -100
    tr.zero_grad()
    loss, dx = tr.forward_backward(emb.to(DEV), _shift(labels))
    rl, rg, rde = O.decoder_train_grads(w, llm, emb[None], labels[None])
    assert abs(float(loss) - float(rl)) < 2e-2 * float(rl)
    assert rel_err(dx.cpu(), rde[0]) < 4e-2
    b = tr.layers[0]
    p = "model.layers.0."
    I = 18944
    gq = b.view(b.g, "wqkv").cpu()
    assert rel_err(gq[:3584], rg[p + "self_attn.q_proj.weight"]) < 4e-2
    assert rel_err(gq[3584:4096], rg[p + "self_attn.k_proj.weight"]) < 4e-2
    assert rel_err(gq[4096:], rg[p + "self_attn.v_proj.weight"]) < 4e-2
    assert rel_err(b.view(b.g, "wo").cpu(), rg[p + "self_attn.o_proj.weight"]) < 4e-2
    gu = b.view(b.g, "wgu").cpu().view(I // 16, 2, 16, -1)
    assert rel_err(gu[:, 0].reshape(I, -1), rg[p + "mlp.gate_proj.weight"]) < 4e-2
    assert rel_err(gu[:, 1].reshape(I, -1), rg[p + "mlp.up_proj.weight"]) < 4e-2
    assert rel_err(b.view(b.g, "wd").cpu(), rg[p + "mlp.down_proj.weight"]) < 4e-2
    assert rel_err(tr.head.view(tr.head.g, "lm_head")[:1024].cpu(), rg["lm_head.weight"]) < 4e-2
    assert rel_err(tr.small.view(tr.small.g, "ln2.0").cpu(), rg[p + "post_attention_layernorm.weight"]) < 4e-2
    assert rel_err(tr.small.view(tr.small.g, "bqkv.0").cpu()[:3584], rg[p + "self_attn.q_proj.bias"]) < 4e-2


@pytest.mark.parametrize("M,N,K,ns", [(2399, 128, 16896, 16), (300, 256, 1024, 4), (300, 128, 1088, 16), (37, 16, 72, 4)])
def test_gemm_splitk_vs_torch(M, N, K, ns):
    """C (+)= A W^T with K split over blocks (partials + ordered sum); the last case falls back to one ordinary GEMM"""
    g = torch.Generator().manual_seed(M + K)
    a, w = bfr(torch.randn(M, K, generator=g)), bfr(torch.randn(N, K, generator=g))
    c0 = torch.randn(M, N, generator=g)
    ad, wd = a.to(torch.bfloat16).to(DEV), w.to(torch.bfloat16).to(DEV)
    prod = a.double() @ w.double().t()
    out = c0.clone().to(DEV)
    ops.gemm_splitk(ad, wd, out, ns, accumulate=True)
    assert rel_err(out.cpu().double(), c0.double() + prod) < 1e-5
    ops.gemm_splitk(ad, wd, out, ns)
    assert rel_err(out.cpu().double(), prod) < 1e-5
    ob = torch.empty(M, N + 8, device=DEV, dtype=torch.bfloat16)[:, :N]
    ops.gemm_splitk(ad, wd, ob, ns)
    assert rel_err(ob.float().cpu().double(), prod) < 1e-2


def test_gradient_accumulation_equals_sum_of_micro_batches():
    """two forward_backward() calls in one window: the first overwrites the matrix gradients, the second accumulates"""
    a, _ = load_golden("train_grad_tiny")
    m, _, _ = tiny_model()
    tr = DecoderTrainer(m)
    emb = t(a["inputs_embeds"])[0].to(DEV)
    labels = _shift(t(a["labels"])[0])
    eids = spliced_embed_ids(a["ids"][0], emb.shape[0])
    for b in tr.layers:
        b.g.fill_(123.0)                                   # stale values from an earlier step must not survive
    tr.zero_grad()
    tr.forward_backward(emb, labels, embed_ids=eids)
    g1 = [b.g.clone() for b in tr.buckets()]
    tr.forward_backward(emb * 0.5, labels, embed_ids=eids)
    g12 = [b.g.clone() for b in tr.buckets()]
    tr.zero_grad()
    tr.forward_backward(emb * 0.5, labels, embed_ids=eids)
    g2 = [b.g.clone() for b in tr.buckets()]
    for x1, x12, x2, b in zip(g1, g12, g2, tr.buckets()):
        views = [n for n, *_ in b.entries]
        for n in views:
            s_ = b.view(x1, n) + b.view(x2, n)
            assert torch.allclose(b.view(x12, n), s_, rtol=1e-4, atol=1e-6 * float(s_.abs().max() + 1)), n


@pytest.mark.parametrize("lora", [None, dict(r=4, alpha=8, seed=3)])
def test_gradient_checkpointing_is_bit_identical_and_keeps_one_stash(lora):
    """DecoderTrainer(gradient_checkpointing=True) -- the reference's --gradient_checkpointing True (scripts/train/train_1121v1.sh; transformers wraps every decoder
    layer in torch.utils.checkpoint): per layer only the input stream is kept, a layer's backward first re-runs its forward into the ONE shared stash.  Same launches on
    the same inputs: loss, every gradient and d(inputs_embeds) are `torch.equal` to the stashing form; the activation buffers shrink from n_layers stashes to one."""
    a, _ = load_golden("train_grad_tiny")
    emb = t(a["inputs_embeds"])[0].to(DEV)
    labels = _shift(t(a["labels"])[0])
    eids = spliced_embed_ids(a["ids"][0], emb.shape[0])
    res = []
    for ck in (False, True):
        m, _, _ = tiny_model()
        tr = DecoderTrainer(m, lora=lora, gradient_checkpointing=ck)
        if lora is not None:                                             # B = 0 at init would leave the adapters' A without gradient: give B values
            tr.lora_bucket.w.copy_(torch.randn(tr.lora_bucket.w.shape, generator=torch.Generator().manual_seed(5)).to(tr.lora_bucket.w) * 0.05)
            tr._refresh_lora()
        tr.zero_grad()
        loss, dx = tr.forward_backward(emb, labels, embed_ids=eids)
        loss2, dx2 = tr.forward_backward(emb * 0.5, labels, embed_ids=eids)          # a second micro-batch accumulates through the same path
        stash_ptrs = {st["gu"].data_ptr() for st in tr.st}
        x_ptrs = {st["x_in"].data_ptr() for st in tr.st}
        res.append((float(loss), float(loss2), dx.clone(), dx2.clone(), [b.g.clone() for b in tr.buckets()], len(stash_ptrs), len(x_ptrs)))
        tr.detach()
    (l0, l0b, d0, d0b, g0, n0, x0), (l1, l1b, d1, d1b, g1, n1, x1) = res
    assert l0 == l1 and l0b == l1b and torch.equal(d0, d1) and torch.equal(d0b, d1b)
    assert len(g0) == len(g1) and all(torch.equal(x, y) for x, y in zip(g0, g1))
    assert n0 == x0 == x1 and n0 > 1 and n1 == 1, (n0, n1, x0, x1)        # one stash for all layers, an input stream per layer


def test_hf_gradient_checkpointing_switch_selects_the_recomputing_engine():
    """transformers' Trainer calls model.gradient_checkpointing_enable() under --gradient_checkpointing True and the reference then asks for
    enable_input_require_grads() (ufvideo/train.py:820-826): both exist on the drop-in model; the autograd path's engine is re-built with the re-computing backward and
    `.grad` comes out bit-identical."""
    a, _ = load_golden("train_grad_tiny")
    m, arrs, _ = tiny_model()
    ids, labels = t(a["ids"]).to(DEV), t(a["labels_in"]).to(DEV)
    batch = dict(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids), images=[(t(arrs["video"]).to(DEV), "video")],
                 images_sam=torch.zeros(1, 4, 3, 8, 8, device=DEV), offset=[0, 1], masks_list=[torch.zeros(0, 56, 56)], label_list=[torch.zeros(56, 56)])
    names = [n for n, _ in m.named_parameters() if n.startswith(("model.layers.", "model.norm.", "lm_head."))]
    own = dict(m.named_parameters())
    for n in names:
        own[n].requires_grad_(True)
    assert m.supports_gradient_checkpointing and not m.is_gradient_checkpointing
    m(**batch)["loss"].backward()
    plain = {n: own[n].grad.clone() for n in names}
    assert not m._engine[1].gradient_checkpointing
    for n in names:
        own[n].grad = None
    m.gradient_checkpointing_enable()
    m.enable_input_require_grads()
    assert m.is_gradient_checkpointing
    m(**batch)["loss"].backward()
    assert m._engine[1].gradient_checkpointing and len({st["gu"].data_ptr() for st in m._engine[1].st}) == 1
    assert all(torch.equal(own[n].grad, plain[n]) for n in names)
    m.gradient_checkpointing_disable()
    m.release_grad_engine()


def test_train_step_on_the_collator_batch_contract():
    """train_step(**batch) with the reference collator's keys: loss = the reference's ce_loss on the same sample (our tower +
    projector feed the splice), and repeated steps on the sample drive it down."""
    a, am_ = load_golden("train_grad_tiny")
    m, arrs, _ = tiny_model()
    tr = DecoderTrainer(m, lr=1e-3, weight_decay=0.01)
    ids, labels = t(a["ids"]).to(DEV), t(a["labels_in"]).to(DEV)
    video = t(arrs["video"]).to(DEV)
    batch = dict(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids), images=[(video, "video")], images_sam=None,
                 offset=[0, 1], masks_list=None, label_list=None)
    r1 = tr.train_step(**batch)
    assert abs(float(r1["loss"]) - float(a["ce_loss"])) < 2e-2 * float(a["ce_loss"])
    assert abs(float(r1["grad_norm"]) - float(a["grad_norm"])) < 5e-2 * float(a["grad_norm"])
    r2 = tr.train_step(**batch)
    r3 = tr.train_step(**batch)
    assert float(r3["loss"]) < float(r2["loss"]) < float(r1["loss"])
    # a batch of two samples of different length: loss is the token-weighted mean of the per-sample losses
    ids2 = torch.cat([ids, ids], 0); ids2[1, -3:] = 0
    am2 = torch.ones_like(ids2); am2[1, -3:] = 0
    lab2 = torch.cat([labels, labels], 0); lab2[1, -3:] = -100
    r = tr.train_step(input_ids=ids2, labels=lab2, attention_mask=am2, images=[(video, "video"), (video, "video")])
    assert torch.isfinite(r["loss"]) and float(r["grad_norm"]) > 0


def test_trainer_hands_weights_back_saves_resumes_and_refuses_seg_batches():
    """(i) while a trainer is attached, anything that would re-pack the model from its stale nn.Parameters raises; sync_to_model() puts the
    trained weights under the reference's names; detach() releases the model and the re-packed model computes what the trainer's
    buffers computed; (ii) state_dict() / load_state_dict() of the optimizer shards: a resumed trainer takes bit-identical steps;
    (iii) a batch with [SEG] targets or ground-truth masks is refused (its mask-loss backward does not exist) instead of being
    trained on the CE term alone."""
    a, _ = load_golden("train_grad_tiny")
    m, arrs, _ = tiny_model()
    ids, labels = t(a["ids"]).to(DEV), t(a["labels_in"]).to(DEV)
    video = t(arrs["video"]).to(DEV)
    batch = dict(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids), images=[(video, "video")], images_sam=None,
                 offset=[0, 1], masks_list=None, label_list=None)
    before = m.state_dict()["model.layers.0.mlp.down_proj.weight"].clone()
    tr = DecoderTrainer(m, lr=1e-3, weight_decay=0.01)
    tr.train_step(**batch)
    for f in (lambda: m.to(DEV), lambda: m.set_gemm_dtype("bf16"), lambda: m.get_model().invalidate(), lambda: m.load_state_dict({}, strict=False)):
        with pytest.raises(RuntimeError, match="DecoderTrainer owns"):
            f()
    assert torch.equal(m.state_dict()["model.layers.0.mlp.down_proj.weight"], before)           # the Parameters are stale ...
    tr.sync_to_model()
    after = m.state_dict()["model.layers.0.mlp.down_proj.weight"]
    assert not torch.equal(after, before) and torch.equal(after, tr.export_state_dict()["model.layers.0.mlp.down_proj.weight"])   # ... until synced
    # (ii) resume: save, step, restore into a fresh trainer on a fresh model, step -> same parameters bit for bit
    snap_w = {k: v.clone() for k, v in tr.export_state_dict().items()}
    snap_o = tr.state_dict()
    tr.train_step(**batch)
    want = tr.export_state_dict()
    sam = torch.zeros(1, 4, 3, 8, 8, device=DEV)
    logits_attached = m(input_ids=ids, attention_mask=torch.ones_like(ids), images=[(video, "video")], images_sam=sam, inference=True).logits.clone()
    tr.detach()
    logits_detached = m(input_ids=ids, attention_mask=torch.ones_like(ids), images=[(video, "video")], images_sam=sam, inference=True).logits
    # detach() loses nothing but the fp32 precision of the norm weights / biases (the trainer keeps them as fp32 masters, the
    # model's parameters are bf16 like the reference's checkpoints)
    assert rel_err(logits_detached, logits_attached) < 1e-2
    m.get_model().invalidate()                                                                  # allowed again
    m2, _, _ = tiny_model()
    m2.load_state_dict({**m2.state_dict(), **snap_w}, strict=True)
    tr2 = DecoderTrainer(m2, lr=1e-3, weight_decay=0.01)
    tr2.load_state_dict(snap_o)
    assert tr2.t == 1
    tr2.train_step(**batch)
    got = tr2.export_state_dict()
    for k in want:
        assert torch.equal(got[k], want[k]), k
    with pytest.raises(ValueError):
        tr2.load_state_dict(dict(snap_o, world=2))
    # (iii)
    seg = labels.clone(); seg[0, -2] = m2.config.seg_token_id
    with pytest.raises(NotImplementedError, match="mask-loss"):
        tr2.train_step(**dict(batch, labels=seg))
    with pytest.raises(NotImplementedError, match="mask-loss"):
        tr2.train_step(**dict(batch, masks_list=[torch.ones(1, 8, 8)]))
    tr2.train_step(**dict(batch, masks_list=[torch.zeros(0, 8, 8)]))                            # an empty ground truth is the no-[SEG] sample


# ---- projector backward ------------------------------------------------------------------------------------------------------

class _PCfg:
    def __init__(self, cin, hid):
        self.mm_hidden_size, self.hidden_size = cin, hid


@pytest.mark.parametrize("cin,hid,t,hw,depth", [(64, 64, 4, 4, 4), (128, 256, 4, 8, 2), (64, 64, 2, 6, 0)])
def test_projector_grad_vs_oracle_autograd(cin, hid, t, hw, depth):
    """every parameter gradient and the input gradient of the STC connector (v35: Conv3d 2x2x2 / padding 0) against torch autograd
    over the oracle restatement; the middle case runs its 1x1 convs on the MFMA tile kernels"""
    from ufvideo_amd.model.projector import STCConnectorV35
    from ufvideo_amd.train_projector import ProjectorGrad
    sd = O.make_stc_weights(cin, hid, seed=31, depth=depth)
    sd = {k: bfr(v * (3.0 if v.ndim >= 2 else 1.0)) for k, v in sd.items()}
    pj = STCConnectorV35(_PCfg(cin, hid), depth=depth)
    pj.load_state_dict(sd)
    pj = pj.to(DEV)
    g_ = torch.Generator().manual_seed(32)
    x = bfr(torch.randn(t * hw * hw, cin, generator=g_))
    n_out = (t // 2) * (hw // 2) ** 2
    dout = torch.randn(n_out, hid, generator=g_)
    with torch.enable_grad():
        p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xr = x.clone().requires_grad_(True)
        y = O.stc_connector(p, xr.view(1, t, hw * hw, cin), downsample=(2, 2, 2), padding=0, depth=depth)
        y[0].backward(dout)
    pg = ProjectorGrad(pj)
    out, st = pg.forward(x.to(DEV), t, hw)
    assert rel_err(out.cpu(), y[0].detach()) < 3e-2
    grads, dx = pg.backward(dout.to(DEV), st)
    assert set(grads) == set(sd), set(sd) ^ set(grads)
    worst = max((rel_err(grads[k].cpu().reshape(p[k].shape), p[k].grad), k) for k in sd)
    assert worst[0] < 6e-2, worst
    assert rel_err(dx.float().cpu(), xr.grad) < 6e-2


@pytest.mark.parametrize("depth,cin,hid,t,hw", [(2, 64, 128, 4, 6), (1, 128, 256, 3, 8), (3, 64, 64, 2, 5)])
def test_mlp_projector_grad_vs_autograd(depth, cin, hid, t, hw):
    """`mlpNx_gelu` / `linear` projectors in training (ref projector.py:95-108 + the mean over frames of temporal_aggregator, videorefer_arch.py:199-201):
    every parameter gradient and the input gradient against torch autograd of the same graph"""
    from ufvideo_amd.model.projector import MlpProjector
    from ufvideo_amd.train_projector import ProjectorGrad, MlpProjectorGrad
    g_ = torch.Generator().manual_seed(70 + depth)
    sd = {}
    for i in range(depth):
        sd[f"{2 * i}.weight"] = bfr(torch.randn(hid, cin if i == 0 else hid, generator=g_) * 0.1)
        sd[f"{2 * i}.bias"] = bfr(torch.randn(hid, generator=g_) * 0.1)
    pj = MlpProjector(cin, hid, depth)
    pj.load_state_dict(sd)
    pj = pj.to(DEV)
    x = bfr(torch.randn(t * hw * hw, cin, generator=g_))
    dout = torch.randn(hw * hw, hid, generator=g_)
    with torch.enable_grad():
        p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xr = x.clone().requires_grad_(True)
        h = bfr(xr.view(t, hw * hw, cin).mean(0))
        for i in range(depth):
            h = F.linear(h, p[f"{2 * i}.weight"], p[f"{2 * i}.bias"])
            if i < depth - 1:
                h = F.gelu(h)
        h.backward(dout)
    pg = ProjectorGrad(pj)
    assert isinstance(pg, MlpProjectorGrad)
    out, st = pg.forward(x.to(DEV), t, hw)
    assert out.shape == (hw * hw, hid) and rel_err(out.cpu(), h.detach()) < 2e-2
    assert rel_err(pj(x.view(1, t, hw * hw, cin).to(DEV).float().mean(1)).cpu()[0], h.detach()) < 2e-2        # == the inference path
    grads, dx = pg.backward(dout.to(DEV), st)
    assert set(grads) == set(sd)
    worst = max((rel_err(grads[k].cpu().reshape(p[k].shape), p[k].grad), k) for k in sd)
    assert worst[0] < 4e-2, worst
    assert rel_err(dx.float().cpu(), xr.grad) < 4e-2


def test_train_step_with_an_mlp2x_gelu_projector():
    """the drop-in step with the reference's `mlp2x_gelu` projector trained (adapter stage: tower + decoder frozen): the visual tokens are
    the MLP of the frame-mean features (16 tokens for the tiny tower), gradients land under the reference's names, the loss falls"""
    from test_model_gpu import TINY_LLM, TINY_VIT, Tok
    from ufvideo_amd.model import VideoReferQwen2Config, VideoReferQwen2ForCausalLM
    from ufvideo_amd.model.projector import MlpProjector
    a, w = load_golden("model_tiny")
    cfg = VideoReferQwen2Config(**TINY_LLM, mm_vision_tower="siglip", mm_vision_select_layer=-2, mm_vision_select_feature="patch",
                                mm_projector_type="mlp2x_gelu", mm_hidden_size=64, mm_region_encoder_type="pooling", image_aspect_ratio="square",
                                train_mask_decoder=False, sam_pretrained=None, sam_out_dim=256, num_frames=4, seg_token_id=299, vision_config=TINY_VIT, sam2_trunk=None)
    m = VideoReferQwen2ForCausalLM(cfg)
    m.get_vision_tower().load_model()
    m.load_state_dict({k: v for k, v in w.items() if not k.startswith("model.mm_projector.")}, strict=False)
    m = m.to(DEV)
    for mod in m.modules():
        mod.tokenizer = Tok()
    assert isinstance(m.get_model().mm_projector, MlpProjector)
    tr = DecoderTrainer(m, lr=3e-3, weight_decay=0.0, max_grad_norm=1.0, train_projector=True, train_decoder=False)
    ids = torch.tensor([[5, 6, -201, 7, 8, 9, 10, 11]], device=DEV)
    labels = ids.clone(); labels[labels < 0] = -100; labels[:, :3] = -100
    video = t(a["video"]).to(DEV)
    before = tr.proj_bucket.master.clone()
    losses = [float(tr.train_step(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids), images=[(video, "video")])["loss"]) for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert float((tr.proj_bucket.master - before).abs().max()) > 0
    assert set(k for k in tr.proj_params) == {"mm_projector.0.weight", "mm_projector.0.bias", "mm_projector.2.weight", "mm_projector.2.bias"}


@pytest.mark.parametrize("kind,cin,hid,t,hw,depth", [("stc_connector", 64, 64, 4, 4, 2), ("stc_connector", 128, 256, 3, 5, 1), ("spatial_conv", 64, 64, 3, 6, 0),
                                                     ("stp_connector", 64, 64, 4, 6, 2), ("stp_connector", 64, 128, 5, 5, 1), ("spatial_pool", 64, 64, 3, 5, 0)])
def test_projector_grad_other_connectors_vs_oracle_autograd(kind, cin, hid, t, hw, depth):
    """the rest of the STC family in training (VERDICT r1 missing #4): Conv3d with padding 1 (windows straddle the zero border: stc_connector,
    spatial_conv) and the AvgPool3d samplers (stp_connector, spatial_pool; odd sizes floor, so the last frame / row / column gets no gradient)
    against torch autograd over the oracle restatement of the reference's forward (projector.py:133-250)"""
    from ufvideo_amd.model.projector import STCConnector, STPConnector, SpatialConv, SpatialPool
    from ufvideo_amd.train_projector import ProjectorGrad
    cls, ds, pad, avg = {"stc_connector": (STCConnector, (2, 2, 2), 1, False), "spatial_conv": (SpatialConv, (1, 2, 2), 1, False),
                         "stp_connector": (STPConnector, (2, 2, 2), 0, True), "spatial_pool": (SpatialPool, (1, 2, 2), 0, True)}[kind]
    sd = O.make_stc_weights(cin, hid, seed=41, depth=depth, downsample=ds)
    if avg:
        sd = {k: v for k, v in sd.items() if not k.startswith("sampler.")}
    sd = {k: bfr(v * (3.0 if v.ndim >= 2 else 1.0)) for k, v in sd.items()}
    pj = cls(_PCfg(cin, hid), depth=depth) if kind in ("stc_connector", "stp_connector") else cls(_PCfg(cin, hid))
    pj.load_state_dict(sd)
    pj = pj.to(DEV)
    g_ = torch.Generator().manual_seed(42)
    x = bfr(torch.randn(t * hw * hw, cin, generator=g_))
    with torch.enable_grad():
        p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xr = x.clone().requires_grad_(True)
        y = O.stc_connector(p, xr.view(1, t, hw * hw, cin), downsample=ds, padding=pad, depth=depth, avgpool=avg)
        dout = torch.randn(y.shape[1:], generator=g_)
        y[0].backward(dout)
    pg = ProjectorGrad(pj)
    out, st = pg.forward(x.to(DEV), t, hw)
    assert out.shape == y.shape[1:] and rel_err(out.cpu(), y[0].detach()) < 3e-2
    with torch.no_grad():
        assert rel_err(pj(x.view(1, t, hw * hw, cin).to(DEV))[0].cpu(), y[0].detach()) < 3e-2          # and the inference path of the same module
    grads, dx = pg.backward(dout.to(DEV), st)
    assert set(grads) == set(sd), set(sd) ^ set(grads)
    worst = max((rel_err(grads[k].cpu().reshape(p[k].shape), p[k].grad), k) for k in sd)
    assert worst[0] < 6e-2, worst
    assert rel_err(dx.float().cpu(), xr.grad) < 6e-2
    if avg and depth == 0 and (t % ds[0] or hw % ds[1]):       # floor: pixels past the last window carry exactly no gradient (no RegStage in front)
        d4 = dx.float().cpu().view(t, hw, hw, cin)
        assert float(d4[:, (hw // ds[1]) * ds[1]:].abs().max() if hw % ds[1] else 0.0) == 0.0
        assert float(d4[(t // ds[0]) * ds[0]:].abs().max() if t % ds[0] else 0.0) == 0.0


def _mm_tiny_model():
    """tiny tower + STC-v35 connector + decoder with a one-video sample (weights from the oracle's generators)"""
    from ufvideo_amd.model import VideoReferQwen2Config, VideoReferQwen2ForCausalLM
    vit = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4, image_size=56, patch_size=14)
    llm = dict(vocab_size=300, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
               rope_theta=10000.0, rms_norm_eps=1e-6)
    cfg = VideoReferQwen2Config(**llm, mm_vision_tower="siglip", mm_vision_select_layer=-2, mm_vision_select_feature="patch",
                                mm_projector_type="stc_connector_v35", mm_hidden_size=64, mm_region_encoder_type="pooling",
                                image_aspect_ratio="square", train_mask_decoder=False, sam_out_dim=256, num_frames=4, sam2_trunk=None,
                                seg_token_id=299, vision_config=vit)
    sd = {}
    sd.update({"model.vision_tower.vision_tower.vision_model." + k: v for k, v in O.make_siglip_weights(vit, seed=1).items()})
    psd = {k: bfr(v * (3.0 if v.ndim >= 2 else 1.0)) for k, v in O.make_stc_weights(64, 64, seed=2).items()}
    sd.update({"model.mm_projector." + k: v for k, v in psd.items()})
    lsd = {k: bfr(v * 3 if v.ndim == 2 else v) for k, v in O.make_qwen2_weights(llm, seed=3).items()}
    sd.update(lsd)
    m = VideoReferQwen2ForCausalLM(cfg)
    m.get_vision_tower().load_model()
    m.load_state_dict(sd, strict=False)
    m = m.to(DEV)

    class Tok:
        def convert_tokens_to_ids(self, t_):
            return [290]
    for mod in m.modules():
        mod.tokenizer = Tok()
    g_ = torch.Generator().manual_seed(4)
    video = torch.randn(4, 3, 56, 56, generator=g_)
    ids = torch.tensor([[5, 6, -201, 7, 8, 9, 12, 13]])
    labels = ids.clone(); labels[labels < 0] = -100; labels[:, :2] = -100
    return m, sd, psd, lsd, vit, llm, video, ids, labels


def test_train_step_with_projector_grads_vs_oracle_autograd():
    """train_projector=True: tower (frozen) -> projector -> splice -> decoder -> CE; the projector's gradients that reach the
    optimizer equal torch autograd through the oracle's connector + splice + decoder, and the step lowers the loss"""
    m, sd, psd, lsd, vit, llm, video, ids, labels = _mm_tiny_model()
    tr = DecoderTrainer(m, lr=1e-3, train_projector=True)
    batch = dict(input_ids=ids.to(DEV), labels=labels.to(DEV), attention_mask=torch.ones_like(ids).to(DEV), images=[(video.to(DEV), "video")])
    r1 = tr.train_step(**batch)
    # oracle: autograd through connector + splice + decoder
    feats = O.siglip_tower(sd, vit, video, prefix="model.vision_tower.vision_tower.vision_model.")
    with torch.enable_grad():
        p = {k: v.clone().requires_grad_(True) for k, v in psd.items()}
        mm = O.stc_connector(p, feats[None])
        am, emb, lab2, _ = O.splice(lsd["model.embed_tokens.weight"].float(), ids, torch.ones_like(ids), labels, mm, [], [], 290, False)
        out = O.qwen2_forward(lsd, llm, emb, am)
        loss = O.causal_lm_loss(out["logits"], lab2)
        loss.backward()
    assert abs(float(r1["loss"]) - float(loss)) < 3e-2 * float(loss)
    pb = tr.proj_bucket
    worst = max((rel_err(pb.view(pb.g, "mm_projector." + k).cpu(), p[k].grad), k) for k in psd)
    assert worst[0] < 8e-2, worst
    r2 = tr.train_step(**batch)
    r3 = tr.train_step(**batch)
    assert float(r3["loss"]) < float(r1["loss"])
    # the connector's own parameters moved (and its packed copies were rebuilt)
    moved = sum(int((v.detach().float().cpu() != psd[k]).any()) for k, v in m.get_model().mm_projector.named_parameters())
    assert moved > len(psd) // 2


def test_adapter_only_step_freezes_the_decoder():
    """train_decoder=False = the reference's tune_mm_mlp_adapter stage (train.py:882-885: everything frozen, then the projector
    re-enabled).  Same kernels carry dL/dx through the frozen layers, so loss, projector gradients and the updated projector are
    bit-identical to the full trainer's (clipping off: the clip norm covers the trainable set, which differs); the decoder has
    no gradient / optimizer buffers and its weights do not move."""
    outs = []
    for td in (True, False):
        m, sd, psd, lsd, vit, llm, video, ids, labels = _mm_tiny_model()
        tr = DecoderTrainer(m, lr=1e-3, train_projector=True, train_decoder=td, max_grad_norm=0.0)
        before = [b.w.clone() for b in tr.layers] + [tr.head.w.clone(), tr.small.w.clone()]
        batch = dict(input_ids=ids.to(DEV), labels=labels.to(DEV), attention_mask=torch.ones_like(ids).to(DEV), images=[(video.to(DEV), "video")])
        r = tr.train_step(**batch)
        after = [b.w for b in tr.layers] + [tr.head.w, tr.small.w]
        outs.append((float(r["loss"]), tr.proj_bucket.g.clone(), tr.proj_bucket.w.clone(), [torch.equal(x, y) for x, y in zip(before, after)], tr))
    (l1, g1, w1, same1, _), (l2, g2, w2, same2, tr2) = outs
    assert l1 == l2 and torch.equal(g1, g2) and torch.equal(w1, w2)
    assert not any(same1[:-1]) and all(same2)                      # full trainer moved every matrix bucket; adapter-only moved none
    assert all(b.g is None and b.master is None for b in tr2.layers) and tr2.head.g is None
    assert tr2.buckets() == [tr2.proj_bucket]
    with pytest.raises(ValueError):
        DecoderTrainer(tr2.model, train_decoder=False)


def test_mm_projector_lr_group_and_lr_ratio():
    """--mm_projector_lr gives mm_projector.* its own rate (videorefer_trainer.py:278-306); set_lr_ratio scales every group"""
    ws = []
    for kw, ratio in ((dict(lr=2e-3), 1.0), (dict(lr=1e-3, mm_projector_lr=2e-3), 1.0), (dict(lr=8e-3, mm_projector_lr=4e-3), 0.5)):
        m, sd, psd, lsd, vit, llm, video, ids, labels = _mm_tiny_model()
        tr = DecoderTrainer(m, train_projector=True, train_decoder=False, max_grad_norm=0.0, **kw)
        tr.set_lr_ratio(ratio)
        tr.train_step(input_ids=ids.to(DEV), labels=labels.to(DEV), attention_mask=torch.ones_like(ids).to(DEV), images=[(video.to(DEV), "video")])
        ws.append(tr.proj_bucket.w.clone())
    assert torch.equal(ws[0], ws[1]) and torch.equal(ws[0], ws[2])


def test_train_step_with_region_encoder_grads_vs_oracle_autograd():
    """train_region_encoder=True on the golden `vid_region` sample (two region tokens): the MLP's gradients equal torch autograd
    through oracle mask_extractor + splice + decoder"""
    m, a, w = tiny_model()
    video, frame, mask = t(a["video"]).to(DEV), t(a["frame"]).to(DEV), t(a["mask"]).to(DEV)
    ids = t(a["sp_vid_region_ids"])
    labels = ids.clone(); labels[labels < 0] = -100; labels[:, :2] = -100
    tr = DecoderTrainer(m, lr=1e-3, train_region_encoder=True)
    r1 = tr.train_step(input_ids=ids.to(DEV), labels=labels.to(DEV), attention_mask=torch.ones_like(ids).to(DEV), images=[(video, "video")],
                       masks=[mask], frame=[frame], ann_indices=[[[0], [1]]], frame_nums=[2])
    vt = "model.vision_tower.vision_tower.vision_model."
    vt = vt if any(k.startswith(vt) for k in w) else "model.vision_tower.vision_tower."
    feats = O.siglip_tower(w, dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4, image_size=56, patch_size=14),
                           t(a["frame"]), prefix=vt)
    names = [k for k in w if k.startswith("model.region_encoder.")]
    with torch.enable_grad():
        p = {k: bfr(w[k]).requires_grad_(True) for k in names}
        full = dict(w); full.update(p)
        mf, nums = O.mask_extractor(full, feats, [t(a["mask"])], [[[0], [1]]], prefix="model.region_encoder.")
        wl = {k: bfr(v) for k, v in w.items()}
        am, emb, lab2, _ = O.splice(wl["model.embed_tokens.weight"].float(), ids, torch.ones_like(ids), labels, t(a["mm_features"]), mf, nums, 290, True)
        loss = O.causal_lm_loss(O.qwen2_forward(wl, TINY_LLM, emb, am)["logits"], lab2)
        loss.backward()
    assert abs(float(r1["loss"]) - float(loss)) < 3e-2 * float(loss)
    pb = tr.proj_bucket
    for k in names:
        kk = k[len("model."):]
        assert rel_err(pb.view(pb.g, kk).cpu(), p[k].grad) < 8e-2, k


# ---- projector-backward kernels alone vs torch ------------------------------------------------------------------------------

@pytest.mark.parametrize("act,fn", [("silu", F.silu), ("sigmoid", torch.sigmoid), ("gelu", F.gelu), ("relu", F.relu)])
def test_act_fwd_bwd_vs_autograd(act, fn):
    g = torch.Generator().manual_seed(41)
    pre, dout = bfr(torch.randn(33, 72, generator=g) * 2), bfr(torch.randn(33, 72, generator=g))
    pr = pre.clone().requires_grad_(True)
    y = fn(pr); y.backward(dout)
    assert rel_err(ops.act_fwd(pre.to(torch.bfloat16).to(DEV), act).float().cpu(), y.detach()) < 1e-2
    assert rel_err(ops.act_bwd(pre.to(torch.bfloat16).to(DEV), dout.to(torch.bfloat16).to(DEV), act).float().cpu(), pr.grad) < 1e-2


@pytest.mark.parametrize("M,C,act", [(37, 64, None), (300, 1152, "silu"), (1100, 3584, "silu"), (5000, 1152, "silu")])
def test_layernorm_bwd_vs_autograd(M, C, act):
    g = torch.Generator().manual_seed(M + C)
    x, dout = bfr(torch.randn(M, C, generator=g) * 2 + 0.3), bfr(torch.randn(M, C, generator=g))
    w, b = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.layer_norm(xr, (C,), wr, br, 1e-5)
    (F.silu(y) if act else y).backward(dout)
    dw, db = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    dx = ops.layernorm_bwd(x.to(torch.bfloat16).to(DEV), w.to(DEV), b.to(DEV), dout.to(torch.bfloat16).to(DEV), dw, db, 1e-5, act=act)
    assert rel_err(dx.float().cpu(), xr.grad) < 2e-2
    assert rel_err(dw.cpu() - 1, wr.grad) < 1e-2 and rel_err(db.cpu(), br.grad) < 1e-2          # dw / db are added to


@pytest.mark.parametrize("Fr,H,W,C", [(3, 5, 6, 64), (6, 12, 10, 328)])
def test_dwconv_se_gather_kernels_vs_torch(Fr, H, W, C):
    """the second shape spans two 32-chunk block columns (the last one partial) and several pixel slices"""
    g = torch.Generator().manual_seed(43)
    x = bfr(torch.randn(Fr, H, W, C, generator=g)); w9 = torch.randn(9, C, generator=g)
    dy = bfr(torch.randn(Fr, H, W, C, generator=g))
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr = w9.t().reshape(C, 1, 3, 3).clone().requires_grad_(True)
    y = F.conv2d(xr, wr, padding=1, groups=C)
    y.backward(dy.permute(0, 3, 1, 2))
    xd, dyd = x.reshape(-1, C).to(torch.bfloat16).to(DEV), dy.reshape(-1, C).to(torch.bfloat16).to(DEV)
    assert rel_err(ops.dwconv3x3(xd, w9.to(DEV), Fr, H, W).float().cpu().view(Fr, H, W, C), y.detach().permute(0, 2, 3, 1)) < 1e-2
    assert rel_err(ops.dwconv3x3(dyd, w9.to(DEV), Fr, H, W, flip=True).float().cpu().view(Fr, H, W, C), xr.grad.permute(0, 2, 3, 1)) < 1e-2
    dw9 = ops.dwconv3x3_dw(xd, dyd, torch.zeros(9, C, device=DEV), Fr, H, W)
    assert rel_err(dw9.cpu(), wr.grad.reshape(C, 9).t()) < 1e-2
    # SE pieces
    P = H * W
    a, b_ = bfr(torch.randn(Fr, P, C, generator=g)), bfr(torch.randn(Fr, P, C, generator=g))
    assert rel_err(ops.prod_colsum(a.to(torch.bfloat16).to(DEV), b_.to(torch.bfloat16).to(DEV), Fr, P).cpu(), (a * b_).sum(1)) < 1e-4
    gate, s = bfr(torch.rand(Fr, C, generator=g)), torch.randn(Fr, C, generator=g)
    out = ops.scale_add_bcast(a.reshape(-1, C).to(torch.bfloat16).to(DEV), gate.to(torch.bfloat16).to(DEV), s.to(DEV), 0.25, Fr, P)
    assert rel_err(out.float().cpu().view(Fr, P, C), a * gate[:, None] + 0.25 * s[:, None]) < 1e-2
    # conv3d scatter is the inverse of the non-overlapping gather (odd trailing rows get zeros)
    T, Hh, Ww = 5, 4, 6
    h = bfr(torch.randn(T * Hh * Ww, C, generator=g)).to(torch.bfloat16).to(DEV)
    A, (To, Ho, Wo) = ops.conv3d_gather(h, T, Hh, Ww, C, (2, 2, 2), 0)
    back = ops.conv3d_scatter(A, T, Hh, Ww, C, (2, 2, 2)).view(T, Hh, Ww, C)
    ref = h.view(T, Hh, Ww, C).clone(); ref[2 * To:] = 0
    assert torch.equal(back, ref)
    s2 = ops.add_bf16(h, h)
    assert torch.equal(s2.float(), (h.float() * 2).to(torch.bfloat16).float())


def test_forward_training_is_an_autograd_node_loss_backward_fills_grad_like_the_reference():
    """`ufvideo/train.py` under the HF Trainer calls `loss = model(**batch)["loss"]; loss.backward(); optimizer.step()`
    (videorefer_trainer.py:244-413 -> Trainer.training_step).  Here forward(inference=False) under autograd returns the loss attached to ONE
    autograd node; backward() leaves d(loss)/d(parameter) in `.grad` under the reference's parameter names -- checked against the REFERENCE's own
    loss.backward() (golden train_grad_tiny: every decoder gradient), accumulation over two backward calls, a torch optimizer stepping the
    nn.Parameters (the engine re-reads them: the next loss is lower), and no-grad forward still returning plain values."""
    a, _ = load_golden("train_grad_tiny")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_grad_tiny.npz"))
    m, arrs, _ = tiny_model()
    ids, labels = t(a["ids"]).to(DEV), t(a["labels_in"]).to(DEV)
    video = t(arrs["video"]).to(DEV)
    batch = dict(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids), images=[(video, "video")],
                 images_sam=torch.zeros(1, 4, 3, 8, 8, device=DEV), offset=[0, 1], masks_list=[torch.zeros(0, 56, 56)],
                 label_list=[torch.zeros(56, 56)])
    with torch.no_grad():
        plain = m(**batch)                                            # values only, no graph
    assert not plain["loss"].requires_grad
    names = [n for n, _ in m.named_parameters() if n.startswith(("model.layers.", "model.norm.", "lm_head.", "model.embed_tokens."))]
    own = dict(m.named_parameters())
    for n in names:
        own[n].requires_grad_(True)
    out = m(**batch)
    loss = out["loss"]
    assert loss.requires_grad and loss.grad_fn is not None and set(out) >= {"loss", "ce_loss", "mask_loss"}
    assert abs(float(loss) - float(a["ce_loss"])) < 2e-2 * float(a["ce_loss"]) and abs(float(loss) - float(plain["loss"])) < 2e-2 * float(loss)
    loss.backward()
    ref = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("g::")}
    assert set(ref) == set(names)
    worst = max((rel_err(own[k].grad.float().cpu(), ref[k]), k) for k in ref)
    assert worst[0] < 8e-2, worst                                      # bf16 .grad of a bf16 parameter; the fp32 buffers hold 5e-2 (test_tiny_model_step_vs_reference_golden)
    assert all(p.grad is None for n, p in m.named_parameters() if n not in names)          # tower / projector stay frozen
    # accumulation: a second backward of a scaled loss adds 0.5 x the same gradient
    g1 = {k: own[k].grad.clone() for k in names}
    (m(**batch)["loss"] * 0.5).backward()
    k = "model.layers.1.mlp.down_proj.weight"
    assert rel_err(own[k].grad.float(), 1.5 * g1[k].float()) < 1e-2
    # a torch optimizer on the nn.Parameters: the engine picks the new values up
    for p in own.values():
        p.grad = None
    opt = torch.optim.SGD([own[n] for n in names], lr=0.05)
    l0 = m(**batch)["loss"]; l0.backward(); opt.step(); opt.zero_grad()
    l1 = m(**batch)["loss"]; l1.backward(); opt.step(); opt.zero_grad()
    l2 = m(**batch)["loss"]
    assert float(l2) < float(l1) < float(l0), (float(l0), float(l1), float(l2))
    # giving the packed weights back: inference sees the trained parameters
    m.release_grad_engine()
    for p in own.values():
        p.requires_grad_(False)
    with torch.no_grad():
        after = m(**batch)
    assert abs(float(after["loss"]) - float(l2)) < 2e-2 * float(l2)
