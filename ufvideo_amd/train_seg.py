"""Mask-loss training of the [SEG] head (the part after the causal-LM term).

The reference adds  bce_loss_weight * BCE + dice_loss_weight * DICE  of the SAM2 masks of every [SEG] target to the loss
(ufvideo/model/videorefer_qwen2.py:225-226, 279-338; losses :34-77) and, with `train_mask_decoder`, trains `text_hidden_fcs` and
`sam_mask_decoder` through it (videorefer_arch.py:124-149); the gradient also flows into the decoder through the last hidden state
of the positions in front of each [SEG].  The masks come from `get_sam2_embeddings_train` / `inject_language_embd_train`
(sam2.py:343-447): every object's embedding is queried on every SAM frame as an initial conditioning frame, the best-IoU
high-resolution mask is resized to the label size.

Here: the frozen SAM2 image encoder runs on the inference kernels (no gradient reaches it); the head -- text_hidden_fcs, the
two-way transformer (prompt tokens <-> image tokens), the two transposed convolutions with the high-resolution skips, the
hyper-network product, two bilinear resizes, BCE + DICE -- runs forward with its activations kept and backward through
`torch.autograd` used as a TAPE ONLY: every node is a `torch.autograd.Function` whose forward and backward are HIP kernels
(MFMA GEMMs, LayerNorm / activation forward + backward of the training path, the few-token attention / mask-product / resize /
loss kernels of csrc/seg_train.hip).  torch itself only moves data (views, permutes, cat / index of small tensors, the
accumulation of a gradient that has two consumers).  Activations are bf16 between kernels, statistics / losses / parameter
gradients fp32, like the decoder's training step.  The IoU and object-score heads get no gradient (the mask is PICKED by arg-max
IoU), exactly as in the reference.  Parity: tests/test_seg_train_gpu.py against torch autograd over the oracle restatement, which
tests/test_oracle_golden.py pins to the reference's own backward."""
import torch

from . import ops
from .train_projector import _lin_bwd

BF, F32 = torch.bfloat16, torch.float32


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _Cast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return ops.convert(_c(x), dtype)

    @staticmethod
    def backward(ctx, dy):
        return ops.convert(_c(dy), ctx.src), None


class _Lin(torch.autograd.Function):
    """y = act(x W^T + b): x bf16 [M, K], W fp32 [N, K] (leaf or a reshaped view of one), b fp32 [N] | None -> bf16 [M, N]"""

    @staticmethod
    def forward(ctx, x, W, b, act):
        x = _c(x)
        Wb = ops.convert(_c(W), BF)
        pre = ops.gemm(x, Wb, bias=None if b is None else _c(b))
        ctx.act, ctx.has_b = act, b is not None
        ctx.save_for_backward(x, Wb, pre if act else None)
        return ops.act_fwd(pre, act) if act else pre

    @staticmethod
    def backward(ctx, dy):
        x, Wb, pre = ctx.saved_tensors
        d = ops.act_bwd(pre, _c(dy), ctx.act) if ctx.act else _c(dy)
        dx, dW = _lin_bwd(x, Wb, d, want_dx=ctx.needs_input_grad[0])
        db = None
        if ctx.has_b:
            db = ops.colsum(d, torch.zeros((Wb.shape[0],), device=d.device, dtype=F32))
        return dx, dW, db, None


class _LN(torch.autograd.Function):
    """row LayerNorm (+ activation): x bf16 [M, C]; w, b fp32"""

    @staticmethod
    def forward(ctx, x, w, b, eps, act):
        x, w, b = _c(x), _c(w), _c(b)
        ctx.eps, ctx.act = eps, act
        ctx.save_for_backward(x, w, b)
        return ops.layernorm(x, w, b, eps, act=act)

    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        dw, db = torch.zeros_like(w), torch.zeros_like(b)
        dx = ops.layernorm_bwd(x, w, b, _c(dy), dw, db, ctx.eps, act=ctx.act)
        return dx, dw, db, None, None


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pre, act):
        pre = _c(pre)
        ctx.act = act
        ctx.save_for_backward(pre)
        return ops.act_fwd(pre, act)

    @staticmethod
    def backward(ctx, dy):
        (pre,) = ctx.saved_tensors
        return ops.act_bwd(pre, _c(dy), ctx.act), None


class _Add(torch.autograd.Function):
    """bf16 a + b (same shape)"""

    @staticmethod
    def forward(ctx, a, b):
        return ops.add_bf16(_c(a), _c(b))

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class _AddTable(torch.autograd.Function):
    """x bf16 [M, C] + constant fp32 table[m % rows] (position terms): the table gets no gradient"""

    @staticmethod
    def forward(ctx, x, table):
        return ops.add_bcast(_c(x), table, out_dtype=BF)

    @staticmethod
    def backward(ctx, dy):
        return dy, None


class _Attn(torch.autograd.Function):
    """softmax(q k^T / sqrt(hd)) v between few and many tokens: q [B*Nq, H*hd], k / v [B*Nk, H*hd] bf16"""

    @staticmethod
    def forward(ctx, q, k, v, B, H, Nq, Nk, hd):
        q, k, v = _c(q), _c(k), _c(v)
        o, lse = ops.small_attn_fwd(q, k, v, B, H, Nq, Nk, hd)
        ctx.dims = (B, H, Nq, Nk, hd)
        ctx.save_for_backward(q, k, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, dO):
        q, k, v, o, lse = ctx.saved_tensors
        dq, dk, dv = ops.small_attn_bwd(q, k, v, o, _c(dO), lse, *ctx.dims)
        return dq, dk, dv, None, None, None, None, None


class _MaskDot(torch.autograd.Function):
    """masks[b, p] = sum_c up[b*P + p, c] h[b, c]: up bf16, h fp32 -> fp32 [B, P]"""

    @staticmethod
    def forward(ctx, up, h, B, P):
        up, h = _c(up), _c(h)
        ctx.dims = (B, P)
        ctx.save_for_backward(up, h)
        return ops.mask_dot_fwd(up, h, B, P)

    @staticmethod
    def backward(ctx, dm):
        up, h = ctx.saved_tensors
        dup, dh = ops.mask_dot_bwd(up, h, _c(dm), *ctx.dims)
        return dup, dh, None, None


class _Resize(torch.autograd.Function):
    """F.interpolate(bilinear, align_corners=False) on fp32 [N, 1, H, W]"""

    @staticmethod
    def forward(ctx, x, size):
        x = _c(x)
        ctx.in_hw = tuple(x.shape[-2:])
        return ops.resize_bilinear(x, tuple(size))

    @staticmethod
    def backward(ctx, dy):
        return ops.resize_bilinear_bwd(_c(dy), ctx.in_hw), None


def lin(x, W, b=None, act=None):
    return _Lin.apply(x, W, b, act)


def ln(x, w, b, eps=1e-5, act=None):
    return _LN.apply(x, w, b, eps, act)


class SegHeadGrad:
    """Forward with loss + backward of the mask branch for ONE sample.  `params`: dict name -> fp32 leaf tensor (requires_grad) under
    the reference's names relative to `model.`: `text_hidden_fcs.0.{0,2}.{weight,bias}` and `mask_encoder.sam2_model.sam_mask_decoder.*`;
    frozen tensors (prompt-encoder embeddings, no_mem_embed, dense position table) are read from the model."""

    FCS = "text_hidden_fcs."
    DEC = "mask_encoder.sam2_model.sam_mask_decoder."

    def __init__(self, model):
        self.model = model
        self.enc = model.get_model().mask_encoder

    @staticmethod
    def trainable(model):
        """[(name relative to model.get_model(), parameter)] of the modules the reference trains through the mask loss"""
        inner = model.get_model()
        out = [("text_hidden_fcs." + k, v) for k, v in inner.text_hidden_fcs.named_parameters()]
        if inner.mask_encoder is not None:
            out += [(SegHeadGrad.DEC + k, v) for k, v in inner.mask_encoder.sam2_model.sam_mask_decoder.named_parameters()]
        return out

    # ---- pieces ---------------------------------------------------------------------------------------------------------------
    def _attention(self, P, pre, q_in, k_in, v_in, B, Nq, Nk, heads):
        g = lambda n: P[pre + n]                                     # noqa: E731
        q = lin(q_in, g("q_proj.weight"), g("q_proj.bias"))
        k = lin(k_in, g("k_proj.weight"), g("k_proj.bias"))
        v = lin(v_in, g("v_proj.weight"), g("v_proj.bias"))
        internal = q.shape[1]
        o = _Attn.apply(q, k, v, B, heads, Nq, Nk, internal // heads)
        return lin(o, g("out_proj.weight"), g("out_proj.bias"))

    def _two_way(self, P, src, pos, tokens, B, T, hw, heads, depth):
        """sam2.py TwoWayTransformer.forward (:1298-1333); src bf16 [B*hw, C] (constant), pos fp32 [hw, C] (constant), tokens bf16 [B*T, C]"""
        t = self.DEC + "transformer."
        queries, keys = tokens, src
        for i in range(depth):
            lp = f"{t}layers.{i}."
            if i == 0:
                queries = self._attention(P, lp + "self_attn.", queries, queries, queries, B, T, T, heads)
            else:
                q = _Add.apply(queries, tokens)
                queries = _Add.apply(queries, self._attention(P, lp + "self_attn.", q, q, queries, B, T, T, heads))
            queries = ln(queries, P[lp + "norm1.weight"], P[lp + "norm1.bias"])
            q, k = _Add.apply(queries, tokens), _AddTable.apply(keys, pos)
            queries = ln(_Add.apply(queries, self._attention(P, lp + "cross_attn_token_to_image.", q, k, keys, B, T, hw, heads)),
                         P[lp + "norm2.weight"], P[lp + "norm2.bias"])
            m = lin(lin(queries, P[lp + "mlp.layers.0.weight"], P[lp + "mlp.layers.0.bias"], "relu"), P[lp + "mlp.layers.1.weight"], P[lp + "mlp.layers.1.bias"])
            queries = ln(_Add.apply(queries, m), P[lp + "norm3.weight"], P[lp + "norm3.bias"])
            q, k = _Add.apply(queries, tokens), _AddTable.apply(keys, pos)
            keys = ln(_Add.apply(keys, self._attention(P, lp + "cross_attn_image_to_token.", k, q, queries, B, hw, T, heads)),
                      P[lp + "norm4.weight"], P[lp + "norm4.bias"])
        q, k = _Add.apply(queries, tokens), _AddTable.apply(keys, pos)
        queries = ln(_Add.apply(queries, self._attention(P, t + "final_attn_token_to_image.", q, k, keys, B, T, hw, heads)),
                     P[t + "norm_final_attn.weight"], P[t + "norm_final_attn.bias"])
        return queries, keys

    @staticmethod
    def _deconv(x, W, b, B, h, w):
        """ConvTranspose2d(k = s = 2) on token-major x bf16 [B*h*w, Cin]: one GEMM to [.., (dy, dx, Cout)] + pixel shuffle -> [B*2h*2w, Cout]"""
        cin, cout = W.shape[0], W.shape[1]
        y = lin(x, W.permute(2, 3, 1, 0).reshape(4 * cout, cin), b.repeat(4))
        return y.view(B, h, w, 2, 2, cout).permute(0, 1, 3, 2, 4, 5).reshape(B * 4 * h * w, cout)

    # ---- one sample ---------------------------------------------------------------------------------------------------------------
    def forward_backward(self, P, hidden_rows, images_sam, gt, label_hw, w_bce, w_dice, num_masks_total):
        """hidden_rows fp32 [n_obj, D] (requires_grad): final-norm hidden states of the positions in front of the [SEG] targets;
        images_sam [T, 3, S, S]; gt fp32 [T * n_obj, h, w] frame-major.  Runs forward + backward; parameter gradients land in
        `.grad` of the leaves in P, the hidden-state gradient in `hidden_rows.grad`.
        -> (mask_bce_loss, mask_dice_loss) contributions of this sample (already divided by the batch's mask count)"""
        enc, sam = self.enc, self.enc.sam2_model
        dec, pe = sam.sam_mask_decoder, sam.sam_prompt_encoder
        dev = hidden_rows.device
        n_obj = hidden_rows.shape[0]
        T = images_sam.shape[0]
        B = T * n_obj
        C, heads, depth = sam.hidden_dim, dec.num_heads, dec.depth
        with torch.no_grad():
            feats = sam.forward_image_tokens(images_sam.to(device=dev, dtype=next(enc.parameters()).dtype))      # frozen encoder (inference kernels)
            (t0, h0, w0), (t1, h1, w1), (t2, h, w) = feats
            hw = h * w
            per_obj = lambda tok, n: tok.view(T, n, -1)[:, None].expand(T, n_obj, n, tok.shape[-1]).reshape(B * n, tok.shape[-1])    # noqa: E731
            vec = (sam.no_mem_embed.detach().float().reshape(1, C) + pe.no_mask_embed.weight.detach().float().reshape(1, C)).contiguous()
            src = per_obj(ops.add_bcast(t2, vec, out_dtype=BF), hw).contiguous()              # image tokens + no_mem_embed + dense no-mask embedding
            pos = pe.dense_pe_tokens(h, w).float().contiguous()                                # [hw, C]
            nap = pe.not_a_point_embed.weight.detach().to(BF)
            f0, f1 = ops.convert(t0, BF), ops.convert(t1, BF)
        # ---- [SEG] embeddings and prompt tokens: [obj_score, iou, 4 x mask] + 2 x not-a-point + language
        fp = self.FCS + "0."
        x = _Cast.apply(hidden_rows, BF)
        emb = lin(lin(x, P[fp + "0.weight"], P[fp + "0.bias"], "relu"), P[fp + "2.weight"], P[fp + "2.bias"])          # [n_obj, C]
        d = self.DEC
        out_tok = _Cast.apply(torch.cat([P[d + "obj_score_token.weight"], P[d + "iou_token.weight"], P[d + "mask_tokens.weight"]], 0), BF)   # [6, C]
        nt = out_tok.shape[0] + 3
        tokens = torch.cat([out_tok[None].expand(B, -1, -1), nap[None].expand(B, 2, C), emb.repeat(T, 1)[:, None]], 1).reshape(B * nt, C)
        hs, keys = self._two_way(P, src, pos, tokens, B, nt, hw, heads, depth)
        hs = hs.view(B, nt, C)
        # ---- which of the three multimask outputs: arg-max of the IoU head (no gradient: a choice, not a value)
        with torch.no_grad():
            ip = d + "iou_prediction_head.layers."
            z = lin(lin(lin(hs[:, 1], P[ip + "0.weight"], P[ip + "0.bias"], "relu"), P[ip + "1.weight"], P[ip + "1.bias"], "relu"),
                    P[ip + "2.weight"], P[ip + "2.bias"])
            best = torch.argmax(z.float()[:, 1:], dim=-1) + 1                                   # sigmoid is monotone; token 0 is dropped
        # ---- upscaling with the high-resolution skips (conv_s0 / conv_s1 belong to the mask decoder and train)
        s0 = lin(f0, P[d + "conv_s0.weight"].reshape(C // 8, C), P[d + "conv_s0.bias"])                            # [T*16hw, C/8]
        s1 = lin(f1, P[d + "conv_s1.weight"].reshape(C // 4, C), P[d + "conv_s1.bias"])                            # [T*4hw, C/4]
        up = self._deconv(keys, P[d + "output_upscaling.0.weight"], P[d + "output_upscaling.0.bias"], B, h, w)      # [B*4hw, C/4]
        up = ln(_Add.apply(up, per_obj(s1, 4 * hw)), P[d + "output_upscaling.1.weight"], P[d + "output_upscaling.1.bias"], 1e-6, "gelu")
        up = self._deconv(up, P[d + "output_upscaling.3.weight"], P[d + "output_upscaling.3.bias"], B, 2 * h, 2 * w)   # [B*16hw, C/8]
        up = _Act.apply(_Add.apply(up, per_obj(s0, 16 * hw)), "gelu")
        # ---- hyper-network of the chosen mask token, mask logits, two resizes (sam2.py :3409-3421, videorefer_qwen2.py:293)
        hyper = []
        for j in range(dec.num_mask_tokens):
            hp = f"{d}output_hypernetworks_mlps.{j}.layers."
            xj = hs[:, 2 + j]
            hyper.append(lin(lin(lin(xj, P[hp + "0.weight"], P[hp + "0.bias"], "relu"), P[hp + "1.weight"], P[hp + "1.bias"], "relu"),
                             P[hp + "2.weight"], P[hp + "2.bias"]))
        hsel = torch.stack(hyper, 1)[torch.arange(B, device=dev), best]                          # [B, C/8]
        Pm = 16 * hw
        masks = _MaskDot.apply(up, _Cast.apply(hsel, F32), B, Pm).view(B, 1, 4 * h, 4 * w)
        high = _Resize.apply(masks, (sam.image_size, sam.image_size))
        pred = _Resize.apply(high, tuple(label_hw))[:, 0].contiguous()                            # [B, h', w'] fp32
        # ---- BCE + DICE (values) and their gradient w.r.t. the logits; the tape does the rest
        gt = gt.to(device=dev, dtype=F32).contiguous()
        n = gt.shape[0]
        assert n == B, "gt_mask.shape: {}, pred_mask.shape: {}".format(tuple(gt.shape), tuple(pred.shape))
        with torch.no_grad():
            sums = ops.mask_loss_sums(pred, gt)
            HW = float(label_hw[0] * label_hw[1])
            k = n / (n + 1e-8) / (num_masks_total + 1e-8)
            num = 2.0 * sums[:, 1] / 1000.0 + 1e-6
            den = sums[:, 2] / 1000.0 + sums[:, 3] / 1000.0 + 1e-6
            bce = w_bce * k * (sums[:, 0] / HW).sum()
            dice = w_dice * k * (1.0 - num / den).sum()
            cd = w_dice * k
            coef = torch.stack([-cd * 2.0 / (1000.0 * den), cd * num / (1000.0 * den * den)], 1).contiguous()
            dpred = ops.mask_loss_bwd(pred, gt, coef, w_bce * k / HW)
        pred.backward(dpred)
        return bce, dice
