"""CPU oracle: a plain-torch fp32 restatement of the UFVideo hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``ufvideo_amd``) never imports anything from ``oracle/``.

What it restates (reference = /root/reference, third-party = transformers 4.46.3 /
timm 1.0.15 as pinned by the reference's requirements.txt):

* SigLIP vision tower            ufvideo/model/encoder.py:126-146  ->  HF modeling_siglip
                                 (embeddings :116-187, attention :227-307, MLP :310-322,
                                 layer :325-357)
* CLIP vision tower              ufvideo/model/encoder.py:12-93   ->  HF modeling_clip
* STC connector (+v35, spatial)  ufvideo/model/projector.py:133-250 -> timm RegStage
* region encoder                 ufvideo/model/layer.py:6-152
* embedding splice               ufvideo/model/videorefer_arch.py:218-370
* Qwen2 decoder                  ufvideo/model/videorefer_qwen2.py:129-197 -> HF modeling_qwen2
                                 (MLP :35-48, rotary :51-135, attention :150-235,
                                 RMSNorm :238-254, layer :258-299)
* greedy generate                ufvideo/model/videorefer_qwen2.py:357-459 (QA branch)
* frame sampling / tokenisation  ufvideo/mm_utils.py:43-54,135-158,381-406

Pinning status (see DESIGN.md "Oracle"): every function here except ``regstage`` is
checked against outputs of the reference itself (imported in the build container by
``oracle/gen_fixtures.py``; vectors committed under ``tests/golden/``).
``regstage`` restates timm's published RegNet bottleneck from its documented semantics;
timm is absent offline, so for that one function: PARITY UNPINNED.

All tensors are fp32 on CPU.  Weights arrive as flat ``dict[str, Tensor]`` with the
reference's state-dict key names (HF / timm naming) under a caller-chosen prefix.

Two precisions of the SAME graph:
* default: fp32 everywhere -- this is what is pinned against the reference's goldens;
* ``with bf16_mirror():`` -- the identical functions, with values rounded to bf16 (round-to-nearest-even,
  kept in fp32 storage) at exactly the points where the HIP path stores bf16: every parameter (the model keeps
  bf16 parameters, as the reference's checkpoints do), GEMM outputs that are not residual-stream updates, norm
  outputs, q / k after RoPE, the softmax numerators P before the PV product (the row sum uses the un-rounded
  values, as the kernel does), attention outputs, activation outputs.  The residual streams, norm statistics,
  softmax, and every accumulation stay fp32, as in the kernels (and as HF itself does at modeling_qwen2.py:238-254,
  modeling_siglip.py:241).  The mirror exists so the north-star tolerance ("within 1e-3 bf16 tolerance") can be
  asserted: HIP vs mirror isolates implementation error from the bf16 quantisation that any bf16 run carries
  (SURVEY 7.2 item (ii)); tests/test_oracle_golden.py pins the mirror against the fp32 graph on the goldens.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

IGNORE_INDEX = -100                                   # ufvideo/constants.py:7
MODAL_INDEX_MAP = {"<image>": -200, "<video>": -201, "<audio>": -202}   # constants.py:37-41

SD = Dict[str, torch.Tensor]


_MIRROR = False          # set by `bf16_mirror`; never by product code (the product never imports this module)
_LOG2E = 1.4426950408889634


class bf16_mirror:
    """Context manager: run the restatement with bf16 rounding at the HIP path's storage points (module docstring)."""

    def __enter__(self):
        global _MIRROR
        self._saved, _MIRROR = _MIRROR, True
        return self

    def __exit__(self, *exc):
        global _MIRROR
        _MIRROR = self._saved
        return False


def _rb(x: torch.Tensor) -> torch.Tensor:
    """Round to bf16 (RNE) in mirror mode; identity otherwise."""
    return x.to(torch.bfloat16).float() if _MIRROR else x


def _g(sd: SD, prefix: str, name: str) -> torch.Tensor:
    return _rb(sd[prefix + name].float())


def _flash_pv_mirror(s2: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """The MFMA flash kernels' arithmetic (csrc/attn.hip `attn_fwd_mfma*`), tile for tile, so that the bf16 rounding of P happens
    against the same running maximum as on the GPU: keys in tiles of 64; a wave owns 32 consecutive query rows; the running max
    of a row only moves when SOME row of its wave sees a tile max more than 2^6 above its running max (deferred rescale,
    RESCALE_THR = 6 in the log2 domain), and then every row of the wave takes max(m, tile max); O and l are rescaled in fp32.
    s2 = scores * scale * log2(e) with masked entries at -inf, [..., Sq, Sk]; v [..., Sk, hd]."""
    Sq, Sk = s2.shape[-2:]
    lead = s2.shape[:-2]
    ninf = float("-inf")
    m = torch.full(lead + (Sq,), ninf)
    l = torch.zeros(lead + (Sq,))
    acc = torch.zeros(lead + (Sq, v.shape[-1]))
    G = (Sq + 31) // 32
    for t0 in range(0, Sk, 64):
        st = s2[..., t0:t0 + 64]
        tmax = st.amax(dim=-1)
        trig_row = tmax > m + 6.0
        pad = G * 32 - Sq
        tr = torch.cat([trig_row, trig_row[..., -1:].expand(lead + (pad,))], dim=-1) if pad else trig_row
        trig = tr.reshape(lead + (G, 32)).any(dim=-1, keepdim=True).expand(lead + (G, 32)).reshape(lead + (G * 32,))[..., :Sq]
        mnew = torch.where(trig, torch.maximum(m, tmax), m)
        alpha = torch.where(mnew == ninf, torch.ones_like(m), torch.exp2(m - mnew))
        alpha = torch.where(trig, alpha, torch.ones_like(alpha))
        l = l * alpha
        acc = acc * alpha[..., None]
        m = mnew
        msub = torch.where(m == ninf, torch.zeros_like(m), m)
        p = torch.exp2(st - msub[..., None])
        l = l + p.sum(dim=-1)
        acc = acc + torch.matmul(_rb(p), v[..., t0:t0 + 64, :])
    return acc / l[..., None]


def _flash_vit72_mirror(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float) -> torch.Tensor:
    """The second-generation ViT kernel's arithmetic (csrc/attn_vit.inc `attn_fwd_vit72`: head_dim 72, non-causal; query counts that are a
    multiple of 288, and every count above 576 -- 729 tokens at 384 px), tile for tile: q' = bf16(q * (scale * log2 e)) (one rounding, the product of the two constants taken in
    fp32); scores s = q' k^T - m accumulated in fp32 with the running maximum m kept as a bf16 value; tiles of 64 keys, a wave owns 32
    query rows; tile 0 sets m = bf16(row max); afterwards m only moves when SOME row of the wave sees a tile maximum above 2^6
    (then every row takes m <- bf16(m + max(tile max, 0)) and O, l are rescaled by exp2(m_old - m_new) in fp32); P = bf16(exp2(s));
    the row sum l is the fp32 sum of the SAME bf16 P (the kernel gets it from the PV MFMAs through a ones column); O / l at the end.
    q, k, v [..., S, 72] fp32 holding bf16 values."""
    c = (torch.tensor(scale, dtype=torch.float32) * torch.tensor(1.4426950408889634, dtype=torch.float32))
    s_all = torch.matmul(_rb(q.float() * c), k.float().transpose(-1, -2))
    Sq, Sk = s_all.shape[-2:]
    lead = s_all.shape[:-2]
    m = torch.zeros(lead + (Sq,))
    l = torch.zeros(lead + (Sq,))
    acc = torch.zeros(lead + (Sq, v.shape[-1]))
    G = (Sq + 31) // 32
    for t0 in range(0, Sk, 64):
        st = s_all[..., t0:t0 + 64] - m[..., None]
        tmax = st.amax(dim=-1)
        if t0 == 0:
            m_new = _rb(m + tmax)
            trig = torch.ones_like(tmax, dtype=torch.bool)
        else:
            # a 32-row unit past the end of the sequence is filled with rows that never trigger on their own (attn_vit.inc repeats row Sq - 1,
            # the generated kernel reads zeros: q = 0 gives s - m = 0 in every tile)
            over = tmax > 6.0
            if G * 32 != Sq:
                over = torch.cat([over, torch.zeros(lead + (G * 32 - Sq,), dtype=torch.bool)], dim=-1)
            trig = over.reshape(lead + (G, 32)).any(dim=-1, keepdim=True).expand(lead + (G, 32)).reshape(lead + (G * 32,))[..., :Sq]
            m_new = torch.where(trig, _rb(m + tmax.clamp_min(0.0)), m)
        delta = m_new - m
        if t0:
            alpha = torch.where(trig, torch.exp2(-delta), torch.ones_like(delta))
            l = l * alpha
            acc = acc * alpha[..., None]
        st = st - delta[..., None]
        m = m_new
        p = _rb(torch.exp2(st))
        l = l + p.sum(dim=-1)
        acc = acc + torch.matmul(p, v[..., t0:t0 + 64, :].float())
    return acc / l[..., None]


def _softmax_pv(scores: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """softmax(scores) @ v; scores [..., Sq, Sk] (scaled, masked entries very negative), v [..., Sk, hd].  Mirror mode follows
    csrc/attn.hip: the MFMA flash kernels (dispatch rule of `ufv_attention`: head_dim in {64,72,80,96,128} and >= 16 queries)
    round p to bf16 as the operand of the second MFMA, tile by tile (`_flash_pv_mirror`); the scalar kernels (decode, other head
    dims) keep p = exp(s - max) in fp32.  In both the row sum uses the un-rounded p and the division comes last."""
    if not _MIRROR:
        return torch.matmul(torch.softmax(scores.float(), dim=-1), v)
    s2 = scores.float() * _LOG2E
    if v.shape[-1] in (64, 72, 80, 96, 128) and scores.shape[-2] >= 16:
        return _flash_pv_mirror(s2, v)
    p = torch.exp2(s2 - s2.amax(dim=-1, keepdim=True))
    return torch.matmul(p, v) / p.sum(dim=-1, keepdim=True)


def attention_noncausal(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float) -> torch.Tensor:
    """softmax(q k^T * scale) v for the vision towers; q, k, v [..., S, hd].  fp32 graph: exactly that (modeling_siglip.py:237-247).
    Mirror mode picks the arithmetic of the kernel `ufv_attention` dispatches to for this shape."""
    if _MIRROR and q.shape[-1] == 72 and (q.shape[-2] % 288 == 0 or q.shape[-2] > 576):       # ufv_attention: attn_vit_p2 / attn_fwd_vit72 (729 tokens at 384 px)
        return _flash_vit72_mirror(q, k, v, scale)
    return _softmax_pv(torch.matmul(q, k.transpose(-1, -2)) * scale, v)


# --------------------------------------------------------------------------------------
# activations
# --------------------------------------------------------------------------------------
def gelu_tanh(x):            # HF ACT2FN["gelu_pytorch_tanh"]
    return F.gelu(x, approximate="tanh")


def quick_gelu(x):           # HF ACT2FN["quick_gelu"]  (CLIP)
    return x * torch.sigmoid(1.702 * x)


_ACT = {"gelu_pytorch_tanh": gelu_tanh, "quick_gelu": quick_gelu, "gelu": F.gelu,
        "silu": F.silu, "relu": F.relu}


# --------------------------------------------------------------------------------------
# SigLIP / CLIP vision tower
# --------------------------------------------------------------------------------------
def vit_encoder_layer(sd: SD, p: str, x: torch.Tensor, heads: int, eps: float, act: str, stream_bf16: bool = False) -> torch.Tensor:
    """One pre-LN encoder layer (modeling_siglip.py:325-357; CLIP's layer is the same graph).  stream_bf16 (mirror mode only): the residual stream is stored
    in bf16 after both adds -- the product's opt-in UFV_TOWER_STREAM=bf16 (ufv_gemm_stream_bf16: one rounding of the fp32 sum), what a bf16 HF tower keeps."""
    rs = (lambda t: _rb(t)) if stream_bf16 else (lambda t: t)
    B, N, D = x.shape
    hd = D // heads
    h = _rb(F.layer_norm(x, (D,), _g(sd, p, "layer_norm1.weight"), _g(sd, p, "layer_norm1.bias"), eps))
    q = _rb(F.linear(h, _g(sd, p, "self_attn.q_proj.weight"), _g(sd, p, "self_attn.q_proj.bias")))
    k = _rb(F.linear(h, _g(sd, p, "self_attn.k_proj.weight"), _g(sd, p, "self_attn.k_proj.bias")))
    v = _rb(F.linear(h, _g(sd, p, "self_attn.v_proj.weight"), _g(sd, p, "self_attn.v_proj.bias")))
    q = q.view(B, N, heads, hd).transpose(1, 2)
    k = k.view(B, N, heads, hd).transpose(1, 2)
    v = v.view(B, N, heads, hd).transpose(1, 2)
    o = _rb(attention_noncausal(q, k, v, hd ** -0.5)).transpose(1, 2).reshape(B, N, D)   # :237-247 (fp32 softmax)
    o = F.linear(o, _g(sd, p, "self_attn.out_proj.weight"), _g(sd, p, "self_attn.out_proj.bias"))
    x = rs(x + o)
    h = _rb(F.layer_norm(x, (D,), _g(sd, p, "layer_norm2.weight"), _g(sd, p, "layer_norm2.bias"), eps))
    h = F.linear(h, _g(sd, p, "mlp.fc1.weight"), _g(sd, p, "mlp.fc1.bias"))
    h = _rb(_ACT[act](h))
    h = F.linear(h, _g(sd, p, "mlp.fc2.weight"), _g(sd, p, "mlp.fc2.bias"))
    return rs(x + h)


def siglip_embeddings(sd: SD, p: str, pixel_values: torch.Tensor, patch: int) -> torch.Tensor:
    """Conv2d(k=s=patch, valid) -> flatten -> + learned position table (modeling_siglip.py:175-186)."""
    w = _g(sd, p, "embeddings.patch_embedding.weight")
    b = _g(sd, p, "embeddings.patch_embedding.bias")
    e = F.conv2d(_rb(pixel_values.float()), w, b, stride=patch)        # mirror: the im2col matrix is bf16
    e = e.flatten(2).transpose(1, 2)
    return e + _g(sd, p, "embeddings.position_embedding.weight")[None]


def siglip_tower(sd: SD, cfg: dict, pixel_values: torch.Tensor, prefix: str = "",
                 select_layer: int = -2, return_all: bool = False, stream_bf16: bool = False):
    """SiglipVisionTower.forward + feature_select (encoder.py:126-146).

    HF ``hidden_states`` = [embeddings, layer1 out, ..., layerL out]; the reference picks
    ``hidden_states[select_layer]`` (=-2 -> output of layer L-1), so the last layer,
    post_layernorm and the pooling head never influence the result and are skipped.
    """
    p = prefix
    L = cfg["num_hidden_layers"]
    x = siglip_embeddings(sd, p, pixel_values, cfg["patch_size"])
    if stream_bf16:
        x = _rb(x)
    hs = [x]
    n_run = L if return_all else (L + 1 + select_layer if select_layer < 0 else select_layer)
    for i in range(n_run):
        x = vit_encoder_layer(sd, f"{p}encoder.layers.{i}.", x, cfg["num_attention_heads"],
                              cfg.get("layer_norm_eps", 1e-6), cfg.get("hidden_act", "gelu_pytorch_tanh"), stream_bf16)
        hs.append(x)
    if return_all:
        return hs
    return hs[select_layer] if select_layer >= 0 else hs[n_run]


def clip_tower(sd: SD, cfg: dict, pixel_values: torch.Tensor, prefix: str = "",
               select_layer: int = -2, select_feature: str = "patch") -> torch.Tensor:
    """CLIPVisionTower.forward + feature_select (encoder.py:36-58): class token + position
    table, pre_layrnorm, encoder layers, CLS dropped for 'patch'."""
    p = prefix
    L = cfg["num_hidden_layers"]
    D = cfg["hidden_size"]
    eps = cfg.get("layer_norm_eps", 1e-5)
    w = _g(sd, p, "embeddings.patch_embedding.weight")
    e = F.conv2d(_rb(pixel_values.float()), w, None, stride=cfg["patch_size"]).flatten(2).transpose(1, 2)
    cls = _g(sd, p, "embeddings.class_embedding").expand(e.shape[0], 1, -1)
    x = torch.cat([cls, e], dim=1) + _g(sd, p, "embeddings.position_embedding.weight")[None]
    x = F.layer_norm(x, (D,), _g(sd, p, "pre_layrnorm.weight"), _g(sd, p, "pre_layrnorm.bias"), eps)
    n_run = L + 1 + select_layer if select_layer < 0 else select_layer
    for i in range(n_run):
        x = vit_encoder_layer(sd, f"{p}encoder.layers.{i}.", x, cfg["num_attention_heads"], eps,
                              cfg.get("hidden_act", "quick_gelu"))
    if select_feature == "patch":
        x = x[:, 1:]
    return x


# --------------------------------------------------------------------------------------
# Projector: timm RegStage restatement + STC connector
# --------------------------------------------------------------------------------------
def layernorm2d(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    """timm LayerNorm2d: LayerNorm over the channel dim of an NCHW tensor."""
    return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, eps).permute(0, 3, 1, 2)


def regstage_block(sd: SD, p: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    """timm.models.regnet.Bottleneck with bottle_ratio=1, group_size=1 (depthwise 3x3),
    se_ratio=0.25, downsample='conv1x1', act=SiLU, norm=LayerNorm2d.  PARITY UNPINNED
    (timm 1.0.15 absent offline).  Graph:
        conv1 1x1 (no bias) -> LN2d -> SiLU
        conv2 3x3 depthwise pad 1 (no bias) -> LN2d -> SiLU
        se:   mean(H,W) -> fc1 (+bias) -> SiLU -> fc2 (+bias) -> sigmoid -> scale
        conv3 1x1 (no bias) -> LN2d
        + shortcut (1x1 conv + LN2d when in_chs != out_chs, identity otherwise) -> SiLU
    """
    sc = x
    y = _rb(F.conv2d(x, _g(sd, p, "conv1.conv.weight")))
    y = _rb(F.silu(layernorm2d(y, _g(sd, p, "conv1.bn.weight"), _g(sd, p, "conv1.bn.bias"), eps)))
    C = y.shape[1]
    y = F.conv2d(y, _g(sd, p, "conv2.conv.weight"), padding=1, groups=C)          # conv + LN + SiLU are one kernel: no rounding between
    y = _rb(F.silu(layernorm2d(y, _g(sd, p, "conv2.bn.weight"), _g(sd, p, "conv2.bn.bias"), eps)))
    s = _rb(y.mean((2, 3), keepdim=True))
    s = _rb(F.silu(F.conv2d(s, _g(sd, p, "se.fc1.weight"), _g(sd, p, "se.fc1.bias"))))
    s = _rb(torch.sigmoid(F.conv2d(s, _g(sd, p, "se.fc2.weight"), _g(sd, p, "se.fc2.bias"))))
    y = _rb(y * s)
    y = _rb(F.conv2d(y, _g(sd, p, "conv3.conv.weight")))
    y = layernorm2d(y, _g(sd, p, "conv3.bn.weight"), _g(sd, p, "conv3.bn.bias"), eps)
    if (p + "downsample.conv.weight") in sd:
        sc = _rb(F.conv2d(sc, _g(sd, p, "downsample.conv.weight")))
        sc = layernorm2d(sc, _g(sd, p, "downsample.bn.weight"), _g(sd, p, "downsample.bn.bias"), eps)
    return _rb(F.silu(y + sc))                                                    # LN(y) + LN(sc) + SiLU: one kernel


def regstage(sd: SD, p: str, x: torch.Tensor, depth: int, eps: float = 1e-5) -> torch.Tensor:
    for i in range(depth):
        x = regstage_block(sd, f"{p}b{i + 1}.", x, eps)
    return x


def stc_connector(sd: SD, x: torch.Tensor, prefix: str = "", downsample=(2, 2, 2), padding: int = 0,
                  depth: int = 4, mlp_depth: int = 2, ln_eps: float = 1e-5, avgpool: bool = False) -> torch.Tensor:
    """STCConnector.forward (projector.py:189-215).  v35 = padding 0, depth 4 (:225-238);
    'spatial_conv' = downsample (1,2,2), padding 1, depth 0 (:241-244); avgpool=True = the STPConnector / SpatialPool
    sampler nn.AvgPool3d(downsample) + SiLU (:218-222, :247-250)."""
    p = prefix
    b, t, n, d = x.shape
    hw = int(n ** 0.5)
    x = _rb(x.float()).view(b, t, hw, hw, d).permute(0, 1, 4, 2, 3).reshape(b * t, d, hw, hw)     # (b t) d h w
    if depth:
        x = regstage(sd, p + "s1.", x, depth, ln_eps)
    C = x.shape[1]
    x = x.view(b, t, C, hw, hw).permute(0, 2, 1, 3, 4)                              # b d t h w
    if avgpool:
        x = F.avg_pool3d(x, tuple(downsample))
    else:
        x = F.conv3d(x, _g(sd, p, "sampler.0.weight"), _g(sd, p, "sampler.0.bias"),
                     stride=downsample, padding=padding)
    x = _rb(F.silu(x))
    nt, nh, nw = x.shape[2:]
    x = x.permute(0, 2, 1, 3, 4).reshape(b * nt, C, nh, nw)
    if depth:
        x = regstage(sd, p + "s2.", x, depth, ln_eps)
    x = x.view(b, nt, C, nh * nw).permute(0, 1, 3, 2).reshape(b, nt * nh * nw, C)   # b (t h w) d
    for i in range(mlp_depth):                                                       # build_mlp :125-130
        if i:
            x = _rb(F.gelu(x))
        x = F.linear(x, _g(sd, p, f"readout.{2 * i}.weight"), _g(sd, p, f"readout.{2 * i}.bias"))
    return x


# --------------------------------------------------------------------------------------
# Region encoder  (layer.py)
# --------------------------------------------------------------------------------------
def token_merge(x: torch.Tensor, r: int) -> torch.Tensor:
    """layer.py:6-33 — greedy merge of adjacent tokens with cosine similarity >= the r-th largest."""
    x1, x2 = x[:, :-1, :], x[:, 1:, :]
    sim = torch.sum(F.normalize(x1, p=2, dim=-1) * F.normalize(x2, p=2, dim=-1), dim=-1)
    values, _ = torch.topk(sim.flatten(), r)
    kth = values[-1]
    new_tokens, merged = [], []
    for i in range(sim.shape[1]):
        merged.append(x[:, i:i + 1, :])
        if sim[0, i] < kth:
            new_tokens.append(torch.mean(torch.cat(merged, dim=1), dim=1, keepdim=True))
            merged = []
    merged.append(x[:, sim.shape[1]:sim.shape[1] + 1, :])
    new_tokens.append(torch.mean(torch.cat(merged, dim=1), dim=1, keepdim=True))
    return torch.cat(new_tokens, dim=1)


def mask_pooling(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """layer.py:135-152: bilinear resize -> (>0) -> masked mean. x [n,C,h,w], mask [1,q,H,W]."""
    if x.shape[-2:] != mask.shape[-2:]:
        mask = F.interpolate(mask, size=x.shape[-2:], mode="bilinear", align_corners=False)
    mask = (mask > 0).to(mask.dtype).permute(1, 0, 2, 3)
    denorm = mask.sum(dim=(-1, -2), keepdim=True) + 1e-8
    return (x * mask / denorm).sum(-1).sum(-1)


def mask_extractor(sd: SD, feats: torch.Tensor, masks: Sequence[torch.Tensor], ann_indices, prefix: str = "",
                   image_aspect_ratio: str = "square", region_token_num: int = 4):
    """MaskExtractor.forward (layer.py:63-128). Returns (region tokens [sum, D_llm], region_token_nums)."""
    p = prefix
    query_feats, region_token_nums = [], []
    for idx in range(len(masks)):
        mask = masks[idx].unsqueeze(0).float()
        if image_aspect_ratio == "pad":
            _h, w = mask.shape[-2:]
            m = max(_h, w)
            mask = F.pad(mask, ((m - w) // 2, (m - w) - (m - w) // 2, (m - _h) // 2, (m - _h) - (m - _h) // 2, 0, 0, 0, 0))
        ann_index = [i for index in ann_indices[idx] for i in index]
        feat = feats[ann_index].float()
        N = int(pow(feat.shape[1], 0.5))
        feat = feat.reshape(feat.shape[0], N, N, -1).permute(0, 3, 1, 2)
        raw = mask_pooling(feat, mask)
        merged, start = [], 0
        for index in ann_indices[idx]:
            mf = raw[start:start + len(index), :].unsqueeze(0)
            if mf.shape[1] > region_token_num:
                mf = token_merge(mf, mf.shape[1] - region_token_num)
            region_token_nums.append(mf.shape[1])
            merged.append(mf)
            start += len(index)
        query_feats.append(torch.cat(merged, dim=1).reshape(-1, raw.shape[-1]))
    mf = _rb(torch.cat(query_feats, dim=0))
    mf = F.linear(mf, _g(sd, p, "feat_linear.0.weight"), _g(sd, p, "feat_linear.0.bias"))
    mf = _rb(F.gelu(mf))
    mf = F.linear(mf, _g(sd, p, "feat_linear.2.weight"), _g(sd, p, "feat_linear.2.bias"))
    return mf, region_token_nums


# --------------------------------------------------------------------------------------
# Embedding splice  (videorefer_arch.py:218-370)
# --------------------------------------------------------------------------------------
def splice(embed_table: torch.Tensor, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor],
           labels: Optional[torch.Tensor], mm_features: torch.Tensor, mask_feats, region_token_nums,
           region_token_id: int, have_frame: bool):
    """prepare_inputs_labels_for_multimodal after the encoders have run.

    Returns (attention_mask, inputs_embeds, labels, mark_mm_token_indices) exactly as the
    reference builds them, including its quirks: the trailing-span bookkeeping of
    ``mark_mm_token_indices`` (:320-324), left-padding of the mask with True (:366), and the
    cur_region_idx/cur_region_pos counters advancing on samples without '<region>' (:293-298).
    """
    emb = lambda ids: embed_table[ids]
    mm_ids = list(MODAL_INDEX_MAP.values())
    new_embeds, new_labels = [], ([] if labels is not None else None)
    cur_mm, cur_ridx, cur_rpos = 0, 0, 0
    mark = []
    if not have_frame:
        mask_feats = []
    for b, ids in enumerate(input_ids):
        is_mm = torch.zeros_like(ids, dtype=torch.bool)
        for t in mm_ids:
            is_mm |= ids == t
        if int(is_mm.sum()) == 0:                                         # :251-267
            half = ids.shape[0] // 2
            new_embeds.append(torch.cat([emb(ids[:half]), emb(ids[half:])], dim=0))
            if labels is not None:
                new_labels.append(labels[b])
            cur_mm += 1; cur_ridx += 1; cur_rpos += 1
            continue
        parts, lparts = [], []
        cur_labels = labels[b] if labels is not None else None
        pos = torch.where(is_mm)[0]
        while pos.numel() > 0:                                            # :276-289
            feats = mm_features[cur_mm]
            s = int(pos[0])
            parts.append(emb(ids[:s])); parts.append(feats)
            if labels is not None:
                lparts.append(cur_labels[:s])
                lparts.append(torch.full((feats.shape[0],), IGNORE_INDEX, dtype=labels.dtype))
                cur_labels = cur_labels[s + 1:]
            cur_mm += 1
            ids = ids[s + 1:]
            is_mm = torch.zeros_like(ids, dtype=torch.bool)
            for t in mm_ids:
                is_mm |= ids == t
            pos = torch.where(is_mm)[0]
        if ids.numel() > 0:                                               # :291-318
            ridx = torch.nonzero(ids == region_token_id)
            if len(ridx) == 0:
                if have_frame:
                    parts.append(mask_feats[cur_ridx:cur_ridx + 1][0:0])
                cur_ridx += 1; cur_rpos += 1
            _l = 0
            for r in ridx:
                r0 = int(r[0])
                parts.append(emb(ids[_l:r0]))
                if labels is not None:
                    lparts.append(cur_labels[_l:r0])
                n = region_token_nums[cur_rpos]
                parts.append(mask_feats[cur_ridx:cur_ridx + n])
                if labels is not None:
                    lparts.append(torch.full((n,), IGNORE_INDEX, dtype=labels.dtype))
                cur_ridx += n; cur_rpos += 1
                _l = r0 + 1
            if _l < len(ids):
                parts.append(emb(ids[_l:]))
                if labels is not None:
                    lparts.append(cur_labels[_l:])
        last = parts[-1].shape[0]
        cat = torch.cat(parts, dim=0)
        mark.append([cat.shape[0] - last, last])
        new_embeds.append(cat)
        if labels is not None:
            new_labels.append(torch.cat(lparts, dim=0))

    if any(x.shape != new_embeds[0].shape for x in new_embeds):           # :334-359
        max_len = max(x.shape[0] for x in new_embeds)
        embeds = torch.stack([torch.cat((x, torch.zeros((max_len - x.shape[0], x.shape[1]), dtype=x.dtype)), 0)
                              for x in new_embeds], 0)
        if labels is not None:
            _nl = new_labels
            new_labels = torch.stack([torch.cat((x, torch.full((max_len - x.shape[0],), IGNORE_INDEX, dtype=x.dtype)), 0)
                                      for x in new_labels], 0)
            if attention_mask is not None:
                am = []
                for cur_am, cur_nl, cur_nla in zip(attention_mask, _nl, new_labels):
                    left = torch.full((cur_nl.shape[0] - labels.shape[1],), True, dtype=attention_mask.dtype)
                    right = torch.full((cur_nla.shape[0] - cur_nl.shape[0],), False, dtype=attention_mask.dtype)
                    am.append(torch.cat((left, cur_am, right), 0))
                attention_mask = torch.stack(am, 0)
    else:                                                                 # :360-368
        embeds = torch.stack(new_embeds, 0)
        if labels is not None:
            new_labels = torch.stack(new_labels, 0)
        if attention_mask is not None:
            left = torch.full((attention_mask.shape[0], embeds.shape[1] - input_ids.shape[1]), True,
                              dtype=attention_mask.dtype)
            attention_mask = torch.cat((left, attention_mask), dim=1)
    return attention_mask, embeds, new_labels, mark


# --------------------------------------------------------------------------------------
# Qwen2 decoder
# --------------------------------------------------------------------------------------
def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """Qwen2RMSNorm (modeling_qwen2.py:238-254)."""
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x.float() * torch.rsqrt(v + eps))


def rope_cos_sin(positions: torch.Tensor, head_dim: int, theta: float):
    """Qwen2RotaryEmbedding.forward (:51-102): fp32 outer product, cat(freqs, freqs)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    freqs = positions.float()[:, None] * inv[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def qwen2_layer(sd: SD, p: str, x: torch.Tensor, cfg: dict, cos, sin, kv: Optional[list], bias_mask):
    """Qwen2DecoderLayer (:258-299). x [B,S,D]; kv = [k_cache, v_cache] or None; returns (x, new_kv)."""
    B, S, D = x.shape
    H, KV = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    hd = cfg.get("head_dim", D // H)
    eps = cfg.get("rms_norm_eps", 1e-6)
    h = _rb(rmsnorm(x, _g(sd, p, "input_layernorm.weight"), eps))
    q = _rb(F.linear(h, _g(sd, p, "self_attn.q_proj.weight"), _g(sd, p, "self_attn.q_proj.bias"))).view(B, S, H, hd).transpose(1, 2)
    k = _rb(F.linear(h, _g(sd, p, "self_attn.k_proj.weight"), _g(sd, p, "self_attn.k_proj.bias"))).view(B, S, KV, hd).transpose(1, 2)
    v = _rb(F.linear(h, _g(sd, p, "self_attn.v_proj.weight"), _g(sd, p, "self_attn.v_proj.bias"))).view(B, S, KV, hd).transpose(1, 2)
    c, s = cos[None, None], sin[None, None]
    q = _rb(q * c + rotate_half(q) * s)
    k = _rb(k * c + rotate_half(k) * s)
    if kv is not None:
        k = torch.cat([kv[0], k], dim=2)
        v = torch.cat([kv[1], v], dim=2)
    new_kv = [k, v]
    rep = H // KV
    kk = k[:, :, None].expand(B, KV, rep, k.shape[2], hd).reshape(B, H, k.shape[2], hd)
    vv = v[:, :, None].expand(B, KV, rep, v.shape[2], hd).reshape(B, H, v.shape[2], hd)
    att = torch.matmul(q, kk.transpose(2, 3)) * (hd ** -0.5)
    att = att + bias_mask
    o = _rb(_softmax_pv(att, vv)).transpose(1, 2).reshape(B, S, H * hd)
    x = x + F.linear(o, _g(sd, p, "self_attn.o_proj.weight"))
    h = _rb(rmsnorm(x, _g(sd, p, "post_attention_layernorm.weight"), eps))
    g = F.linear(h, _g(sd, p, "mlp.gate_proj.weight"))
    u = F.linear(h, _g(sd, p, "mlp.up_proj.weight"))
    x = x + F.linear(_rb(F.silu(g) * u), _g(sd, p, "mlp.down_proj.weight"))       # SwiGLU is the gate/up GEMM's epilogue
    return x, new_kv


def qwen2_forward(sd: SD, cfg: dict, inputs_embeds: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                  past: Optional[list] = None, prefix: str = "model.", lm_head_key: str = "lm_head.weight",
                  all_logits: bool = True):
    """Qwen2ForCausalLM.forward with inputs_embeds (the reference never forwards position_ids,
    videorefer_qwen2.py:187-197, so positions = arange(past_len, past_len+S)).

    attention_mask: [B, past_len+S] (1 = attend).  Returns dict(logits, hidden_states, past).
    hidden_states follows HF: [embeds, layer1, ..., layer_{L-1}, final_norm(layer_L)].
    """
    B, S, D = inputs_embeds.shape
    L = cfg["num_hidden_layers"]
    H = cfg["num_attention_heads"]
    hd = cfg.get("head_dim", D // H)
    past_len = 0 if past is None else past[0][0].shape[2]
    T = past_len + S
    pos = torch.arange(past_len, T)
    cos, sin = rope_cos_sin(pos, hd, cfg.get("rope_theta", 1e6))
    neg = torch.finfo(torch.float32).min
    causal = torch.full((S, T), 0.0)
    causal = causal.masked_fill(torch.arange(T)[None, :] > pos[:, None], neg)
    bias = causal[None, None].expand(B, 1, S, T).clone()
    if attention_mask is not None:
        pad = (attention_mask[:, None, None, :T] == 0)
        bias = bias.masked_fill(pad, neg)
    x = inputs_embeds.float()
    hs = [x]
    new_past = []
    for i in range(L):
        x, kv = qwen2_layer(sd, f"{prefix}layers.{i}.", x, cfg, cos, sin, None if past is None else past[i], bias)
        new_past.append(kv)
        hs.append(x)
    x = rmsnorm(x, _g(sd, prefix, "norm.weight"), cfg.get("rms_norm_eps", 1e-6))
    hs[-1] = x
    head = _rb(sd[lm_head_key].float())
    logits = F.linear(_rb(x if all_logits else x[:, -1:]), head)
    return {"logits": logits, "hidden_states": hs, "past": new_past}


def greedy_generate(sd: SD, cfg: dict, inputs_embeds: torch.Tensor, attention_mask: torch.Tensor,
                    max_new_tokens: int, eos_token_ids: Sequence[int] = (), stop_fn=None,
                    embed_key: str = "model.embed_tokens.weight", **kw):
    """HF GenerationMixin greedy search as driven by videorefer_qwen2.py:414-426 (batch 1):
    prefill with inputs_embeds, then one token at a time through the KV cache; returns only
    the NEW tokens (HF returns no prompt ids when generation starts from inputs_embeds),
    plus per-step last-layer hidden states (what `output.hidden_states[o_idx][-1]` holds)."""
    table = _rb(sd[embed_key].float())
    out = qwen2_forward(sd, cfg, inputs_embeds, attention_mask, None, all_logits=False, **kw)
    tokens, hiddens = [], [out["hidden_states"][-1]]
    am = attention_mask
    for step in range(max_new_tokens):
        nxt = int(torch.argmax(out["logits"][0, -1]))
        tokens.append(nxt)
        if nxt in eos_token_ids or (stop_fn is not None and stop_fn(tokens)):
            break
        if step == max_new_tokens - 1:
            break
        am = torch.cat([am, torch.ones((am.shape[0], 1), dtype=am.dtype)], dim=1)
        out = qwen2_forward(sd, cfg, table[torch.tensor([[nxt]])], am, out["past"], all_logits=False, **kw)
        hiddens.append(out["hidden_states"][-1])
    return torch.tensor([tokens], dtype=torch.long), hiddens


# --------------------------------------------------------------------------------------
# Integer / host helpers  (mm_utils.py)
# --------------------------------------------------------------------------------------
def frame_sample(duration: int, mode: str = "uniform", num_frames: Optional[int] = None, fps=None) -> np.ndarray:
    """mm_utils.py:135-158."""
    if mode == "uniform":
        seg = float(duration - 1) / num_frames
        ids = [(seg * i + seg * (i + 1)) / 2 for i in range(num_frames)]
        return np.round(np.array(ids) + 1e-6).astype(int)
    if mode == "fps":
        seg_len = min(fps // 1, duration)                                 # NUM_FRAMES_PER_SECOND = 1
        return np.arange(seg_len // 2, duration, seg_len, dtype=int)
    raise ImportError(f"Unsupported frame sampling mode: {mode}")


def tokenizer_multimodal_token(prompt: str, tokenize, multimodal_token: str = "<image>") -> List[int]:
    """mm_utils.py:381-406 with `tokenize(str) -> list[int]`."""
    idx = MODAL_INDEX_MAP.get(multimodal_token, None)
    if idx is None:
        return list(tokenize(prompt))
    chunks = [list(tokenize(c)) for c in prompt.split(multimodal_token)]
    ids: List[int] = []
    for i in range(1, 2 * len(chunks)):
        if i % 2 == 1:
            ids.extend(chunks[i // 2])
        else:
            ids.append(idx)
    return ids


def text_hidden_fcs(sd: SD, x: torch.Tensor, prefix: str = "model.text_hidden_fcs.0.") -> torch.Tensor:
    """videorefer_arch.py:137-149: Linear -> ReLU -> Linear -> Dropout(0)."""
    h = _rb(F.relu(F.linear(_rb(x.float()), _g(sd, prefix, "0.weight"), _g(sd, prefix, "0.bias"))))
    return F.linear(h, _g(sd, prefix, "2.weight"), _g(sd, prefix, "2.bias"))


def siglip_preprocess(frames_u8: np.ndarray) -> torch.Tensor:
    """The arithmetic tail of SiglipImageProcessor after resize: x/255 -> (x-0.5)/0.5, HWC->CHW."""
    x = torch.from_numpy(np.ascontiguousarray(frames_u8)).float()
    x = (x * (1.0 / 255.0) - 0.5) / 0.5
    return x.permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------------------
# Synthetic weights (shared by tests/bench so GPU and CPU paths see identical parameters)
# --------------------------------------------------------------------------------------
def _randn(gen, *shape, std=0.02):
    return torch.randn(*shape, generator=gen) * std


def make_siglip_weights(cfg: dict, seed: int = 0, prefix: str = "", std: float = 0.02) -> SD:
    g = torch.Generator().manual_seed(seed)
    D, I, P = cfg["hidden_size"], cfg["intermediate_size"], cfg["patch_size"]
    n = (cfg["image_size"] // P) ** 2
    sd = {prefix + "embeddings.patch_embedding.weight": _randn(g, D, 3, P, P, std=std),
          prefix + "embeddings.patch_embedding.bias": _randn(g, D, std=std),
          prefix + "embeddings.position_embedding.weight": _randn(g, n, D, std=std)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"{prefix}encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"self_attn.{nm}.weight"] = _randn(g, D, D, std=std)
            sd[p + f"self_attn.{nm}.bias"] = _randn(g, D, std=std)
        for ln in ("layer_norm1", "layer_norm2"):
            sd[p + ln + ".weight"] = 1.0 + _randn(g, D, std=std)
            sd[p + ln + ".bias"] = _randn(g, D, std=std)
        sd[p + "mlp.fc1.weight"] = _randn(g, I, D, std=std); sd[p + "mlp.fc1.bias"] = _randn(g, I, std=std)
        sd[p + "mlp.fc2.weight"] = _randn(g, D, I, std=std); sd[p + "mlp.fc2.bias"] = _randn(g, D, std=std)
    return sd


def make_clip_weights(cfg: dict, seed: int = 0, prefix: str = "", std: float = 0.02) -> SD:
    """seeded CLIPVisionModel weights under HF's names (no patch bias, class embedding, n + 1 positions, pre_layrnorm)"""
    sd = make_siglip_weights(cfg, seed=seed, prefix=prefix, std=std)
    g = torch.Generator().manual_seed(seed + 1000)
    D, n = cfg["hidden_size"], (cfg["image_size"] // cfg["patch_size"]) ** 2
    del sd[prefix + "embeddings.patch_embedding.bias"]
    sd[prefix + "embeddings.class_embedding"] = _randn(g, D, std=std)
    sd[prefix + "embeddings.position_embedding.weight"] = _randn(g, n + 1, D, std=std)
    sd[prefix + "pre_layrnorm.weight"] = 1.0 + _randn(g, D, std=std)
    sd[prefix + "pre_layrnorm.bias"] = _randn(g, D, std=std)
    return sd


def make_regstage_weights(sd: SD, g, p: str, depth: int, cin: int, cout: int, std: float):
    for i in range(depth):
        b = f"{p}b{i + 1}."
        ci = cin if i == 0 else cout
        rd = int(round(ci * 0.25))
        sd[b + "conv1.conv.weight"] = _randn(g, cout, ci, 1, 1, std=std)
        sd[b + "conv2.conv.weight"] = _randn(g, cout, 1, 3, 3, std=0.2)
        sd[b + "conv3.conv.weight"] = _randn(g, cout, cout, 1, 1, std=std)
        for c in ("conv1", "conv2", "conv3"):
            sd[b + c + ".bn.weight"] = 1.0 + _randn(g, cout, std=std)
            sd[b + c + ".bn.bias"] = _randn(g, cout, std=std)
        sd[b + "se.fc1.weight"] = _randn(g, rd, cout, 1, 1, std=std); sd[b + "se.fc1.bias"] = _randn(g, rd, std=std)
        sd[b + "se.fc2.weight"] = _randn(g, cout, rd, 1, 1, std=std); sd[b + "se.fc2.bias"] = _randn(g, cout, std=std)
        if ci != cout:
            sd[b + "downsample.conv.weight"] = _randn(g, cout, ci, 1, 1, std=std)
            sd[b + "downsample.bn.weight"] = 1.0 + _randn(g, cout, std=std)
            sd[b + "downsample.bn.bias"] = _randn(g, cout, std=std)


def make_stc_weights(d_in: int, d_out: int, seed: int = 1, prefix: str = "", depth: int = 4,
                     downsample=(2, 2, 2), std: float = 0.02) -> SD:
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}
    if depth:
        make_regstage_weights(sd, g, prefix + "s1.", depth, d_in, d_out, std)
        make_regstage_weights(sd, g, prefix + "s2.", depth, d_out, d_out, std)
        c = d_out
    else:
        c = d_out                                                            # reference builds Conv3d(hidden, hidden)
    sd[prefix + "sampler.0.weight"] = _randn(g, d_out, c, *downsample, std=std)
    sd[prefix + "sampler.0.bias"] = _randn(g, d_out, std=std)
    sd[prefix + "readout.0.weight"] = _randn(g, d_out, d_out, std=std); sd[prefix + "readout.0.bias"] = _randn(g, d_out, std=std)
    sd[prefix + "readout.2.weight"] = _randn(g, d_out, d_out, std=std); sd[prefix + "readout.2.bias"] = _randn(g, d_out, std=std)
    return sd


def make_qwen2_weights(cfg: dict, seed: int = 2, std: float = 0.02) -> SD:
    g = torch.Generator().manual_seed(seed)
    D, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    H, KV = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    hd = cfg.get("head_dim", D // H)
    sd = {"model.embed_tokens.weight": _randn(g, V, D, std=std), "lm_head.weight": _randn(g, V, D, std=std),
          "model.norm.weight": 1.0 + _randn(g, D, std=std)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        sd[p + "self_attn.q_proj.weight"] = _randn(g, H * hd, D, std=std); sd[p + "self_attn.q_proj.bias"] = _randn(g, H * hd, std=std)
        sd[p + "self_attn.k_proj.weight"] = _randn(g, KV * hd, D, std=std); sd[p + "self_attn.k_proj.bias"] = _randn(g, KV * hd, std=std)
        sd[p + "self_attn.v_proj.weight"] = _randn(g, KV * hd, D, std=std); sd[p + "self_attn.v_proj.bias"] = _randn(g, KV * hd, std=std)
        sd[p + "self_attn.o_proj.weight"] = _randn(g, D, H * hd, std=std)
        sd[p + "mlp.gate_proj.weight"] = _randn(g, I, D, std=std)
        sd[p + "mlp.up_proj.weight"] = _randn(g, I, D, std=std)
        sd[p + "mlp.down_proj.weight"] = _randn(g, D, I, std=std)
        sd[p + "input_layernorm.weight"] = 1.0 + _randn(g, D, std=std)
        sd[p + "post_attention_layernorm.weight"] = 1.0 + _randn(g, D, std=std)
    return sd


# --------------------------------------------------------------------------------------
# SAM2 image encoder: Hiera trunk + FPN neck  (ufvideo/model/sam2.py:784-1258, 1736-1797)
# --------------------------------------------------------------------------------------
def hiera_schedule(cfg: dict):
    """Per-block (dim, dim_out, heads, window, q_stride) exactly as Hiera.__init__ derives them (sam2.py:1167-1226)."""
    stages, window_spec = cfg["stages"], cfg["window_spec"]
    depth = sum(stages)
    stage_ends = [sum(stages[:i]) - 1 for i in range(1, len(stages) + 1)]
    q_pool_blocks = [x + 1 for x in stage_ends[:-1]][:cfg.get("q_pool", 3)]
    glob = cfg.get("global_att_blocks", ())
    embed_dim, heads, cur_stage = cfg["embed_dim"], cfg["num_heads"], 1
    blocks = []
    for i in range(depth):
        dim_out = embed_dim
        window = window_spec[cur_stage - 1]            # lags one block behind the stage change
        if glob is not None and i in glob:
            window = 0
        if i - 1 in stage_ends:
            dim_out = int(embed_dim * cfg.get("dim_mul", 2.0))
            heads = int(heads * cfg.get("head_mul", 2.0))
            cur_stage += 1
        blocks.append(dict(dim=embed_dim, dim_out=dim_out, heads=heads, window=window, q_stride=2 if i in q_pool_blocks else 0))
        embed_dim = dim_out
    return blocks, stage_ends


def _window_partition(x, ws):
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)
    return x, (Hp, Wp)


def _window_unpartition(w, ws, pad_hw, hw):
    Hp, Wp = pad_hw
    H, W = hw
    B = w.shape[0] // (Hp * Wp // ws // ws)
    x = w.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :]


def hiera_block(sd: SD, p: str, x: torch.Tensor, blk: dict) -> torch.Tensor:
    """MultiScaleBlock.forward (sam2.py:1099-1131); x [B,H,W,C]."""
    dim, dim_out, heads, ws, qs = blk["dim"], blk["dim_out"], blk["heads"], blk["window"], blk["q_stride"]
    pool = (lambda t: F.max_pool2d(t.permute(0, 3, 1, 2), qs, qs).permute(0, 2, 3, 1)) if qs else (lambda t: t)
    shortcut = x
    x = F.layer_norm(x, (dim,), _g(sd, p, "norm1.weight"), _g(sd, p, "norm1.bias"), 1e-6)
    if dim != dim_out:
        shortcut = pool(F.linear(x, _g(sd, p, "proj.weight"), _g(sd, p, "proj.bias")))
    H, W = x.shape[1], x.shape[2]
    pad_hw = (H, W)
    if ws > 0:
        x, pad_hw = _window_partition(x, ws)
    B, h, w, _ = x.shape
    qkv = F.linear(x, _g(sd, p, "attn.qkv.weight"), _g(sd, p, "attn.qkv.bias")).reshape(B, h * w, 3, heads, -1)
    q, k, v = torch.unbind(qkv, 2)
    if qs:
        q = pool(q.reshape(B, h, w, -1))
        h, w = q.shape[1:3]
        q = q.reshape(B, h * w, heads, -1)
    hd = q.shape[-1]
    att = torch.softmax((q.transpose(1, 2) @ k.transpose(1, 2).transpose(-1, -2)) * hd ** -0.5, dim=-1)
    o = (att @ v.transpose(1, 2)).transpose(1, 2).reshape(B, h, w, -1)
    o = F.linear(o, _g(sd, p, "attn.proj.weight"), _g(sd, p, "attn.proj.bias"))
    if qs:
        ws2 = ws // qs
        H, W = shortcut.shape[1:3]
        pad_hw = (H + (ws2 - H % ws2) % ws2, W + (ws2 - W % ws2) % ws2) if ws2 > 0 else (H, W)
        if ws > 0:
            o = _window_unpartition(o, ws2, pad_hw, (H, W))
    elif ws > 0:
        o = _window_unpartition(o, ws, pad_hw, (H, W))
    x = shortcut + o
    h2 = F.layer_norm(x, (dim_out,), _g(sd, p, "norm2.weight"), _g(sd, p, "norm2.bias"), 1e-6)
    h2 = F.gelu(F.linear(h2, _g(sd, p, "mlp.layers.0.weight"), _g(sd, p, "mlp.layers.0.bias")))
    return x + F.linear(h2, _g(sd, p, "mlp.layers.1.weight"), _g(sd, p, "mlp.layers.1.bias"))


def hiera_pos_embed(sd: SD, p: str, hw) -> torch.Tensor:
    """Hiera._get_pos_embed (sam2.py:1232-1241) -> [1, h, w, C]."""
    pe = F.interpolate(_g(sd, p, "pos_embed"), size=hw, mode="bicubic")
    win = _g(sd, p, "pos_embed_window")
    pe = pe + win.tile([x // y for x, y in zip(pe.shape, win.shape)])
    return pe.permute(0, 2, 3, 1)


def hiera_forward(sd: SD, cfg: dict, img: torch.Tensor, prefix: str = "") -> List[torch.Tensor]:
    """Hiera.forward (sam2.py:1243-1258): stage-end features, NCHW, highest resolution first."""
    p = prefix
    x = F.conv2d(img.float(), _g(sd, p, "patch_embed.proj.weight"), _g(sd, p, "patch_embed.proj.bias"), stride=4, padding=3)
    x = x.permute(0, 2, 3, 1)
    x = x + hiera_pos_embed(sd, p, x.shape[1:3])
    blocks, stage_ends = hiera_schedule(cfg)
    outs = []
    for i, blk in enumerate(blocks):
        x = hiera_block(sd, f"{p}blocks.{i}.", x, blk)
        if i in stage_ends:
            outs.append(x.permute(0, 3, 1, 2))
    return outs


def position_embedding_sine(shape_bchw, num_pos_feats: int = 256, temperature: float = 10000.0) -> torch.Tensor:
    """PositionEmbeddingSine.forward with normalize=True, scale=2*pi (sam2.py:1797-1830)."""
    B, _, H, W = shape_bchw
    npf = num_pos_feats // 2
    y = torch.arange(1, H + 1, dtype=torch.float32).view(1, -1, 1).repeat(B, 1, W)
    x = torch.arange(1, W + 1, dtype=torch.float32).view(1, 1, -1).repeat(B, H, 1)
    eps, scale = 1e-6, 2 * math.pi
    y = y / (y[:, -1:, :] + eps) * scale
    x = x / (x[:, :, -1:] + eps) * scale
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / npf)
    px, py = x[:, :, :, None] / dim_t, y[:, :, :, None] / dim_t
    px = torch.stack((px[:, :, :, 0::2].sin(), px[:, :, :, 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[:, :, :, 0::2].sin(), py[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


def fpn_neck(sd: SD, feats: List[torch.Tensor], prefix: str = "", top_down_levels=(2, 3), d_model: int = 256):
    """FpnNeck.forward, nearest top-down, fuse 'sum' (sam2.py:871-903).  feats: highest resolution first.
    Returns (out, pos) lists in the same order."""
    n = len(feats) - 1
    out, pos, prev = [None] * len(feats), [None] * len(feats), None
    for i in range(n, -1, -1):
        lat = F.conv2d(feats[i], _g(sd, prefix, f"convs.{n - i}.conv.weight"), _g(sd, prefix, f"convs.{n - i}.conv.bias"))
        if i in top_down_levels and prev is not None:
            prev = lat + F.interpolate(prev.float(), scale_factor=2.0, mode="nearest")
        else:
            prev = lat
        out[i] = prev
        pos[i] = position_embedding_sine(prev.shape, d_model)
    return out, pos


def sam2_image_encoder(sd: SD, cfg: dict, img: torch.Tensor, prefix: str = "", scalp: int = 1):
    """ImageEncoder.forward (sam2.py:798-812)."""
    feats, pos = fpn_neck(sd, hiera_forward(sd, cfg, img, prefix + "trunk."), prefix + "neck.", cfg.get("fpn_top_down_levels", (2, 3)),
                          cfg.get("d_model", 256))
    if scalp > 0:
        feats, pos = feats[:-scalp], pos[:-scalp]
    return {"vision_features": feats[-1], "vision_pos_enc": pos, "backbone_fpn": feats}


def make_hiera_weights(cfg: dict, seed: int = 20, prefix: str = "trunk.", std: float = 0.02) -> SD:
    g = torch.Generator().manual_seed(seed)
    blocks, _ = hiera_schedule(cfg)
    E = cfg["embed_dim"]
    bs = cfg.get("window_pos_embed_bkg_spatial_size", (7, 7))
    sd = {prefix + "patch_embed.proj.weight": _randn(g, E, 3, 7, 7, std=0.05), prefix + "patch_embed.proj.bias": _randn(g, E, std=std),
          prefix + "pos_embed": _randn(g, 1, E, *bs, std=std), prefix + "pos_embed_window": _randn(g, 1, E, cfg["window_spec"][0], cfg["window_spec"][0], std=std)}
    for i, b in enumerate(blocks):
        p = f"{prefix}blocks.{i}."
        d, do = b["dim"], b["dim_out"]
        sd[p + "norm1.weight"] = 1 + _randn(g, d, std=std); sd[p + "norm1.bias"] = _randn(g, d, std=std)
        sd[p + "attn.qkv.weight"] = _randn(g, 3 * do, d, std=std); sd[p + "attn.qkv.bias"] = _randn(g, 3 * do, std=std)
        sd[p + "attn.proj.weight"] = _randn(g, do, do, std=std); sd[p + "attn.proj.bias"] = _randn(g, do, std=std)
        sd[p + "norm2.weight"] = 1 + _randn(g, do, std=std); sd[p + "norm2.bias"] = _randn(g, do, std=std)
        sd[p + "mlp.layers.0.weight"] = _randn(g, 4 * do, do, std=std); sd[p + "mlp.layers.0.bias"] = _randn(g, 4 * do, std=std)
        sd[p + "mlp.layers.1.weight"] = _randn(g, do, 4 * do, std=std); sd[p + "mlp.layers.1.bias"] = _randn(g, do, std=std)
        if d != do:
            sd[p + "proj.weight"] = _randn(g, do, d, std=std); sd[p + "proj.bias"] = _randn(g, do, std=std)
    return sd


def make_fpn_weights(channel_list, d_model=256, seed=21, prefix="neck.", std=0.02) -> SD:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for i, c in enumerate(channel_list):
        sd[prefix + f"convs.{i}.conv.weight"] = _randn(g, d_model, c, 1, 1, std=std)
        sd[prefix + f"convs.{i}.conv.bias"] = _randn(g, d_model, std=std)
    return sd


# --------------------------------------------------------------------------------------
# SAM2 heads with a language token as sparse prompt  (sam2.py:1260-1497 two-way transformer,
# :1565-1730 prompt encoder, :1940-2174 mask decoder, :3276-3452 _forward_sam_heads, :3174-3275 track_step)
# --------------------------------------------------------------------------------------
def _sam_attention(sd: SD, p: str, q, k, v, heads: int):
    """sam2.py `Attention.forward` (:1430-1497): projections to internal_dim, SDPA, out_proj."""
    q = F.linear(q, _g(sd, p, "q_proj.weight"), _g(sd, p, "q_proj.bias"))
    k = F.linear(k, _g(sd, p, "k_proj.weight"), _g(sd, p, "k_proj.bias"))
    v = F.linear(v, _g(sd, p, "v_proj.weight"), _g(sd, p, "v_proj.bias"))
    B, Nq, C = q.shape
    hd = C // heads
    sp = lambda t: t.reshape(B, t.shape[1], heads, hd).transpose(1, 2)
    att = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * hd ** -0.5, dim=-1)
    o = (att @ sp(v)).transpose(1, 2).reshape(B, Nq, C)
    return F.linear(o, _g(sd, p, "out_proj.weight"), _g(sd, p, "out_proj.bias"))


def _ln(sd, p, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), _g(sd, p, "weight"), _g(sd, p, "bias"), eps)


def _mlp(sd: SD, p: str, x, n_layers: int, sigmoid: bool = False):
    for i in range(n_layers):
        x = F.linear(x, _g(sd, p, f"layers.{i}.weight"), _g(sd, p, f"layers.{i}.bias"))
        if i < n_layers - 1:
            x = F.relu(x)
    return torch.sigmoid(x) if sigmoid else x


def sam_two_way_transformer(sd: SD, p: str, src, pos_src, tokens, depth: int = 2, heads: int = 8):
    """TwoWayTransformer.forward (:1298-1333); src/pos_src [B, HW, C], tokens [B, N, C]."""
    queries, keys = tokens, src
    for i in range(depth):
        lp = f"{p}layers.{i}."
        if i == 0:                                            # skip_first_layer_pe
            queries = _sam_attention(sd, lp + "self_attn.", queries, queries, queries, heads)
        else:
            q = queries + tokens
            queries = queries + _sam_attention(sd, lp + "self_attn.", q, q, queries, heads)
        queries = _ln(sd, lp + "norm1.", queries)
        q, k = queries + tokens, keys + pos_src
        queries = _ln(sd, lp + "norm2.", queries + _sam_attention(sd, lp + "cross_attn_token_to_image.", q, k, keys, heads))
        queries = _ln(sd, lp + "norm3.", queries + _mlp(sd, lp + "mlp.", queries, 2))
        q, k = queries + tokens, keys + pos_src
        keys = _ln(sd, lp + "norm4.", keys + _sam_attention(sd, lp + "cross_attn_image_to_token.", k, q, queries, heads))
    q, k = queries + tokens, keys + pos_src
    queries = _ln(sd, p + "norm_final_attn.", queries + _sam_attention(sd, p + "final_attn_token_to_image.", q, k, keys, heads))
    return queries, keys


def sam_dense_pe(gauss: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """PositionEmbeddingRandom.forward (:1869-1879) -> [C, h, w]."""
    grid = torch.ones((h, w), dtype=torch.float32)
    y, x = (grid.cumsum(0) - 0.5) / h, (grid.cumsum(1) - 0.5) / w
    c = (2 * torch.stack([x, y], dim=-1) - 1) @ gauss.float()
    c = 2 * np.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1).permute(2, 0, 1)


def _layernorm2d_cf(x, w, b, eps=1e-6):
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return w[:, None, None] * ((x - u) / torch.sqrt(s + eps)) + b[:, None, None]


def sam_heads_language(sd: SD, backbone_features, high_res_features, language_embd, image_size: int, prefix: str = ""):
    """_forward_sam_heads (:3276-3452) for the only prompt form the reference's inference uses: one dummy point
    (label -1) + its padding point + the language embedding, no mask prompt, multimask_output=True.
    backbone_features [B,C,h,w]; high_res_features = [feat_s0 [B,C/8,4h,4w], feat_s1 [B,C/4,2h,2w]]; language_embd [B,1,C].
    Returns dict(low_res_multimasks, ious, low_res_masks, high_res_masks, object_score_logits)."""
    p, d = prefix, prefix + "sam_mask_decoder."
    B, C, h, w = backbone_features.shape
    nap = _g(sd, p, "sam_prompt_encoder.not_a_point_embed.weight")                       # [1, C]
    sparse = torch.cat([nap[None].expand(B, 2, C), language_embd.float()], dim=1)       # 2 not-a-point tokens + language
    dense = _g(sd, p, "sam_prompt_encoder.no_mask_embed.weight").reshape(1, C, 1, 1)
    image_pe = sam_dense_pe(sd[p + "sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"], h, w)[None]
    out_tok = torch.cat([_g(sd, d, "obj_score_token.weight"), _g(sd, d, "iou_token.weight"), _g(sd, d, "mask_tokens.weight")], 0)
    tokens = torch.cat([out_tok[None].expand(B, -1, -1), sparse], dim=1)
    src = (backbone_features.float() + dense).flatten(2).permute(0, 2, 1)
    pos = image_pe.expand(B, -1, -1, -1).flatten(2).permute(0, 2, 1)
    hs, src = sam_two_way_transformer(sd, d + "transformer.", src, pos, tokens)
    iou_tok, mask_toks = hs[:, 1], hs[:, 2:6]
    src = src.transpose(1, 2).reshape(B, C, h, w)
    feat_s0, feat_s1 = high_res_features
    up = F.conv_transpose2d(src, _g(sd, d, "output_upscaling.0.weight"), _g(sd, d, "output_upscaling.0.bias"), stride=2)
    up = F.gelu(_layernorm2d_cf(up + feat_s1.float(), _g(sd, d, "output_upscaling.1.weight"), _g(sd, d, "output_upscaling.1.bias")))
    up = F.gelu(F.conv_transpose2d(up, _g(sd, d, "output_upscaling.3.weight"), _g(sd, d, "output_upscaling.3.bias"), stride=2) + feat_s0.float())
    hyper = torch.stack([_mlp(sd, d + f"output_hypernetworks_mlps.{i}.", mask_toks[:, i], 3) for i in range(4)], dim=1)
    b, c, H, W = up.shape
    masks = (hyper @ up.reshape(b, c, H * W)).reshape(b, -1, H, W)
    iou = _mlp(sd, d + "iou_prediction_head.", iou_tok, 3, sigmoid=True)
    obj = _mlp(sd, d + "pred_obj_score_head.", hs[:, 0], 3)
    masks, iou = masks[:, 1:], iou[:, 1:]                                               # multimask_output=True
    best = torch.argmax(iou, dim=-1)
    bi = torch.arange(B)
    low = masks[bi, best].unsqueeze(1)
    high = F.interpolate(masks, size=(image_size, image_size), mode="bilinear", align_corners=False)[bi, best].unsqueeze(1)
    return dict(low_res_multimasks=masks, ious=iou, low_res_masks=low, high_res_masks=high, object_score_logits=obj, best=best)


def sam2_language_masks(sd: SD, cfg: dict, images, language_embd, prefix: str = ""):
    """What `SAM2.get_sam2_embeddings` + `language_embd_inference` (sam2.py:378-410) compute for frames that are all
    initial conditioning frames: forward_image (+conv_s0/conv_s1, :2804-2816) -> + no_mem_embed (:2980-2984) ->
    SAM heads with the language token -> best-IoU low-res mask -> bilinear to the frame size.  images [F,3,S,S],
    language_embd [F,1,C] -> mask logits [F,1,S,S]."""
    p = prefix
    enc = sam2_image_encoder(sd, cfg, images, prefix=p + "image_encoder.")
    fpn = enc["backbone_fpn"]
    d = p + "sam_mask_decoder."
    s0 = F.conv2d(fpn[0], _g(sd, d, "conv_s0.weight"), _g(sd, d, "conv_s0.bias"))
    s1 = F.conv2d(fpn[1], _g(sd, d, "conv_s1.weight"), _g(sd, d, "conv_s1.bias"))
    pix = fpn[2] + _g(sd, p, "no_mem_embed").reshape(1, -1, 1, 1)
    out = sam_heads_language(sd, pix, [s0, s1], language_embd, images.shape[-1], prefix=p)
    out["video_res_masks"] = F.interpolate(out["low_res_masks"], size=images.shape[-2:], mode="bilinear", align_corners=False)
    return out


def make_sam_head_weights(C: int = 256, seed: int = 22, prefix: str = "", std: float = 0.05) -> SD:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    R = lambda *s, sc=std: _randn(g, *s, std=sc)
    def lin(p, o, i):
        sd[p + "weight"] = R(o, i); sd[p + "bias"] = R(o)
    def lnorm(p, n):
        sd[p + "weight"] = 1 + R(n); sd[p + "bias"] = R(n)
    def attn(p, internal):
        lin(p + "q_proj.", internal, C); lin(p + "k_proj.", internal, C); lin(p + "v_proj.", internal, C); lin(p + "out_proj.", C, internal)
    pe, d = prefix + "sam_prompt_encoder.", prefix + "sam_mask_decoder."
    sd[pe + "pe_layer.positional_encoding_gaussian_matrix"] = R(2, C // 2, sc=1.0)
    sd[pe + "not_a_point_embed.weight"] = R(1, C, sc=0.5); sd[pe + "no_mask_embed.weight"] = R(1, C, sc=0.5)
    for i in range(4):
        sd[pe + f"point_embeddings.{i}.weight"] = R(1, C)
    sd[prefix + "no_mem_embed"] = R(1, 1, C, sc=0.2)
    t = d + "transformer."
    for i in range(2):
        lp = t + f"layers.{i}."
        attn(lp + "self_attn.", C); attn(lp + "cross_attn_token_to_image.", C // 2); attn(lp + "cross_attn_image_to_token.", C // 2)
        for n in ("norm1.", "norm2.", "norm3.", "norm4."):
            lnorm(lp + n, C)
        lin(lp + "mlp.layers.0.", 2048 if C == 256 else 4 * C, C); lin(lp + "mlp.layers.1.", C, 2048 if C == 256 else 4 * C)
    attn(t + "final_attn_token_to_image.", C // 2); lnorm(t + "norm_final_attn.", C)
    sd[d + "iou_token.weight"] = R(1, C, sc=0.5); sd[d + "mask_tokens.weight"] = R(4, C, sc=0.5); sd[d + "obj_score_token.weight"] = R(1, C, sc=0.5)
    sd[d + "output_upscaling.0.weight"] = R(C, C // 4, 2, 2); sd[d + "output_upscaling.0.bias"] = R(C // 4)
    lnorm(d + "output_upscaling.1.", C // 4)
    sd[d + "output_upscaling.3.weight"] = R(C // 4, C // 8, 2, 2); sd[d + "output_upscaling.3.bias"] = R(C // 8)
    sd[d + "conv_s0.weight"] = R(C // 8, C, 1, 1); sd[d + "conv_s0.bias"] = R(C // 8)
    sd[d + "conv_s1.weight"] = R(C // 4, C, 1, 1); sd[d + "conv_s1.bias"] = R(C // 4)
    for i in range(4):
        hp = d + f"output_hypernetworks_mlps.{i}."
        lin(hp + "layers.0.", C, C); lin(hp + "layers.1.", C, C); lin(hp + "layers.2.", C // 8, C)
    ip = d + "iou_prediction_head."
    lin(ip + "layers.0.", 256, C); lin(ip + "layers.1.", 256, 256); lin(ip + "layers.2.", 4, 256)
    op = d + "pred_obj_score_head."
    lin(op + "layers.0.", C, C); lin(op + "layers.1.", C, C); lin(op + "layers.2.", 1, C)
    return sd


def seg_language_logits(sam_sd: SD, sam_cfg: dict, images_sam: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """`SAM2.language_embd_inference(state, [emb] * T)` (sam2.py:378-406): emb [n, C] is n objects on each of the T frames;
    each (frame, object) is an independent initial conditioning frame; `propagate_in_video` yields [n, 1, S, S] per frame and
    the results are concatenated over frames -> [T * n, 1, S, S], frame-major."""
    T, n = images_sam.shape[0], emb.shape[0]
    per_obj = [sam2_language_masks(sam_sd, sam_cfg, images_sam, emb[o].reshape(1, 1, -1).expand(T, 1, -1))["video_res_masks"] for o in range(n)]
    return torch.cat(per_obj, dim=1).reshape(T * n, 1, *images_sam.shape[-2:])


def seg_masks_generated(sd: SD, tokens, hiddens, seg_id: int, sam_sd: SD, sam_cfg: dict, images_sam, out_hw):
    """videorefer_qwen2.py:428-458 after greedy generation: step o contributes `hidden_states[o][-1]` when token o+1 is [SEG]
    (step 0's state covers the whole prompt), each row of text_hidden_fcs(states) is one single-object SAM2 query.
    -> (list of bool [T, h, w], list of logits [T, S, S])"""
    toks = tokens[0].tolist()
    hit = [o for o in range(len(toks) - 1) if toks[o + 1] == seg_id]
    if not hit:
        return [], []
    states = torch.cat([hiddens[o][0] for o in hit], dim=0)
    emb = text_hidden_fcs(sd, states)
    masks, logits = [], []
    for e in emb:
        lg = seg_language_logits(sam_sd, sam_cfg, images_sam, e[None])
        logits.append(lg[:, 0])
        masks.append(torch.sigmoid(F.interpolate(lg, size=tuple(out_hw), mode="bilinear", align_corners=False)[:, 0]) > 0.5)
    return masks, logits


def seg_masks_prompt(sd: SD, ids, mark, hidden_last, seg_id: int, sam_sd: SD, sam_cfg: dict, images_sam, out_hw):
    """videorefer_qwen2.py:461-518 ([SEG] already in the prompt, batch 1): the shifted [SEG] mask restricted to the trailing
    text segment (`mark` = [start, length] from the splice) selects rows of text_hidden_fcs(last hidden state); all n of them
    go to SAM2 together -> bool [T * n, h, w] + logits [T * n, S, S]."""
    m = torch.as_tensor(ids)[0] == seg_id
    m = torch.cat([m[1:], torch.zeros(1, dtype=torch.bool)])
    sel = torch.cat([torch.zeros(int(mark[0]), dtype=torch.bool), m[-int(mark[1]):]])
    emb = text_hidden_fcs(sd, hidden_last[0])[sel]
    lg = seg_language_logits(sam_sd, sam_cfg, images_sam, emb)
    return torch.sigmoid(F.interpolate(lg, size=tuple(out_hw), mode="bilinear", align_corners=False)[:, 0]) > 0.5, lg[:, 0]


# --------------------------------------------------------------------------------------
# W8A8 fp8 (OCP e4m3) restatement for config #5a (not a reference feature: the reference runs bf16/fp16; this states
# the arithmetic the fp8 GEMM path must reproduce, include/ufv.h `ufv_quantize_fp8` / `ufv_gemm_fp8`)
# --------------------------------------------------------------------------------------
def quantize_fp8_rows(x: torch.Tensor):
    """scale[m] = max|x[m]| / 448; q = round-to-nearest-even e4m3fn(x * (1 / scale)) -> (dequantised values fp32, scale, codes uint8)."""
    x = x.float()
    amax = x.abs().amax(dim=1, keepdim=True)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    q8 = (x * (1.0 / scale)).to(torch.float8_e4m3fn)
    return q8.float(), scale[:, 0], q8.view(torch.uint8)


def gemm_fp8(a: torch.Tensor, w: torch.Tensor, bias=None):
    """(q(a) q(w)^T) * sa[m] * sw[n] (+ bias) with per-token / per-output-channel e4m3 quantisation, fp32 accumulation."""
    qa, sa, _ = quantize_fp8_rows(a)
    qw, sw, _ = quantize_fp8_rows(w)
    y = (qa @ qw.t()) * sa[:, None] * sw[None, :]
    return y if bias is None else y + bias.float()


def mx_quantize(x: torch.Tensor):
    """MX-style block quantisation as the HIP path defines it (csrc/common.h mx_scale_byte; round 5): per (row, 32 consecutive elements) one e8m0 byte
    e = biased exponent of amax / 448, + 1 when its mantissa is not zero (amax / 448 rounded UP to a power of two: no element saturates), clamped to [1, 253];
    codes = round-to-nearest-even e4m3fn(x * 2^(127 - e)).  The layout is the OCP microscaling one that v_mfma_scale_f32_16x16x128_f8f6f4 consumes; the scale
    choice (round up instead of the spec's floor(log2 amax) - 8, which lets the block maximum saturate) is this builder's and is restated, not pinned.
    -> (dequantised values fp32 [M, K], scale bytes uint8 [M, K / 32], codes uint8 [M, K])"""
    x = x.float()
    M, K = x.shape
    assert K % 32 == 0
    xb = x.view(M, K // 32, 32)
    amax = xb.abs().amax(dim=2)
    bits = (amax * (1.0 / 448.0)).view(torch.int32)
    e = (bits >> 23) + ((bits & 0x7FFFFF) != 0).to(torch.int32)
    e = e.clamp(1, 253)
    inv = ((254 - e) << 23).view(torch.float32)                     # 2^(127 - e)
    q8 = (xb * inv[:, :, None]).to(torch.float8_e4m3fn)
    scale = (e << 23).view(torch.float32)                            # 2^(e - 127)
    deq = (q8.float() * scale[:, :, None]).view(M, K)
    return deq, e.to(torch.uint8), q8.view(torch.uint8).view(M, K)


def gemm_fp8_mx(a: torch.Tensor, w: torch.Tensor, bias=None):
    """(mx(a) q(w)^T) * sw[n] (+ bias): MX block-scaled activations, per-output-channel e4m3 weights, fp32 accumulation (ufv_gemm_fp8_mx with a_bscale)."""
    qa, _, _ = mx_quantize(a)
    qw, sw, _ = quantize_fp8_rows(w)
    y = (qa @ qw.t()) * sw[None, :]
    return y if bias is None else y + bias.float()


class fp8_linear_mode:
    """Context manager: inside it every `F.linear` of this module whose weight fits the fp8 MFMA tiles (N % 128 == 0,
    K % 128 == 0 -- the rule `PackedModule.gw` applies) runs the W8A8 restatement `gemm_fp8`; activations are first rounded
    to bf16, as the HIP path quantises the bf16 tensor the previous kernel wrote."""

    class _Shim:
        def __init__(self, real):
            self._real = real

        def __getattr__(self, name):
            return getattr(self._real, name)

        def linear(self, x, w, b=None):
            if w.dim() == 2 and w.shape[0] % 128 == 0 and w.shape[1] % 128 == 0:
                shp = x.shape
                y = gemm_fp8(x.reshape(-1, shp[-1]).to(torch.bfloat16), w.to(torch.bfloat16), b)
                return y.reshape(*shp[:-1], w.shape[0])
            return self._real.linear(x, w, b)

    def __enter__(self):
        g = globals()
        self._saved = g["F"]
        g["F"] = fp8_linear_mode._Shim(self._saved)
        return self

    def __exit__(self, *exc):
        globals()["F"] = self._saved
        return False


# --------------------------------------------------------------------------------------
# Training-loss forward (videorefer_qwen2.py:34-77 losses, :198-352 forward(inference=False)); forward values only
# --------------------------------------------------------------------------------------
def dice_loss(inputs: torch.Tensor, targets: torch.Tensor, num_masks: float, scale=1000, eps=1e-6):
    """videorefer_qwen2.py:34-57."""
    inputs = inputs.sigmoid().flatten(1, 2)
    targets = targets.flatten(1, 2)
    numerator = 2 * (inputs / scale * targets).sum(-1)
    denominator = (inputs / scale).sum(-1) + (targets / scale).sum(-1)
    loss = 1 - (numerator + eps) / (denominator + eps)
    return loss.sum() / (num_masks + 1e-8)


def sigmoid_ce_loss(inputs: torch.Tensor, targets: torch.Tensor, num_masks: float):
    """videorefer_qwen2.py:60-76."""
    loss = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    return loss.flatten(1, 2).mean(1).sum() / (num_masks + 1e-8)


def causal_lm_loss(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """HF Qwen2ForCausalLM loss: shift by one, mean cross entropy over labels != -100."""
    return F.cross_entropy(logits[..., :-1, :].reshape(-1, logits.shape[-1]).float(), labels[..., 1:].reshape(-1), ignore_index=-100)


def sampling_distribution(logits: torch.Tensor, temperature: float, top_k: int, top_p: float) -> torch.Tensor:
    """Probabilities HF's sampling chain draws from (GenerationMixin with the kwargs the reference forwards,
    ufvideo/__init__.py:113-127): TemperatureLogitsWarper -> TopKLogitsWarper (top_k > 0) -> TopPLogitsWarper (top_p < 1)
    -> softmax.  logits [V] fp32.  Pinned against transformers' own warpers in tests/test_oracle_golden.py."""
    x = logits.float() / temperature
    if top_k and top_k > 0:
        k = min(top_k, x.numel())
        x = x.masked_fill(x < torch.topk(x, k)[0][-1], float("-inf"))
    if top_p < 1.0:
        sl, si = torch.sort(x, descending=False)
        cum = sl.softmax(-1).cumsum(-1)
        rem = cum <= (1 - top_p)
        rem[-1:] = False
        x = x.masked_fill(torch.zeros_like(rem).scatter(0, si, rem), float("-inf"))
    return x.softmax(-1)


def sample_inverse_cdf(probs: torch.Tensor, u: float) -> int:
    """first index (vocabulary order) whose running mass exceeds u * total"""
    c = probs.double().cumsum(0)
    idx = int(torch.searchsorted(c, torch.tensor(u * float(c[-1]), dtype=torch.float64), right=True))
    nz = torch.nonzero(probs > 0).reshape(-1)
    return min(idx, int(nz[-1]))


DECODER_TRAINABLE = ("model.layers.", "model.norm.weight", "model.embed_tokens.weight", "lm_head.weight")


def decoder_train_grads(sd: SD, cfg: dict, inputs_embeds: torch.Tensor, labels: torch.Tensor):
    """torch autograd of the causal-LM loss through the decoder restatement: what `ce_loss.backward()` gives the reference
    (videorefer_qwen2.py:198-215 + HF Qwen2ForCausalLM).  inputs_embeds [1,S,D], labels [1,S] (un-shifted, -100 = ignored).
    Returns (loss, {name: grad} for every decoder parameter used, d loss / d inputs_embeds).  Pinned to the reference's own
    backward by tests/golden/train_grad_tiny.npz (oracle/gen_fixtures_train_grad.py)."""
    with torch.enable_grad():
        p = {k: v.detach().clone().float().requires_grad_(True) for k, v in sd.items()
             if k.startswith(DECODER_TRAINABLE[0]) or k in DECODER_TRAINABLE[1:]}
        e = inputs_embeds.detach().clone().float().requires_grad_(True)
        full = dict(sd); full.update(p)
        out = qwen2_forward(full, cfg, e, None)
        loss = causal_lm_loss(out["logits"], labels)
        loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()}
    return loss.detach(), grads, e.grad


def adamw_first_step(params: SD, grads: SD, lr: float, wd: float, betas=(0.9, 0.999), eps: float = 1e-8, clip: float = 1.0):
    """clip_grad_norm_(clip) + the first torch.optim.AdamW step (zero moments), weight decay on matrices only (HF Trainer's
    decay/no-decay split).  Returns (new params, pre-clip gradient norm)."""
    norm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(clip / (norm + 1e-6), max=1.0) if clip else 1.0
    b1, b2 = betas
    new = {}
    for k, p in params.items():
        g = grads[k] * coef
        p = p.float() * (1 - lr * (wd if p.ndim >= 2 else 0.0))
        m = (1 - b1) * g
        v = (1 - b2) * g * g
        new[k] = p - (lr / (1 - b1)) * m / (v.sqrt() / math.sqrt(1 - b2) + eps)
    return new, norm


def training_losses(sd: SD, cfg: dict, inputs_embeds, attention_mask, labels, seg_id: int, sam_sd: SD, sam_cfg: dict, images_sam,
                    masks_list, label_list, weights=(1.0, 1.0, 1.0)):
    """forward(inference=False) after the splice, batch 1 (the accelerated decoder's batch): CE on all positions, [SEG]
    embeddings = text_hidden_fcs(last hidden state) at positions whose NEXT label is [SEG], each object queried on every
    SAM frame (frame-major), best-IoU high-res mask -> bilinear to the label size -> BCE + DICE vs masks_list.
    images_sam [T,3,S,S].  -> dict(loss, ce_loss, mask_bce_loss, mask_dice_loss, mask_loss)"""
    w_ce, w_bce, w_dice = weights
    out = qwen2_forward(sd, cfg, inputs_embeds, attention_mask, None, all_logits=True)
    ce = causal_lm_loss(out["logits"], labels) * w_ce
    m = labels[0] == seg_id
    m = torch.cat([m[1:], torch.zeros(1, dtype=torch.bool)])
    emb = text_hidden_fcs(sd, out["hidden_states"][-1][0])[m]                       # [n_obj, 256]
    n_obj = emb.shape[0]
    T = images_sam.shape[0]
    bce = dice = torch.zeros(())
    if n_obj > 0:
        per_obj = [sam2_language_masks(sam_sd, sam_cfg, images_sam, emb[o].reshape(1, 1, -1).expand(T, 1, -1))["high_res_masks"] for o in range(n_obj)]
        high = torch.cat(per_obj, dim=1).reshape(T * n_obj, 1, *images_sam.shape[-2:])            # frame-major, as the reference expands
        pred = F.interpolate(high, size=tuple(label_list[0].shape), mode="bilinear", align_corners=False)[:, 0]
        gt = masks_list[0].float()
        n = gt.shape[0]
        bce = w_bce * (sigmoid_ce_loss(pred, gt, n) * n) / (n + 1e-8)
        dice = w_dice * (dice_loss(pred, gt, n) * n) / (n + 1e-8)
    return dict(loss=ce + bce + dice, ce_loss=ce, mask_bce_loss=bce, mask_dice_loss=dice, mask_loss=bce + dice)


# --------------------------------------------------------------------------------------
# Pillow's 8-bit bicubic resize (Image.resize(resample=BICUBIC) under HF SiglipImageProcessor, reference mm_utils.py:269-295):
# restated from the published algorithm of Pillow's Resample.c (precompute_coeffs / normalize_coeffs_8bpc /
# ImagingResampleHorizontal_8bpc / ...Vertical_8bpc); the GPU frame-batching path must match it bit for bit.
# --------------------------------------------------------------------------------------
PIL_PRECISION_BITS = 32 - 8 - 2


def _pil_bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_resize_coeffs(in_size: int, out_size: int):
    """-> (bounds int32 [out, 2] = (first input index, tap count), coeffs int32 [out, ksize] fixed point 2^22)"""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_pil_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(0.5 + k * (1 << PIL_PRECISION_BITS)) if k >= 0 else int(-0.5 + k * (1 << PIL_PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def pil_resize_bicubic_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """img uint8 [H, W, C] -> uint8 [out_h, out_w, C]: horizontal pass, round/clip to uint8, vertical pass, round/clip."""
    H, W, C = img.shape
    x = img.astype(np.int64)
    if W != out_w:
        b, k = pil_resize_coeffs(W, out_w)
        tmp = np.empty((H, out_w, C), dtype=np.int64)
        for xo in range(out_w):
            x0, n = b[xo]
            acc = (x[:, x0:x0 + n, :] * k[xo, :n].astype(np.int64)[None, :, None]).sum(1) + (1 << (PIL_PRECISION_BITS - 1))
            tmp[:, xo, :] = np.clip(acc >> PIL_PRECISION_BITS, 0, 255)
        x = tmp
    if H != out_h:
        b, k = pil_resize_coeffs(H, out_h)
        out = np.empty((out_h, x.shape[1], C), dtype=np.int64)
        for yo in range(out_h):
            y0, n = b[yo]
            acc = (x[y0:y0 + n] * k[yo, :n].astype(np.int64)[:, None, None]).sum(0) + (1 << (PIL_PRECISION_BITS - 1))
            out[yo] = np.clip(acc >> PIL_PRECISION_BITS, 0, 255)
        x = out
    return x.astype(np.uint8)
