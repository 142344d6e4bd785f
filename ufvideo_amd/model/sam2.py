"""SAM2 image encoder — Hiera trunk + FPN neck — behind the reference's module names
(ufvideo/model/sam2.py: `Hiera` :1134-1258, `MultiScaleBlock` :1049-1131, `MultiScaleAttention` :1000-1046,
`PatchEmbed` :954-984, `FpnNeck` :815-903, `ImageEncoder` :784-812, `PositionEmbeddingSine` :1736-1830).

This is the heavy part of the segmentation head (~1.8 TFLOP per 1024x1024 frame).  The SAM
heads driven by the LLM's [SEG] embedding (`PromptEncoder` :1565-1735, `TwoWayTransformer` :1260-1497, `MaskDecoder`
:1940-2224, `_forward_sam_heads` :3276-3452) and the `SAM2` wrapper's inference entry points
(`get_sam2_embeddings` / `language_embd_inference`, :378-413) follow below the encoder.

Execution model: tokens stay NHWC / token-major; the residual stream is fp32; every Linear (qkv, proj, MLP, the
dim-changing shortcut, the 7x7/s4 patch-embed conv via im2col, the FPN 1x1 convs) is an MFMA GEMM; window
partition is a row gather (index -1 = zero padding), window un-partition + residual is a permuted row add;
windowed / global attention (head_dim 72 for Hiera-L at every stage) runs on the flash kernel with the window as
the batch; q-pooling is a 2x2 max-pool over the window's token grid.  Channel counts that are not multiples of
128 (144/288/576 in Hiera-L) are zero-padded once at pack time (256/384/640) so that all GEMMs take the tiled
kernels; LayerNorm statistics use the true width.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ._params import Holder, PackedModule, init_tensor, bf, f32, round_up


def _pad_dim(d, wide192=False):
    """width a channel dimension is padded to for the GEMM kernels.  e4m3 operands (K-tiles 128 deep): multiples of 128.  bf16 operands (`wide192`, round 6): the smallest
    multiple of 64 (the K-tile) that a tile shape takes as an N -- a multiple of 128 (the 128- / 256-wide kernels) or, through the ping-pong kernel's 192-wide shapes, a multiple of
    192 or 128 past one (csrc/gemm.hip launch_any).  Hiera-L's widths 144 / 288 / 576 / 1152 become 192 / 320 / 576 / 1152 instead of 256 / 384 / 640 / 1152: its third stage
    (36 of 48 blocks) ran +11 % on every K and N of its GEMMs, the first two +33 / +78 %."""
    if d < 128:
        return d
    if not wide192:
        return round_up(d, 128)
    k = (d + 63) // 64
    while not (k % 2 == 0 or k % 3 in (0, 2)):
        k += 1
    return 64 * k


def _pad2(w, n, k):
    """zero-pad a [N, K] matrix to [n, k]"""
    if w.shape == (n, k):
        return w.contiguous()
    out = torch.zeros((n, k), device=w.device, dtype=w.dtype)
    out[: w.shape[0], : w.shape[1]] = w
    return out


def _pad1(v, n):
    if v.shape[0] == n:
        return v.contiguous()
    out = torch.zeros((n,), device=v.device, dtype=v.dtype)
    out[: v.shape[0]] = v
    return out


def hiera_schedule(embed_dim, num_heads, stages, window_spec, global_att_blocks, q_pool=3, dim_mul=2.0, head_mul=2.0):
    """Per-block (dim, dim_out, heads, window, q_stride) as the reference constructor derives them (sam2.py:1167-1226):
    the window size lags one block behind a stage change, blocks listed in global_att_blocks have window 0."""
    depth = sum(stages)
    stage_ends = [sum(stages[:i]) - 1 for i in range(1, len(stages) + 1)]
    q_pool_blocks = [x + 1 for x in stage_ends[:-1]][:q_pool]
    blocks, cur_stage = [], 1
    for i in range(depth):
        dim_out = embed_dim
        window = window_spec[cur_stage - 1]
        if global_att_blocks is not None and i in global_att_blocks:
            window = 0
        if i - 1 in stage_ends:
            dim_out = int(embed_dim * dim_mul)
            num_heads = int(num_heads * head_mul)
            cur_stage += 1
        blocks.append(dict(dim=embed_dim, dim_out=dim_out, heads=num_heads, window=window, q_stride=2 if i in q_pool_blocks else 0))
        embed_dim = dim_out
    return blocks, stage_ends


def window_index(B, H, W, ws, device):
    """Row indices of the window-partitioned order (sam2.py:905-925): position i of [B*nW, ws, ws] -> source row in
    [B, H, W] or -1 for zero padding.  Returns (idx int64 [B*nWh*nWw*ws*ws], (Hp, Wp))."""
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    Hp, Wp = H + ph, W + pw
    b = torch.arange(B).view(B, 1, 1, 1, 1)
    wy = torch.arange(Hp // ws).view(1, -1, 1, 1, 1)
    wx = torch.arange(Wp // ws).view(1, 1, -1, 1, 1)
    iy = torch.arange(ws).view(1, 1, 1, -1, 1)
    ix = torch.arange(ws).view(1, 1, 1, 1, -1)
    y, x = wy * ws + iy, wx * ws + ix
    idx = (b * H + y) * W + x
    idx = torch.where((y < H) & (x < W), idx, torch.full_like(idx, -1))
    return idx.reshape(-1).to(device), (Hp, Wp)


class Hiera(PackedModule):
    def __init__(self, embed_dim=96, num_heads=1, drop_path_rate=0.0, q_pool=3, q_stride=(2, 2), stages=(2, 3, 16, 3), dim_mul=2.0,
                 head_mul=2.0, window_pos_embed_bkg_spatial_size=(14, 14), window_spec=(8, 4, 14, 7), global_att_blocks=(12, 16, 20),
                 return_interm_layers=True, device=None, dtype=torch.bfloat16, seed=20, std=0.02):
        super().__init__()
        assert len(stages) == len(window_spec) and tuple(q_stride) == (2, 2)
        self.window_spec = tuple(window_spec)
        self.q_stride = tuple(q_stride)
        self.return_interm_layers = return_interm_layers
        self.global_att_blocks = global_att_blocks
        self.window_pos_embed_bkg_spatial_size = tuple(window_pos_embed_bkg_spatial_size)
        self.schedule, self.stage_ends = hiera_schedule(embed_dim, num_heads, stages, window_spec, global_att_blocks, q_pool, dim_mul, head_mul)
        self.q_pool_blocks = [x + 1 for x in self.stage_ends[:-1]][:q_pool]
        gen = torch.Generator(device=device if device is not None else "cpu").manual_seed(seed)
        mk = lambda shape, kind="w": init_tensor(shape, kind, gen, std, device, dtype)
        self.put("patch_embed.proj.weight", mk((embed_dim, 3, 7, 7))); self.put("patch_embed.proj.bias", mk((embed_dim,), "zero"))
        self.put("pos_embed", mk((1, embed_dim, *self.window_pos_embed_bkg_spatial_size)))
        self.put("pos_embed_window", mk((1, embed_dim, window_spec[0], window_spec[0])))
        for i, b in enumerate(self.schedule):
            p, d, do = f"blocks.{i}.", b["dim"], b["dim_out"]
            self.put(p + "norm1.weight", mk((d,), "one")); self.put(p + "norm1.bias", mk((d,), "zero"))
            self.put(p + "attn.qkv.weight", mk((3 * do, d))); self.put(p + "attn.qkv.bias", mk((3 * do,), "zero"))
            self.put(p + "attn.proj.weight", mk((do, do))); self.put(p + "attn.proj.bias", mk((do,), "zero"))
            self.put(p + "norm2.weight", mk((do,), "one")); self.put(p + "norm2.bias", mk((do,), "zero"))
            self.put(p + "mlp.layers.0.weight", mk((4 * do, do))); self.put(p + "mlp.layers.0.bias", mk((4 * do,), "zero"))
            self.put(p + "mlp.layers.1.weight", mk((do, 4 * do))); self.put(p + "mlp.layers.1.bias", mk((do,), "zero"))
            if d != do:
                self.put(p + "proj.weight", mk((do, d))); self.put(p + "proj.bias", mk((do,), "zero"))
        self.channel_list = ([self.schedule[i]["dim_out"] for i in self.stage_ends[::-1]] if return_interm_layers
                             else [self.schedule[-1]["dim_out"]])
        self._idx_cache = {}
        self._zbufs = {}

    def _zbuf(self, name, shape, device, dtype=torch.bfloat16):
        """Channel-padded scratch whose pad columns must read zero: allocated (zeroed) once per shape and reused -- kernels only
        ever write the valid columns, and everything runs on one stream, so the pads stay zero and no per-block memset is needed."""
        key = (name, tuple(shape), str(device), dtype)
        if key not in self._zbufs:
            self._zbufs[key] = torch.zeros(shape, device=device, dtype=dtype)
        return self._zbufs[key]

    # ---- packing -----------------------------------------------------------------------------------------------
    def _pack(self):
        E = self.schedule[0]["dim"]
        w192 = getattr(self, "gemm_dtype", "bf16") != "fp8"
        Ep = _pad_dim(E, w192)
        K = 3 * 49
        pk = {"Kp": round_up(K, 64), "Ep": Ep}
        pk["patch_w"] = bf(_pad2(self.patch_embed.proj.weight.reshape(E, K), Ep, pk["Kp"]))
        pk["patch_b"] = f32(_pad1(self.patch_embed.proj.bias, Ep))
        blocks = []
        for i, b in enumerate(self.schedule):
            L = self.blocks.get(str(i))
            d, do = b["dim"], b["dim_out"]
            dp, dop, hp = _pad_dim(d, w192), _pad_dim(do, w192), _pad_dim(4 * do, w192)
            wq = L.attn.qkv.weight
            bq = L.attn.qkv.bias
            wqkv = torch.cat([_pad2(wq[j * do:(j + 1) * do], dop, dp) for j in range(3)], 0)
            bqkv = torch.cat([_pad1(bq[j * do:(j + 1) * do], dop) for j in range(3)], 0)
            blk = dict(n1=(f32(L.norm1.weight), f32(L.norm1.bias)), n2=(f32(L.norm2.weight), f32(L.norm2.bias)),
                       # (self.gw: bf16, or e4m3 + per-channel scale under set_gemm_dtype("fp8") -- config #5: every padded dim is a multiple of 128)
                       wqkv=self.gw(wqkv), bqkv=f32(bqkv), wo=self.gw(_pad2(L.attn.proj.weight, dop, dop)), bo=f32(_pad1(L.attn.proj.bias, dop)),
                       w1=self.gw(_pad2(L.mlp.layers.get("0").weight, hp, dop)), b1=f32(_pad1(L.mlp.layers.get("0").bias, hp)),
                       w2=self.gw(_pad2(L.mlp.layers.get("1").weight, dop, hp)), b2=f32(_pad1(L.mlp.layers.get("1").bias, dop)),
                       dp=dp, dop=dop, hp=hp, proj=None)
            if d != do:
                blk["proj"] = (self.gw(_pad2(L.proj.weight, dop, dp)), f32(_pad1(L.proj.bias, dop)))
            blocks.append(blk)
        pk["blocks"] = blocks
        pk["pos"] = {}
        return pk

    def _get_pos_embed(self, hw):
        """[h*w, Ep] fp32 table: bicubic-resized background embedding + tiled window embedding (sam2.py:1232-1241).
        Depends on the input size only -> computed once per size with torch and cached with the packed weights."""
        pk = self.packed()
        if hw not in pk["pos"]:
            h, w = hw
            pe = F.interpolate(self.pos_embed.detach().float(), size=(h, w), mode="bicubic")
            win = self.pos_embed_window.detach().float()
            pe = pe + win.tile([x // y for x, y in zip(pe.shape, win.shape)])
            tab = pe.permute(0, 2, 3, 1).reshape(h * w, -1)
            out = torch.zeros((h * w, pk["Ep"]), device=tab.device, dtype=torch.float32)
            out[:, : tab.shape[1]] = tab
            pk["pos"][hw] = out
        return pk["pos"][hw]

    def _windows(self, B, H, W, ws, device):
        key = (B, H, W, ws, str(device))
        if key not in self._idx_cache:
            self._idx_cache[key] = window_index(B, H, W, ws, device)
        return self._idx_cache[key]

    keep_window_order = True      # consecutive blocks of one window size leave the token stream in window order (see _block)

    def _reorder(self, x, B, H, W, cur, want):
        """the fp32 stream from token order `cur` to `want` (0 = row-major, ws > 0 = the window-partitioned order of that window size; only sizes that
        divide the grid are kept as orders, so the index is a permutation)"""
        if cur == want:
            return x
        if cur:
            idx, _ = self._windows(B, H, W, cur, x.device)
            x = ops.gather_rows(x, None, torch.empty_like(x), idx)
        if want:
            idx, _ = self._windows(B, H, W, want, x.device)
            x = ops.gather_rows(x, idx, torch.empty_like(x), None)
        return x

    # ---- one MultiScaleBlock (sam2.py:1099-1131) on the fp32 stream x [B*H*W, dp] ----------------------------------------
    # `order`: the token order x arrives in.  LayerNorm, the MLP and the residual adds work row by row, and global attention sees all tokens of a frame
    # whatever their order inside the frame, so a run of blocks with the same window size keeps the stream in that window order: the per-block window
    # partition (a row gather) and un-partition (a permuted row add) become the identity (2 of ~13 passes over the stream per block); blocks that pool
    # or change the width, and the stage outputs, get row-major order back.  Every row goes through the same arithmetic: bit-identical outputs.
    def _block(self, x, blk, w, B, H, W, order=0):
        d, do, heads, ws, qs = blk["dim"], blk["dim_out"], blk["heads"], blk["window"], blk["q_stride"]
        dp, dop = w["dp"], w["dop"]
        dev = x.device
        N = B * H * W
        plain = w["proj"] is None and not qs
        if not (self.keep_window_order and plain):
            want = 0
        elif ws > 0:
            want = ws if (H % ws == 0 and W % ws == 0) else 0
        else:
            want = order
        x = self._reorder(x, B, H, W, order, want)
        order = want
        xn = self._zbuf("xn", (N, dp), dev) if dp != d else torch.empty((N, dp), device=dev, dtype=torch.bfloat16)
        ops.layernorm(x[:, :d] if dp != d else x, w["n1"][0], w["n1"][1], 1e-6, out=xn[:, :d] if dp != d else xn)
        Ho, Wo = (H // 2, W // 2) if qs else (H, W)
        if w["proj"] is not None:                       # dim change: shortcut = pool(proj(norm1(x)))
            sc = ops.gemm(xn, w["proj"][0], bias=w["proj"][1], out_dtype=torch.float32)
            x = ops.maxpool2x2(sc, B, H, W, dop) if qs else sc
        # window partition
        if ws > 0 and order == ws:
            Bw, wh, xw = N // (ws * ws), ws, xn
        elif ws > 0:
            idx, (Hp, Wp) = self._windows(B, H, W, ws, dev)
            Bw, wh = idx.numel() // (ws * ws), ws
            xw = torch.empty((idx.numel(), dp), device=dev, dtype=torch.bfloat16)
            ops.gather_rows(xn, idx, xw, None)
        else:
            Bw, wh, xw = B, None, xn
        Sk = ws * ws if ws > 0 else H * W
        gh, gw = (ws, ws) if ws > 0 else (H, W)           # token grid of one attention problem
        qkv = ops.gemm(xw, w["wqkv"], bias=w["bqkv"])     # [Bw*Sk, 3*dop]  (q | k | v, each padded to dop)
        q, Sq, ldq = qkv, Sk, 3 * dop
        if qs:
            q = ops.maxpool2x2(qkv, Bw, gh, gw, do, out=self._zbuf("q", (Bw * (gh // 2) * (gw // 2), dop), dev))
            Sq, ldq = (gh // 2) * (gw // 2), dop
        hd = do // heads
        o = self._zbuf("o", (Bw * Sq, dop), dev) if dop != do else None
        o = ops.attention(q, qkv[:, dop:], qkv[:, 2 * dop:], Bw, heads, heads, Sq, Sk, hd, (Sq * ldq, ldq), (Sk * 3 * dop, 3 * dop),
                          (Sk * 3 * dop, 3 * dop), out=o)
        y = ops.gemm(o, w["wo"], bias=w["bo"])            # [Bw*Sq, dop] in window order
        # window un-partition + residual
        if ws > 0 and order != ws:
            ws2 = ws // 2 if qs else ws
            idx2, _ = self._windows(B, Ho, Wo, ws2, dev)
            ops.add_rows(y, x, idx2)
        else:
            ops.add_rows(y, x, None)
        # MLP
        h2 = self._zbuf("h2", (x.shape[0], dop), dev) if dop != do else torch.empty((x.shape[0], dop), device=dev, dtype=torch.bfloat16)
        ops.layernorm(x[:, :do] if dop != do else x, w["n2"][0], w["n2"][1], 1e-6, out=h2[:, :do] if dop != do else h2)
        f = ops.gemm(h2, w["w1"], bias=w["b1"], act="gelu")
        ops.gemm(f, w["w2"], bias=w["b2"], resid=x, out=x)
        return x, Ho, Wo, order

    def forward_tokens(self, img):
        """img [B,3,H,W] -> list of (fp32 tokens [B*h*w, C_pad], h, w, C) at the stage ends, highest resolution first."""
        pk = self.packed()
        B = img.shape[0]
        cols, (H, W) = ops.im2col(img.contiguous(), 7, 4, 3, pk["Kp"])
        x = ops.gemm(cols, pk["patch_w"], bias=pk["patch_b"], resid=self._get_pos_embed((H, W)), resid_rows=H * W, out_dtype=torch.float32)
        outs = []
        order = 0
        for i, (blk, w) in enumerate(zip(self.schedule, pk["blocks"])):
            x, H, W, order = self._block(x, blk, w, B, H, W, order)
            if (i == self.stage_ends[-1]) or (i in self.stage_ends and self.return_interm_layers):
                x, order = self._reorder(x, B, H, W, order, 0), 0
                outs.append((x.clone() if i != self.stage_ends[-1] else x, H, W, blk["dim_out"]))
        return outs

    @torch.no_grad()
    def forward(self, x):
        """-> list of NCHW fp32 feature maps like the reference (sam2.py:1243-1258)."""
        B = x.shape[0]
        return [t[:, :C].reshape(B, h, w, C).permute(0, 3, 1, 2) for t, h, w, C in self.forward_tokens(x)]


class PositionEmbeddingSine(nn.Module):
    """Input-independent sine position encoding (normalize=True, scale=2*pi), cached per shape (sam2.py:1736-1830)."""

    def __init__(self, num_pos_feats, temperature=10000, normalize=True, scale=None):
        super().__init__()
        assert num_pos_feats % 2 == 0 and normalize
        self.num_pos_feats = num_pos_feats // 2
        self.temperature = temperature
        self.scale = 2 * math.pi if scale is None else scale
        self.cache = {}

    @torch.no_grad()
    def forward(self, x):
        key = (x.shape[-2], x.shape[-1], str(x.device))
        if key not in self.cache:
            H, W = x.shape[-2:]
            y = torch.arange(1, H + 1, dtype=torch.float32, device=x.device).view(-1, 1).repeat(1, W)
            xx = torch.arange(1, W + 1, dtype=torch.float32, device=x.device).view(1, -1).repeat(H, 1)
            eps = 1e-6
            y = y / (y[-1:, :] + eps) * self.scale
            xx = xx / (xx[:, -1:] + eps) * self.scale
            dim_t = torch.arange(self.num_pos_feats, dtype=torch.float32, device=x.device)
            dim_t = self.temperature ** (2 * (dim_t // 2) / self.num_pos_feats)
            px, py = xx[:, :, None] / dim_t, y[:, :, None] / dim_t
            px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
            py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
            self.cache[key] = torch.cat((py, px), dim=2).permute(2, 0, 1)
        return self.cache[key][None].repeat(x.shape[0], 1, 1, 1)


class FpnNeck(PackedModule):
    """Lateral 1x1 convs + nearest top-down sum on the listed levels (sam2.py:815-903)."""

    def __init__(self, position_encoding, d_model, backbone_channel_list, kernel_size=1, stride=1, padding=0,
                 fpn_interp_model="bilinear", fuse_type="sum", fpn_top_down_levels=None, device=None, dtype=torch.bfloat16, seed=21, std=0.02):
        super().__init__()
        if kernel_size != 1 or fuse_type != "sum" or fpn_interp_model != "nearest":
            raise NotImplementedError("only the SAM2 configuration (1x1 lateral convs, nearest top-down, sum) is accelerated")
        self.position_encoding = position_encoding
        self.backbone_channel_list = list(backbone_channel_list)
        self.d_model = d_model
        gen = torch.Generator(device=device if device is not None else "cpu").manual_seed(seed)
        for i, c in enumerate(self.backbone_channel_list):
            self.put(f"convs.{i}.conv.weight", init_tensor((d_model, c, 1, 1), "w", gen, std, device, dtype))
            self.put(f"convs.{i}.conv.bias", init_tensor((d_model,), "zero", gen, std, device, dtype))
        self.fpn_top_down_levels = list(range(len(self.backbone_channel_list))) if fpn_top_down_levels is None else list(fpn_top_down_levels)

    def _pack(self):
        out = []
        for i, c in enumerate(self.backbone_channel_list):
            cv = self.convs.get(str(i)).conv
            out.append((self.gw(_pad2(cv.weight.reshape(self.d_model, c), self.d_model, _pad_dim(c, self._w192()))), f32(cv.bias)))
        return out

    def _w192(self):
        return getattr(self, "gemm_dtype", "bf16") != "fp8"

    def forward_tokens(self, feats, B):
        """feats: [(tokens fp32 [B*h*w, Cpad], h, w, C)] highest resolution first -> [(tokens fp32 [B*h*w, d_model], h, w)]"""
        pk = self.packed()
        n = len(feats) - 1
        out, prev = [None] * len(feats), None
        for i in range(n, -1, -1):
            t, h, w, C = feats[i]
            lat = ops.gemm(ops.convert(t, torch.bfloat16), pk[n - i][0], bias=pk[n - i][1], out_dtype=torch.float32)
            if i in self.fpn_top_down_levels and prev is not None:
                ops.upsample2x_add(lat, prev[0], B, h, w, self.d_model)
            prev = (lat, h, w)
            out[i] = prev
        return out

    @torch.no_grad()
    def forward(self, xs):
        B = xs[0].shape[0]
        feats = [(x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous().float(), x.shape[2], x.shape[3], x.shape[1]) for x in xs]
        feats = [(self._pad_tokens(t, C, self._w192()), h, w, C) for t, h, w, C in feats]
        outs = self.forward_tokens(feats, B)
        maps = [t.view(B, h, w, self.d_model).permute(0, 3, 1, 2) for t, h, w in outs]
        return maps, [self.position_encoding(m).to(m.dtype) for m in maps]

    @staticmethod
    def _pad_tokens(t, C, w192=False):
        Cp = _pad_dim(C, w192)
        if Cp == C:
            return t
        out = torch.zeros((t.shape[0], Cp), device=t.device, dtype=t.dtype)
        out[:, :C] = t
        return out


class ImageEncoder(nn.Module):
    """trunk + neck, dropping the `scalp` lowest-resolution levels (sam2.py:784-812)."""

    def __init__(self, trunk, neck, scalp=0):
        super().__init__()
        self.trunk, self.neck, self.scalp = trunk, neck, scalp
        assert self.trunk.channel_list == self.neck.backbone_channel_list

    @torch.no_grad()
    def forward(self, sample):
        B = sample.shape[0]
        outs = self.neck.forward_tokens(self.trunk.forward_tokens(sample), B)
        d = self.neck.d_model
        feats = [t.view(B, h, w, d).permute(0, 3, 1, 2) for t, h, w in outs]
        pos = [self.neck.position_encoding(m).to(m.dtype) for m in feats]
        if self.scalp > 0:
            feats, pos = feats[: -self.scalp], pos[: -self.scalp]
        return {"vision_features": feats[-1], "vision_pos_enc": pos, "backbone_fpn": feats}


# ======================================================================================================================
# SAM heads with the language token as the sparse prompt
# ======================================================================================================================
class PromptEncoder(Holder):
    """Parameters of sam2.py `PromptEncoder` (:1565-1735).  The reference's inference path feeds it one dummy point with
    label -1 (+ its padding point) and no mask, so only `not_a_point_embed`, `no_mask_embed` and the random-Fourier dense
    position encoding are ever used; the point / mask-downscaling parameters are kept so checkpoints load by name."""

    def __init__(self, embed_dim=256, mask_in_chans=16, device=None, dtype=torch.bfloat16, seed=22, std=0.02):
        super().__init__()
        self.embed_dim = embed_dim
        gen = torch.Generator(device=device if device is not None else "cpu").manual_seed(seed)
        mk = lambda shape, kind="w", sd=std: init_tensor(shape, kind, gen, sd, device, dtype)
        self.put("pe_layer.positional_encoding_gaussian_matrix", init_tensor((2, embed_dim // 2), "w", gen, 1.0, device, torch.float32))
        for i in range(4):
            self.put(f"point_embeddings.{i}.weight", mk((1, embed_dim)))
        self.put("not_a_point_embed.weight", mk((1, embed_dim)))
        self.put("no_mask_embed.weight", mk((1, embed_dim)))
        c4 = mask_in_chans // 4
        self.put("mask_downscaling.0.weight", mk((c4, 1, 2, 2))); self.put("mask_downscaling.0.bias", mk((c4,), "zero"))
        self.put("mask_downscaling.1.weight", mk((c4,), "one")); self.put("mask_downscaling.1.bias", mk((c4,), "zero"))
        self.put("mask_downscaling.3.weight", mk((mask_in_chans, c4, 2, 2))); self.put("mask_downscaling.3.bias", mk((mask_in_chans,), "zero"))
        self.put("mask_downscaling.4.weight", mk((mask_in_chans,), "one")); self.put("mask_downscaling.4.bias", mk((mask_in_chans,), "zero"))
        self.put("mask_downscaling.6.weight", mk((embed_dim, mask_in_chans, 1, 1))); self.put("mask_downscaling.6.bias", mk((embed_dim,), "zero"))
        self._pe_cache = {}

    def _apply(self, fn, *a, **k):
        self._pe_cache = {}
        return super()._apply(fn, *a, **k)

    @torch.no_grad()
    def dense_pe_tokens(self, h, w):
        """PositionEmbeddingRandom over the (h, w) grid (:1869-1879), token-major fp32 [h*w, C]; input independent."""
        if (h, w) not in self._pe_cache:
            g = self.pe_layer.positional_encoding_gaussian_matrix.detach().float()
            y = (torch.arange(h, device=g.device, dtype=torch.float32) + 0.5) / h
            x = (torch.arange(w, device=g.device, dtype=torch.float32) + 0.5) / w
            c = torch.stack([x.view(1, w).expand(h, w), y.view(h, 1).expand(h, w)], dim=-1)
            c = (2 * math.pi) * ((2 * c - 1) @ g)
            self._pe_cache[(h, w)] = torch.cat([torch.sin(c), torch.cos(c)], dim=-1).reshape(h * w, -1).contiguous()
        return self._pe_cache[(h, w)]

    def get_dense_pe(self, size):
        h, w = size
        return self.dense_pe_tokens(h, w).view(h, w, -1).permute(2, 0, 1)[None]


class MaskDecoder(PackedModule):
    """sam2.py `MaskDecoder` (:1940-2224) with its `TwoWayTransformer` (depth 2, 8 heads, mlp 2048, attention
    downsample 2), 4 mask tokens, IoU head with sigmoid, object-score MLP and the high-res-feature convs."""

    def __init__(self, transformer_dim=256, depth=2, num_heads=8, mlp_dim=2048, num_multimask_outputs=3, iou_head_hidden_dim=256,
                 device=None, dtype=torch.bfloat16, seed=23, std=0.02):
        super().__init__()
        C = self.transformer_dim = transformer_dim
        self.depth, self.num_heads, self.mlp_dim = depth, num_heads, mlp_dim
        self.num_mask_tokens = num_multimask_outputs + 1
        gen = torch.Generator(device=device if device is not None else "cpu").manual_seed(seed)
        mk = lambda shape, kind="w": init_tensor(shape, kind, gen, std, device, dtype)

        def lin(p, o, i):
            self.put(p + "weight", mk((o, i))); self.put(p + "bias", mk((o,), "zero"))

        def norm(p, n):
            self.put(p + "weight", mk((n,), "one")); self.put(p + "bias", mk((n,), "zero"))

        def attn(p, internal):
            for nme in ("q_proj.", "k_proj.", "v_proj."):
                lin(p + nme, internal, C)
            lin(p + "out_proj.", C, internal)

        for i in range(depth):
            lp = f"transformer.layers.{i}."
            attn(lp + "self_attn.", C); attn(lp + "cross_attn_token_to_image.", C // 2); attn(lp + "cross_attn_image_to_token.", C // 2)
            for n in ("norm1.", "norm2.", "norm3.", "norm4."):
                norm(lp + n, C)
            lin(lp + "mlp.layers.0.", mlp_dim, C); lin(lp + "mlp.layers.1.", C, mlp_dim)
        attn("transformer.final_attn_token_to_image.", C // 2); norm("transformer.norm_final_attn.", C)
        self.put("iou_token.weight", mk((1, C))); self.put("mask_tokens.weight", mk((self.num_mask_tokens, C)))
        self.put("obj_score_token.weight", mk((1, C)))
        self.put("output_upscaling.0.weight", mk((C, C // 4, 2, 2))); self.put("output_upscaling.0.bias", mk((C // 4,), "zero"))
        norm("output_upscaling.1.", C // 4)
        self.put("output_upscaling.3.weight", mk((C // 4, C // 8, 2, 2))); self.put("output_upscaling.3.bias", mk((C // 8,), "zero"))
        self.put("conv_s0.weight", mk((C // 8, C, 1, 1))); self.put("conv_s0.bias", mk((C // 8,), "zero"))
        self.put("conv_s1.weight", mk((C // 4, C, 1, 1))); self.put("conv_s1.bias", mk((C // 4,), "zero"))
        for i in range(self.num_mask_tokens):
            hp = f"output_hypernetworks_mlps.{i}."
            lin(hp + "layers.0.", C, C); lin(hp + "layers.1.", C, C); lin(hp + "layers.2.", C // 8, C)
        lin("iou_prediction_head.layers.0.", iou_head_hidden_dim, C); lin("iou_prediction_head.layers.1.", iou_head_hidden_dim, iou_head_hidden_dim)
        lin("iou_prediction_head.layers.2.", self.num_mask_tokens, iou_head_hidden_dim)
        lin("pred_obj_score_head.layers.0.", C, C); lin("pred_obj_score_head.layers.1.", C, C); lin("pred_obj_score_head.layers.2.", 1, C)

    def _pack(self):
        C = self.transformer_dim
        g = self.get
        wb = lambda name: (bf(g(name).weight), f32(g(name).bias))
        cat_w = lambda *names: bf(torch.cat([g(n).weight for n in names], 0))
        cat_b = lambda *names: f32(torch.cat([g(n).bias for n in names], 0))

        def cross(p):     # token->image attention: q on tokens, (k | v) on the image tokens in one GEMM
            return dict(q=wb(p + "q_proj"), wkv=cat_w(p + "k_proj", p + "v_proj"), wk32=f32(g(p + "k_proj").weight),
                        bkv=cat_b(p + "k_proj", p + "v_proj"), o=wb(p + "out_proj"))

        layers = []
        for i in range(self.depth):
            lp = f"transformer.layers.{i}."
            sa, i2t = lp + "self_attn.", lp + "cross_attn_image_to_token."
            layers.append(dict(
                sa_qk=(cat_w(sa + "q_proj", sa + "k_proj"), cat_b(sa + "q_proj", sa + "k_proj")), sa_v=wb(sa + "v_proj"), sa_o=wb(sa + "out_proj"),
                t2i=cross(lp + "cross_attn_token_to_image."),
                i2t=dict(q=wb(i2t + "q_proj"), wq32=f32(g(i2t + "q_proj").weight), k=wb(i2t + "k_proj"), v=wb(i2t + "v_proj"), o=wb(i2t + "out_proj")),
                n=[(f32(g(lp + f"norm{j}").weight), f32(g(lp + f"norm{j}").bias)) for j in (1, 2, 3, 4)],
                m1=wb(lp + "mlp.layers.0"), m2=wb(lp + "mlp.layers.1")))
        pk = dict(layers=layers, final=cross("transformer.final_attn_token_to_image."),
                  nf=(f32(g("transformer.norm_final_attn").weight), f32(g("transformer.norm_final_attn").bias)))
        pk["out_tok"] = f32(torch.cat([self.obj_score_token.weight, self.iou_token.weight, self.mask_tokens.weight], 0))
        # 1x1 convs on the high-res FPN levels; N padded to one 128-wide tile
        pk["s0"] = (bf(_pad2(self.conv_s0.weight.reshape(C // 8, C), 128, C)), f32(_pad1(self.conv_s0.bias, 128)))
        pk["s1"] = (bf(_pad2(self.conv_s1.weight.reshape(C // 4, C), 128, C)), f32(_pad1(self.conv_s1.bias, 128)))
        # ConvTranspose2d(k=2,s=2) as a GEMM: output column (dy*2+dx)*Cout + co
        up = self.output_upscaling
        w0, w3 = up.get("0").weight, up.get("3").weight
        pk["dc1"] = (bf(w0.permute(2, 3, 1, 0).reshape(4 * (C // 4), C)), f32(up.get("0").bias.repeat(4)))
        pk["dc2"] = (bf(w3.permute(2, 3, 1, 0).reshape(4 * (C // 8), C // 4)), f32(up.get("3").bias.repeat(4)))
        pk["ln_up"] = (f32(up.get("1").weight), f32(up.get("1").bias))
        pk["hyper"] = [[wb(f"output_hypernetworks_mlps.{i}.layers.{j}") for j in range(3)] for i in range(self.num_mask_tokens)]
        pk["iou"] = [wb(f"iou_prediction_head.layers.{j}") for j in range(3)]
        pk["obj"] = [wb(f"pred_obj_score_head.layers.{j}") for j in range(3)]
        pk["tables"] = {}
        return pk

    def pos_tables(self, pe_tokens):
        """k_proj(keys + pos) = k_proj(keys) + (pos @ Wk^T + bk): the position term is input independent, so it is a
        per-grid fp32 table added in the GEMM epilogue (row-modulo residual) instead of a pass over the image tokens."""
        pk = self.packed()
        key = (pe_tokens.data_ptr(), pe_tokens.shape[0])
        if key not in pk["tables"]:
            C = self.transformer_dim
            pe = pe_tokens.float()
            t = {}
            for name, blk in [(f"t2i{i}", L["t2i"]) for i, L in enumerate(pk["layers"])] + [("final", pk["final"])]:
                tab = blk["bkv"][None].repeat(pe.shape[0], 1)
                tab[:, : C // 2] += pe @ blk["wk32"].t()
                t[name] = tab.contiguous()
            for i, L in enumerate(pk["layers"]):
                t[f"i2t{i}"] = (pe @ L["i2t"]["wq32"].t() + L["i2t"]["q"][1][None]).contiguous()
            pk["tables"][key] = t
        return pk["tables"][key]


def _pixel_shuffle_index(B, h, w, device):
    """destination row in [B, 2h, 2w] of source row (b, y, x, dy, dx) of a k2/s2 transposed conv run as a GEMM"""
    b = torch.arange(B).view(B, 1, 1, 1, 1)
    y = torch.arange(h).view(1, h, 1, 1, 1)
    x = torch.arange(w).view(1, 1, w, 1, 1)
    dy = torch.arange(2).view(1, 1, 1, 2, 1)
    dx = torch.arange(2).view(1, 1, 1, 1, 2)
    return ((b * 2 * h + 2 * y + dy) * 2 * w + 2 * x + dx).reshape(-1).to(device)


class SAM2Base(nn.Module):
    """The slice of sam2.py `SAM2Base` / `SAM2VideoPredictor` that the reference's inference reaches
    (`language_embd_inference` :378-406): every frame receives the language token as an *initial conditioning* frame, so
    `track_step` (:3174-3275) adds `no_mem_embed`, runs the SAM heads and `propagate_in_video` (:4071-4153) returns those
    stored masks resized to the frame size.  The memory encoder / memory attention only feed non-conditioning frames,
    which this call pattern never has, so they are not built (their checkpoint keys are ignored on load)."""

    def __init__(self, image_encoder, image_size=1024, device=None, dtype=torch.bfloat16, seed=22):
        super().__init__()
        self.image_encoder = image_encoder
        self.image_size = image_size
        self.hidden_dim = image_encoder.neck.d_model
        self.sam_prompt_encoder = PromptEncoder(self.hidden_dim, device=device, dtype=dtype, seed=seed)
        self.sam_mask_decoder = MaskDecoder(self.hidden_dim, device=device, dtype=dtype, seed=seed + 1)
        gen = torch.Generator(device=device if device is not None else "cpu").manual_seed(seed + 2)
        self.no_mem_embed = nn.Parameter(init_tensor((1, 1, self.hidden_dim), "w", gen, 0.02, device, dtype), requires_grad=False)
        self._ps_cache = {}

    # ---- forward_image (:2804-2816) in token-major form -------------------------------------------------------------------
    @torch.no_grad()
    def forward_image_tokens(self, images):
        """images [F,3,S,S] -> [(fp32 tokens [F*h*w, 256], h, w)] for the 3 kept FPN levels, highest resolution first."""
        enc = self.image_encoder
        B = images.shape[0]
        outs = enc.neck.forward_tokens(enc.trunk.forward_tokens(images), B)
        return outs[: len(outs) - enc.scalp] if enc.scalp > 0 else outs

    def _ps_index(self, B, h, w, device):
        key = (B, h, w, str(device))
        if key not in self._ps_cache:
            self._ps_cache[key] = _pixel_shuffle_index(B, h, w, device)
        return self._ps_cache[key]

    @staticmethod
    def _attn(q, k, v, B, H, Sq, Sk, hd):
        return ops.attention(q, k, v, B, H, H, Sq, Sk, hd, (Sq * q.stride(0), q.stride(0)), (Sk * k.stride(0), k.stride(0)),
                             (Sk * v.stride(0), v.stride(0)))

    # ---- _forward_sam_heads (:3276-3452) with point_inputs=None, mask_inputs=None, multimask_output=True ------------------
    @torch.no_grad()
    def forward_sam_heads_tokens(self, feats, B, language_embd, out_size=None):
        """feats from `forward_image_tokens`; language_embd [B, 1, C] (or [B, C]).  Returns dict with low_res_multimasks
        [B,3,4h,4w], ious [B,3], object_score_logits [B,1], best int32 [B], low_res_masks [B,1,4h,4w] and high_res_masks
        [B,1,*out_size] (default image_size), all fp32."""
        dec, pe = self.sam_mask_decoder, self.sam_prompt_encoder
        pk = dec.packed()
        (t0, h0, w0), (t1, h1, w1), (t2, h, w) = feats
        C, H, hw, dev = self.hidden_dim, dec.num_heads, feats[2][1] * feats[2][2], t2.device
        assert (h0, w0, h1, w1) == (4 * h, 4 * w, 2 * h, 2 * w) and t2.shape[0] == B * hw
        BF, F32 = torch.bfloat16, torch.float32
        tabs = dec.pos_tables(pe.dense_pe_tokens(h, w))
        # prompt tokens: [obj_score, iou, 4 x mask] output tokens + 2 x not_a_point + the language embedding
        nap = pe.not_a_point_embed.weight.detach().float()
        lang = language_embd.reshape(B, 1, C).to(device=dev, dtype=F32)
        tokens = torch.cat([pk["out_tok"][None].expand(B, -1, -1), nap[None].expand(B, 2, C), lang], dim=1).reshape(-1, C).contiguous()
        T = tokens.shape[0] // B
        # image tokens + no_mem_embed (:2980-2984) + dense no-mask embedding (:1716-1720), both per-channel vectors
        vec = (self.no_mem_embed.detach().float().reshape(1, C) + pe.no_mask_embed.weight.detach().float().reshape(1, C)).contiguous()
        keys = ops.add_bcast(t2, vec, out_dtype=F32)
        keys_bf = ops.convert(keys, BF)
        queries = tokens
        for i, L in enumerate(pk["layers"]):
            # (1) self attention of the prompt tokens (layer 0: no position term, output replaces the queries)
            q_in = ops.add_bcast(queries, tokens if i > 0 else None, out_dtype=BF)
            v_in = ops.convert(queries, BF) if i > 0 else q_in
            qk = ops.gemm(q_in, L["sa_qk"][0], bias=L["sa_qk"][1])
            vv = ops.gemm(v_in, L["sa_v"][0], bias=L["sa_v"][1])
            o = self._attn(qk, qk[:, C:], vv, B, H, T, T, C // H)
            queries = ops.gemm(o, L["sa_o"][0], bias=L["sa_o"][1], resid=queries if i > 0 else None, out_dtype=F32)
            queries = ops.layernorm(queries, *L["n"][0], 1e-5, out_dtype=F32)
            # (2) tokens attend to the image
            queries = self._token_to_image(queries, tokens, keys_bf, L["t2i"], tabs[f"t2i{i}"], L["n"][1], B, T, hw)
            # (3) token MLP
            f = ops.gemm(ops.convert(queries, BF), L["m1"][0], bias=L["m1"][1], act="relu")
            queries = ops.layernorm(ops.gemm(f, L["m2"][0], bias=L["m2"][1], resid=queries, out_dtype=F32), *L["n"][2], 1e-5, out_dtype=F32)
            # (4) image attends to the tokens
            I = L["i2t"]
            qi = ops.gemm(keys_bf, I["q"][0], resid=tabs[f"i2t{i}"], resid_rows=hw)
            kt = ops.gemm(ops.add_bcast(queries, tokens, out_dtype=BF), I["k"][0], bias=I["k"][1])
            vt = ops.gemm(ops.convert(queries, BF), I["v"][0], bias=I["v"][1])
            o = self._attn(qi, kt, vt, B, H, hw, T, (C // 2) // H)
            keys = ops.layernorm(ops.gemm(o, I["o"][0], bias=I["o"][1], resid=keys, out_dtype=F32), *L["n"][3], 1e-5, out_dtype=F32)
            keys_bf = ops.convert(keys, BF)
        hs = self._token_to_image(queries, tokens, keys_bf, pk["final"], tabs["final"], pk["nf"], B, T, hw)
        hs_bf = ops.convert(hs, BF).view(B, T * C)
        tok = lambda j: hs_bf[:, j * C:(j + 1) * C]
        # upscaling with the high-res skips (:2137-2149)
        s0 = ops.gemm(ops.convert(t0, BF), pk["s0"][0], bias=pk["s0"][1])                                # bf16 [B*16hw, 128], 32 used
        s1 = ops.gemm(ops.convert(t1, BF), pk["s1"][0], bias=pk["s1"][1], out_dtype=F32)                 # f32 [B*4hw, 128], 64 used
        up1 = ops.gemm(keys_bf, pk["dc1"][0], bias=pk["dc1"][1])                                         # [B*hw, 4*64]
        ops.add_rows(up1.view(-1, C // 4), s1, self._ps_index(B, h, w, dev), D=C // 4)
        u = ops.layernorm(s1[:, : C // 4], *pk["ln_up"], 1e-6, act="gelu")                               # LayerNorm2d + GELU
        up2 = ops.gemm(u, pk["dc2"][0], bias=pk["dc2"][1])                                               # [B*4hw, 4*32]
        nm = dec.num_mask_tokens
        hyper = torch.empty((B, nm, C // 8), device=dev, dtype=F32)
        for j in range(nm):
            x = tok(2 + j)
            for li, (wj, bj) in enumerate(pk["hyper"][j]):
                x = ops.gemm(x, wj, bias=bj, act="relu") if li < 2 else ops.gemm(x, wj, bias=bj, out=hyper[:, j])
        masks = ops.sam_mask_head(up2, s0, hyper, B, h1, w1, C // 8)         # [B, 4, 4h, 4w]
        x = tok(1)
        x = ops.gemm(ops.gemm(x, pk["iou"][0][0], bias=pk["iou"][0][1], act="relu"), pk["iou"][1][0], bias=pk["iou"][1][1], act="relu")
        ious = ops.gemm(x, pk["iou"][2][0], bias=pk["iou"][2][1], act="sigmoid", out_dtype=F32)          # [B, 4]
        x = tok(0)
        x = ops.gemm(ops.gemm(x, pk["obj"][0][0], bias=pk["obj"][0][1], act="relu"), pk["obj"][1][0], bias=pk["obj"][1][1], act="relu")
        obj = ops.gemm(x, pk["obj"][2][0], bias=pk["obj"][2][1], out_dtype=F32)
        best = ops.argmax_rows(ious[:, 1:])                                                             # multimask_output: drop token 0
        S = (self.image_size, self.image_size) if out_size is None else tuple(out_size)
        return dict(low_res_multimasks=masks[:, 1:], ious=ious[:, 1:], object_score_logits=obj, best=best,
                    low_res_masks=ops.resize_bilinear(masks, (h0, w0), sel=best, sel_off=1),
                    high_res_masks=ops.resize_bilinear(masks, S, sel=best, sel_off=1))

    def _token_to_image(self, queries, tokens, keys_bf, A, table, norm, B, T, hw):
        C, H = self.hidden_dim, self.sam_mask_decoder.num_heads
        q = ops.gemm(ops.add_bcast(queries, tokens, out_dtype=torch.bfloat16), A["q"][0], bias=A["q"][1])
        kv = ops.gemm(keys_bf, A["wkv"], resid=table, resid_rows=hw)
        o = self._attn(q, kv, kv[:, C // 2:], B, H, T, hw, (C // 2) // H)
        return ops.layernorm(ops.gemm(o, A["o"][0], bias=A["o"][1], resid=queries, out_dtype=torch.float32), norm[0], norm[1], 1e-5,
                             out_dtype=torch.float32)

    # ---- SAM2VideoPredictor surface used by the wrapper -----------------------------------------------------------------------
    def init_state(self, images):
        """sam2.py:3792-3843 (only the fields this path reads)."""
        return {"images": images, "num_frames": len(images), "video_height": self.image_size, "video_width": self.image_size,
                "cached_features": {}}


class SAM2(nn.Module):
    """sam2.py `SAM2` wrapper (:86-460): `get_sam2_embeddings(images)` -> state, `language_embd_inference(state, embeds)`
    -> mask logits [F * n_obj, 1, S, S] (frame-major, what concatenating `propagate_in_video`'s per-frame
    [n_obj, 1, S, S] outputs gives, :398-406).  Frames are encoded in batches of `frame_batch` and their FPN tokens are kept in the
    state, so several [SEG] embeddings share one encoder pass (the reference re-encodes every frame per embedding)."""

    def __init__(self, ckpt_path=None, device=None, dtype=torch.bfloat16, image_encoder=None, image_size=1024, frame_batch=8, seed=20):
        super().__init__()
        if ckpt_path is not None:
            raise NotImplementedError("load SAM2 weights through VideoReferQwen2ForCausalLM.from_pretrained / load_state_dict")
        enc = image_encoder if image_encoder is not None else build_sam2_image_encoder(device=device, dtype=dtype, seed=seed)
        self.sam2_model = SAM2Base(enc, image_size=image_size, device=device, dtype=dtype, seed=seed + 2)
        self.hidden_dim = self.sam2_model.hidden_dim
        self.img_mean = (0.485, 0.456, 0.406)
        self.img_std = (0.229, 0.224, 0.225)
        self.frame_batch = frame_batch

    def preprocess_image(self, image, dtype=torch.bfloat16):
        image = image / 255.0
        mean = torch.tensor(self.img_mean, dtype=dtype, device=image.device)[:, None, None]
        std = torch.tensor(self.img_std, dtype=dtype, device=image.device)[:, None, None]
        return (image - mean) / std

    def get_sam2_embeddings(self, images):
        return self.sam2_model.init_state(images)

    get_sam2_embeddings_inference = get_sam2_embeddings

    # ---- training-path entry points (ref :343-377, :408-447); forward values only ----------------------------------------------
    def get_sam2_embeddings_train(self, images, expand_size=1, obj_num_list=None, BS=1, T=None):
        """images [F,3,S,S] of ONE sample -> state; the per-object feature expansion of the reference (each frame's features
        repeated once per object) is done lazily by `inject_language_embd_train`, which reuses the cached FPN tokens."""
        return self.sam2_model.init_state(images)

    @torch.no_grad()
    def inject_language_embd_train(self, sam_states, language_embd, nf_nobj=None):
        """language_embd [n_obj, C]: every object on every frame of the state, frame-major -> high-res mask logits
        [F * n_obj, 1, S, S] (the reference's `high_res_masks`, index 4 of `_forward_sam_heads`)."""
        F_ = sam_states["num_frames"]
        return self.language_embd_inference(sam_states, [language_embd] * F_)

    @torch.no_grad()
    def _features(self, state, lo, hi):
        key = (lo, hi)
        if key not in state["cached_features"]:
            imgs = state["images"][lo:hi]
            imgs = imgs if torch.is_tensor(imgs) else torch.stack(list(imgs))
            state["cached_features"][key] = self.sam2_model.forward_image_tokens(imgs.to(torch.bfloat16).contiguous())
        return state["cached_features"][key]

    @torch.no_grad()
    def language_embd_inference(self, inference_state, language_embd):
        """language_embd: per frame, per object, a [C] embedding (list of lists / list of [n_obj, C] tensors)."""
        F_, n_obj = len(language_embd), len(language_embd[0])
        assert F_ == inference_state["num_frames"]
        S = (inference_state["video_height"], inference_state["video_width"])
        out = []
        for lo in range(0, F_, self.frame_batch):
            hi = min(F_, lo + self.frame_batch)
            feats = self._features(inference_state, lo, hi)
            per_obj = []
            for oi in range(n_obj):
                emb = torch.stack([torch.as_tensor(language_embd[f][oi]).reshape(-1) for f in range(lo, hi)])
                per_obj.append(self.sam2_model.forward_sam_heads_tokens(feats, hi - lo, emb, out_size=S)["high_res_masks"])
            out.append(torch.cat(per_obj, dim=1).reshape(-1, 1, *S))
        return torch.cat(out, dim=0)


def build_sam2_image_encoder(device=None, dtype=torch.bfloat16, seed=20):
    """Hiera-L + FPN as `SAM2.build_image_encoder` configures them (sam2.py:147-195)."""
    trunk = Hiera(embed_dim=144, num_heads=2, stages=[2, 6, 36, 4], global_att_blocks=[23, 33, 43],
                  window_pos_embed_bkg_spatial_size=[7, 7], window_spec=[8, 4, 16, 8], device=device, dtype=dtype, seed=seed)
    neck = FpnNeck(PositionEmbeddingSine(num_pos_feats=256), d_model=256, backbone_channel_list=[1152, 576, 288, 144],
                   fpn_top_down_levels=[2, 3], fpn_interp_model="nearest", device=device, dtype=dtype, seed=seed + 1)
    return ImageEncoder(trunk=trunk, neck=neck, scalp=1)
