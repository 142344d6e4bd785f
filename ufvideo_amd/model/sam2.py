"""SAM2 image encoder — Hiera trunk + FPN neck — behind the reference's module names
(ufvideo/model/sam2.py: `Hiera` :1134-1258, `MultiScaleBlock` :1049-1131, `MultiScaleAttention` :1000-1046,
`PatchEmbed` :954-984, `FpnNeck` :815-903, `ImageEncoder` :784-812, `PositionEmbeddingSine` :1736-1830).

This is the heavy part of the segmentation head (~1.8 TFLOP per 1024x1024 frame).  The prompt
encoder / mask decoder / video-predictor bookkeeping are not built yet.

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


def _pad_dim(d):
    return d if (d < 128 or d % 128 == 0) else round_up(d, 128)


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

    # ---- packing -----------------------------------------------------------------------------------------------
    def _pack(self):
        E = self.schedule[0]["dim"]
        Ep = _pad_dim(E)
        K = 3 * 49
        pk = {"Kp": round_up(K, 64), "Ep": Ep}
        pk["patch_w"] = bf(_pad2(self.patch_embed.proj.weight.reshape(E, K), Ep, pk["Kp"]))
        pk["patch_b"] = f32(_pad1(self.patch_embed.proj.bias, Ep))
        blocks = []
        for i, b in enumerate(self.schedule):
            L = self.blocks.get(str(i))
            d, do = b["dim"], b["dim_out"]
            dp, dop, hp = _pad_dim(d), _pad_dim(do), _pad_dim(4 * do)
            wq = L.attn.qkv.weight
            bq = L.attn.qkv.bias
            wqkv = torch.cat([_pad2(wq[j * do:(j + 1) * do], dop, dp) for j in range(3)], 0)
            bqkv = torch.cat([_pad1(bq[j * do:(j + 1) * do], dop) for j in range(3)], 0)
            blk = dict(n1=(f32(L.norm1.weight), f32(L.norm1.bias)), n2=(f32(L.norm2.weight), f32(L.norm2.bias)),
                       wqkv=bf(wqkv), bqkv=f32(bqkv), wo=bf(_pad2(L.attn.proj.weight, dop, dop)), bo=f32(_pad1(L.attn.proj.bias, dop)),
                       w1=bf(_pad2(L.mlp.layers.get("0").weight, hp, dop)), b1=f32(_pad1(L.mlp.layers.get("0").bias, hp)),
                       w2=bf(_pad2(L.mlp.layers.get("1").weight, dop, hp)), b2=f32(_pad1(L.mlp.layers.get("1").bias, dop)),
                       dp=dp, dop=dop, hp=hp, proj=None)
            if d != do:
                blk["proj"] = (bf(_pad2(L.proj.weight, dop, dp)), f32(_pad1(L.proj.bias, dop)))
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

    # ---- one MultiScaleBlock (sam2.py:1099-1131) on the fp32 stream x [B*H*W, dp] ----------------------------------------
    def _block(self, x, blk, w, B, H, W):
        d, do, heads, ws, qs = blk["dim"], blk["dim_out"], blk["heads"], blk["window"], blk["q_stride"]
        dp, dop = w["dp"], w["dop"]
        dev = x.device
        N = B * H * W
        xn = torch.zeros((N, dp), device=dev, dtype=torch.bfloat16) if dp != d else torch.empty((N, dp), device=dev, dtype=torch.bfloat16)
        ops.layernorm(x[:, :d] if dp != d else x, w["n1"][0], w["n1"][1], 1e-6, out=xn[:, :d] if dp != d else xn)
        Ho, Wo = (H // 2, W // 2) if qs else (H, W)
        if w["proj"] is not None:                       # dim change: shortcut = pool(proj(norm1(x)))
            sc = ops.gemm(xn, w["proj"][0], bias=w["proj"][1], out_dtype=torch.float32)
            x = ops.maxpool2x2(sc, B, H, W, dop) if qs else sc
        # window partition
        if ws > 0:
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
            q = ops.maxpool2x2(qkv, Bw, gh, gw, do, out=torch.zeros((Bw * (gh // 2) * (gw // 2), dop), device=dev, dtype=torch.bfloat16))
            Sq, ldq = (gh // 2) * (gw // 2), dop
        hd = do // heads
        o = torch.zeros((Bw * Sq, dop), device=dev, dtype=torch.bfloat16) if dop != do else None
        o = ops.attention(q, qkv[:, dop:], qkv[:, 2 * dop:], Bw, heads, heads, Sq, Sk, hd, (Sq * ldq, ldq), (Sk * 3 * dop, 3 * dop),
                          (Sk * 3 * dop, 3 * dop), out=o)
        y = ops.gemm(o, w["wo"], bias=w["bo"])            # [Bw*Sq, dop] in window order
        # window un-partition + residual
        if ws > 0:
            ws2 = ws // 2 if qs else ws
            idx2, _ = self._windows(B, Ho, Wo, ws2, dev)
            ops.add_rows(y, x, idx2)
        else:
            ops.add_rows(y, x, None)
        # MLP
        h2 = torch.zeros((x.shape[0], dop), device=dev, dtype=torch.bfloat16) if dop != do else torch.empty((x.shape[0], dop), device=dev, dtype=torch.bfloat16)
        ops.layernorm(x[:, :do] if dop != do else x, w["n2"][0], w["n2"][1], 1e-6, out=h2[:, :do] if dop != do else h2)
        f = ops.gemm(h2, w["w1"], bias=w["b1"], act="gelu")
        ops.gemm(f, w["w2"], bias=w["b2"], resid=x, out=x)
        return x, Ho, Wo

    def forward_tokens(self, img):
        """img [B,3,H,W] -> list of (fp32 tokens [B*h*w, C_pad], h, w, C) at the stage ends, highest resolution first."""
        pk = self.packed()
        B = img.shape[0]
        cols, (H, W) = ops.im2col(img.contiguous(), 7, 4, 3, pk["Kp"])
        x = ops.gemm(cols, pk["patch_w"], bias=pk["patch_b"], resid=self._get_pos_embed((H, W)), resid_rows=H * W, out_dtype=torch.float32)
        outs = []
        for i, (blk, w) in enumerate(zip(self.schedule, pk["blocks"])):
            x, H, W = self._block(x, blk, w, B, H, W)
            if (i == self.stage_ends[-1]) or (i in self.stage_ends and self.return_interm_layers):
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
            out.append((bf(_pad2(cv.weight.reshape(self.d_model, c), self.d_model, _pad_dim(c))), f32(cv.bias)))
        return out

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
        feats = [(self._pad_tokens(t, C), h, w, C) for t, h, w, C in feats]
        outs = self.forward_tokens(feats, B)
        maps = [t.view(B, h, w, self.d_model).permute(0, 3, 1, 2) for t, h, w in outs]
        return maps, [self.position_encoding(m).to(m.dtype) for m in maps]

    @staticmethod
    def _pad_tokens(t, C):
        Cp = _pad_dim(C)
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


def build_sam2_image_encoder(device=None, dtype=torch.bfloat16, seed=20):
    """Hiera-L + FPN as `SAM2.build_image_encoder` configures them (sam2.py:147-195)."""
    trunk = Hiera(embed_dim=144, num_heads=2, stages=[2, 6, 36, 4], global_att_blocks=[23, 33, 43],
                  window_pos_embed_bkg_spatial_size=[7, 7], window_spec=[8, 4, 16, 8], device=device, dtype=dtype, seed=seed)
    neck = FpnNeck(PositionEmbeddingSine(num_pos_feats=256), d_model=256, backbone_channel_list=[1152, 576, 288, 144],
                   fpn_top_down_levels=[2, 3], fpn_interp_model="nearest", device=device, dtype=dtype, seed=seed + 1)
    return ImageEncoder(trunk=trunk, neck=neck, scalp=1)
