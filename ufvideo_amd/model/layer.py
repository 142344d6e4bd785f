"""Region encoder behind the reference's interface (ufvideo/model/layer.py): bilinear-resized,
binarised object masks -> masked mean pooling of tower features -> greedy adjacent token_merge to
<= 4 tokens per object -> 2-layer GELU MLP.  Pooling and the MLP are HIP kernels; the merge is
data-dependent control flow over <= a few dozen tokens and stays on the host side of the ABI."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ._params import Holder, PackedModule, init_tensor, bf, f32


def token_merge(x, r):
    """x [1, n, d]: repeatedly average runs of adjacent tokens whose cosine similarity is among the
    r largest (ref layer.py:6-33).  Returns [1, n - (#merged), d]."""
    sim = torch.sum(F.normalize(x[:, :-1], p=2, dim=-1) * F.normalize(x[:, 1:], p=2, dim=-1), dim=-1)
    kth = torch.topk(sim.flatten(), r).values[-1]
    below = (sim[0] < kth).tolist()                 # one host sync for the whole run structure
    out, run = [], []
    for i, cut in enumerate(below):
        run.append(i)
        if cut:
            out.append(x[:, run].mean(dim=1, keepdim=True))
            run = []
    run.append(len(below))
    out.append(x[:, run].mean(dim=1, keepdim=True))
    return torch.cat(out, dim=1)


class MaskExtractor(PackedModule):
    def __init__(self, image_aspect_ratio, config, mask_shape=112, depth=2, region_token_num=4, device=None,
                 dtype=torch.bfloat16, seed=3, std=0.02):
        super().__init__()
        gen = torch.Generator(device=device if device is not None else "cpu").manual_seed(seed)
        self.mask_shape = mask_shape
        self.feat_linear = Holder()
        for i in range(depth):
            din = config.mm_hidden_size if i == 0 else config.hidden_size
            self.feat_linear.put(f"{2 * i}.weight", init_tensor((config.hidden_size, din), "w", gen, std, device, dtype))
            self.feat_linear.put(f"{2 * i}.bias", init_tensor((config.hidden_size,), "zero", gen, std, device, dtype))
        self.depth = depth
        self.image_aspect_ratio = image_aspect_ratio
        self.region_token_num = region_token_num

    def _pack(self):
        return [(bf(self.feat_linear.get(f"{2 * i}.weight")), f32(self.feat_linear.get(f"{2 * i}.bias"))) for i in range(self.depth)]

    def forward(self, feats, masks, X_features, ann_indices, frame_nums, stash=None):
        """feats [n_frames, P, C] tower features; masks: list of [q, H, W]; -> (tokens [sum, hidden] fp32, region_token_nums)"""
        dev = feats.device
        N = int(pow(feats.shape[1], 0.5))
        query_feats, region_token_nums = [], []
        for idx in range(len(masks)):
            mask = masks[idx].unsqueeze(0).float().to(dev)
            if len(mask[0]) == 0:
                mask = torch.zeros((1, 1, 336, 336), device=dev)
            if self.image_aspect_ratio == "pad":
                _h, w = mask.shape[-2:]
                m = max(_h, w)
                mask = F.pad(mask, ((m - w) // 2, (m - w) - (m - w) // 2, (m - _h) // 2, (m - _h) - (m - _h) // 2, 0, 0, 0, 0))
            ann_index = [i for index in ann_indices[idx] for i in index]
            if mask.shape[-2:] != (N, N):
                mask = ops.resize_bilinear(mask.float().contiguous(), (N, N))      # F.interpolate(bilinear, align_corners=False), ref :139
            mbin = (mask[0] > 0).float().reshape(mask.shape[1], N * N).contiguous()
            frame_of = torch.tensor(ann_index, dtype=torch.int32, device=dev)
            raw = ops.mask_pool(feats.contiguous(), mbin, frame_of)          # [q, C] fp32
            merged, start = [], 0
            for index in ann_indices[idx]:
                mf = raw[start:start + len(index)].unsqueeze(0)
                if mf.shape[1] > self.region_token_num:
                    mf = token_merge(mf, mf.shape[1] - self.region_token_num)
                region_token_nums.append(mf.shape[1])
                merged.append(mf)
                start += len(index)
            query_feats.append(torch.cat(merged, dim=1).reshape(-1, raw.shape[-1]))
        mf = ops.convert(torch.cat(query_feats, dim=0).contiguous(), torch.bfloat16)
        if stash is not None:                       # training (ufvideo_amd.train): keep the MLP input and its pre-activations
            stash["in"], stash["pre"] = [], []
        pk = self.packed()
        for i, (w, b) in enumerate(pk):
            last = i == len(pk) - 1
            if stash is None:
                mf = ops.gemm(mf, w, bias=b, act=None if last else "gelu", out_dtype=torch.float32 if last else torch.bfloat16)
            else:
                stash["in"].append(mf)
                if last:
                    mf = ops.gemm(mf, w, bias=b, out_dtype=torch.float32)
                else:
                    pre = ops.gemm(mf, w, bias=b)
                    stash["pre"].append(pre)
                    mf = ops.act_fwd(pre, "gelu")
        return mf, region_token_nums

    def backward(self, dout, stash):
        """dout fp32 [tokens, hidden] -> {feat_linear.N.weight / bias: fp32 gradient} (the pooled tower features are constants:
        the tower is frozen in the reference, encoder.py:122,134)"""
        from ..train_projector import _lin_bwd
        pk = self.packed()
        g = {}
        d = ops.convert(dout.contiguous(), torch.bfloat16)
        for i in range(len(pk) - 1, -1, -1):
            w, _ = pk[i]
            if i < len(pk) - 1:
                d = ops.act_bwd(stash["pre"][i], d, "gelu")
            g[f"feat_linear.{2 * i}.bias"] = ops.colsum(d, torch.zeros((w.shape[0],), device=d.device, dtype=torch.float32))
            d, dw = _lin_bwd(stash["in"][i], w, d, want_dx=i > 0)
            g[f"feat_linear.{2 * i}.weight"] = dw
        return g


def build_region_encoder(config, image_aspect_ratio, **kw):
    kind = getattr(config, "mm_region_encoder_type", "pooling")
    if kind == "pooling":
        return MaskExtractor(image_aspect_ratio, config, **kw)
    raise ValueError(f"Unknown region encoder type: {kind}")
