"""Multimodal projectors behind the reference's interface (ufvideo/model/projector.py).

Production type is `stc_connector_v35`: RegStage x4 -> Conv3d 2x2x2/s2/p0 + SiLU -> RegStage x4 ->
Linear-GELU-Linear.  Everything runs token-major (NHWC): the tower's [T, N, C] output IS the NHWC
image, so none of the reference's five einops permute copies (projector.py:199-213) exist here.
1x1 convs and the Conv3d are MFMA GEMMs; LayerNorm2d == row LayerNorm in this layout.
RegStage follows timm 1.0.15's RegNet Bottleneck (bottle_ratio 1, depthwise 3x3, SE 0.25, conv1x1
shortcut, SiLU, LayerNorm2d); timm is unavailable offline, so that part is parity-unpinned
 and the LayerNorm eps is a parameter.
"""
import os
import re

import torch
import torch.nn as nn

from .. import ops
from ._params import Holder, PackedModule, init_tensor, bf, f32


def load_mm_projector(model_path, cache_dir=None, token=None):
    f = os.path.join(model_path, "mm_projector.bin")
    if not os.path.exists(f):
        raise FileNotFoundError(f"{f} not found (remote hub download is not supported offline)")
    w = torch.load(f, map_location="cpu")
    return {k: v.to(torch.float16) for k, v in w.items()}


class IdentityMap(nn.Module):
    def forward(self, x, *args, **kwargs):
        return x

    @property
    def config(self):
        return {"mm_projector_type": "identity"}


class MlpProjector(PackedModule):
    """`linear` / `mlpNx_gelu`: nn.Sequential(Linear, [GELU, Linear]...) keys '0','2',... (projector.py:95-108)"""

    def __init__(self, d_in, d_out, depth, device=None, dtype=torch.bfloat16, seed=1, std=0.02):
        super().__init__()
        gen = torch.Generator(device=device if device is not None else "cpu").manual_seed(seed)
        self.depth = depth
        for i in range(depth):
            self.put(f"{2 * i}.weight", init_tensor((d_out, d_in if i == 0 else d_out), "w", gen, std, device, dtype))
            self.put(f"{2 * i}.bias", init_tensor((d_out,), "zero", gen, std, device, dtype))

    def _pack(self):
        return [(bf(self.get(f"{2 * i}.weight")), f32(self.get(f"{2 * i}.bias"))) for i in range(self.depth)]

    def forward(self, x):
        shp = x.shape
        h = ops.convert(x.reshape(-1, shp[-1]).contiguous(), torch.bfloat16)
        pk = self.packed()
        for i, (w, b) in enumerate(pk):
            last = i == len(pk) - 1
            h = ops.gemm(h, w, bias=b, act=None if last else "gelu", out_dtype=torch.float32 if last else torch.bfloat16)
        return h.view(*shp[:-1], -1)


def _regstage_params(h, prefix, depth, cin, cout, mk):
    for i in range(depth):
        b = f"{prefix}b{i + 1}."
        ci = cin if i == 0 else cout
        rd = int(round(ci * 0.25))
        h.put(b + "conv1.conv.weight", mk((cout, ci, 1, 1)))
        h.put(b + "conv2.conv.weight", mk((cout, 1, 3, 3)))
        h.put(b + "conv3.conv.weight", mk((cout, cout, 1, 1)))
        for c in ("conv1", "conv2", "conv3"):
            h.put(b + c + ".bn.weight", mk((cout,), "one")); h.put(b + c + ".bn.bias", mk((cout,), "zero"))
        h.put(b + "se.fc1.weight", mk((rd, cout, 1, 1))); h.put(b + "se.fc1.bias", mk((rd,), "zero"))
        h.put(b + "se.fc2.weight", mk((cout, rd, 1, 1))); h.put(b + "se.fc2.bias", mk((cout,), "zero"))
        if ci != cout:
            h.put(b + "downsample.conv.weight", mk((cout, ci, 1, 1)))
            h.put(b + "downsample.bn.weight", mk((cout,), "one")); h.put(b + "downsample.bn.bias", mk((cout,), "zero"))


class STCConnector(PackedModule):
    """ref projector.py:133-215 (base: Conv3d padding 1)."""
    PADDING = 1
    AVGPOOL = False          # STP / spatial_pool: nn.AvgPool3d(downsample) + SiLU instead of the Conv3d sampler

    def __init__(self, config, downsample=(2, 2, 2), depth=4, mlp_depth=2, device=None, dtype=torch.bfloat16, seed=1,
                 std=0.02, ln_eps=1e-5):
        super().__init__()
        self.encoder_hidden_size = cin = config.mm_hidden_size
        self.hidden_size = hid = config.hidden_size
        self.output_hidden_size = config.hidden_size
        self.depth, self.mlp_depth, self.downsample, self.ln_eps = depth, mlp_depth, tuple(downsample), ln_eps
        gen = torch.Generator(device=device if device is not None else "cpu").manual_seed(seed)
        mk = lambda shape, kind="w": init_tensor(shape, kind, gen, std, device, dtype)
        if depth:
            self.s1 = Holder(); _regstage_params(self.s1, "", depth, cin, hid, mk)
        self.sampler = Holder()
        if not self.AVGPOOL:
            self.sampler.put("0.weight", mk((hid, hid, *self.downsample))); self.sampler.put("0.bias", mk((hid,), "zero"))
        if depth:
            self.s2 = Holder(); _regstage_params(self.s2, "", depth, hid, hid, mk)
        self.readout = Holder()
        for i in range(mlp_depth):
            self.readout.put(f"{2 * i}.weight", mk((hid, hid))); self.readout.put(f"{2 * i}.bias", mk((hid,), "zero"))

    # ---- packing -------------------------------------------------------------------------------
    def _pack_stage(self, st):
        blocks = []
        for i in range(self.depth):
            b = st.get(f"b{i + 1}")
            C = b.conv1.conv.weight.shape[0]
            blk = dict(
                w1=self.gw(b.conv1.conv.weight.reshape(C, -1)), n1=(f32(b.conv1.bn.weight), f32(b.conv1.bn.bias)),
                w9=f32(b.conv2.conv.weight.reshape(C, 9).t()), n2=(f32(b.conv2.bn.weight), f32(b.conv2.bn.bias)),
                se1=(bf(b.se.fc1.weight.reshape(b.se.fc1.weight.shape[0], C)), f32(b.se.fc1.bias)),
                se2=(bf(b.se.fc2.weight.reshape(C, -1)), f32(b.se.fc2.bias)),
                w3=self.gw(b.conv3.conv.weight.reshape(C, C)), n3=(f32(b.conv3.bn.weight), f32(b.conv3.bn.bias)), ds=None)
            if hasattr(b, "downsample"):
                blk["ds"] = (self.gw(b.downsample.conv.weight.reshape(C, -1)), f32(b.downsample.bn.weight), f32(b.downsample.bn.bias))
            blocks.append(blk)
        return blocks

    def _pack(self):
        pk = {}
        if self.depth:
            pk["s1"], pk["s2"] = self._pack_stage(self.s1), self._pack_stage(self.s2)
        if not self.AVGPOOL:
            w = self.sampler.get("0.weight")                   # [Co, Ci, kt, kh, kw] -> [Co, (kt kh kw Ci)]
            pk["samp_w"] = self.gw(w.permute(0, 2, 3, 4, 1).reshape(w.shape[0], -1))
            pk["samp_b"] = f32(self.sampler.get("0.bias"))
        pk["readout"] = [(self.gw(self.readout.get(f"{2 * i}.weight")), f32(self.readout.get(f"{2 * i}.bias")))
                         for i in range(self.mlp_depth)]
        return pk

    # ---- one RegNet bottleneck, token-major [F*P, C] ------------------------------------------------
    def _block(self, x, blk, F, H, W):
        eps, P = self.ln_eps, H * W
        y = ops.gemm(x, blk["w1"])
        y = ops.layernorm(y, blk["n1"][0], blk["n1"][1], eps, act="silu")
        y = ops.dwconv3x3_ln_silu(y, blk["w9"], blk["n2"][0], blk["n2"][1], F, H, W, y.shape[1], eps)
        s = ops.colmean(y, F, P)
        s = ops.gemm(s, blk["se1"][0], bias=blk["se1"][1], act="silu")
        s = ops.gemm(s, blk["se2"][0], bias=blk["se2"][1], act="sigmoid")
        ops.scale_channels(y, s, F, P)
        z = ops.gemm(y, blk["w3"])
        if blk["ds"] is not None:
            sc = ops.gemm(x, blk["ds"][0])
            return ops.ln_add_silu(z, blk["n3"][0], blk["n3"][1], sc, blk["ds"][1], blk["ds"][2], eps)
        return ops.ln_add_silu(z, blk["n3"][0], blk["n3"][1], x, None, None, eps)

    def c_model(self):
        """ctypes view (include/ufv.h ufv_stc_model) of the packed connector -> (struct, objects to keep alive)"""
        import ctypes
        from .. import _lib
        pk = self.packed()
        keep = []

        def stage(blocks):
            arr = (_lib.StcBlock * max(len(blocks), 1))()
            for i, b in enumerate(blocks):
                ds = b["ds"]
                arr[i] = _lib.StcBlock(b["w1"].shape[1], b["w1"].shape[0], b["se1"][0].shape[0], 0, b["w1"].data_ptr(), b["n1"][0].data_ptr(), b["n1"][1].data_ptr(),
                                       b["w9"].data_ptr(), b["n2"][0].data_ptr(), b["n2"][1].data_ptr(), b["se1"][0].data_ptr(), b["se1"][1].data_ptr(),
                                       b["se2"][0].data_ptr(), b["se2"][1].data_ptr(), b["w3"].data_ptr(), b["n3"][0].data_ptr(), b["n3"][1].data_ptr(),
                                       ds[0].data_ptr() if ds else None, ds[1].data_ptr() if ds else None, ds[2].data_ptr() if ds else None)
            keep.append(arr)
            return arr
        n = len(pk["readout"])
        rw = (ctypes.c_void_p * n)(*[w.data_ptr() for w, _ in pk["readout"]])
        rb = (ctypes.c_void_p * n)(*[b.data_ptr() for _, b in pk["readout"]])
        keep += [rw, rb]
        m = _lib.StcModel(depth=self.depth, mlp_depth=n, kt=self.downsample[0], kh=self.downsample[1], kw=self.downsample[2], pad=self.PADDING,
                          avgpool=int(self.AVGPOOL), c_in=self.encoder_hidden_size, c_hid=self.hidden_size, eps=self.ln_eps,
                          s1=stage(pk["s1"]) if self.depth else None, s2=stage(pk["s2"]) if self.depth else None,
                          samp_w=pk["samp_w"].data_ptr() if not self.AVGPOOL else None, samp_b=pk["samp_b"].data_ptr() if not self.AVGPOOL else None,
                          readout_w=rw, readout_b=rb)
        return m, keep

    def _stage_call_ok(self, pk):
        ws = [w for w, _ in pk["readout"]] + ([] if self.AVGPOOL else [pk["samp_w"]]) + ([b["w1"] for b in pk["s1"] + pk["s2"]] if self.depth else [])
        return not any(isinstance(w, ops.Fp8Weight) for w in ws) and os.environ.get("UFV_STAGE_CALLS", "1") != "0"

    def forward_one(self, x, t, hw):
        """x: [t*hw*hw, C_in] any float dtype (one video, token-major) -> fp32 [tokens, hidden]"""
        pk = self.packed()
        if self._stage_call_ok(pk):
            # the whole connector as ONE C call (ufv_stc_forward, csrc/stages.hip): the same launches in the same order, bit-identical
            import ctypes
            from .. import _lib
            m, keep = self.c_model()
            x = x.contiguous()
            nbytes = _lib.load().ufv_stc_forward_ws_bytes(ctypes.byref(m), t, hw)
            ws = torch.empty((nbytes,), device=x.device, dtype=torch.uint8)
            k, p = self.downsample, (0 if self.AVGPOOL else self.PADDING)
            To, Ho, Wo = [(n_ // k_) if self.AVGPOOL else ((n_ + 2 * p - k_) // k_ + 1) for n_, k_ in zip((t, hw, hw), k)]
            out = torch.empty((To * Ho * Wo, self.hidden_size), device=x.device, dtype=torch.float32)
            _lib.call("ufv_stc_forward", ctypes.byref(m), x.data_ptr(), ops._DT[x.dtype], t, hw, out.data_ptr(), ws.data_ptr(), nbytes,
                      torch.cuda.current_stream().cuda_stream)
            return out
        h = ops.convert(x.contiguous(), torch.bfloat16)
        if self.depth:
            for blk in pk["s1"]:
                h = self._block(h, blk, t, hw, hw)
        C = h.shape[1]
        if self.AVGPOOL:
            h, (To, Ho, Wo) = ops.avgpool3d_silu(h.contiguous(), t, hw, hw, C, self.downsample)
        else:
            A, (To, Ho, Wo) = ops.conv3d_gather(h, t, hw, hw, C, self.downsample, self.PADDING)
            h = ops.gemm(A, pk["samp_w"], bias=pk["samp_b"], act="silu")
        if self.depth:
            for blk in pk["s2"]:
                h = self._block(h, blk, To, Ho, Wo)
        n = len(pk["readout"])
        for i, (w, b) in enumerate(pk["readout"]):
            last = i == n - 1
            h = ops.gemm(h, w, bias=b, act=None if last else "gelu", out_dtype=torch.float32 if last else torch.bfloat16)
        return h

    def forward(self, x):
        """x: [b, t, l, d] or [b, t, h, w, d] -> [b, tokens, hidden] fp32"""
        b, t = x.shape[0], x.shape[1]
        if x.ndim == 4:
            hw = int(x.shape[2] ** 0.5)
        else:
            hw = x.shape[2]
        outs = [self.forward_one(x[i].reshape(t * hw * hw, x.shape[-1]), t, hw) for i in range(b)]
        return outs[0].unsqueeze(0) if b == 1 else torch.stack(outs, 0)          # (one video: a view, not a 33 MB device copy)


class STCConnectorV35(STCConnector):
    """ref projector.py:225-238: same with Conv3d padding 0."""
    PADDING = 0


class STPConnector(STCConnector):
    """ref projector.py:218-222: the Conv3d sampler replaced by AvgPool3d(downsample) + SiLU."""
    AVGPOOL = True


class SpatialConv(STCConnector):
    """ref projector.py:241-244: depth 0, downsample (1,2,2), padding 1."""

    def __init__(self, config, downsample=(1, 2, 2), depth=0, mlp_depth=2, **kw):
        super().__init__(config, downsample=downsample, depth=depth, mlp_depth=mlp_depth, **kw)


def build_vision_projector(config, delay_load=False, **kwargs):
    kind = getattr(config, "mm_projector_type", "linear")
    m = re.match(r"^mlp(\d+)x_gelu$", kind)
    if m:
        return MlpProjector(config.mm_hidden_size, config.hidden_size, int(m.group(1)), **kwargs)
    if kind == "linear":
        return MlpProjector(config.mm_hidden_size, config.hidden_size, 1, **kwargs)
    if kind == "stc_connector":
        return STCConnector(config, **kwargs)
    if kind == "stc_connector_v35":
        return STCConnectorV35(config, **kwargs)
    if kind == "spatial_conv":
        return SpatialConv(config, **kwargs)
    if kind == "identity":
        return IdentityMap()
    if kind == "stp_connector":
        return STPConnector(config, **kwargs)
    if kind == "spatial_pool":
        return SpatialPool(config, **kwargs)
    raise ValueError(f"Unknown projector type: {kind}")


class SpatialPool(STPConnector):
    """ref projector.py:247-250: depth 0, downsample (1,2,2), AvgPool3d sampler."""

    def __init__(self, config, downsample=(1, 2, 2), depth=0, mlp_depth=2, **kw):
        super().__init__(config, downsample=downsample, depth=depth, mlp_depth=mlp_depth, **kw)
