"""Vision towers (SigLIP / CLIP) behind the reference's wrapper interface (ufvideo/model/encoder.py).

The arithmetic — patch-embed conv as an im2col GEMM, pre-LN encoder layers with MFMA GEMMs and the
hd=72 flash-attention kernel — runs in csrc/ through the C ABI; this file only holds parameters
under the reference's state-dict names, packs them once, and sequences kernel launches.
Only `hidden_states[select_layer]` is ever used by the reference (encoder.py:126-132), so the layers
after it, post_layernorm and the pooling head are never executed.
"""
import json
import os

import torch
import torch.nn as nn

from .. import ops
from ..mm_utils import UfvImageProcessor
from ._params import Holder, PackedModule, init_tensor, bf, f32, pad_rows, pad_cols, round_up


class VisionConfig:
    def __init__(self, hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16,
                 image_size=384, patch_size=14, num_channels=3, layer_norm_eps=1e-6, hidden_act="gelu_pytorch_tanh",
                 model_type="siglip_vision_model", **unused):
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.image_size = image_size
        self.patch_size = patch_size
        self.num_channels = num_channels
        self.layer_norm_eps = layer_norm_eps
        self.hidden_act = hidden_act
        self.model_type = model_type

    @classmethod
    def from_pretrained(cls, path, **kw):
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        d = d.get("vision_config", d)
        return cls(**d)

    def to_dict(self):
        return dict(self.__dict__)


SIGLIP_SO400M = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16,
                     image_size=384, patch_size=14)
CLIP_L_336 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336,
                  patch_size=14, layer_norm_eps=1e-5, hidden_act="quick_gelu", model_type="clip_vision_model")


class _VitBody(PackedModule):
    """Parameters of the HF vision model (names as in transformers 4.46.3: `vision_model.*`) + runner."""

    def __init__(self, cfg: VisionConfig, clip: bool, device=None, dtype=torch.bfloat16, seed=0, std=0.02):
        super().__init__()
        self.cfg, self.clip = cfg, clip
        gen = torch.Generator(device=device if device is not None else "cpu").manual_seed(seed)
        vm = Holder()
        self.vision_model = vm
        D, I, P = cfg.hidden_size, cfg.intermediate_size, cfg.patch_size
        n = (cfg.image_size // P) ** 2
        mk = lambda shape, kind="w": init_tensor(shape, kind, gen, std, device, dtype)
        vm.put("embeddings.patch_embedding.weight", mk((D, cfg.num_channels, P, P)))
        if clip:
            vm.put("embeddings.class_embedding", mk((D,)))
            vm.put("embeddings.position_embedding.weight", mk((n + 1, D)))
            vm.put("pre_layrnorm.weight", mk((D,), "one")); vm.put("pre_layrnorm.bias", mk((D,), "zero"))
        else:
            vm.put("embeddings.patch_embedding.bias", mk((D,), "zero"))
            vm.put("embeddings.position_embedding.weight", mk((n, D)))
        for i in range(cfg.num_hidden_layers):
            p = f"encoder.layers.{i}."
            for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
                vm.put(p + f"self_attn.{nm}.weight", mk((D, D))); vm.put(p + f"self_attn.{nm}.bias", mk((D,), "zero"))
            for ln in ("layer_norm1", "layer_norm2"):
                vm.put(p + ln + ".weight", mk((D,), "one")); vm.put(p + ln + ".bias", mk((D,), "zero"))
            vm.put(p + "mlp.fc1.weight", mk((I, D))); vm.put(p + "mlp.fc1.bias", mk((I,), "zero"))
            vm.put(p + "mlp.fc2.weight", mk((D, I))); vm.put(p + "mlp.fc2.bias", mk((D,), "zero"))

    # -- packing: q|k|v fused, d_ff padded to a multiple of 128 (4304 -> 4352), patch K padded to 64
    def _pack(self):
        cfg, vm = self.cfg, self.vision_model
        D, I = cfg.hidden_size, cfg.intermediate_size
        Ip = round_up(I, 128)
        K = cfg.num_channels * cfg.patch_size ** 2
        Kp = round_up(K, 64)
        pk = {"Ip": Ip, "Kp": Kp}
        pk["patch_w"] = bf(pad_cols(vm.embeddings.patch_embedding.weight.reshape(D, K), Kp))
        pos = f32(vm.embeddings.position_embedding.weight)
        if self.clip:
            pk["patch_b"] = None
            pk["cls_row"] = (f32(vm.embeddings.class_embedding) + pos[0])[None].contiguous()
            pk["pos"] = pos[1:].contiguous()
            pk["pre_ln"] = (f32(vm.pre_layrnorm.weight), f32(vm.pre_layrnorm.bias))
        else:
            pk["patch_b"] = f32(vm.embeddings.patch_embedding.bias)
            pk["pos"] = pos
        layers = []
        for i in range(cfg.num_hidden_layers):
            L = vm.encoder.layers.get(str(i))
            a = L.self_attn
            layers.append(dict(
                ln1=(f32(L.layer_norm1.weight), f32(L.layer_norm1.bias)), ln2=(f32(L.layer_norm2.weight), f32(L.layer_norm2.bias)),
                wqkv=self.gw(torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0)),
                bqkv=f32(torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0)),
                # out_proj stays bf16 in the W8A8 mode too (round 5): K = 1152 makes the GEMM seam-bound -- the e4m3 kernel (63 us) is no faster than the bf16 one (65 us)
                # and needed a quantise launch (13 us) for the attention kernel's bf16 output in front of it
                wo=bf(a.out_proj.weight), bo=f32(a.out_proj.bias),
                w1=self.gw(pad_rows(L.mlp.fc1.weight, Ip)), b1=f32(pad_rows(L.mlp.fc1.bias, Ip)),
                w2=self.gw(pad_cols(L.mlp.fc2.weight, Ip)), b2=f32(L.mlp.fc2.bias)))
        pk["layers"] = layers
        return pk

    def _encode_c(self, pixels, n_layers, n):
        import ctypes
        from .. import _lib
        m, keep = _vit_c_model(self)
        T = pixels.shape[0]
        nbytes = _lib.load().ufv_vit_forward_ws_bytes(ctypes.byref(m), T)
        ws = torch.empty((nbytes,), device=pixels.device, dtype=torch.uint8)
        x = torch.empty((T * n, self.cfg.hidden_size), device=pixels.device, dtype=torch.float32)
        _lib.call("ufv_vit_forward", ctypes.byref(m), pixels.data_ptr(), ops._DT[pixels.dtype], T, pixels.shape[-2], pixels.shape[-1], n_layers,
                  x.data_ptr(), ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
        return x

    def encode(self, pixels, n_layers):
        """pixels [T,3,H,W] (f32/f16/bf16, device) -> fp32 residual stream [T*S, D] after `n_layers` layers
        (S = patches (+1 with CLS))."""
        cfg, pk = self.cfg, self.packed()
        T = pixels.shape[0]
        D, H, P = cfg.hidden_size, cfg.num_attention_heads, cfg.patch_size
        n = (pixels.shape[-1] // P) * (pixels.shape[-2] // P)
        if n != pk["pos"].shape[0]:
            raise ValueError(f"vision tower built for {pk['pos'].shape[0]} patches, got {n} (image {tuple(pixels.shape[-2:])})")
        hd = D // H
        # UFV_TOWER_STREAM=bf16 (opt-in, NOT the default): the residual stream kept in bf16 as the reference's bf16 tower keeps it -- in-place updates through
        # ufv_gemm_stream_bf16, half the stream traffic of out_proj / fc2 / the LayerNorms (-0.67 ms per 32-frame clip, same box) for twice the distance from the
        # fp32 oracle (LABNOTES round 5, tests/test_kernels_gpu.py::test_tower_bf16_stream_option); the default stream is fp32
        # (works in the W8A8 mode too -- ufv_gemm_fp8_mx takes a bf16 residual --, where it is worth -0.2 ms: 33.5 -> 33.3, same box)
        sb = os.environ.get("UFV_TOWER_STREAM") == "bf16" and not self.clip
        if sb and any(isinstance(L["w2"], ops.Fp8Weight) for L in pk["layers"][:n_layers]) and (
                os.environ.get("UFV_FP8_NO_MX") is not None or T * n < 256 or pk["Ip"] % 256 != 0):
            sb = False        # W8A8 without the MX chain (switched off, < 256 rows, unaligned d_ff): fc2 would meet an e4m3 weight in the bf16-stream GEMM -- the fp32 stream there
        if (not self.clip and not any(isinstance(L["wqkv"], ops.Fp8Weight) for L in pk["layers"][:n_layers])
                and os.environ.get("UFV_STAGE_CALLS", "1") != "0" and not sb):
            # the whole tower as ONE C call (ufv_vit_forward, csrc/stages.hip): the same launches in the same order, bit-identical
            return self._encode_c(pixels.contiguous(), n_layers, n), n
        cols = ops.patchify(pixels.contiguous(), P, pk["Kp"])
        x = ops.gemm(cols, pk["patch_w"], bias=pk["patch_b"], resid=pk["pos"], resid_rows=n, out_dtype=torch.bfloat16 if sb else torch.float32)
        S = n
        if self.clip:                     # [CLS] row + patches, then pre-LN (in place on the fp32 stream)
            S = n + 1
            xs = torch.empty((T * S, D), device=x.device, dtype=torch.float32)
            t_idx = torch.arange(T, device=x.device)
            ops.gather_rows(pk["cls_row"], torch.zeros(T, dtype=torch.int64, device=x.device), xs, t_idx * S)
            dst = (t_idx[:, None] * S + 1 + torch.arange(n, device=x.device)[None]).reshape(-1)
            ops.gather_rows(x, None, xs, dst)
            x = ops.layernorm(xs, pk["pre_ln"][0], pk["pre_ln"][1], cfg.layer_norm_eps, out_dtype=torch.float32)
        M = T * S
        h = torch.empty((M, D), device=x.device, dtype=torch.bfloat16)
        qkv = torch.empty((M, 3 * D), device=x.device, dtype=torch.bfloat16)
        o = torch.empty((M, D), device=x.device, dtype=torch.bfloat16)
        ff = torch.empty((M, pk["Ip"]), device=x.device, dtype=torch.bfloat16)
        st = (S * 3 * D, 3 * D)
        for L in pk["layers"][:n_layers]:
            q8 = isinstance(L["wqkv"], ops.Fp8Weight)          # W8A8: the LayerNorms emit e4m3 + row scale directly
            hq = (ops.layernorm(x, L["ln1"][0], L["ln1"][1], cfg.layer_norm_eps, quant=True) if q8
                  else ops.layernorm(x, L["ln1"][0], L["ln1"][1], cfg.layer_norm_eps, out=h))
            ops.gemm(hq, L["wqkv"], bias=L["bqkv"], out=qkv)
            ops.attention(qkv, qkv[:, D:], qkv[:, 2 * D:], T, H, H, S, S, hd, st, st, st, out=o)
            if sb:
                ops.gemm_stream_bf16(o, L["wo"], x, bias=L["bo"])
            else:
                ops.gemm(o, L["wo"], bias=L["bo"], resid=x, out=x)
            hq = (ops.layernorm(x, L["ln2"][0], L["ln2"][1], cfg.layer_norm_eps, quant=True) if q8
                  else ops.layernorm(x, L["ln2"][0], L["ln2"][1], cfg.layer_norm_eps, out=h))
            if (q8 and isinstance(L["w1"], ops.Fp8Weight) and isinstance(L["w2"], ops.Fp8Weight) and M >= 256 and pk["Ip"] % 256 == 0
                    and os.environ.get("UFV_FP8_NO_MX") is None):
                # W8A8, fused (round 5): fc1's GELU epilogue writes e4m3 codes + MX block scales, fc2 takes them as its block-scaled A operand: no quantise launch
                ffm = ops.gemm_fp8_mx(hq, L["w1"], bias=L["b1"], act=cfg.hidden_act, mx_out=True)
                ops.gemm_fp8_mx(ffm, L["w2"], bias=L["b2"], resid=x, out=x)
            else:
                ops.gemm(hq, L["w1"], bias=L["b1"], act=cfg.hidden_act, out=ff)
                if sb:
                    ops.gemm_stream_bf16(ff, L["w2"], x, bias=L["b2"])
                else:
                    ops.gemm(ff, L["w2"], bias=L["b2"], resid=x, out=x)
        return x, S


def _vit_c_model(tower):
    """ctypes view (include/ufv.h ufv_vit_model) of a packed SigLIP tower -> (struct, layer array to keep alive)"""
    from .. import _lib
    cfg, pk = tower.cfg, tower.packed()
    n = len(pk["layers"])
    layers = (_lib.VitLayer * n)()
    for i, L in enumerate(pk["layers"]):
        layers[i] = _lib.VitLayer(L["ln1"][0].data_ptr(), L["ln1"][1].data_ptr(), L["ln2"][0].data_ptr(), L["ln2"][1].data_ptr(),
                                  L["wqkv"].data_ptr(), L["bqkv"].data_ptr(), L["wo"].data_ptr(), L["bo"].data_ptr(),
                                  L["w1"].data_ptr(), L["b1"].data_ptr(), L["w2"].data_ptr(), L["b2"].data_ptr())
    m = _lib.VitModel(n_layers=n, d=cfg.hidden_size, n_heads=cfg.num_attention_heads, d_ff_pad=pk["Ip"], patch=cfg.patch_size,
                      channels=cfg.num_channels, kpad=pk["Kp"], n_patches=pk["pos"].shape[0], act=ops.ACT[cfg.hidden_act],
                      eps=cfg.layer_norm_eps, patch_w=pk["patch_w"].data_ptr(), patch_b=pk["patch_b"].data_ptr(), pos=pk["pos"].data_ptr(),
                      layers=layers)
    return m, layers


class _TowerBase(nn.Module):
    CLIP = False
    DEFAULT = SIGLIP_SO400M

    def __init__(self, vision_tower, args, delay_load=False, vision_config=None, **kwargs):
        super().__init__()
        self.is_loaded = False
        self.vision_tower_name = vision_tower
        self.select_layer = args.mm_vision_select_layer
        self.select_feature = getattr(args, "mm_vision_select_feature", "patch")
        cfg = vision_config or getattr(args, "vision_config", None)
        if isinstance(cfg, dict):
            cfg = VisionConfig(**cfg)
        if cfg is None:
            path = self._local_path()
            if path is not None:
                cfg = VisionConfig.from_pretrained(path)
            else:
                cfg = VisionConfig(**self.DEFAULT)
        self.cfg_only = cfg
        self._init_kw = kwargs
        if not delay_load:
            self.load_model()

    def _local_path(self):
        for p in (getattr(self, "path", None), self.vision_tower_name):
            if p and os.path.isfile(os.path.join(p, "config.json")):
                return p
        return None

    def load_model(self, device=None, dtype=torch.bfloat16, seed=0):
        """Builds the tower (random init) and, when a local HF directory with safetensors exists, loads it."""
        if self.is_loaded:
            return
        cfg = self.cfg_only
        self.image_processor = UfvImageProcessor(size=cfg.image_size,
                                                 image_mean=(0.5, 0.5, 0.5) if not self.CLIP else (0.48145466, 0.4578275, 0.40821073),
                                                 image_std=(0.5, 0.5, 0.5) if not self.CLIP else (0.26862954, 0.26130258, 0.27577711))
        self.vision_tower = _VitBody(cfg, self.CLIP, device=device, dtype=dtype, seed=seed)
        self.vision_tower.requires_grad_(False)
        path = self._local_path()
        if path is not None:
            f = os.path.join(path, "model.safetensors")
            if os.path.isfile(f):
                from safetensors.torch import load_file
                self.load_hf_state_dict(load_file(f))
        self.is_loaded = True

    def load_hf_state_dict(self, sd):
        """Accepts HF vision-model keys with or without the `vision_model.` prefix; ignores the
        unused post_layernorm / pooling head."""
        own = self.vision_tower.state_dict()
        new = {}
        for k, v in sd.items():
            k2 = k if k.startswith("vision_model.") else "vision_model." + k
            if k2 in own:
                new[k2] = v
        missing = [k for k in own if k not in new]
        if missing:
            raise KeyError(f"vision tower weights missing: {missing[:4]} ...")
        self.vision_tower.load_state_dict(new, strict=True)

    def _n_layers(self):
        L = self.config.num_hidden_layers
        return L + 1 + self.select_layer if self.select_layer < 0 else self.select_layer

    def encode(self, images):
        """-> fp32 [T, N, D] features of `hidden_states[select_layer]` (internal fast path)."""
        x, S = self.vision_tower.encode(images, self._n_layers())
        x = x.view(images.shape[0], S, -1)
        return self.feature_select_tensor(x)

    def feature_select_tensor(self, x):
        return x

    @torch.no_grad()
    def forward(self, images):
        if type(images) is list:
            return [self.encode(im.unsqueeze(0)).to(im.dtype) for im in images]
        return self.encode(images).to(images.dtype)

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        return next(self.vision_tower.parameters()).dtype

    @property
    def device(self):
        return next(self.vision_tower.parameters()).device

    @property
    def config(self):
        return self.vision_tower.cfg if self.is_loaded else self.cfg_only

    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches(self):
        return (self.config.image_size // self.config.patch_size) ** 2

    @property
    def num_patches_per_side(self):
        return self.config.image_size // self.config.patch_size

    @property
    def image_size(self):
        return self.config.image_size


class SiglipVisionTower(_TowerBase):
    """ref encoder.py:96-181 (the reference hard-codes the cwd-relative directory below)."""
    path = "siglip-so400m-patch14-384"

    def feature_select_tensor(self, x):
        if self.select_feature != "patch":
            raise ValueError(f"Unexpected select feature: {self.select_feature}")
        return x


class CLIPVisionTower(_TowerBase):
    """ref encoder.py:12-93: CLS token dropped for 'patch', kept for 'cls_patch'."""
    CLIP = True
    DEFAULT = CLIP_L_336
    path = None

    def feature_select_tensor(self, x):
        if self.select_feature == "patch":
            return x[:, 1:]
        if self.select_feature == "cls_patch":
            return x
        raise ValueError(f"Unexpected select feature: {self.select_feature}")


def build_vision_tower(vision_tower_cfg, **kwargs):
    name = getattr(vision_tower_cfg, "mm_vision_tower", getattr(vision_tower_cfg, "vision_tower", None))
    if "clip" in name:
        return CLIPVisionTower(name, args=vision_tower_cfg, **kwargs)
    if "siglip" in name:
        return SiglipVisionTower(name, args=vision_tower_cfg, **kwargs)
    raise ValueError(f"Unknown vision tower: {name}")
