"""Parameter containers: nested nn.Modules whose state_dict keys equal the reference's
(HF / timm names), plus helpers to turn them into kernel-ready (packed) device buffers."""
import torch
import torch.nn as nn


class Holder(nn.Module):
    """Empty named container; parameters are attached with `put`."""

    def put(self, dotted, tensor):
        mod = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, Holder())
            mod = getattr(mod, p)
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))

    def get(self, dotted):
        mod = self
        for p in dotted.split("."):
            mod = getattr(mod, p)
        return mod


def init_tensor(shape, kind, gen, std, device, dtype):
    """kind: 'w' normal(0,std), 'one' ones, 'zero' zeros."""
    if kind == "one":
        return torch.ones(shape, device=device, dtype=dtype)
    if kind == "zero":
        return torch.zeros(shape, device=device, dtype=dtype)
    t = torch.empty(shape, device=device, dtype=torch.float32).normal_(0.0, std, generator=gen)
    return t.to(dtype)


class PackedModule(Holder):
    """Base for modules that keep reference-named parameters and derive packed kernel buffers from
    them on first use.  Moving / casting / loading the module invalidates the packed buffers."""

    gemm_dtype = "bf16"          # "fp8": large GEMM weights are packed as e4m3 + per-channel scale (W8A8 path, config #5a)

    def __init__(self):
        super().__init__()
        self._packed = None
        self._owner = None       # set by ufvideo_amd.train.DecoderTrainer while the packed buffers ARE its flat training buffers

    def _check_owner(self, what):
        if getattr(self, "_owner", None) is not None:
            raise RuntimeError(f"{what}: a DecoderTrainer owns this module's packed weights (they are views of its gradient buckets). "
                               "Call trainer.sync_to_model() to write the trained weights into the parameters and trainer.detach() to release "
                               "them -- re-packing now would silently fall back to the parameters from before training.")

    def gw(self, w):
        """A GEMM weight [N, K] in this module's compute format: bf16, or `ops.Fp8Weight` when gemm_dtype == "fp8" and the
        shape fits the fp8 MFMA tiles (N % 128 == 0, K % 128 == 0)."""
        w = bf(w)
        if self.gemm_dtype == "fp8" and w.is_cuda and w.shape[0] % 128 == 0 and w.shape[1] % 128 == 0:
            from .. import ops
            return ops.Fp8Weight(w)
        return w

    def _apply(self, fn, *a, **k):
        self._check_owner("moving / casting the module")
        self._packed = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._check_owner("load_state_dict")
        self._packed = None
        return super().load_state_dict(*a, **k)

    def invalidate(self):
        self._check_owner("invalidate")
        self._packed = None
        for m in self.modules():
            if isinstance(m, PackedModule):
                m._packed = None

    def packed(self):
        if self._packed is None:
            with torch.no_grad():
                self._packed = self._pack()
        return self._packed

    def _pack(self):
        raise NotImplementedError


def bf(t):
    return t.detach().to(torch.bfloat16).contiguous()


def f32(t):
    return t.detach().to(torch.float32).contiguous()


def pad_rows(w, n):
    """zero-pad dim 0 of a 2-D/1-D tensor to n"""
    if w.shape[0] == n:
        return w
    out = torch.zeros((n,) + tuple(w.shape[1:]), device=w.device, dtype=w.dtype)
    out[: w.shape[0]] = w
    return out


def pad_cols(w, k):
    if w.shape[1] == k:
        return w
    out = torch.zeros((w.shape[0], k), device=w.device, dtype=w.dtype)
    out[:, : w.shape[1]] = w
    return out


def round_up(x, m):
    return (x + m - 1) // m * m


def set_gemm_dtype(root, mode):
    """Switch every packed module under `root` between the bf16 and the W8A8 fp8 GEMM path (repacks lazily)."""
    assert mode in ("bf16", "fp8")
    for m in root.modules():
        if isinstance(m, PackedModule):
            m._check_owner("set_gemm_dtype")
            m.gemm_dtype = mode
            m._packed = None
