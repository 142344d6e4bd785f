"""LoRA adapters merged into the base weights at load time -- what the reference's `load_lora` gets from
`peft.PeftModel.from_pretrained(model, path).merge_and_unload()` (ufvideo/model/__init__.py:82-105).

peft is a third-party dependency that is absent from this image (not vendored in the reference either; the reference's requirements pin
no version), so this restates its published LoRA merge: for every adapted Linear,  W <- W + scaling * (B @ A),  scaling = lora_alpha / r
(lora_alpha / sqrt(r) with `use_rslora`), per-module `rank_pattern` / `alpha_pattern` overrides, the product transposed for
`fan_in_fan_out` layers; `modules_to_save` entries replace the module's weights.  Parity unpinned (no peft here to run): the arithmetic is
checked against its definition in tests/test_host_cpu.py.  One-time host-side weight preparation, like packing -- not the hot path."""
import json
import math
import os
import re

import torch

PEFT_PREFIX = "base_model.model."


def read_adapter(path):
    """-> (adapter_config dict, adapter state dict) from a peft checkpoint directory (adapter_config.json + adapter_model.safetensors | .bin)"""
    with open(os.path.join(path, "adapter_config.json")) as f:
        cfg = json.load(f)
    st = os.path.join(path, "adapter_model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        b = os.path.join(path, "adapter_model.bin")
        if not os.path.exists(b):
            raise FileNotFoundError(f"{path}: neither adapter_model.safetensors nor adapter_model.bin")
        sd = torch.load(b, map_location="cpu")
    return cfg, sd


def _pattern_value(patterns, key, default):
    # peft: the first pattern that matches the module name as a suffix (regex `(.*\.)?pattern$`) wins
    for pat, val in (patterns or {}).items():
        if re.match(rf"(.*\.)?({pat})$", key):
            return val
    return default


def merge_lora(params, cfg, adapter_sd):
    """params: {name: tensor} of the model (modified in place).  Returns the sorted list of merged weight names."""
    if cfg.get("peft_type", "LORA") != "LORA":
        raise NotImplementedError(f"peft_type {cfg.get('peft_type')}: only LoRA adapters are merged")
    if cfg.get("use_dora"):
        raise NotImplementedError("DoRA adapters (use_dora) are not supported")
    r0, a0 = cfg["r"], cfg["lora_alpha"]
    pairs, saved = {}, {}
    for k, v in adapter_sd.items():
        k = k[len(PEFT_PREFIX):] if k.startswith(PEFT_PREFIX) else k
        m = re.match(r"(.*)\.lora_([AB])(?:\.[^.]+)?\.weight$", k)              # saved adapters drop the adapter name, live ones carry `.default`
        if m:
            pairs.setdefault(m.group(1), {})[m.group(2)] = v
            continue
        m = re.match(r"(.*)\.modules_to_save(?:\.[^.]+)?\.(weight|bias)$", k)
        if m:
            saved[f"{m.group(1)}.{m.group(2)}"] = v
            continue
        if "lora_embedding_" in k or "lora_magnitude" in k:
            raise NotImplementedError(f"adapter tensor {k}: embedding / DoRA adapters are not supported")
        raise KeyError(f"adapter tensor {k} is neither a lora_A / lora_B pair nor a modules_to_save entry")
    merged = []
    for mod, ab in sorted(pairs.items()):
        if set(ab) != {"A", "B"}:
            raise KeyError(f"{mod}: lora_A / lora_B incomplete")
        name = mod + ".weight"
        if name not in params:
            raise KeyError(f"adapter targets {name}, which the model does not have")
        A, B = ab["A"].float(), ab["B"].float()                               # A [r, in], B [out, r]
        r = _pattern_value(cfg.get("rank_pattern"), mod, r0)
        if A.shape[0] != r or B.shape[1] != r:
            raise ValueError(f"{mod}: adapter rank {tuple(A.shape)} x {tuple(B.shape)} does not match r = {r}")
        alpha = _pattern_value(cfg.get("alpha_pattern"), mod, a0)
        scaling = alpha / math.sqrt(r) if cfg.get("use_rslora") else alpha / r
        delta = (B @ A) * scaling
        if cfg.get("fan_in_fan_out"):
            delta = delta.t()
        w = params[name]
        if tuple(delta.shape) != tuple(w.shape):
            raise ValueError(f"{name}: adapter product {tuple(delta.shape)} vs weight {tuple(w.shape)}")
        w.copy_((w.float() + delta.to(w.device)).to(w.dtype))
        merged.append(name)
    for name, v in sorted(saved.items()):
        if name not in params:
            raise KeyError(f"modules_to_save holds {name}, which the model does not have")
        params[name].copy_(v.to(params[name].dtype))
        merged.append(name)
    return merged


def strip_non_lora_prefixes(sd):
    """key normalisation of the reference's `non_lora_trainables.bin` loader (ufvideo/model/__init__.py:92-95)"""
    sd = {(k[11:] if k.startswith("base_model.") else k): v for k, v in sd.items()}
    if any(k.startswith("model.model.") for k in sd):
        sd = {(k[6:] if k.startswith("model.") else k): v for k, v in sd.items()}
    return sd
