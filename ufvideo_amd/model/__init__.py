"""Model loading with the reference's signature (ufvideo/model/__init__.py:39-156), offline only."""
import os

import torch

from .encoder import build_vision_tower, SiglipVisionTower, CLIPVisionTower, VisionConfig  # noqa: F401
from .layer import MaskExtractor, token_merge, build_region_encoder  # noqa: F401
from .projector import build_vision_projector, load_mm_projector, STCConnector, STCConnectorV35, SpatialConv  # noqa: F401
from .sam2 import Hiera, FpnNeck, ImageEncoder, PositionEmbeddingSine, build_sam2_image_encoder  # noqa: F401
from .videorefer_qwen2 import (VideoReferQwen2Config, VideoReferQwen2Model, VideoReferQwen2ForCausalLM,  # noqa: F401
                               UFVideoForCausalLM, KVCache, QWEN2_7B)

VLLMs = {"videorefer": VideoReferQwen2ForCausalLM, "videorefer_qwen2": VideoReferQwen2ForCausalLM}
VLLMConfigs = {"videorefer": VideoReferQwen2Config, "videorefer_qwen2": VideoReferQwen2Config}


def load_pretrained_model(model_path, model_base, model_name, load_8bit=False, load_4bit=False, device_map="auto",
                          device="cuda", use_flash_attn=False, lora=False, args=None, **kwargs):
    """-> (tokenizer, model, processor, context_len), the reference's three loading forms (ufvideo/model/__init__.py:31-157), local directories
    only (config.json + *.safetensors + tokenizer files; no hub):
      * SFT checkpoint (model_base None): everything from `model_path`;
      * base / pre-training form (`model_base` given or config.tune_mm_mlp_adapter): language model from `model_base` with `model_path`'s
        config, then `mm_projector.bin` of `model_path` on top;
      * lora=True: base model from `model_base`, `initialize_MM_tokenizer(tokenizer of model_path)`, `non_lora_trainables.bin`, then the LoRA
        adapter merged into the weights (model/lora.py).  The reference overwrites `model_base` with an empty string on this branch
        (:83), which cannot load anything; here the caller's `model_base` is used.
    8 / 4-bit (bitsandbytes) loading is not available: the GEMMs run on bf16 or e4m3 (`model.set_gemm_dtype("fp8")`)."""
    if load_8bit or load_4bit:
        raise NotImplementedError("8 / 4-bit bitsandbytes loading is not available; W8A8 e4m3 is model.set_gemm_dtype('fp8')")
    if isinstance(device_map, dict) and "" in device_map:
        device = device_map[""]
    from transformers import AutoTokenizer

    def tok_from(path, **kw):
        try:                                    # tokenisation is not on the hot path; any HF tokenizer works
            return AutoTokenizer.from_pretrained(path, local_files_only=True, **kw)
        except Exception as ex:  # pragma: no cover
            raise RuntimeError(f"could not load a tokenizer from {path}: {ex}")

    def set_sam(cfg):
        cfg.train_mask_decoder = False
        cfg.sam_pretrained = "sam2-hiera-large/sam2_hiera_large.pt"
        cfg.sam_out_dim = 256
        return cfg
    # modules a plain language-model base does not hold: they come from non_lora_trainables / mm_projector.bin (or stay as built, as in the reference)
    MM = ("model.mm_projector.", "model.region_encoder.", "model.text_hidden_fcs.", "model.mask_encoder.", "model.vision_tower.")
    config = set_sam(VideoReferQwen2Config.from_pretrained(model_path))
    is_pretraining = bool(getattr(config, "tune_mm_mlp_adapter", False))
    if lora:
        if not model_base:
            raise ValueError("lora=True needs model_base (the directory of the base model the adapter was trained on)")
        from .lora import read_adapter, merge_lora, strip_non_lora_prefixes
        base_config = set_sam(VideoReferQwen2Config.from_pretrained(model_base))
        for k, v in config.to_dict().items():       # the multimodal settings live in the adapter directory's config
            if k.startswith(("mm_", "image_aspect", "num_frames", "seg_token", "vision_config", "sam2_trunk")) and not hasattr(base_config, k):
                setattr(base_config, k, v)
        tokenizer = tok_from(model_path, use_fast=False)
        model = VideoReferQwen2ForCausalLM.from_pretrained(model_base, config=base_config, device=device, dtype=torch.bfloat16, allow_missing=MM)
        model.initialize_MM_tokenizer(tokenizer)
        nl = os.path.join(model_path, "non_lora_trainables.bin")
        if os.path.exists(nl):
            info = model.load_state_dict(strip_non_lora_prefixes(torch.load(nl, map_location="cpu")), strict=False)
            print(f"Load non_lora_trainables unexpected_keys: {info.unexpected_keys}")
        acfg, asd = read_adapter(model_path)
        with torch.no_grad():
            merged = merge_lora(dict(model.named_parameters()), acfg, asd)
        model.invalidate(); model.get_model().invalidate()
        print(f"Have merged pretrained LoRA weights ({len(merged)} tensors)")
    elif model_base is not None or is_pretraining:
        from .projector import load_mm_projector
        model_base = model_base if model_base is not None else getattr(config, "_name_or_path", None)
        if not model_base:
            raise ValueError("a pre-training checkpoint needs model_base (or config._name_or_path) to find the language model")
        tokenizer = tok_from(model_base, use_fast=False)
        model = VideoReferQwen2ForCausalLM.from_pretrained(model_base, config=config, device=device, dtype=torch.bfloat16, allow_missing=MM)
        info = model.load_state_dict({k: v for k, v in load_mm_projector(model_path).items()}, strict=False)
        if info.unexpected_keys:
            raise KeyError(f"{model_path}/mm_projector.bin: tensors with no counterpart in the model: {info.unexpected_keys[:6]}")
    else:
        tokenizer = tok_from(model_path, use_fast=False)
        model = VideoReferQwen2ForCausalLM.from_pretrained(model_path, config=config, device=device, dtype=torch.bfloat16)
    vision_tower = model.get_vision_tower()
    if not vision_tower.is_loaded:
        vision_tower.load_model(device=device)
    processor = vision_tower.image_processor
    context_len = getattr(model.config, "max_sequence_length", 2048)
    return tokenizer, model, processor, context_len
