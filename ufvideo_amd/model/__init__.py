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
    """-> (tokenizer, model, processor, context_len).  Local directories only (config.json +
    safetensors + tokenizer files); quantised / LoRA / hub loading are outside the hot path."""
    if load_8bit or load_4bit or lora or model_base is not None:
        raise NotImplementedError("8/4-bit, LoRA-merge and base-model loading are outside the accelerated hot path")
    if isinstance(device_map, dict) and "" in device_map:
        device = device_map[""]
    config = VideoReferQwen2Config.from_pretrained(model_path)
    config.train_mask_decoder = False
    config.sam_pretrained = "sam2-hiera-large/sam2_hiera_large.pt"
    config.sam_out_dim = 256
    tokenizer = None
    try:                                        # tokenisation is not on the hot path; any HF tokenizer works
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(model_path, use_fast=False, local_files_only=True)
    except Exception as ex:  # pragma: no cover
        raise RuntimeError(f"could not load a tokenizer from {model_path}: {ex}")
    model = VideoReferQwen2ForCausalLM.from_pretrained(model_path, config=config, device=device, dtype=torch.bfloat16)
    vision_tower = model.get_vision_tower()
    if not vision_tower.is_loaded:
        vision_tower.load_model(device=device)
    processor = vision_tower.image_processor
    context_len = getattr(model.config, "max_sequence_length", 2048)
    return tokenizer, model, processor, context_len
