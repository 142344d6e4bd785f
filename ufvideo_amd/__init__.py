"""ufvideo_amd — MI355X-native hot path for UFVideo-style video LLMs.

Top-level API mirrors the reference's `ufvideo/__init__.py`: `model_init`, `mm_infer`."""
import copy

import torch

from .constants import NUM_FRAMES, DEFAULT_IMAGE_TOKEN, DEFAULT_VIDEO_TOKEN, MODAL_INDEX_MAP  # noqa: F401
from .mm_utils import (process_image, process_video, tokenizer_multimodal_token, get_model_name_from_path,  # noqa: F401
                       KeywordsStoppingCriteria)

__all__ = ["model_init", "mm_infer"]


def model_init(model_path=None, region=None, lora=False, args=None, **kwargs):
    """-> (model, processor, tokenizer)   (ref ufvideo/__init__.py:14-31)"""
    from .model import load_pretrained_model
    model_name = get_model_name_from_path(model_path)
    tokenizer, model, processor, context_len = load_pretrained_model(model_path, None, model_name, lora=lora, args=args, **kwargs)
    if tokenizer.pad_token is None and tokenizer.unk_token is not None:
        tokenizer.pad_token = tokenizer.unk_token
    return model, processor, tokenizer


def mm_infer(image_or_video, instruct, model, tokenizer, modal="video", masks=None, ann_indices=None, frame_nums=None,
             frame=None, choice=1, images_sam=None, offset=None, masks_list=None, label_list=None, seg=False, **kwargs):
    """Inference API (ref ufvideo/__init__.py:34-149): prompt build -> tokenise with the modality
    sentinel -> model.generate -> decode.  Returns (text, output_dict) or output_dict when seg=True."""
    if modal == "image":
        modal_token = DEFAULT_IMAGE_TOKEN
    elif modal == "video":
        modal_token = DEFAULT_VIDEO_TOKEN
    elif modal == "text":
        modal_token = ""
    else:
        raise ValueError(f"Unsupported modal: {modal}")
    dev = model.device
    tensor = None if modal == "text" else [(image_or_video.to(dev), modal)]

    if choice in (1, 2):
        if isinstance(instruct, str):
            content = (modal_token + "\n" + instruct) if choice == 1 else instruct
            message = [{"role": "user", "content": content}]
        elif isinstance(instruct, list):
            message = copy.deepcopy(instruct)
            message[0]["content"] = modal_token + "\n" + message[0]["content"]
        else:
            raise ValueError(f"Unsupported type of instruct: {type(instruct)}")
    elif choice == 3:
        message = [{"role": s["from"][0], "content": s["value"][0]} for s in instruct]
    else:
        raise ValueError(f"Unsupported choice: {choice}")

    prompt = tokenizer.apply_chat_template(message, tokenize=False, add_generation_prompt=True)
    ids_host = tokenizer_multimodal_token(prompt, tokenizer, modal_token, return_tensors="pt").unsqueeze(0).long()
    am_host = ids_host.ne(tokenizer.pad_token_id).long()
    input_ids, attention_masks = ids_host.to(dev), am_host.to(dev)      # the host copies go along: the splice plan is built from them, no device read-back
    stopping_criteria = KeywordsStoppingCriteria([tokenizer.eos_token], tokenizer, input_ids)

    do_sample = kwargs.get("do_sample", False)
    temperature = kwargs.get("temperature", 0.2 if do_sample else 0.0)
    top_p = kwargs.get("top_p", 0.9)
    max_new_tokens = kwargs.get("max_new_tokens", 2048)
    if frame is not None:
        frame = [frame.to(dev)]
    with torch.inference_mode():
        output_ids = model.generate(
            input_ids, attention_mask=attention_masks, images=tensor, do_sample=do_sample, temperature=temperature,
            max_new_tokens=max_new_tokens, top_p=top_p, use_cache=True, stopping_criteria=[stopping_criteria],
            pad_token_id=tokenizer.eos_token_id, masks=masks, ann_indices=ann_indices, frame_nums=frame_nums, frame=frame,
            images_sam=images_sam, offset=offset, masks_list=masks_list, label_list=label_list, video_file="",
            input_ids_host=ids_host, attention_mask_host=am_host)
    if not seg:
        return tokenizer.batch_decode(output_ids["output"], skip_special_tokens=True)[0].strip(), output_ids
    return output_ids
