"""Host-side preprocessing helpers with the reference's call surface (ufvideo/mm_utils.py).

Integer/index helpers (frame_sample, tokenizer_multimodal_token, expand2square) are bit-exact
restatements; file decoding (decord / gif / moviepy) is outside the hot path and only supported for
inputs that are already frames (arrays, PIL images, image paths, frame directories).
The tensor-producing tail ("frame batching": resize -> 1/255 -> normalise -> NCHW) is
`UfvImageProcessor`; its arithmetic tail also exists as a HIP kernel (ops.preprocess_u8).
"""
import math
import os

import numpy as np
import torch
from PIL import Image

from .constants import NUM_FRAMES, NUM_FRAMES_PER_SECOND, MODAL_INDEX_MAP, DEFAULT_IMAGE_TOKEN


def _rle_counts_from_string(s):
    """COCO compressed-RLE string -> run lengths (the published LEB128-like coding of pycocotools' rleFrString: 5 data bits
    + continuation bit per character offset by 48, sign-extended, runs after the second stored as a difference to counts[i-2])."""
    if isinstance(s, bytes):
        s = s.decode("ascii")
    cnts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return cnts


def annToMask(mask_ann, h=None, w=None):
    """COCO segmentation -> uint8 [h, w] mask (ref mm_utils.py:22-33, which calls pycocotools).  RLE inputs (uncompressed
    counts list or compressed counts string) are decoded natively: runs alternate 0/1 starting with 0 in column-major order.
    Polygon lists need pycocotools' rasteriser and are delegated to it when it is installed."""
    if isinstance(mask_ann, list):
        try:
            from pycocotools import mask as maskUtils
        except ImportError as ex:
            raise NotImplementedError("polygon annotations need pycocotools (frPyObjects); RLE annotations are decoded natively") from ex
        return maskUtils.decode(maskUtils.merge(maskUtils.frPyObjects(mask_ann, h, w)))
    counts = mask_ann["counts"]
    hh, ww = mask_ann.get("size", (h, w))
    runs = list(counts) if isinstance(counts, (list, tuple)) else _rle_counts_from_string(counts)
    flat = np.zeros(hh * ww, dtype=np.uint8)
    pos, val = 0, 0
    for r in runs:
        if val:
            flat[pos:pos + r] = 1
        pos += r
        val ^= 1
    return flat.reshape(ww, hh).T.copy()                     # column-major (Fortran) order


def load_image_from_base64(image):
    import base64
    from io import BytesIO
    return Image.open(BytesIO(base64.b64decode(image)))


def create_photo_grid(arr, rows=None, cols=None):
    """Tile t frames [t, h, w, c] (ndarray, list of ndarrays or list of PIL images) row-major into one
    [rows*h, cols*w, c] contact sheet; missing grid dimensions default to a near-square layout and unused
    cells stay zero (same call contract as ref mm_utils.py:57-104)."""
    if isinstance(arr, list):
        if not isinstance(arr[0], (Image.Image, np.ndarray)):
            raise ValueError("Invalid input type. Expected list of Images or numpy arrays.")
        arr = np.stack([np.asarray(a) for a in arr])
    t, h, w, c = arr.shape
    if rows is None:
        rows = math.ceil(math.sqrt(t)) if cols is None else math.ceil(t / cols)
    if cols is None:
        cols = math.ceil(t / rows)
    if rows * cols < t:
        raise ValueError(f"Not enough grid cells ({rows}x{cols}) to hold all images ({t}).")
    cells = np.zeros((rows * cols, h, w, c), dtype=arr.dtype)
    cells[:t] = arr
    return cells.reshape(rows, cols, h, w, c).transpose(0, 2, 1, 3, 4).reshape(rows * h, cols * w, c)


def chunk_list(input_list, chunk_size):
    return [input_list[i:i + chunk_size] for i in range(0, len(input_list), chunk_size)]


def expand2square(pil_img, background_color):
    """Centre the image on a square canvas (ref mm_utils.py:43-54)."""
    w, h = pil_img.size
    if w == h:
        return pil_img
    side = max(w, h)
    canvas = Image.new(pil_img.mode, (side, side), background_color)
    canvas.paste(pil_img, (0, (w - h) // 2) if w > h else ((h - w) // 2, 0))
    return canvas


def frame_sample(duration, mode="uniform", num_frames=None, fps=None):
    """Frame indices to keep (ref mm_utils.py:135-158): segment midpoints rounded with a +1e-6 nudge
    for 'uniform'; one frame per second-long segment for 'fps'."""
    if mode == "uniform":
        assert num_frames is not None, "Number of frames must be provided for uniform sampling."
        seg = float(duration - 1) / num_frames
        mids = np.array([(seg * i + seg * (i + 1)) / 2 for i in range(num_frames)])
        return np.round(mids + 1e-6).astype(int)
    if mode == "fps":
        assert fps is not None, "FPS must be provided for FPS sampling."
        seg_len = min(fps // NUM_FRAMES_PER_SECOND, duration)
        return np.arange(seg_len // 2, duration, seg_len, dtype=int)
    raise ImportError(f"Unsupported frame sampling mode: {mode}")


def _pil_bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_resize_coeffs(in_size, out_size, precision_bits=22):
    """Coefficient tables of Pillow's 8-bit bicubic resample for one axis (the algorithm of Pillow's Resample.c:
    precompute_coeffs + normalize_coeffs_8bpc, in the same double-precision operation order).  -> (bounds int32 [out, 2] =
    (first input index, tap count), coeffs int32 [out, ksize], fixed point 2^22) for `ufv_resize_bicubic_u8`."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    one = float(1 << precision_bits)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_pil_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(0.5 + k * one) if k >= 0 else int(-0.5 + k * one)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


class UfvImageProcessor:
    """Stand-in for the HF SiglipImageProcessor the reference obtains from the tower
    (encoder.py:120): bicubic PIL resize -> x/255 -> (x-mean)/std -> float32 NCHW."""

    def __init__(self, size=384, image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5), resample=Image.BICUBIC):
        if isinstance(size, dict):
            size = size.get("height", size.get("shortest_edge"))
        self.size = {"height": int(size), "width": int(size)}
        self.image_mean = list(image_mean)
        self.image_std = list(image_std)
        self.resample = resample
        self.rescale_factor = 1 / 255

    def resize_u8(self, images):
        """list of PIL / HWC uint8 arrays -> uint8 [T, S, S, 3]"""
        S = self.size["height"]
        out = []
        for im in images:
            if not isinstance(im, Image.Image):
                im = Image.fromarray(np.asarray(im))
            im = im.convert("RGB")
            if im.size != (S, S):
                im = im.resize((S, S), resample=self.resample)
            out.append(np.asarray(im, dtype=np.uint8))
        return np.stack(out)

    def preprocess(self, images, return_tensors="pt", **kwargs):
        if isinstance(images, (Image.Image, np.ndarray)) and not (isinstance(images, np.ndarray) and images.ndim == 4):
            images = [images]
        u8 = self.resize_u8(list(images))
        x = u8.astype(np.float32) * np.float32(self.rescale_factor)
        mean = np.asarray(self.image_mean, dtype=np.float32)
        std = np.asarray(self.image_std, dtype=np.float32)
        x = ((x - mean) / std).transpose(0, 3, 1, 2)
        x = np.ascontiguousarray(x)
        return {"pixel_values": torch.from_numpy(x) if return_tensors == "pt" else x}

    __call__ = preprocess

    def preprocess_device(self, frames_u8, device="cuda"):
        """Device-side frame batching: uint8 HWC frames [T,H,W,3] (numpy or tensor, all one size) -> bf16 NCHW [T,3,S,S] on the
        GPU: Pillow-exact bicubic resize (`ufv_resize_bicubic_u8`) + normalise / layout (`ufv_preprocess_u8`).  The uint8 image
        after the resize is bit-identical to `resize_u8`; the final values are `preprocess`'s float32 values within bf16 rounding."""
        from . import ops
        t = torch.as_tensor(np.ascontiguousarray(frames_u8) if isinstance(frames_u8, np.ndarray) else frames_u8)
        t = t.to(device=device, dtype=torch.uint8).contiguous()
        S = self.size["height"]
        if tuple(t.shape[1:3]) != (S, S):
            t = ops.resize_bicubic_u8(t, S, S)
        return ops.preprocess_u8(t, self.image_mean, self.image_std)


def _to_pil_list(video_path, frame_idx):
    """The already-decoded input forms of process_video (ref mm_utils.py:230-267)."""
    if isinstance(video_path, np.ndarray):
        video = [Image.fromarray(f) for f in video_path]
        frames = [video[i] for i in frame_idx] if frame_idx is not None else None
    elif isinstance(video_path, list) and isinstance(video_path[0], np.ndarray):
        video = [Image.fromarray(f) for f in video_path]
        frames = [np.array(video[i]) for i in frame_idx] if frame_idx is not None else None
    elif isinstance(video_path, list) and isinstance(video_path[0], str):
        video = [Image.open(f) for f in video_path]
        frames = [np.array(video[i].convert("RGB")) for i in frame_idx] if frame_idx is not None else None
    elif isinstance(video_path, list) and isinstance(video_path[0], Image.Image):
        video = video_path
        frames = [np.array(video[i]) for i in frame_idx] if frame_idx is not None else None
    else:
        raise ValueError(f"Unsupported video path type: {type(video_path)}")
    return video, frames


def process_video(video_path, processor, s=None, e=None, aspect_ratio="pad", num_frames=NUM_FRAMES, frame_idx=None, device=None):
    """-> (video [T,3,S,S] f32, frame_data [n,3,S,S] | None, height, width, frames_list)  (ref mm_utils.py:161-295).
    `device` (not in the reference): resize + normalise on that GPU instead of PIL/numpy -> bf16 device tensors."""
    if isinstance(video_path, str):
        if s is not None and e is not None:
            s = max(s, 0.0); e = max(e, 0.0)
            if s > e:
                s, e = e, s
            elif s == e:
                e = s + 1
        if os.path.isdir(video_path):
            files = sorted(os.listdir(video_path))
            fps, total = 3, len(files)
            read = lambda i: Image.open(os.path.join(video_path, files[i]))
        else:
            try:
                from decord import VideoReader, cpu      # optional, outside the hot path
            except ImportError as ex:
                raise RuntimeError("decoding video files needs `decord`; pass frames (arrays / PIL / a frame "
                                   "directory) instead") from ex
            vr = VideoReader(video_path, ctx=cpu(0), num_threads=1)
            fps, total = vr.get_avg_fps(), len(vr)
            read = lambda i: Image.fromarray(vr[i].asnumpy())
        f_start = 0 if s is None else max(int(s * fps) - 1, 0)
        f_end = total - 1 if e is None else min(int(e * fps) - 1, total - 1)
        indices = list(range(f_start, f_end + 1))
        if num_frames is None:
            picked = [indices[i] for i in frame_sample(len(indices), mode="fps", fps=fps)]
        else:
            picked = [indices[i] for i in frame_sample(len(indices), mode="uniform", num_frames=num_frames)]
        video_data = [read(i) for i in picked]
        frame_data = [np.array(read(i).convert("RGB")) for i in frame_idx] if frame_idx is not None else None
    else:
        video_data, frame_data = _to_pil_list(video_path, frame_idx)

    video_data = list(video_data)
    while num_frames is not None and len(video_data) < num_frames:        # black padding frames
        video_data.append(Image.fromarray(np.zeros((*video_data[-1].size, 3), dtype=np.uint8)))
    frames_list = list(frame_data) if frame_data is not None else []
    video_data = video_data[:num_frames]
    height, width = np.array(video_data[0]).shape[:2]

    def prep(imgs):
        imgs = [Image.fromarray(f.numpy() if isinstance(f, torch.Tensor) else f) if not isinstance(f, Image.Image) else f
                for f in imgs]
        if aspect_ratio == "pad":
            bg = tuple(int(x * 255) for x in processor.image_mean)
            imgs = [expand2square(im, bg) for im in imgs]
        if device is not None and len({im.size for im in imgs}) == 1 and hasattr(processor, "preprocess_device"):
            # opt-in device-side frame batching: one H2D copy of the uint8 frames, Pillow-exact resize + normalise on the GPU
            return processor.preprocess_device(np.stack([np.asarray(im.convert("RGB"), dtype=np.uint8) for im in imgs]), device=device)
        return processor.preprocess(imgs, return_tensors="pt")["pixel_values"]

    video = prep(video_data)
    if frame_data is not None:
        frame_data = prep(frame_data)
    return video, frame_data, height, width, frames_list


def process_image(image_path, processor, aspect_ratio="pad", num_frames=NUM_FRAMES, image_grid=False):
    """-> (images, height, width, frame_list)  (ref mm_utils.py:107-131)"""
    image = Image.open(image_path).convert("RGB") if isinstance(image_path, str) else image_path.convert("RGB")
    if image_grid:
        pg = np.stack([np.array(image)] * num_frames)
        g = math.ceil(math.sqrt(num_frames))
        images = [create_photo_grid(pg, g, g), np.array(image)]
    else:
        images = [np.array(image)]
    frame_list = [images[0] for _ in range(4)]
    height, width = images[0].shape[:2]
    images = [Image.fromarray(f) for f in images]
    if aspect_ratio == "pad":
        images = [expand2square(im, tuple(int(x * 255) for x in processor.image_mean)) for im in images]
    images = processor.preprocess(images, return_tensors="pt")["pixel_values"]
    return images, height, width, frame_list


def tokenizer_multimodal_token(prompt, tokenizer, multimodal_token=DEFAULT_IMAGE_TOKEN, return_tensors=None):
    """Tokenise text chunks and put the modality sentinel id between them (ref mm_utils.py:381-406)."""
    sentinel = MODAL_INDEX_MAP.get(multimodal_token, None)
    if sentinel is None:
        input_ids = tokenizer(prompt, add_special_tokens=False).input_ids
    else:
        chunks = [tokenizer(c, add_special_tokens=False).input_ids for c in prompt.split(multimodal_token)]
        input_ids = []
        for i, c in enumerate(chunks):
            if i:
                input_ids.append(sentinel)
            input_ids.extend(c)
    if return_tensors is not None:
        if return_tensors == "pt":
            return torch.tensor(input_ids, dtype=torch.long)
        raise ValueError(f"Unsupported tensor type: {return_tensors}")
    return input_ids


def get_model_name_from_path(model_path):
    parts = model_path.strip("/").split("/")
    if parts[-1].startswith("checkpoint-"):
        return parts[-2] + "_" + parts[-1]
    return parts[-1]


class KeywordsStoppingCriteria:
    """Stop when the tail of the generated ids equals a keyword's ids or the decoded tail contains
    the keyword (ref mm_utils.py:418-449).  Duck-typed: callable(output_ids, scores) -> bool."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = keywords
        self.keyword_ids = []
        self.max_keyword_len = 0
        for kw in keywords:
            ids = tokenizer(kw).input_ids
            if len(ids) > 1 and ids[0] == tokenizer.bos_token_id:
                ids = ids[1:]
            self.max_keyword_len = max(self.max_keyword_len, len(ids))
            self.keyword_ids.append(torch.tensor(ids))
        self.tokenizer = tokenizer
        self.start_len = input_ids.shape[1]

    def call_for_batch(self, output_ids, scores, **kwargs):
        offset = min(output_ids.shape[1] - self.start_len, self.max_keyword_len)
        self.keyword_ids = [k.to(output_ids.device) for k in self.keyword_ids]
        for k in self.keyword_ids:
            if output_ids.shape[1] >= k.shape[0] and (output_ids[0, -k.shape[0]:] == k).all():
                return True
        text = self.tokenizer.batch_decode(output_ids[:, -offset:], skip_special_tokens=True)[0]
        return any(kw in text for kw in self.keywords)

    def __call__(self, output_ids, scores=None, **kwargs):
        return all(self.call_for_batch(output_ids[i].unsqueeze(0), scores) for i in range(output_ids.shape[0]))


class DirectResize:
    """Resize an HxWxC uint8 array to target x target, aspect ignored (ref mm_utils.py:452-461)."""

    def __init__(self, target_length):
        self.target_length = target_length

    def apply_image(self, image):
        return np.array(Image.fromarray(image, mode="RGB").resize((self.target_length, self.target_length)))


def sam_preprocess(x, pixel_mean=torch.Tensor([123.675, 116.28, 103.53]).view(-1, 1, 1),
                   pixel_std=torch.Tensor([58.395, 57.12, 57.375]).view(-1, 1, 1), img_size=1024):
    """Normalise a 0-255 CHW tensor for SAM2 (ref mm_utils.py:464-478; no padding)."""
    return (x - pixel_mean) / pixel_std
