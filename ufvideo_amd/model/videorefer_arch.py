"""Meta-model glue with the reference's interface (ufvideo/model/videorefer_arch.py): builds the
tower / projector / region encoder, encodes frames, and splices visual + region tokens into the
text embedding sequence.

The splice is split in two: `build_splice_plan` is pure host integer bookkeeping that reproduces
the reference loop (videorefer_arch.py:239-370) decision for decision — including its quirks — and
is tested bit-exactly on CPU; `execute` then fills the fp32 embedding buffer on the GPU with three
row-gather launches (text rows from embed_tokens, visual rows, region rows).
"""
from abc import ABC, abstractmethod

import os

import torch

from .. import ops
from ..constants import IGNORE_INDEX, MODAL_INDEX_MAP, NUM_FRAMES, TEMPORAL_TOKEN_FORMAT
from .encoder import build_vision_tower
from .layer import build_region_encoder
from .projector import build_vision_projector, load_mm_projector  # noqa: F401
from ._params import Holder, PackedModule, init_tensor, bf, f32


_UPLOAD_STREAMS = {}


def _upload_stream(dev):
    """one side stream per device for the splice's host-to-device index upload"""
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    if key not in _UPLOAD_STREAMS:
        _UPLOAD_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _UPLOAD_STREAMS[key]


def _upload(host_tensor, dev):
    """host tensor -> device through pinned memory on the side stream; the current stream waits on the copy's event (see the splice below)"""
    main = torch.cuda.current_stream(dev)
    side = _upload_stream(dev)
    pinned = host_tensor.pin_memory()
    with torch.cuda.stream(side):
        out = pinned.to(dev, non_blocking=True)
        done = torch.cuda.Event(); done.record(side)
    main.wait_event(done)
    out.record_stream(main)
    return out


# ------------------------------------------------------------------------------------------------
# splice plan (host, exact)
# ------------------------------------------------------------------------------------------------
class SplicePlan:
    """Per sample: list of segments
         ('text', a, b)      embed_tokens(input_ids[a:b])
         ('mm', k)           visual tokens of the k-th (image|video) in the batch
         ('region', off, n)  region tokens mask_feats[off:off+n]
       plus the reference's by-products."""

    def __init__(self):
        self.segments = []          # per sample
        self.lengths = []           # spliced length per sample
        self.mark = []              # mark_mm_token_indices (multimodal samples only)


def build_splice_plan(input_ids, mm_lens, region_token_nums, region_token_id, have_frame):
    """input_ids: list[list[int]]; mm_lens[k] = #visual tokens of the k-th modal input."""
    mm_ids = set(MODAL_INDEX_MAP.values())
    plan = SplicePlan()
    cur_mm = cur_ridx = cur_rpos = 0
    for ids in input_ids:
        L = len(ids)
        mm_pos = [i for i, t in enumerate(ids) if t in mm_ids]
        if not mm_pos:                                     # ref :251-267 (text-only sample still consumes counters)
            half = L // 2
            plan.segments.append([("text", 0, half), ("text", half, L)])
            plan.lengths.append(L)
            cur_mm += 1; cur_ridx += 1; cur_rpos += 1
            continue
        segs, base = [], 0
        for p in mm_pos:                                   # ref :276-289
            segs.append(("text", base, p))
            segs.append(("mm", cur_mm))
            cur_mm += 1
            base = p + 1
        if base < L:                                       # ref :291-318
            ridx = [i for i in range(base, L) if ids[i] == region_token_id]
            if not ridx:
                if have_frame:
                    segs.append(("region", cur_ridx, 0))
                cur_ridx += 1; cur_rpos += 1
            lo = base
            for r in ridx:
                segs.append(("text", lo, r))
                n = region_token_nums[cur_rpos]
                segs.append(("region", cur_ridx, n))
                cur_ridx += n; cur_rpos += 1
                lo = r + 1
            if lo < L:
                segs.append(("text", lo, L))

        def seg_len(s):
            return s[2] - s[1] if s[0] == "text" else (mm_lens[s[1]] if s[0] == "mm" else s[2])
        total = sum(seg_len(s) for s in segs)
        last = seg_len(segs[-1])
        plan.mark.append([total - last, last])
        plan.segments.append(segs)
        plan.lengths.append(total)
    return plan


def splice_labels_and_mask(plan, input_ids, attention_mask, labels, mm_lens, am_host=None):
    """Labels / attention mask exactly as ref :262-263,282-285,303-309,333-368 (torch CPU or GPU tensors).  `am_host`: the caller's mask as host lists
    (prepare_inputs_labels_for_multimodal has it already): the new mask is then assembled on the host and uploaded ONCE from pinned memory instead of being
    built on the device from 2 fills + a cat + a stack per sample (four tiny launches in the timed path)."""
    B = len(plan.segments)
    max_len = max(plan.lengths)
    uneven = any(l != plan.lengths[0] for l in plan.lengths)
    new_labels = None
    if labels is not None:
        rows = []
        for b, segs in enumerate(plan.segments):
            parts = []
            for s in segs:
                if s[0] == "text":
                    parts.append(labels[b, s[1]:s[2]])
                else:
                    n = mm_lens[s[1]] if s[0] == "mm" else s[2]
                    parts.append(torch.full((n,), IGNORE_INDEX, device=labels.device, dtype=labels.dtype))
            row = torch.cat(parts, 0)
            if uneven:
                row = torch.cat((row, torch.full((max_len - row.shape[0],), IGNORE_INDEX, device=labels.device, dtype=labels.dtype)), 0)
            rows.append(row)
        new_labels = torch.stack(rows, 0)
    if attention_mask is not None and am_host is not None and attention_mask.is_cuda:
        L_in = attention_mask.shape[1]
        rows = [[1] * (plan.lengths[b] - L_in) + [int(bool(v)) for v in am_host[b]] + [0] * (max_len - plan.lengths[b]) for b in range(B)]
        attention_mask = _upload(torch.tensor(rows, dtype=attention_mask.dtype), attention_mask.device)
    elif attention_mask is not None:
        L_in = attention_mask.shape[1]
        rows = []
        for b in range(B):
            left = torch.full((plan.lengths[b] - L_in,), True, dtype=attention_mask.dtype, device=attention_mask.device)
            right = torch.full((max_len - plan.lengths[b],), False, dtype=attention_mask.dtype, device=attention_mask.device)
            rows.append(torch.cat((left, attention_mask[b], right), 0))
        attention_mask = torch.stack(rows, 0)
    return new_labels, attention_mask


def splice_index_arrays(plan, input_ids, mm_lens, mm_offsets):
    """Flatten the plan into gather index lists (destination row = b*max_len + position)."""
    max_len = max(plan.lengths)
    t_src, t_dst, m_src, m_dst, r_src, r_dst = [], [], [], [], [], []
    for b, segs in enumerate(plan.segments):
        pos = b * max_len
        for s in segs:
            if s[0] == "text":
                n = s[2] - s[1]
                t_src.extend(input_ids[b][s[1]:s[2]]); t_dst.extend(range(pos, pos + n))
            elif s[0] == "mm":
                n = mm_lens[s[1]]
                m_src.extend(range(mm_offsets[s[1]], mm_offsets[s[1]] + n)); m_dst.extend(range(pos, pos + n))
            else:
                n = s[2]
                r_src.extend(range(s[1], s[1] + n)); r_dst.extend(range(pos, pos + n))
            pos += n
    return (t_src, t_dst), (m_src, m_dst), (r_src, r_dst)


# ------------------------------------------------------------------------------------------------
# model mixins
# ------------------------------------------------------------------------------------------------
class TextHiddenFcs(PackedModule):
    """model.text_hidden_fcs = ModuleList([Sequential(Linear, ReLU, Linear, Dropout(0))]) (ref :137-149);
    keys '0.0.weight', '0.2.weight', ..."""

    def __init__(self, in_dim, out_dim, device=None, dtype=torch.bfloat16, seed=4, std=0.02):
        super().__init__()
        gen = torch.Generator(device=device if device is not None else "cpu").manual_seed(seed)
        self.put("0.0.weight", init_tensor((in_dim, in_dim), "w", gen, std, device, dtype))
        self.put("0.0.bias", init_tensor((in_dim,), "zero", gen, std, device, dtype))
        self.put("0.2.weight", init_tensor((out_dim, in_dim), "w", gen, std, device, dtype))
        self.put("0.2.bias", init_tensor((out_dim,), "zero", gen, std, device, dtype))

    def _pack(self):
        g = self.get
        return (bf(g("0.0.weight")), f32(g("0.0.bias")), bf(g("0.2.weight")), f32(g("0.2.bias")))

    def __len__(self):
        return 1

    def __getitem__(self, i):
        assert i == 0
        return self

    def forward(self, x):
        w0, b0, w2, b2 = self.packed()
        shp = x.shape
        h = ops.convert(x.reshape(-1, shp[-1]).contiguous(), torch.bfloat16)
        h = ops.gemm(h, w0, bias=b0, act="relu")
        h = ops.gemm(h, w2, bias=b2, out_dtype=torch.float32)
        return h.view(*shp[:-1], -1)


class VideoReferMetaModel:
    """Mixin for the decoder body: owns vision_tower / mm_projector / region_encoder / text_hidden_fcs
    (ref VideoReferMetaModel :31-149)."""

    def init_mm_modules(self, config, device=None, dtype=torch.bfloat16, seed=0):
        if hasattr(config, "mm_vision_tower"):
            self.vision_tower = build_vision_tower(config, delay_load=True)
            self.mm_projector = build_vision_projector(config, device=device, dtype=dtype, seed=seed + 1)
            self.region_encoder = build_region_encoder(config, config.image_aspect_ratio, device=device, dtype=dtype, seed=seed + 3)
        if not getattr(config, "train_mask_decoder", False):
            self.initialize_sam_modules(config, device=device, dtype=dtype, seed=seed + 4)

    def get_vision_tower(self):
        vt = getattr(self, "vision_tower", None)
        if type(vt) is list:
            vt = vt[0]
        return vt

    def initialize_sam_modules(self, config, device=None, dtype=torch.bfloat16, seed=4):
        """ref :124-149.  `config.sam2_trunk`: "hiera_l" (the reference's SAM2-L, default), a dict of Hiera kwargs
        (+ optional image_size) for reduced test configurations, or None to leave the segmentation head out."""
        trunk = getattr(config, "sam2_trunk", "hiera_l")
        if trunk is None or trunk == "none":
            self.mask_encoder = None
        else:
            from .sam2 import SAM2, Hiera, FpnNeck, ImageEncoder, PositionEmbeddingSine
            if isinstance(trunk, dict):
                kw = dict(trunk)
                size = kw.pop("image_size", 1024)
                tr = Hiera(**kw, device=device, dtype=dtype, seed=seed + 10)
                neck = FpnNeck(PositionEmbeddingSine(256), d_model=256, backbone_channel_list=tr.channel_list, fpn_top_down_levels=[2, 3],
                               fpn_interp_model="nearest", device=device, dtype=dtype, seed=seed + 11)
                self.mask_encoder = SAM2(image_encoder=ImageEncoder(tr, neck, scalp=1), image_size=size, device=device, dtype=dtype,
                                         seed=seed + 12)
            else:
                self.mask_encoder = SAM2(device=device, dtype=dtype, seed=seed + 12)
        self.text_hidden_fcs = TextHiddenFcs(config.hidden_size, getattr(config, "sam_out_dim", 256), device=device,
                                             dtype=dtype, seed=seed)


class VideoReferMetaForCausalLM(ABC):

    @abstractmethod
    def get_model(self):
        pass

    def num_frames(self):
        return getattr(self.config, "num_frames", NUM_FRAMES)

    def get_vision_tower(self):
        return self.get_model().get_vision_tower()

    def encode_images_or_videos(self, images):
        """list[(tensor[T,3,H,W] | [1|T,3,H,W], 'video'|'image')] -> [B, tokens, hidden] fp32 (ref :168-191).
        An image is the frame expanded to num_frames copies (ref :173-174)."""
        num_frames = self.num_frames()
        batch = []
        for data, modal in images:
            batch.append(data.expand(num_frames, -1, -1, -1) if modal == "image" else data)
        batch = batch[0].unsqueeze(0) if len(batch) == 1 else torch.stack(batch, dim=0)          # (one clip: a view, not a 22 MB copy)
        assert len(batch.size()) == 5
        B, T = batch.shape[:2]
        feats = self.get_model().get_vision_tower().encode(batch.reshape(B * T, *batch.shape[2:]))   # [(b t), n, d] fp32
        return self.temporal_aggregator(feats.view(B, T, feats.shape[1], feats.shape[2]))

    def temporal_aggregator(self, frames_features):
        """[b, t, n, d] -> [b, tokens, hidden] (ref :193-216)."""
        kind = self.config.mm_projector_type
        proj = self.get_model().mm_projector
        if kind == "mlp2x_gelu" or kind == "linear":
            return proj(frames_features.mean(1))
        if kind in ("spatial_conv", "spatial_pool") or "tc_connector" in kind or "tp_connector" in kind:
            return proj(frames_features)
        raise Exception(f"Unsupported projector type {kind}!!!")

    @staticmethod
    def _host_lists(t, host, what):
        """The prompt as host lists for the splice plan.  `host` (extension, optional): the caller's own host copy of `t` -- a CPU tensor or nested lists, what a
        caller that tokenised on the host still has (mm_infer does: the ids are built on the CPU and then moved) -- used instead of reading the device tensor back,
        which is a device-to-host copy AND a full stream synchronisation per call.  No cache: rounds 3-4 remembered `t.tolist()` per tensor object and `_version`,
        which a write through `.data`, DLPack, shared numpy memory or this library's own raw-pointer kernels does not bump (a stale plan, silently), and which
        only ever hit for a loop that passes the same tensor again.  The host copy is the CALLER's statement of what the device tensor holds: its SHAPE is
        checked, its CONTENTS are trusted -- a host copy that no longer matches the device tensor (edited in place after `.to(dev)`, a reused buffer) gives a splice
        plan for the wrong prompt, silently.  UFV_CHECK_HOST_IDS=1 (debug) compares the two on every call (one synchronising read-back, what the host copy avoids)."""
        if host is None:
            return t.tolist()
        val = host.tolist() if torch.is_tensor(host) else [list(r) for r in host]
        if len(val) != t.shape[0] or any(len(r) != t.shape[1] for r in val):
            raise ValueError(f"{what}_host has shape {len(val)} x {len(val[0]) if val else 0}, the device tensor {tuple(t.shape)}")
        if os.environ.get("UFV_CHECK_HOST_IDS") == "1" and t.tolist() != val:
            raise ValueError(f"{what}_host does not hold what the device tensor holds (UFV_CHECK_HOST_IDS=1)")
        return val

    def prepare_inputs_labels_for_multimodal(self, input_ids, attention_mask, past_key_values, labels, images, masks, frame,
                                             ann_indices, frame_nums, video_file="", mm_features=None, region_stash=None,
                                             input_ids_host=None, attention_mask_host=None):
        """-> (None, attention_mask, past_key_values, inputs_embeds [B,S,D] fp32, labels, mark_mm_token_indices)
        (ref :218-370).  `mm_features` (extension): visual tokens computed elsewhere, e.g. by
        parallel.encode_frame_sharded, skip the local encode.  `input_ids_host` / `attention_mask_host` (extension): the caller's host copies of the
        two prompt tensors (CPU tensors or nested lists); without them the device tensors are read back -- one synchronising copy per call."""
        vision_tower = self.get_vision_tower()
        if vision_tower is None or images is None or input_ids.shape[1] == 1:
            return input_ids, attention_mask, past_key_values, None, labels, None
        model = self.get_model()
        # the ids come to the host BEFORE the encoder is queued (the copy synchronises: at this point the stream is idle); the
        # splice plan below is then built while the GPU is still busy with the tower, instead of stalling it after the encoder
        ids_host = self._host_lists(input_ids, input_ids_host, "input_ids")
        am_host = self._host_lists(attention_mask, attention_mask_host, "attention_mask") if attention_mask is not None else None
        if mm_features is None:
            mm_features = self.encode_images_or_videos(images)                   # [n_mm, tok, D] fp32
        if frame is not None:
            frame_cns = torch.cat(frame, dim=0)
            first = vision_tower.encode(frame_cns)
            mask_feats, region_token_nums = model.region_encoder(first, masks, mm_features, ann_indices, frame_nums, stash=region_stash)
        else:
            mask_feats, region_token_nums = None, []
        n_mm, tok = mm_features.shape[0], mm_features.shape[1]
        mm_lens = [tok] * n_mm
        region_id = self.tokenizer.convert_tokens_to_ids(["<region>"])[0]
        plan = build_splice_plan(ids_host, mm_lens, region_token_nums, region_id, frame is not None)
        new_labels, new_mask = splice_labels_and_mask(plan, input_ids, attention_mask, labels, mm_lens, am_host=am_host)
        # host-side lengths of the mask just built (left fill = True, then the caller's mask, then right padding): the decoder trims by
        # them without reading the device tensor back.  Only right-padded masks qualify; anything else is left to the decoder's own check.
        self._last_mask_info = None
        if new_mask is not None:
            L_in = len(am_host[0])
            valid = [plan.lengths[b] - L_in + sum(1 for v in am_host[b] if v) for b in range(len(am_host))]
            if all(all(bool(v) for v in am_host[b][:valid[b] - (plan.lengths[b] - L_in)]) for b in range(len(am_host))):
                self._last_mask_info = (new_mask, valid)
        (t_src, t_dst), (m_src, m_dst), (r_src, r_dst) = splice_index_arrays(plan, ids_host, mm_lens, [k * tok for k in range(n_mm)])
        dev = mm_features.device
        D = mm_features.shape[-1]
        B, S = len(plan.lengths), max(plan.lengths)
        embeds = (torch.zeros if B > 1 else torch.empty)((B * S, D), device=dev, dtype=torch.float32)     # padding rows exist only in a batch
        # The six index lists go up as ONE pinned buffer on a SIDE stream: the host is far ahead of the GPU here (the encoder is still running), so the
        # copy overlaps the tower and the main stream only waits on an event that has long fired.  Issued on the main stream each upload sat between two
        # kernels and cost a copy-engine hand-off (5-35 us of idle per copy in the rocprofv3 trace: 70 us per step).
        lists = [t_src, t_dst, m_src, m_dst, r_src, r_dst]
        idx = _upload(torch.tensor([v for l in lists for v in l], dtype=torch.int64), dev)
        offs = [0]
        for l in lists:
            offs.append(offs[-1] + len(l))
        ix = [idx[offs[i]:offs[i + 1]] for i in range(6)]
        if t_src:
            ops.gather_rows(model.embed_table(), ix[0], embeds, ix[1])
        if m_src:
            ops.gather_rows(mm_features.view(n_mm * tok, D), ix[2], embeds, ix[3])
        if r_src:
            ops.gather_rows(mask_feats, ix[4], embeds, ix[5])
        # vocabulary row of every spliced position that came from embed_tokens (-1 elsewhere): the scatter map of the
        # embedding gradient in ufvideo_amd.train
        eids = torch.full((B * S,), -1, dtype=torch.int64)
        if t_src:
            eids[torch.tensor(t_dst, dtype=torch.int64)] = torch.tensor(t_src, dtype=torch.int64)
        self._last_embed_ids = eids.view(B, S)
        self._last_mm_map = (list(m_src), list(m_dst), n_mm, tok)       # visual-token rows: mm_features row m_src -> spliced row m_dst
        self._last_region_map = (list(r_src), list(r_dst), 0 if mask_feats is None else mask_feats.shape[0])
        return None, new_mask, past_key_values, embeds.view(B, S, D), new_labels, plan.mark

    def initialize_MM_tokenizer(self, tokenizer):
        """adds '<region>', 100 x '<TEMP-%03d>', '[SEG]' and resizes the embeddings (ref :373-383)."""
        tokenizer.add_tokens("<region>", special_tokens=True)
        self.temporal_tokens = [TEMPORAL_TOKEN_FORMAT.format(i) for i in range(100)]
        tokenizer.add_tokens(self.temporal_tokens, special_tokens=True)
        tokenizer.add_tokens("[SEG]", special_tokens=True)
        self.resize_token_embeddings(len(tokenizer))
        for m in self.modules():
            m.tokenizer = tokenizer
