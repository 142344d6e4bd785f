"""Multi-GPU helpers for the hot path (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

Two modes:
  * clip replicas  — the path shards by clip, no data-path collective (what the reference's eval does, and what
    bench.py --gpus N measures): `shard_clips`.
  * frame-sharded encoder for ONE clip — tower + projector are independent per aligned frame group of the Conv3d
    temporal stride (2 for stc_connector_v35), so each rank encodes a contiguous, stride-aligned chunk of frames and
    a single all-gather of the visual tokens (rank order = temporal order) feeds the decoder, which every rank then
    runs (it does not frame-shard; Amdahl cap 1.5x at 8 GPUs): `frame_chunks`, `encode_frame_sharded`.
"""
import torch


def shard_clips(n_clips, rank, world):
    """Contiguous rank chunks, same arithmetic as the reference's eval split (inference_PixRQA.py:38-45: ceil-sized chunks)."""
    import math
    size = math.ceil(n_clips / world)
    return list(range(n_clips))[rank * size:(rank + 1) * size]


def frame_chunks(num_frames, world, t_stride=2):
    """[(start, end)] per rank: contiguous, multiples of the temporal stride, as even as possible; ranks past the
    number of groups get empty chunks."""
    if num_frames % t_stride:
        raise ValueError(f"num_frames={num_frames} must be a multiple of the temporal stride {t_stride}")
    groups = num_frames // t_stride
    base, extra = divmod(groups, world)
    out, s = [], 0
    for r in range(world):
        g = base + (1 if r < extra else 0)
        out.append((s * t_stride, (s + g) * t_stride))
        s += g
    return out


def all_gather_tokens(local_tokens, counts, group=None):
    """Concatenate per-rank token blocks [n_r, D] in rank order.  Blocks may differ in length (uneven frame split):
    padded to the longest for the collective, then trimmed.  One all-gather, no other traffic."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n_max = max(counts)
    D = local_tokens.shape[-1]
    buf = local_tokens.new_zeros((n_max, D))
    buf[: local_tokens.shape[0]] = local_tokens
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)
    return torch.cat([g[:c] for g, c in zip(gathered, counts)], dim=0)


def encode_frame_sharded(model, video, group=None, encode_fn=None):
    """video [T,3,H,W] (identical on every rank) -> visual tokens [tokens, D] of the whole clip on every rank.
    `encode_fn(frames) -> [tokens_of_chunk, D]` defaults to the model's tower + projector."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    proj = model.get_model().mm_projector
    t_stride = getattr(proj, "downsample", (1, 1, 1))[0]
    chunks = frame_chunks(video.shape[0], world, t_stride)
    if encode_fn is None:
        def encode_fn(frames):
            feats = model.get_model().get_vision_tower().encode(frames)                 # [t, n, d]
            return model.temporal_aggregator(feats[None])[0]                              # [tokens, D]
    s, e = chunks[rank]
    # tokens per frame group are shape-determined; compute every rank's count without communication
    with torch.no_grad():
        local = encode_fn(video[s:e]) if e > s else None
    per_group = None
    if local is not None:
        per_group = local.shape[0] // ((e - s) // t_stride)
    # all ranks with a non-empty chunk agree on per_group; share it through the collective-free arithmetic below
    pg = torch.tensor([per_group or 0], device=video.device, dtype=torch.int64)
    dist.all_reduce(pg, op=dist.ReduceOp.MAX, group=group)
    per_group = int(pg.item())
    counts = [((ce - cs) // t_stride) * per_group for cs, ce in chunks]
    if local is None:
        D = model.config.hidden_size
        local = torch.zeros((0, D), device=video.device, dtype=torch.float32)
    return all_gather_tokens(local, counts, group)
