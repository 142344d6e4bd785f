"""Multi-GPU helpers for the hot path (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

Two modes:
  * clip replicas  — the path shards by clip, no data-path collective (what the reference's eval does, and what
    bench.py --gpus N measures): `shard_clips`.
  * frame-sharded encoder for ONE clip — tower + projector are independent per aligned frame group of the Conv3d
    temporal stride (2 for stc_connector_v35), so each rank encodes a contiguous, stride-aligned chunk of frames and
    ONE all-gather of the visual tokens (rank order = temporal order) feeds the decoder, which every rank then
    runs (it does not frame-shard; Amdahl cap 1.5x at 8 GPUs): `frame_chunks`, `encode_frame_sharded`.
No scaling curve has been measured on hardware yet (no multi-GPU node in any round so far): the exchange is covered by
world-size-2 gloo tests (tests/test_parallel_cpu.py), and the same calls run on RCCL with the one rank a 1-GPU box offers
(tests/test_parallel_gpu.py).
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
    """Concatenate per-rank token blocks [n_r, D] in rank order with ONE collective (`all_gather_into_tensor`): equal blocks
    (the production case: 32 or 64 frames over 1/2/4/8 ranks) land directly in the result; unequal blocks are padded to the
    longest for the collective and trimmed afterwards."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    assert len(counts) == world and local_tokens.shape[0] == counts[dist.get_rank(group)]
    if local_tokens.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no device all-gather: the exchange goes through the host (the one-GPU rehearsals of tests/test_parallel_gpu.py -- two ranks on one device, which RCCL
        # refuses; on a multi-GPU node the backend is "nccl" = RCCL and the tokens never leave HBM)
        return all_gather_tokens(local_tokens.cpu(), counts, group).to(local_tokens.device)
    n_max = max(counts)
    D = local_tokens.shape[-1]
    if min(counts) == n_max:
        out = local_tokens.new_empty((world * n_max, D))
        dist.all_gather_into_tensor(out, local_tokens.contiguous(), group=group)
        return out
    buf = local_tokens.new_zeros((n_max, D))
    buf[: local_tokens.shape[0]] = local_tokens
    gathered = local_tokens.new_empty((world * n_max, D))
    dist.all_gather_into_tensor(gathered, buf, group=group)
    return torch.cat([gathered[r * n_max: r * n_max + c] for r, c in enumerate(counts)], dim=0)


def check_frame_shardable(proj):
    """Frame sharding is only valid when the projector treats aligned groups of `t_stride` frames independently: RegStage and
    the readout are per frame / per token; the sampler must not look across the group boundary, i.e. Conv3d with padding 0
    (stc_connector_v35) or the AvgPool3d samplers with a temporal kernel > 1.  `stc_connector` / `spatial_conv` (Conv3d padding 1:
    windows straddle the shard boundary and zero frames are added at both ends) and the `mlpNx_gelu` / `linear` projectors
    (mean over ALL frames, videorefer_arch.py:203-205) would silently give different tokens: refuse."""
    pad = getattr(proj, "PADDING", None)
    avg = getattr(proj, "AVGPOOL", False)
    down = getattr(proj, "downsample", None)
    if down is None or not (avg or pad == 0):
        raise ValueError(f"{type(proj).__name__} cannot be frame-sharded: its output for a frame group depends on frames outside the group "
                         "(only Conv3d padding 0 / AvgPool3d samplers are independent per aligned group)")
    return down


def encode_frame_sharded(model, video, group=None, encode_fn=None):
    """video [T,3,H,W] (identical on every rank) -> visual tokens [tokens, D] of the whole clip on every rank.
    `encode_fn(frames) -> [tokens_of_chunk, D]` defaults to the model's tower + projector.  The only communication is the one
    all-gather: every rank derives every rank's token count from the geometry (frames per rank x tokens per frame group)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    proj = model.get_model().mm_projector
    down = check_frame_shardable(proj)
    t_stride = down[0]
    chunks = frame_chunks(video.shape[0], world, t_stride)
    tower = model.get_model().get_vision_tower()
    side = video.shape[-1] // tower.config.patch_size
    per_group = (side // down[1]) * (side // down[2])           # tokens one aligned frame group yields (no padding: floor)
    if encode_fn is None:
        def encode_fn(frames):
            feats = tower.encode(frames)                                                  # [t, n, d]
            return model.temporal_aggregator(feats[None])[0]                              # [tokens, D]
    s, e = chunks[rank]
    counts = [((ce - cs) // t_stride) * per_group for cs, ce in chunks]
    with torch.no_grad():
        if e > s:
            local = encode_fn(video[s:e])
            assert local.shape[0] == counts[rank], (local.shape, counts[rank])
        else:
            local = torch.zeros((0, model.config.hidden_size), device=video.device, dtype=torch.float32)
    return all_gather_tokens(local, counts, group)
