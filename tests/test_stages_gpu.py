"""GPU: the whole-stage C entry points (SURVEY 8b minimum set: `ufv_vit_forward`, `ufv_qwen2_prefill`; csrc/stages.hip) against the same
op sequence issued from the host layer loops (UFV_STAGE_CALLS=0): bit-identical, at tiny and at full dimensions; their optional outputs
(`normed`, `logits_last`, `hidden_layers`) against the model's own forward; argument checks."""
import ctypes
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden, t  # noqa: E402
from ufvideo_amd import ops, _lib  # noqa: E402
from ufvideo_amd.model import KVCache  # noqa: E402
from test_model_gpu import tiny_model  # noqa: E402

DEV = "cuda"


class host_loops:
    """run the layer loops from Python (the op-level calls) instead of the stage calls"""

    def __enter__(self):
        self.old = os.environ.get("UFV_STAGE_CALLS")
        os.environ["UFV_STAGE_CALLS"] = "0"

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("UFV_STAGE_CALLS")
        else:
            os.environ["UFV_STAGE_CALLS"] = self.old


def test_tiny_model_forward_is_bit_identical_through_the_stage_calls():
    m, a, w = tiny_model()
    video = t(a["video"]).to(DEV)
    ids, am = t(a["sp_vid_only_ids"]).to(DEV), t(a["sp_vid_only_am_in"]).to(DEV)
    kw = dict(input_ids=ids, attention_mask=am, images=[(video, "video")], images_sam=torch.zeros(1, 1, 3, 8, 8), inference=True,
              output_hidden_states=True, use_cache=True)
    with torch.no_grad():
        r1 = m(**kw)
        with host_loops():
            r0 = m(**kw)
    assert torch.equal(r1["logits"], r0["logits"])
    assert len(r1["hidden_states"]) == len(r0["hidden_states"]) and all(torch.equal(x, y) for x, y in zip(r1["hidden_states"], r0["hidden_states"]))
    c1, c0 = r1["past_key_values"], r0["past_key_values"]
    assert c1.len == c0.len and all(torch.equal(x[:c1.len], y[:c0.len]) for x, y in zip(c1.buf, c0.buf))
    # the tower on its own, and a continuation of the cache (pos0 > 0) through the prefill call
    tower = m.get_vision_tower()
    with torch.no_grad():
        f1 = tower(video)
        with host_loops():
            f0 = tower(video)
        assert torch.equal(f1, f0)
        more = torch.randn(1, 5, m.config.hidden_size, device=DEV)
        am2 = torch.ones(1, c1.len + 5, dtype=torch.long, device=DEV)
        l1 = m._decode_batch(more, am2, c1, False, 1)[0]
        with host_loops():
            l0 = m._decode_batch(more, am2, c0, False, 1)[0]
    assert torch.equal(l1, l0) and c1.len == c0.len


def test_full_dimension_stages_bit_identical_and_direct_call_outputs():
    """SigLIP-so400m dims (1152, 16 x 72, 4304 -> 4352, 336^2 -> 576 patches), 2 layers, 4 frames; Qwen2-7B dims (3584, 28 / 4 x 128, 18944),
    2 layers, S = 2399: stage call == host loop bit for bit; then the prefill entry point called directly with all optional outputs."""
    import bench
    from ufvideo_amd.model import VideoReferQwen2Config, VideoReferQwen2ForCausalLM, QWEN2_7B
    dev = torch.device("cuda", 0)
    cfg = VideoReferQwen2Config(**dict(QWEN2_7B, num_hidden_layers=2), mm_vision_tower="siglip-so400m-patch14-384", mm_vision_select_layer=-2,
                                mm_vision_select_feature="patch", mm_projector_type="stc_connector_v35", mm_hidden_size=1152,
                                mm_region_encoder_type="pooling", image_aspect_ratio="square", train_mask_decoder=False, sam_pretrained=None,
                                sam_out_dim=256, num_frames=4, seg_token_id=151747, sam2_trunk=None, vision_config=dict(bench.VISION, num_hidden_layers=3))
    model = VideoReferQwen2ForCausalLM(cfg, device=dev, seed=0)
    model.get_vision_tower().load_model(device=dev, seed=7)
    g = torch.Generator().manual_seed(11)
    frames = torch.randn(4, 3, 336, 336, generator=g).to(dev).to(torch.bfloat16)
    tower = model.get_vision_tower()
    with torch.no_grad():
        f1 = tower.encode(frames)
        with host_loops():
            f0 = tower.encode(frames)
    assert f1.shape == (4, 576, 1152) and torch.equal(f1, f0)
    S, D, V = 2399, cfg.hidden_size, cfg.vocab_size
    emb = (torch.randn(1, S, D, generator=g) * 0.5).to(dev)
    with torch.no_grad():
        l1, c1, h1, n1 = model._decode_batch(emb, None, None, True, 1)
        with host_loops():
            l0, c0, h0, n0 = model._decode_batch(emb, None, None, True, 1)
    assert torch.equal(l1, l0) and torch.equal(n1, n0) and all(torch.equal(x, y) for x, y in zip(h1, h0))
    assert all(torch.equal(x[:S], y[:S]) for x, y in zip(c1.buf, c0.buf))
    # direct call: hidden_layers, normed (all rows) and logits_last from the entry point itself
    inner, head = model.get_model(), model.packed()
    cache = KVCache(cfg.num_hidden_layers, S + 8, 2 * cfg.num_key_value_heads * cfg.head_dim, dev)
    cm, keep = inner.c_model(cache, lm_head=head["lm_head"], vocab=V)
    nbytes = _lib.load().ufv_qwen2_prefill_ws_bytes(ctypes.byref(cm), S)
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    x = emb[0].clone()
    hid = torch.empty(cfg.num_hidden_layers - 1, S, D, device=dev)
    normed, logits = torch.empty(S, D, device=dev), torch.empty(V, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("ufv_qwen2_prefill", ctypes.byref(cm), x.data_ptr(), S, 0, ws.data_ptr(), nbytes, hid.data_ptr(), normed.data_ptr(), logits.data_ptr(), st)
    assert torch.equal(logits, l0.view(-1)) and torch.equal(normed, n0) and torch.equal(hid[0], h0[1]) and torch.equal(cache.buf[1][:S], c0.buf[1][:S])
    # logits without the all-rows norm: the same last row
    x2, logits2 = emb[0].clone(), torch.empty(V, device=dev)
    _lib.call("ufv_qwen2_prefill", ctypes.byref(cm), x2.data_ptr(), S, 0, ws.data_ptr(), nbytes, None, None, logits2.data_ptr(), st)
    assert torch.equal(logits2, logits) and torch.equal(x2, x)
    # argument checks: workspace, cache bounds, layer count of the tower
    with pytest.raises(_lib.UfvError, match="workspace too small"):
        _lib.call("ufv_qwen2_prefill", ctypes.byref(cm), x2.data_ptr(), S, 0, ws.data_ptr(), nbytes - 1, None, None, None, st)
    with pytest.raises(_lib.UfvError, match="outside the KV cache"):
        _lib.call("ufv_qwen2_prefill", ctypes.byref(cm), x2.data_ptr(), S, 16, ws.data_ptr(), nbytes, None, None, None, st)
    from ufvideo_amd.model.encoder import _vit_c_model
    vm, keep2 = _vit_c_model(tower.vision_tower)
    vb = _lib.load().ufv_vit_forward_ws_bytes(ctypes.byref(vm), 4)
    vws = torch.empty(vb, device=dev, dtype=torch.uint8)
    out = torch.empty(4 * 576, 1152, device=dev)
    with pytest.raises(_lib.UfvError, match="layers asked"):
        _lib.call("ufv_vit_forward", ctypes.byref(vm), frames.data_ptr(), ops._DT[frames.dtype], 4, 336, 336, 4, out.data_ptr(), vws.data_ptr(), vb, st)
    with pytest.raises(_lib.UfvError, match="patches"):
        _lib.call("ufv_vit_forward", ctypes.byref(vm), frames.data_ptr(), ops._DT[frames.dtype], 4, 224, 224, 2, out.data_ptr(), vws.data_ptr(), vb, st)
    _lib.call("ufv_vit_forward", ctypes.byref(vm), frames.data_ptr(), ops._DT[frames.dtype], 4, 336, 336, 2, out.data_ptr(), vws.data_ptr(), vb, st)
    assert torch.equal(out.view(4, 576, 1152), f0)                  # select_layer -2 of 3 layers = after 2 layers


def test_connector_stage_call_bit_identical_for_every_sampler_and_at_full_dimensions():
    """`ufv_stc_forward` == the host op loop of projector.py for: RegStage blocks with and without a shortcut conv (v35, depth 4), Conv3d padding 0
    and padding 1, the AvgPool3d samplers (odd sizes floor), depth 0; then stc_connector_v35 at production dimensions (1152 -> 3584, 4 frames 24x24)."""
    from oracle import ref_cpu as O
    from ufvideo_amd.model.projector import STCConnector, STCConnectorV35, STPConnector, SpatialConv, SpatialPool

    class Cfg:
        mm_hidden_size = 64
        hidden_size = 128

    class Cfg32:
        mm_hidden_size = 32
        hidden_size = 32
    g = torch.Generator().manual_seed(6)
    cases = [(STCConnectorV35(Cfg(), seed=3), torch.randn(2, 4, 36, 64, generator=g)), (STCConnector(Cfg(), seed=4), torch.randn(1, 4, 25, 64, generator=g)),
             (STPConnector(Cfg(), seed=5), torch.randn(1, 5, 49, 64, generator=g)), (SpatialConv(Cfg32(), seed=6), torch.randn(1, 3, 36, 32, generator=g)),
             (SpatialPool(Cfg32(), seed=7), torch.randn(1, 3, 25, 32, generator=g)), (STCConnectorV35(Cfg32(), depth=0, seed=8), torch.randn(1, 4, 16, 32, generator=g))]
    for m, x in cases:
        m = m.to(DEV)
        with torch.no_grad():
            y1 = m(x.to(DEV))
            with host_loops():
                y0 = m(x.to(DEV))
            yb = m(x.to(DEV).to(torch.bfloat16))                         # the tower hands bf16 / fp16 features
        assert y1.shape == y0.shape and torch.equal(y1, y0), type(m).__name__
        assert torch.isfinite(yb).all()
    # v35 against the oracle through the stage call (as test_model_gpu does through whichever path is active)
    sd = O.make_stc_weights(64, 128, seed=5)
    m = STCConnectorV35(Cfg()); m.load_state_dict(sd); m = m.to(DEV)
    x = torch.randn(1, 4, 36, 64, generator=torch.Generator().manual_seed(6))
    from conftest import rel_err
    assert rel_err(m(x.to(DEV)).cpu(), O.stc_connector(sd, x)) < 2e-2

    class Prod:
        mm_hidden_size = 1152
        hidden_size = 3584
    m = STCConnectorV35(Prod(), device=torch.device("cuda", 0), seed=9)
    x = torch.randn(1, 4, 576, 1152, generator=g).to(DEV)
    with torch.no_grad():
        y1 = m(x)
        with host_loops():
            y0 = m(x)
    assert y1.shape == (1, 2 * 12 * 12, 3584) and torch.equal(y1, y0)
    import ctypes
    cm, keep = m.c_model()
    nb = _lib.load().ufv_stc_forward_ws_bytes(ctypes.byref(cm), 4, 24)
    with pytest.raises(_lib.UfvError, match="workspace too small"):
        _lib.call("ufv_stc_forward", ctypes.byref(cm), x.data_ptr(), ops._DT[x.dtype], 4, 24, y1.data_ptr(), x.data_ptr(), nb - 1, torch.cuda.current_stream().cuda_stream)
