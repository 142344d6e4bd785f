"""GPU: the BENCH WORKLOAD ITSELF against the oracle, and every layer index at production dimensions teacher-forced.

(a) `test_bench_workload_stage_by_stage_vs_oracle`: the model and inputs of `bench.py` (bench.build_model / bench.synthetic_inputs: BASELINE config #2,
    UFVideo-7B dims, 32 frames 336x336 from default_rng(1234), prompt ids from default_rng(1235), S = 2399, DISTINCT seeded weights in every layer) run
    through the product path, and the same weights (streamed layer by layer from the GPU model's state_dict) through `oracle.ref_cpu` in fp32 and under
    `bf16_mirror()`: tower 26 layers -> STC-v35 -> splice -> the first 4 decoder layers -> final norm -> last-position logits.  UFV_PARITY_FULL=1 runs all
    28 decoder layers (fp32 leg only; +32 TFLOP of CPU work).  Reference: videorefer_arch.py:168-191, projector.py:189-238, videorefer_qwen2.py:357-459.
(b) `test_teacher_forced_every_tower_layer` / `..._decoder_layer`: for EVERY one of the 26 + 28 layer indices, the HIP layer's output against the mirror's
    layer applied to the HIP path's own layer input (no chain in front of the layer), at d 1152 / 16 x 72 / 4304 and 3584 / 28:4 x 128 / 18944.
Both write their figures into the table UFV_PARITY_REPORT_BENCH (or UFV_PARITY_REPORT) names (committed as profiles/<round>/parity_table_bench.json; the
UFV_PARITY_FULL=1 run as parity_table_bench_full.json)."""
import json
import os
import sys
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from conftest import rel_err  # noqa: E402
from oracle import ref_cpu as O  # noqa: E402
from ufvideo_amd import ops  # noqa: E402
from test_parity_bf16_gpu import l2_cos  # noqa: E402

DEV = "cuda"
REPORT = []
MEASURE = bool(os.environ.get("UFV_PARITY_MEASURE"))
HALF_TOL = 1e-3          # the north star's figure: every attention block and every MLP block of every layer, teacher-forced, against the bf16 mirror
VT = "model.vision_tower.vision_tower.vision_model."


def record(stage, got, mirror, ref32, **extra):
    g = got.float().cpu().reshape(mirror.shape)
    row = {"stage": stage, "vs_bf16_mirror": rel_err(g, mirror), "rel_l2_vs_mirror": l2_cos(g, mirror)[0]}
    if ref32 is not None:
        row.update(vs_fp32=rel_err(g, ref32), mirror_vs_fp32=rel_err(mirror, ref32), rel_l2_vs_fp32=l2_cos(g, ref32)[0],
                   rel_l2_mirror_vs_fp32=l2_cos(mirror, ref32)[0], one_minus_cos_vs_fp32=1.0 - l2_cos(g, ref32)[1])
    row.update(extra)
    REPORT.append(row)
    print("PARITY " + "  ".join(f"{k} {v:.2e}" if isinstance(v, float) else f"{k} {v}" for k, v in row.items()), flush=True)
    return row


def layer_sd(model_sd, prefix):
    """one layer's (or module's) weights from the GPU model, as fp32 CPU tensors (the parameters are bf16: fp32 and mirror legs see the same values)"""
    return {k: v.detach().float().cpu() for k, v in model_sd.items() if k.startswith(prefix)}


@pytest.fixture(scope="module")
def bench_model():
    import bench
    torch.manual_seed(0)
    model = bench.build_model(torch.device(DEV))
    video, ids, am = bench.synthetic_inputs(torch.device(DEV))
    return bench, model, video, ids, am


def test_bench_workload_stage_by_stage_vs_oracle(bench_model):
    bench, model, video, ids, am = bench_model
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    msd = model.state_dict()
    n_dec = 28 if os.environ.get("UFV_PARITY_FULL") else 4
    t0 = time.time()
    with torch.no_grad():
        # ---- product path ------------------------------------------------------------------------------------------------------------
        tower = model.get_vision_tower()
        feats = tower.encode(video)                                            # fp32 [32, 576, 1152]
        mmf = model.encode_images_or_videos([(video, "video")])                # [1, 2304, 3584]
        _, am2, _, emb, _, mark = model.prepare_inputs_labels_for_multimodal(ids, am, None, None, [(video, "video")], None, None, None, None)
        S = emb.shape[1]
        assert S == 2399 and tuple(mmf.shape) == (1, 2304, 3584)
        logits28, cache, hs, normed28 = model._decode_batch(emb, am2, None, True, 1)
        pk = model.packed()
        x4 = hs[n_dec] if n_dec < 28 else None
        if x4 is not None:
            x4 = x4[0] if x4.dim() == 3 else x4
            normed4 = model.model.final_norm(x4)
            logits4 = ops.gemm(ops.convert(normed4[-1:].contiguous(), torch.bfloat16), pk["lm_head"], out_dtype=torch.float32).view(-1)
        else:
            normed4, logits4 = normed28, logits28.view(-1)
        torch.cuda.synchronize()
        # ---- oracle, fp32 and mirror legs in lock step (one layer's weights on the CPU at a time) -------------------------------------
        vcfg = dict(bench.VISION)
        px = video.float().cpu()
        esd = layer_sd(msd, VT + "embeddings.")
        with O.bf16_mirror():
            hm = O.siglip_embeddings(esd, VT, px, 14)
        h32 = O.siglip_embeddings(esd, VT, px, 14)
        for i in range(26):
            p = f"{VT}encoder.layers.{i}."
            lsd = layer_sd(msd, p)
            with O.bf16_mirror():
                hm = O.vit_encoder_layer(lsd, p, hm, 16, 1e-6, "gelu_pytorch_tanh")
            h32 = O.vit_encoder_layer(lsd, p, h32, 16, 1e-6, "gelu_pytorch_tanh")
        r_t = record("bench workload: tower 26 L (32 f x 576 x 1152)", feats, hm, h32, seconds=round(time.time() - t0, 1))
        psd = layer_sd(msd, "model.mm_projector.")
        # the product feeds the connector its own tower output; the oracle legs carry their own chains
        with O.bf16_mirror():
            mm_m = O.stc_connector(psd, hm[None], prefix="model.mm_projector.")
        mm_32 = O.stc_connector(psd, h32[None], prefix="model.mm_projector.")
        r_p = record("bench workload: tower + STC-v35 (2304 x 3584)", mmf, mm_m, mm_32, seconds=round(time.time() - t0, 1))
        # teacher-forced connector: the oracle on the PRODUCT's tower output
        with O.bf16_mirror():
            mm_tf = O.stc_connector(psd, feats.float().cpu()[None], prefix="model.mm_projector.")
        r_ptf = record("bench workload: STC-v35 teacher-forced on the HIP tower output", mmf, mm_tf, None)
        del psd
        # ---- splice: index-exact, embeddings through a compact table of the 95 text ids ------------------------------------------------
        ids_c = ids.cpu()
        uniq = sorted(set(int(v) for v in ids_c.reshape(-1).tolist() if v >= 0))
        remap = {v: i for i, v in enumerate(uniq)}
        ids_small = torch.tensor([[remap[int(v)] if v >= 0 else int(v) for v in ids_c[0].tolist()]])
        table = msd["model.embed_tokens.weight"][torch.tensor(uniq, device=DEV)].float().cpu()
        with O.bf16_mirror():
            am_o, emb_m, _, mark_o = O.splice(table, ids_small, torch.ones_like(ids_small), None, mm_m, [], [], 10 ** 9, False)
        _, emb_32, _, _ = O.splice(table, ids_small, torch.ones_like(ids_small), None, mm_32, [], [], 10 ** 9, False)
        assert mark_o == mark and torch.equal(am_o, am2.cpu()), "splice bookkeeping (mask / mark indices) must be bit-exact"
        # text rows are copies of bf16 table rows: bit-exact; visual rows are the connector's
        txt = [i for i in range(S) if not (bench.VIDEO_POS <= i < bench.VIDEO_POS + 2304)]
        assert torch.equal(emb[0].float().cpu()[txt], emb_32[0][txt]), "text rows of inputs_embeds are table rows: exact"
        record("bench workload: spliced inputs_embeds (2399 x 3584)", emb, emb_m, emb_32)
        # ---- decoder: n_dec distinct layers ------------------------------------------------------------------------------------------------
        lcfg = dict(hidden_size=3584, num_attention_heads=28, num_key_value_heads=4, rope_theta=1e6, rms_norm_eps=1e-6)
        cos, sin = O.rope_cos_sin(torch.arange(S), 128, 1e6)
        bias = torch.zeros(S, S).masked_fill(torch.arange(S)[None, :] > torch.arange(S)[:, None], torch.finfo(torch.float32).min)[None, None]
        xm, x32 = emb_m.clone(), emb_32.clone()
        do_mirror = n_dec <= 4 or os.environ.get("UFV_PARITY_FULL") == "mirror"        # UFV_PARITY_FULL=mirror: the 28-layer chain with BOTH legs (+ ~3 min of host time)
        for i in range(n_dec):
            p = f"model.layers.{i}."
            lsd = layer_sd(msd, p)
            if do_mirror:
                with O.bf16_mirror():
                    xm, _ = O.qwen2_layer(lsd, p, xm, lcfg, cos, sin, None, bias)
            x32, _ = O.qwen2_layer(lsd, p, x32, lcfg, cos, sin, None, bias)
            if i + 1 < 28 and (i + 1) in (1, 2, 4, 14):
                hi = hs[i + 1]
                record(f"bench workload: decoder stream after {i + 1} layers", hi[0] if hi.dim() == 3 else hi, xm[0] if do_mirror else x32[0], x32[0],
                       seconds=round(time.time() - t0, 1))
        nw = msd["model.norm.weight"].float().cpu()
        with O.bf16_mirror():
            nm = O.rmsnorm(xm[0], O._rb(nw), 1e-6) if do_mirror else None
        n32 = O.rmsnorm(x32[0], nw, 1e-6)
        r_n = record(f"bench workload: final norm after {n_dec} decoder layers (2399 x 3584)", normed4, nm if do_mirror else n32, n32)
        # last-position logits: the 151748 x 3584 lm_head in row chunks
        head = msd["lm_head.weight"]
        lg_m, lg_32 = [], []
        for r0 in range(0, head.shape[0], 16384):
            hw = head[r0:r0 + 16384].float().cpu()
            if do_mirror:
                lg_m.append(nm[-1:].to(torch.bfloat16).float() @ hw.T)
            lg_32.append(n32[-1:] @ hw.T)
        lg_32 = torch.cat(lg_32, 1).view(-1)
        lg_m = torch.cat(lg_m, 1).view(-1) if do_mirror else lg_32
        r_l = record(f"bench workload: last-position logits after {n_dec} decoder layers (151748)", logits4, lg_m, lg_32, seconds=round(time.time() - t0, 1))
        g = logits4.float().cpu()
        top2 = lg_32.topk(2).values
        margin, err = float(top2[0] - top2[1]), float((g - lg_32).abs().max())
        t5 = len(set(g.topk(5).indices.tolist()) & set(lg_32.topk(5).indices.tolist()))
        REPORT.append({"stage": "bench workload: next token", "argmax_equal": bool(g.argmax() == lg_32.argmax()), "fp32_top1_margin": margin, "max_logit_err": err,
                       "top5_overlap": t5, "decoder_layers": n_dec})
        print(f"PARITY next token: arg-max equal {bool(g.argmax() == lg_32.argmax())}, fp32 margin {margin:.3e}, max logit error {err:.3e}, top-5 overlap {t5}")
    if MEASURE:
        return
    # bounds: 2 x what MI355X measured (profiles/r04/parity_table_bench.json) and never further from fp32 than the mirror (x 1.5 + 1e-3)
    assert r_t["rel_l2_vs_mirror"] <= 9e-3 and r_t["rel_l2_vs_fp32"] <= 7.6e-3, r_t                 # measured 4.4e-3 / 3.8e-3 (mirror vs fp32 3.8e-3)
    assert r_t["vs_fp32"] <= 1.5 * r_t["mirror_vs_fp32"] + 1e-3, r_t                                   # max norm 4.1e-3 vs 4.0e-3
    assert r_p["rel_l2_vs_mirror"] <= 2.5e-2 and r_p["rel_l2_vs_fp32"] <= 2e-2, r_p                 # 1.3e-2 / 1.0e-2 (1.0e-2)
    assert r_p["vs_fp32"] <= 1.5 * r_p["mirror_vs_fp32"] + 1e-3, r_p
    assert r_ptf["vs_bf16_mirror"] <= 2.2e-2, r_ptf                                                    # 1.1e-2 (48 storage points, the connector's noise floor)
    if do_mirror:
        if n_dec <= 4:
            assert r_n["rel_l2_vs_fp32"] <= 2.8e-2 and r_l["rel_l2_vs_fp32"] <= 1.2e-2, (r_n, r_l)  # 1.4e-2 (1.35e-2) / 6.0e-3 (5.8e-3)
        else:
            assert r_n["rel_l2_vs_fp32"] <= 5e-2 and r_l["rel_l2_vs_fp32"] <= 2.3e-2, (r_n, r_l)    # 28 layers: 2.47e-2 / 1.08e-2
        assert r_n["vs_fp32"] <= 1.5 * r_n["mirror_vs_fp32"] + 1e-3 and r_l["vs_fp32"] <= 1.5 * r_l["mirror_vs_fp32"] + 1e-3, (r_n, r_l)
    else:                                                                                              # UFV_PARITY_FULL=1: 28 layers, fp32 leg only (no mirror to bound against)
        assert r_n["rel_l2_vs_fp32"] <= 5e-2 and r_l["rel_l2_vs_fp32"] <= 2.3e-2, (r_n, r_l)        # measured 2.47e-2 / 1.13e-2 (profiles/r05/parity_table_bench_full.json)
    if margin > 3.0 * err:                                                                             # measured: margin 0.150, error 0.031 -> decided, equal
        assert bool(g.argmax() == lg_32.argmax())
    assert t5 >= 4                                                                                     # 5 of 5


def test_teacher_forced_every_tower_layer(bench_model):
    """26 layer indices, d 1152 / 16 heads x 72 / d_ff 4304, 2 frames of the bench clip: layer i of the HIP tower on the stream the HIP tower itself produced
    after i layers, against the mirror's layer i on the same stream."""
    bench, model, video, ids, am = bench_model
    body = model.get_vision_tower().vision_tower
    msd = model.state_dict()
    px = video[:2].contiguous()
    worst = worst_half = 0.0
    pk = body.packed()
    M, D = 2 * 576, 1152
    h = torch.empty((M, D), device=DEV, dtype=torch.bfloat16)
    qkv = torch.empty((M, 3 * D), device=DEV, dtype=torch.bfloat16)
    o = torch.empty((M, D), device=DEV, dtype=torch.bfloat16)
    ff = torch.empty((M, pk["Ip"]), device=DEV, dtype=torch.bfloat16)
    st = (576 * 3 * D, 3 * D)
    with torch.no_grad():
        prev = body.encode(px, 0)[0].view(2, 576, 1152).clone()
        for i in range(26):
            cur = body.encode(px, i + 1)[0].view(2, 576, 1152).clone()
            p = f"{VT}encoder.layers.{i}."
            lsd = layer_sd(msd, p)
            xin = prev.cpu()
            with O.bf16_mirror():
                ym = O.vit_encoder_layer(lsd, p, xin, 16, 1e-6, "gelu_pytorch_tanh")
            y32 = O.vit_encoder_layer(lsd, p, xin, 16, 1e-6, "gelu_pytorch_tanh")
            g = cur.cpu()
            # also relative to what the layer ADDS to the stream (the stream's own magnitude grows with depth and flatters the plain measure)
            dm = float(((g - xin) - (ym - xin)).abs().max() / (ym - xin).abs().max())
            r = record(f"ViT layer {i} teacher-forced (d 1152, hd 72)", cur, ym, y32, vs_mirror_rel_to_layer_delta=dm)
            worst = max(worst, r["vs_bf16_mirror"])
            # the two halves of the layer on the HIP path's own inputs (the op sequence of ufv_vit_forward issued from here): attention block
            # LayerNorm -> QKV -> hd-72 flash attention -> out_proj + residual, MLP block LayerNorm -> fc1 + GELU -> fc2 + residual
            L = pk["layers"][i]
            x = prev.reshape(M, D).clone().contiguous()
            ops.gemm(ops.layernorm(x, L["ln1"][0], L["ln1"][1], 1e-6, out=h), L["wqkv"], bias=L["bqkv"], out=qkv)
            ops.attention(qkv, qkv[:, D:], qkv[:, 2 * D:], 2, 16, 16, 576, 576, 72, st, st, st, out=o)
            ops.gemm(o, L["wo"], bias=L["bo"], resid=x, out=x)
            x_mid = x.clone()
            ops.gemm(ops.layernorm(x, L["ln2"][0], L["ln2"][1], 1e-6, out=h), L["w1"], bias=L["b1"], act="gelu_pytorch_tanh", out=ff)
            ops.gemm(ff, L["w2"], bias=L["b2"], resid=x, out=x)
            assert torch.equal(x.view(2, 576, D), cur), f"ViT layer {i}: the op-level sequence and the stage call differ"
            sd_att = dict(lsd); sd_mlp = dict(lsd)
            for k in ("mlp.fc2.weight", "mlp.fc2.bias"):
                sd_att[p + k] = torch.zeros_like(lsd[p + k])
            for k in ("self_attn.out_proj.weight", "self_attn.out_proj.bias"):
                sd_mlp[p + k] = torch.zeros_like(lsd[p + k])
            with O.bf16_mirror():
                am_ = O.vit_encoder_layer(sd_att, p, xin, 16, 1e-6, "gelu_pytorch_tanh")
                mm_ = O.vit_encoder_layer(sd_mlp, p, x_mid.view(2, 576, D).cpu(), 16, 1e-6, "gelu_pytorch_tanh")
            ra = record(f"ViT layer {i}: attention block teacher-forced", x_mid, am_, None)
            rm = record(f"ViT layer {i}: MLP block teacher-forced", x, mm_, None)
            worst_half = max(worst_half, ra["vs_bf16_mirror"], rm["vs_bf16_mirror"])
            if not MEASURE:
                assert r["vs_bf16_mirror"] <= 2e-3, r
                assert ra["vs_bf16_mirror"] <= HALF_TOL and rm["vs_bf16_mirror"] <= HALF_TOL, (ra, rm)
                assert r["vs_fp32"] <= 1.5 * r["mirror_vs_fp32"] + 1e-3, r
            prev = cur
    REPORT.append({"stage": "ViT layers 0..25 teacher-forced: worst vs mirror", "vs_bf16_mirror": worst, "worst_half_layer_vs_mirror": worst_half})


def test_teacher_forced_every_decoder_layer(bench_model):
    """28 layer indices, d 3584 / 28:4 heads x 128 / d_ff 18944, S = 383 (4 frames' 288 visual tokens + 95 text): the stream after every layer of the HIP
    prefill (output_hidden_states) against the mirror's layer i applied to the HIP stream after i layers; layer 27 through the final norm."""
    bench, model, video, ids, am = bench_model
    msd = model.state_dict()
    lcfg = dict(hidden_size=3584, num_attention_heads=28, num_key_value_heads=4, rope_theta=1e6, rms_norm_eps=1e-6)
    S = 383
    xe = (torch.randn(1, S, 3584, generator=torch.Generator().manual_seed(64)) * 0.5).to(DEV)
    cos, sin = O.rope_cos_sin(torch.arange(S), 128, 1e6)
    bias = torch.zeros(S, S).masked_fill(torch.arange(S)[None, :] > torch.arange(S)[:, None], torch.finfo(torch.float32).min)[None, None]
    worst = worst_half = 0.0
    H, KV, hd = 28, 4, 128
    pk = model.model.packed()
    h = torch.empty((S, 3584), device=DEV, dtype=torch.bfloat16)
    qkv = torch.empty((S, (H + 2 * KV) * hd), device=DEV, dtype=torch.bfloat16)
    o = torch.empty((S, H * hd), device=DEV, dtype=torch.bfloat16)
    act = torch.empty((S, 18944), device=DEV, dtype=torch.bfloat16)
    tab = ops.rope_table(pk["inv_freq"], 0, S, hd)
    with torch.no_grad():
        logits, cache, hs, normed = model._decode_batch(xe, None, None, True, 1)
        hs = [h[0] if h.dim() == 3 else h for h in hs]
        # the un-normed stream after the last layer is not part of HF's hidden_states: run 28 layers again without the final norm
        x_last = model.model.run_layers(xe[0].clone(), type(cache)(28, 512, cache.buf[0].shape[1], DEV), 0)
        streams = hs[:28] + [x_last]
        nw = msd["model.norm.weight"].float().cpu()
        for i in range(28):
            p = f"model.layers.{i}."
            lsd = layer_sd(msd, p)
            xin = streams[i].float().cpu()[None]
            with O.bf16_mirror():
                ym, _ = O.qwen2_layer(lsd, p, xin, lcfg, cos, sin, None, bias)
            y32, _ = O.qwen2_layer(lsd, p, xin, lcfg, cos, sin, None, bias)
            g = streams[i + 1].float().cpu()
            dm = float(((g - xin[0]) - (ym[0] - xin[0])).abs().max() / (ym[0] - xin[0]).abs().max())
            r = record(f"decoder layer {i} teacher-forced (d 3584, S 383)", streams[i + 1], ym[0], y32[0], vs_mirror_rel_to_layer_delta=dm)
            worst = max(worst, r["vs_bf16_mirror"])
            # the two halves of the layer, each on the HIP path's own input: attention block (RMSNorm -> QKV -> RoPE -> causal GQA flash attention -> o_proj +
            # residual) and MLP block (RMSNorm -> gate/up + SwiGLU -> down + residual), the op sequence of csrc/stages.hip issued from here; the oracle's
            # half = its layer with the other half's output projection zeroed
            L = pk["layers"][i]
            x = streams[i].float().clone().contiguous()
            kvb = torch.zeros((512, 2 * KV * hd), device=DEV, dtype=torch.bfloat16)
            ops.gemm(ops.rmsnorm(x, L["ln1"], 1e-6, out=h), L["wqkv"], bias=L["bqkv"], out=qkv)
            ops.rope_kv(qkv, S, H, KV, hd, pk["inv_freq"], 0, kvb, table=tab)
            ops.attention(qkv, kvb, kvb[:, KV * hd:], 1, H, KV, S, S, hd, (0, qkv.stride(0)), (0, kvb.stride(0)), (0, kvb.stride(0)), causal=True, q_pos0=0, out=o)
            ops.gemm(o, L["wo"], resid=x, out=x)
            x_mid = x.clone()
            ops.gemm(ops.rmsnorm(x, L["ln2"], 1e-6, out=h), L["wgu"], swiglu=True, out=act)
            ops.gemm(act, L["wd"], resid=x, out=x)
            assert torch.equal(x, streams[i + 1].float()), f"layer {i}: the op-level sequence and the stage call differ"
            sd_att = dict(lsd); sd_att[p + "mlp.down_proj.weight"] = torch.zeros_like(lsd[p + "mlp.down_proj.weight"])
            sd_mlp = dict(lsd); sd_mlp[p + "self_attn.o_proj.weight"] = torch.zeros_like(lsd[p + "self_attn.o_proj.weight"])
            xm_in = x_mid.cpu()[None]
            with O.bf16_mirror():
                am_, _ = O.qwen2_layer(sd_att, p, xin, lcfg, cos, sin, None, bias)
                mm_, _ = O.qwen2_layer(sd_mlp, p, xm_in, lcfg, cos, sin, None, bias)
            ra = record(f"decoder layer {i}: attention block teacher-forced", x_mid, am_[0], None)
            rm = record(f"decoder layer {i}: MLP block teacher-forced", x, mm_[0], None)
            worst_half = max(worst_half, ra["vs_bf16_mirror"], rm["vs_bf16_mirror"])
            if not MEASURE:
                assert r["vs_bf16_mirror"] <= (4e-3 if i == 0 else 2e-3), r          # layer 0 (a 0.5-sigma random stream, the layer adds as much again): 3.05e-3 = the
                assert ra["vs_bf16_mirror"] <= HALF_TOL and rm["vs_bf16_mirror"] <= HALF_TOL, (ra, rm)   # chain noise floor of one layer (PARITY.md); its halves hold 1e-3 (measured <= 8.0e-4)
                assert r["vs_fp32"] <= 1.5 * r["mirror_vs_fp32"] + 1e-3, r
        with O.bf16_mirror():
            nm = O.rmsnorm(x_last.float().cpu(), O._rb(nw), 1e-6)
        r = record("decoder final norm teacher-forced", normed, nm, O.rmsnorm(x_last.float().cpu(), nw, 1e-6))
        if not MEASURE:
            assert r["vs_bf16_mirror"] <= 1e-3, r
    REPORT.append({"stage": "decoder layers 0..27 teacher-forced: worst vs mirror", "vs_bf16_mirror": worst, "worst_half_layer_vs_mirror": worst_half})


def test_zz_write_report():
    path = os.environ.get("UFV_PARITY_REPORT_BENCH") or os.environ.get("UFV_PARITY_REPORT")
    if path and REPORT:
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        with open(path, "w") as f:
            json.dump(REPORT, f, indent=1)
