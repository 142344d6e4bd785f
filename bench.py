#!/usr/bin/env python
"""bench.py — video-tokens/s (encode + project + splice + LLM prefill) on synthetic UFVideo-7B clips.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run with one rank per GPU.  One JSON line on rank 0.

Workload = BASELINE.json configs[1]: UFVideo-7B dims (SigLIP-so400m/14 tower at 336 px, 26 of 27 layers;
stc_connector_v35; Qwen2-7B-dim decoder, vocab 151748), one 32-frame 336x336 clip per step per GPU,
96-id prompt with one <video> sentinel -> S = 2399.  A step = frames (bf16 NCHW, already resident in
HBM) -> ViT -> projector -> embedding splice -> prefill through the last-position logits.
2304 video tokens per clip.  N>1: clip-level replicas (the path shards by clip; no data-path
collective), weak scaling.

`roofline` is reported for the dominant kernel of the step, the gate/up projection GEMM with the SwiGLU
epilogue (`gemm_nt_256<bf16 out, swiglu>`, csrc/gemm256.hip: M=2399, N=37888, K=3584, 28 launches per step, ~25 % of
the step): algorithmic FLOPs 2*M*N*K per launch / its mean launch duration measured with HIP events
on the launch stream inside the timed region.  `cpu_baseline` times the CPU oracle (oracle/ref_cpu.py,
fp32 torch eager, a port of the reference's CPU path) on a bounded sample and extrapolates by FLOPs.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_FRAMES, IMG, PROMPT_LEN, VIDEO_POS = 32, 336, 96, 14
VISION = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16, image_size=IMG, patch_size=14)
MFMA_PEAK_TFLOPS = 2500.0          # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"

# algorithmic work per clip (2*MACs), SURVEY.md §8(d)
FLOPS = dict(vit=15.89e12, proj=4.69e12, llm=32.48e12)


class _Tok:
    def convert_tokens_to_ids(self, toks):
        return [151645 for _ in toks]


def build_model(device, frames=T_FRAMES):
    from ufvideo_amd.model import VideoReferQwen2Config, VideoReferQwen2ForCausalLM, QWEN2_7B
    cfg = VideoReferQwen2Config(**QWEN2_7B, mm_vision_tower="siglip-so400m-patch14-384", mm_vision_select_layer=-2,
                                mm_vision_select_feature="patch", mm_projector_type="stc_connector_v35", mm_hidden_size=1152,
                                mm_region_encoder_type="pooling", image_aspect_ratio="square", train_mask_decoder=False,
                                sam_pretrained=None, sam_out_dim=256, num_frames=frames, seg_token_id=151747, sam2_trunk=None,
                                vision_config=VISION)
    model = VideoReferQwen2ForCausalLM(cfg, device=device, seed=0)
    model.get_vision_tower().load_model(device=device, seed=7)
    for m in model.modules():
        m.tokenizer = _Tok()
    return model


def synthetic_inputs(device, frames=T_FRAMES):
    """uint8 frames from default_rng(1234), (x/255-0.5)/0.5 -> bf16 NCHW; prompt from default_rng(1235)."""
    from ufvideo_amd import ops
    u8 = np.random.default_rng(1234).integers(0, 256, (frames, IMG, IMG, 3), dtype=np.uint8)
    video = ops.preprocess_u8(torch.from_numpy(u8).to(device), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))      # HIP kernel, outside the timed region
    ids = np.random.default_rng(1235).integers(0, 151643, PROMPT_LEN).astype(np.int64)
    ids[VIDEO_POS] = -201
    ids = torch.from_numpy(ids)[None].to(device)
    return video, ids, torch.ones_like(ids)


class KernelTimer:
    """HIP-event timing of one kernel family on the launch stream (torch's current stream)."""

    def __init__(self):
        self.pairs = []
        self.on = False

    def wrap(self, ops_mod):
        orig = ops_mod.gemm
        timer = self

        def gemm(a, w, *args, **kw):
            if timer.on and kw.get("swiglu") and a.shape[0] > 64 and not isinstance(w, ops_mod.Fp8Weight):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                out = orig(a, w, *args, **kw)
                e.record()
                timer.pairs.append((s, e, a.shape[0], w.shape[0], a.shape[1]))
                return out
            return orig(a, w, *args, **kw)
        ops_mod.gemm = gemm
        orig8 = ops_mod.gemm_fp8

        def gemm_fp8(aq, sa, w, *args, **kw):       # fp8 mode: the GEMM alone (the activation quantisation is its own kernel)
            if timer.on and kw.get("swiglu") and aq.shape[0] > 64:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                out = orig8(aq, sa, w, *args, **kw)
                e.record()
                timer.pairs.append((s, e, aq.shape[0], w.shape[0], aq.shape[1]))
                return out
            return orig8(aq, sa, w, *args, **kw)
        ops_mod.gemm_fp8 = gemm_fp8

    def summary(self):
        if not self.pairs:
            return None
        ms = [s.elapsed_time(e) for s, e, *_ in self.pairs]
        M, N, K = self.pairs[0][2:]
        return dict(mean_ms=float(np.mean(ms)), launches=len(ms), M=M, N=N, K=K, flops=2.0 * M * N * K)


def one_step(model, video, ids, am, cache, frameshard=False):
    cache.len = 0
    mmf = None
    if frameshard:
        from ufvideo_amd.parallel import encode_frame_sharded
        mmf = encode_frame_sharded(model, video)[None]
    _, am2, _, emb, _, _ = model.prepare_inputs_labels_for_multimodal(ids, am, None, None, [(video, "video")], None, None, None, None,
                                                                      mm_features=mmf)
    logits, *_ = model._decode_batch(emb, am2, cache, False, 1)
    return logits, emb.shape[1]


def cpu_baseline(threads):
    """Times the CPU oracle on a bounded sample of config #2 (about 10-20 s of CPU work) and extrapolates per stage by
    frames x layers.  Sample: ViT 8 frames x 4 layers (+patch embed); projector (RegStage x4, Conv3d, RegStage x4, readout) on
    the same 8 frames; 3 Qwen2-7B-dim decoder layers at S=2399 (median layer time)."""
    from oracle import ref_cpu as O
    torch.set_num_threads(threads)
    t_all = time.time()
    FS, VL = 8, 4
    with torch.no_grad():
        vcfg = dict(VISION, num_hidden_layers=VL + 1)                        # hidden_states[-2] = output of layer VL
        sd = O.make_siglip_weights(vcfg, seed=11)
        x = torch.randn(FS, 3, IMG, IMG)
        O.siglip_tower(sd, vcfg, x[:1])                                      # warm-up
        t0 = time.time(); f = O.siglip_tower(sd, vcfg, x); t_vit = time.time() - t0
        vit_total = t_vit * (T_FRAMES / FS) * (26 / VL)
        psd = O.make_stc_weights(1152, 3584, seed=5)
        t0 = time.time(); O.stc_connector(psd, f[None]); t_proj = time.time() - t0
        proj_total = t_proj * (T_FRAMES / FS)
        del psd, sd
        lcfg = dict(vocab_size=256, hidden_size=3584, intermediate_size=18944, num_hidden_layers=1, num_attention_heads=28,
                    num_key_value_heads=4, rope_theta=1e6, rms_norm_eps=1e-6)
        lsd = O.make_qwen2_weights(lcfg, seed=12)
        xe = torch.randn(1, 2399, 3584) * 0.5
        ts = []
        for _ in range(3):
            t0 = time.time(); O.qwen2_forward(lsd, lcfg, xe, all_logits=False); ts.append(time.time() - t0)
        t_llm = sorted(ts)[1]
        llm_total = t_llm * 28
    total = vit_total + proj_total + llm_total
    return dict(value=round(2304.0 / total, 3), unit="video-tokens/s", cores=threads, kind="port",
                sample=(f"oracle/ref_cpu.py fp32 eager: ViT {FS} frames x {VL} layers {t_vit:.2f}s, projector on {FS} frames {t_proj:.2f}s, "
                        f"1 of 28 decoder layers at S=2399 {t_llm:.2f}s (median of 3); extrapolated by frames x layers to {total:.1f}s "
                        f"per clip; sample wall {time.time() - t_all:.1f}s"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=T_FRAMES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp8", action="store_true", help="BASELINE config #5a: W8A8 e4m3 GEMMs (not the headline bf16 run)")
    ap.add_argument("--mode", choices=["replica", "frameshard"], default="replica",
                    help="replica: one clip per GPU per step (default, weak scaling); frameshard: ONE clip per step, frames "
                         "sharded over the ranks for tower+projector, RCCL all-gather of visual tokens, decoder on every rank")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    device = torch.device("cuda", torch.cuda.current_device())

    from ufvideo_amd import ops
    from ufvideo_amd.model import KVCache
    timer = KernelTimer(); timer.wrap(ops)

    model = build_model(device, args.frames)
    if args.fp8:
        model.set_gemm_dtype("fp8")
    video, ids, am = synthetic_inputs(device, args.frames)
    cfg = model.config
    cache = KVCache(cfg.num_hidden_layers, 2304 * args.frames // 32 + 128, 2 * cfg.num_key_value_heads * cfg.head_dim, device)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fs = args.mode == "frameshard" and world > 1
    with torch.no_grad():
        for _ in range(args.warmup):
            logits, S = one_step(model, video, ids, am, cache, fs)
        barrier()
        timer.on = True
        t0 = time.perf_counter()
        for _ in range(args.steps):
            logits, S = one_step(model, video, ids, am, cache, fs)
        barrier()
        dt = time.perf_counter() - t0
        timer.on = False
    assert torch.isfinite(logits).all()
    if dist is not None:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    tokens_per_clip = (args.frames // 2) * ((IMG // 14) // 2) ** 2
    value = (1 if fs else world) * args.steps * tokens_per_clip / dt
    out = {
        "metric": "video-tokens/sec (encode+prefill), UFVideo-7B 32f@336px", "value": round(value, 1), "unit": "video-tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong" if fs else "weak", "vs_baseline": None, "dtype": "fp8" if args.fp8 else "bf16", "data": "synthetic",
        "config": {"workload": f"UFVideo-7B dims, {args.frames} frames {IMG}x{IMG}, {'W8A8 e4m3 GEMMs (config #5a), bf16 elsewhere' if args.fp8 else 'bf16'}, prompt 96 ids -> S={S}, "
                               f"encode+project+splice+prefill to last-position logits; clip replicas per GPU",
                   "video_tokens_per_clip": tokens_per_clip, "llm_seq_len": S, "parallelism": (f"frameshard{world}+allgather" if fs else f"clip-dp{world}")},
    }
    if rank == 0:
        ks = timer.summary()
        if ks:
            ach = ks["flops"] / (ks["mean_ms"] * 1e-3) / 1e12
            traffic = None
            peak = 2 * MFMA_PEAK_TFLOPS if args.fp8 else MFMA_PEAK_TFLOPS
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc) and not args.fp8:
                traffic = json.load(open(pmc)).get("gemm_nt_256_swiglu_hbm_bytes_per_launch")
            out["roofline"] = {"bound": "mfma", "kernel": "gemm_nt_256<%s,swiglu> gate/up M=%d N=%d K=%d" % ("fp8" if args.fp8 else "bf16", ks["M"], ks["N"], ks["K"]),
                               "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                               "traffic": traffic, "launch_ms": round(ks["mean_ms"], 4), "launches": ks["launches"]}
        step_tf = sum(FLOPS.values()) * args.frames / 32 / (dt / args.steps) / 1e12
        out["step_tflops"] = round(step_tf, 1)
        out["step_frac_of_mfma_peak"] = round(step_tf / (2 * MFMA_PEAK_TFLOPS if args.fp8 else MFMA_PEAK_TFLOPS), 4)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(min(os.cpu_count() or 1, 64))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
