#!/usr/bin/env python
"""bench.py — video-tokens/s (encode + project + splice + LLM prefill) on synthetic UFVideo-7B clips.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  One process per GPU: under torch.distributed.run (WORLD_SIZE
set) this process IS one rank and WORLD_SIZE must equal --gpus; started plainly with --gpus N > 1 it checks that N GPUs are visible
(fails loudly otherwise), starts the N ranks itself as a `python -m torch.distributed.run` child BEFORE anything touches the GPU, and
exits with the child's code (the reference's eval does the same: one process per GPU, scripts/eval/eval_video_PixRQA.sh:26-38).
One JSON line on rank 0.

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
fp32 torch eager, a port of the reference's CPU path): config #1 in full and ONE WHOLE config-#2 clip (measured,
about 1.5 min on 64 cores); `--cpu-quick` (or a host on which config #1 predicts more than 4 minutes for the clip)
runs a bounded sample instead and labels the figure extrapolated.
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

# algorithmic work per clip (2*MACs); the 384-px row is the released checkpoint's own geometry (siglip-so400m-patch14-384:
# 729 tokens per frame, 2704 visual tokens, S = 2799) and is reported by `--img 384` as a SECONDARY line, never the headline
FLOPS = dict(vit=15.89e12, proj=4.69e12, llm=32.48e12)
FLOPS_384 = dict(vit=20.54e12, proj=5.84e12, llm=38.12e12)


def tokens_per_clip(frames, img=None):
    """(T / 2) * floor((img / 14) / 2)^2: the visual tokens the connector hands to the decoder"""
    return (frames // 2) * (((IMG if img is None else img) // 14) // 2) ** 2


PMC_FILE = "profiles/r06/pmc_bench.json"            # (--fp8: pmc_bench_fp8.json beside it)
GEMM_SOURCES = ("ufvideo_amd/csrc/gemm256_kernel.h", "ufvideo_amd/csrc/gemm256.hip", "ufvideo_amd/csrc/gemm_epi.h", "ufvideo_amd/csrc/gemm.hip",
                "ufvideo_amd/csrc/common.h", "ufvideo_amd/csrc/gemm256_m.hip", "ufvideo_amd/csrc/gemm256_m2.hip")


def gemm_sources_sha256():
    """fingerprint of the dominant kernel's sources: the committed PMC summary (roofline.traffic, mfma_busy_pmc) carries the fingerprint of the tree it
    was measured on (tools/pmc_summarize.py) and is only quoted while it matches this tree's"""
    import hashlib
    h = hashlib.sha256()
    for f in GEMM_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


class _Tok:
    def convert_tokens_to_ids(self, toks):
        return [151645 for _ in toks]


def build_model(device, frames=T_FRAMES, img=IMG):
    from ufvideo_amd.model import VideoReferQwen2Config, VideoReferQwen2ForCausalLM, QWEN2_7B
    cfg = VideoReferQwen2Config(**QWEN2_7B, mm_vision_tower="siglip-so400m-patch14-384", mm_vision_select_layer=-2,
                                mm_vision_select_feature="patch", mm_projector_type="stc_connector_v35", mm_hidden_size=1152,
                                mm_region_encoder_type="pooling", image_aspect_ratio="square", train_mask_decoder=False,
                                sam_pretrained=None, sam_out_dim=256, num_frames=frames, seg_token_id=151747, sam2_trunk=None,
                                vision_config=dict(VISION, image_size=img))
    model = VideoReferQwen2ForCausalLM(cfg, device=device, seed=0)
    model.get_vision_tower().load_model(device=device, seed=7)
    for m in model.modules():
        m.tokenizer = _Tok()
    return model


def synthetic_inputs(device, frames=T_FRAMES, img=IMG):
    """uint8 frames from default_rng(1234), (x/255-0.5)/0.5 -> bf16 NCHW; prompt from default_rng(1235)."""
    from ufvideo_amd import ops
    u8 = np.random.default_rng(1234).integers(0, 256, (frames, img, img, 3), dtype=np.uint8)
    video = ops.preprocess_u8(torch.from_numpy(u8).to(device), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))      # HIP kernel, outside the timed region
    ids = np.random.default_rng(1235).integers(0, 151643, PROMPT_LEN).astype(np.int64)
    ids[VIDEO_POS] = -201
    ids = torch.from_numpy(ids)[None].to(device)
    return video, ids, torch.ones_like(ids)


class KernelTimer:
    """HIP-event timing of the dominant kernel (the gate/up GEMM) on its launch stream, recorded inside the library (ufv_gemm_timing), so the
    launches are timed wherever they are issued from -- the op-level loops or the whole-stage C calls the product path uses."""

    def __init__(self):
        self._on = False

    @property
    def on(self):
        return self._on

    @on.setter
    def on(self, v):
        from ufvideo_amd import _lib
        self._on = bool(v)
        # every 7th gate/up launch (4 of the 28 per step, a different layer each): an event pair costs the stream ~11 us of idle, so bracketing all
        # 28 took 0.3 ms out of every timed step (rocprofv3 kernel trace, profiles/r04); the mean over steps x 4 launches is what `roofline` reports
        _lib.call("ufv_gemm_timing", 7 if (self._on and not os.environ.get("UFV_BENCH_NO_TIMER")) else 0)     # (UFV_BENCH_NO_TIMER: lab A/B runs, tools/lab/ab_bench.sh)

    def summary(self):
        import ctypes
        from ufvideo_amd import _lib
        cap = 1 << 16
        ms = (ctypes.c_float * cap)()
        mnk = (ctypes.c_int32 * (3 * cap))()
        n = _lib.load().ufv_gemm_timing_read(ms, mnk, cap)
        if n <= 0:
            return None
        n = min(n, cap)
        M, N, K = mnk[0], mnk[1], mnk[2]
        return dict(mean_ms=float(np.mean(ms[:n])), launches=n, M=M, N=N, K=K, flops=2.0 * M * N * K)


def one_step(model, video, ids, am, cache, frameshard=False, host=None):
    """host = (ids, attention mask) as CPU tensors: what `ufvideo_amd.mm_infer` hands to generate() on every call (it tokenises on the host and keeps
    that copy), so the splice plan is built without reading the device tensors back.  Nothing is remembered between steps (rounds 3-4 cached the
    read-back per tensor object, which only a loop over one tensor ever hit)."""
    cache.len = 0
    mmf = None
    if frameshard:
        from ufvideo_amd.parallel import encode_frame_sharded
        mmf = encode_frame_sharded(model, video)[None]
    ih, ah = host if host is not None else (None, None)
    _, am2, _, emb, _, _ = model.prepare_inputs_labels_for_multimodal(ids, am, None, None, [(video, "video")], None, None, None, None,
                                                                      mm_features=mmf, input_ids_host=ih, attention_mask_host=ah)
    logits, *_ = model._decode_batch(emb, am2, cache, False, 1, consume=True)          # as generate() hands its splice result on: no 34 MB copy of it
    return logits, emb.shape[1]


def _cpu_clip(O, frames, img, prompt_len, vit_weights, proj_weights, llm_weights, lcfg):
    """One whole clip on the CPU oracle (fp32 torch eager): tower (26 layers) -> STC-v35 -> 28-layer decoder prefill -> last-position
    logits.  Every layer runs; the 26 / 28 layers share ONE layer's synthetic weights (same arithmetic and time, 1/27 of the memory) and the
    lm_head has 4096 rows instead of 151748 (one last-position row product: 1 GFLOP of the clip's 53 TFLOP) -- stated in the JSON `sample` too."""
    vcfg = dict(VISION, image_size=img)
    x = torch.randn(frames, 3, img, img)
    h = O.siglip_embeddings(vit_weights, "", x, 14)
    for _ in range(26):
        h = O.vit_encoder_layer(vit_weights, "encoder.layers.0.", h, 16, 1e-6, "gelu_pytorch_tanh")
    tok = O.stc_connector(proj_weights, h[None])                                  # [1, tokens, 3584]
    S = tok.shape[1] + prompt_len - 1
    xe = torch.cat([torch.randn(1, S - tok.shape[1], 3584) * 0.02, tok], 1)
    cos, sin = O.rope_cos_sin(torch.arange(S), 128, 1e6)
    bias = torch.full((S, S), 0.0).masked_fill(torch.arange(S)[None, :] > torch.arange(S)[:, None], torch.finfo(torch.float32).min)[None, None]
    for _ in range(28):
        xe, _ = O.qwen2_layer(llm_weights, "model.layers.0.", xe, lcfg, cos, sin, None, bias)
    last = O.rmsnorm(xe[:, -1:], llm_weights["model.norm.weight"].float(), 1e-6)
    return torch.nn.functional.linear(last, llm_weights["lm_head.weight"].float()), tok.shape[1]


def cpu_baseline(threads, full_clip=True, budget_s=240.0):
    """The reference's CPU path = HF eager fp32; what is timed here is its restatement (oracle/ref_cpu.py, pinned to the reference on
    the golden fixtures), `kind: port`.  Two figures:
      * `config1`: BASELINE config #1 (4 frames 224x224, S = 223) run IN FULL -- measured, nothing extrapolated (about 4 TFLOP);
      * `value`: config #2 (the bench workload).  By default a bounded sample (ViT 8 frames x 4 layers, the projector on 8 frames,
        one decoder layer at S = 2399) extrapolated by frames x layers and labelled so; `--cpu-full-clip` runs the whole clip
        (about 53 TFLOP: minutes of CPU time) and reports the measured rate instead."""
    from oracle import ref_cpu as O
    torch.set_num_threads(threads)
    t_all = time.time()
    with torch.no_grad():
        vsd = O.make_siglip_weights(dict(VISION, num_hidden_layers=1), seed=11)
        psd = O.make_stc_weights(1152, 3584, seed=5)
        # (vocabulary 4096 instead of 151748: the one last-position lm_head row product is 1 GFLOP of the clip's 53 TFLOP, and
        #  drawing a 0.5 G-element table costs more than the whole timed sample)
        lcfg = dict(vocab_size=4096, hidden_size=3584, intermediate_size=18944, num_hidden_layers=1, num_attention_heads=28,
                    num_key_value_heads=4, rope_theta=1e6, rms_norm_eps=1e-6)
        lsd = O.make_qwen2_weights(lcfg, seed=12)
        # position table of the 224-px tower: 256 rows of the same synthetic table
        vsd224 = dict(vsd); vsd224["embeddings.position_embedding.weight"] = vsd["embeddings.position_embedding.weight"][:256]
        t0 = time.time(); _, ntok1 = _cpu_clip(O, 4, 224, PROMPT_LEN, vsd224, psd, lsd, lcfg); t_c1 = time.time() - t0
        out = dict(unit="video-tokens/s", cores=threads, kind="port",
                   config1={"workload": "config #1: 4 frames 224x224, S = 223, every layer run (the layers share one layer's synthetic weights; 4096-row lm_head)", "seconds": round(t_c1, 2),
                            "value": round(ntok1 / t_c1, 3), "extrapolated": False})
        # config #2 is 53.06 / 4.04 = 13.1 x the work of config #1 and runs at a better rate (longer rows): predicted <= 13.1 x t_c1
        if full_clip and 13.1 * t_c1 > budget_s:
            full_clip = False
            out["note"] = f"whole-clip run skipped: config #1 took {t_c1:.1f}s on this host, predicting > {budget_s:.0f}s for config #2"
        if full_clip:
            t0 = time.time(); _cpu_clip(O, T_FRAMES, IMG, PROMPT_LEN, vsd, psd, lsd, lcfg); total = time.time() - t0
            out.update(value=round(2304.0 / total, 3), extrapolated=False,
                       sample=(f"oracle/ref_cpu.py fp32 eager, one whole config-#2 clip (32 frames 336x336, S = 2399): all 26 tower + 28 decoder layers run, but they "
                               f"share ONE layer's synthetic weights (same arithmetic and time, 1/27 of the memory) and the lm_head has 4096 rows instead of 151748 "
                               f"(1 GFLOP of the clip's 53 TFLOP): {total:.1f}s"))
            return out
        FS, VL = 8, 4
        x = torch.randn(FS, 3, IMG, IMG)
        t0 = time.time()
        f = O.siglip_embeddings(vsd, "", x, 14)
        for _ in range(VL):
            f = O.vit_encoder_layer(vsd, "encoder.layers.0.", f, 16, 1e-6, "gelu_pytorch_tanh")
        t_vit = time.time() - t0
        vit_total = t_vit * (T_FRAMES / FS) * (26 / VL)
        t0 = time.time(); O.stc_connector(psd, f[None]); t_proj = time.time() - t0
        proj_total = t_proj * (T_FRAMES / FS)
        xe = torch.randn(1, 2399, 3584) * 0.5
        cos, sin = O.rope_cos_sin(torch.arange(2399), 128, 1e6)
        bias = torch.full((2399, 2399), 0.0).masked_fill(torch.arange(2399)[None, :] > torch.arange(2399)[:, None], torch.finfo(torch.float32).min)[None, None]
        ts = []
        for _ in range(2):
            t0 = time.time(); O.qwen2_layer(lsd, "model.layers.0.", xe, lcfg, cos, sin, None, bias); ts.append(time.time() - t0)
        t_llm = min(ts)
        llm_total = t_llm * 28
    total = vit_total + proj_total + llm_total
    out.update(value=round(2304.0 / total, 3), extrapolated=True,
               sample=(f"EXTRAPOLATED from a bounded sample of config #2 (oracle/ref_cpu.py fp32 eager): ViT {FS} frames x {VL} layers {t_vit:.2f}s, "
                       f"projector on {FS} frames {t_proj:.2f}s, 1 of 28 decoder layers at S=2399 {t_llm:.2f}s; scaled by frames x layers to "
                       f"{total:.1f}s per clip (the default runs the whole clip); all CPU work of this line {time.time() - t_all:.1f}s"))
    return out


def run(args, rank, world, dist, device, build=None, inputs=None, step=None, sync=None, cache_factory=None):
    """The timed protocol of the driver contract, separated from process set-up so that the world-size-2 gloo test can drive it with
    a stub model: W warm-up steps, barrier + sync, EXACTLY K timed steps, barrier + sync, MAX of the elapsed time over the ranks,
    value = whole-job video tokens / that time.  Returns the JSON dict (rank 0 prints it)."""
    build = build or build_model
    inputs = inputs or synthetic_inputs
    step = step or one_step
    sync = sync or torch.cuda.synchronize
    img = getattr(args, "img", IMG)
    if img != IMG:                                  # the secondary 384-px line: same protocol, the checkpoint's own tower geometry
        import functools
        build = functools.partial(build_model, img=img) if build is build_model else build      # (a test passes its own already-built 384-px model)
        inputs = functools.partial(synthetic_inputs, img=img) if inputs is synthetic_inputs else inputs
    timer = KernelTimer()
    model = build(device, args.frames)
    if args.fp8:
        model.set_gemm_dtype("fp8")
    video, ids, am = inputs(device, args.frames)
    if step is one_step and torch.is_tensor(ids):
        import functools
        step = functools.partial(one_step, host=(ids.cpu(), am.cpu()))       # the prompt's host copy, as mm_infer passes it (copied once, outside the timed region)
    if cache_factory is None:
        from ufvideo_amd.model import KVCache
        cfg = model.config
        cache = KVCache(cfg.num_hidden_layers, tokens_per_clip(args.frames, img) + 128, 2 * cfg.num_key_value_heads * cfg.head_dim, device)
    else:
        cache = cache_factory(model)

    def barrier():
        if dist is not None:
            dist.barrier()
        sync()

    fs = args.mode == "frameshard" and world > 1
    with torch.no_grad():
        for _ in range(args.warmup):
            logits, S = step(model, video, ids, am, cache, fs)
        barrier()
        timer.on = True
        t0 = time.perf_counter()
        for _ in range(args.steps):
            logits, S = step(model, video, ids, am, cache, fs)
        barrier()
        dt = time.perf_counter() - t0
        timer.on = False
    assert torch.isfinite(logits).all()
    if dist is not None:
        # (the rehearsal mode below runs gloo, whose collectives take host tensors; RCCL takes the device scalar)
        tt = torch.tensor([dt], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    ntok = tokens_per_clip(args.frames, img)
    value = (1 if fs else world) * args.steps * ntok / dt
    out = {
        "metric": f"video-tokens/sec (encode+prefill), UFVideo-7B 32f@{img}px", "value": round(value, 1), "unit": "video-tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong" if fs else "weak", "vs_baseline": None, "dtype": "fp8" if args.fp8 else "bf16", "data": "synthetic",
        "config": {"workload": f"UFVideo-7B dims, {args.frames} frames {img}x{img}, {'W8A8 e4m3 GEMMs (config #5a), bf16 elsewhere' if args.fp8 else 'bf16'}, prompt 96 ids -> S={S}, "
                               f"encode+project+splice+prefill to last-position logits; clip replicas per GPU",
                   "video_tokens_per_clip": ntok, "llm_seq_len": S,
                   "prompt": "device tensors + the caller's host copy of the ids / mask, as ufvideo_amd.mm_infer passes them (no per-tensor cache between steps)", "parallelism": (f"frameshard{world}+allgather" if fs else f"clip-dp{world}")},
    }
    if REHEARSAL():
        out["rehearsal"] = f"TEST ONLY: {world} ranks share cuda:0 over gloo (UFV_BENCH_REHEARSAL); not a multi-GPU measurement"
        if fs:      # behind the timed region: what every rank gathered == what ONE process computes from all the frames (tests/test_parallel_gpu.py reads the flag)
            from ufvideo_amd.parallel import encode_frame_sharded
            with torch.no_grad():
                gathered = encode_frame_sharded(model, video)
                whole = model.encode_images_or_videos([(video, "video")])[0]
            ok = torch.tensor([int(torch.equal(gathered, whole))], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            out["frameshard_tokens_equal_single_process"] = bool(ok.item())
    if rank == 0:
        ks = timer.summary()
        if ks:
            ach = ks["flops"] / (ks["mean_ms"] * 1e-3) / 1e12
            traffic = None
            peak = 2 * MFMA_PEAK_TFLOPS if args.fp8 else MFMA_PEAK_TFLOPS
            pmc_file = PMC_FILE.replace("pmc_bench.json", "pmc_bench_fp8.json") if args.fp8 else PMC_FILE
            pmc = os.path.join(ROOT, pmc_file)
            busy, pmc_note = None, f"{pmc_file} (tools/pmc_bench.sh: rocprofv3 --pmc passes of this command, not re-measured per run)"
            if os.path.exists(pmc):
                rec = json.load(open(pmc))
                have, want = rec.get("gemm_sources_sha256"), gemm_sources_sha256()
                gu = rec.get("gate_up", {})
                if have == want and f"M={ks['M']} " not in str(gu.get("kernel", "")) + " ":
                    # the counters were taken on the default workload (32 frames, M = 2399); another M is another launch: not quoted
                    pmc_note = f"not quoted: {pmc_file} holds the counters of {gu.get('kernel')}, this run's dominant launch has M={ks['M']}"
                elif have == want:
                    traffic, busy = gu.get("hbm_bytes_per_launch"), gu.get("mfma_busy_fraction")
                else:
                    # counters of a DIFFERENT build of the kernel are not quoted: null, and say so where the driver's log shows it
                    pmc_note = (f"STALE, not quoted: {pmc_file} was measured on GEMM sources {str(have)[:12]}, this tree has {want[:12]}; "
                                f"re-run tools/pmc_bench.sh")
                    print("bench.py: " + pmc_note, file=sys.stderr, flush=True)
            out["roofline"] = {"bound": "mfma", "kernel": "gemm_nt_256<%s,swiglu> gate/up M=%d N=%d K=%d" % ("fp8" if args.fp8 else "bf16", ks["M"], ks["N"], ks["K"]),
                               "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                               "traffic": traffic, "traffic_source": pmc_note,
                               "mfma_busy_pmc": busy,
                               "launch_ms": round(ks["mean_ms"], 4), "launches": ks["launches"]}
        step_tf = sum((FLOPS_384 if img == 384 else FLOPS).values()) * args.frames / 32 / (dt / args.steps) / 1e12
        out["step_tflops"] = round(step_tf, 1)
        # peak HBM of this process (weights + packed copies + activations + the KV cache): the W8A8 mode keeps the bf16 matrices for the one-call decode step BESIDE the e4m3
        # ones, and down_proj twice in e4m3 (row-scaled for decode / short prompts, column-permuted for the MX chain): about +9.5 GB at 7B -- reported, not hidden
        if device.type == "cuda":
            out["hbm_peak_gb"] = round(torch.cuda.max_memory_allocated(device) / 1e9, 2)
        out["step_frac_of_mfma_peak"] = round(step_tf / (2 * MFMA_PEAK_TFLOPS if args.fp8 else MFMA_PEAK_TFLOPS), 4)
        if img != IMG:
            out["secondary"] = (f"NOT the headline: BASELINE config #2 is quoted at 336 px; this line is the released checkpoint's own tower geometry "
                                f"(siglip-so400m-patch14-384, ufvideo/model/encoder.py:108: {(img // 14) ** 2} tokens per frame, {ntok} visual tokens)")
        if world == 1 and not args.no_cpu_baseline and img == IMG:
            out["cpu_baseline"] = cpu_baseline(min(os.cpu_count() or 1, 64), full_clip=not args.cpu_quick)
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=T_FRAMES)
    ap.add_argument("--img", type=int, default=IMG, choices=[IMG, 384], help="384: the released checkpoint's own tower geometry (729 tokens per frame, 2704 visual tokens) as a SECONDARY line; the headline metric is quoted at 336")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-quick", action="store_true", help="cpu_baseline: a bounded sample of config #2 (about 15 s, extrapolated by frames x layers) instead of one whole clip on the CPU oracle (about 1.5 min)")
    ap.add_argument("--cpu-full-clip", action="store_true", help=argparse.SUPPRESS)      # the default since round 3; accepted and ignored
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)               # tests only: gloo on the CPU, a sleeping stand-in for the step
    ap.add_argument("--fp8", action="store_true", help="BASELINE config #5a: W8A8 e4m3 GEMMs (not the headline bf16 run)")
    ap.add_argument("--mode", choices=["replica", "frameshard"], default="replica",
                    help="replica: one clip per GPU per step (default, weak scaling); frameshard: ONE clip per step, frames "
                         "sharded over the ranks for tower+projector, RCCL all-gather of visual tokens, decoder on every rank")
    return ap.parse_args(argv)


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def REHEARSAL():
    """UFV_BENCH_REHEARSAL=1 (tests/test_parallel_gpu.py only): the N ranks of `--gpus N` all use cuda:0 and talk over gloo (RCCL refuses two ranks on one
    device), so that the launch path an 8-GPU node hits first -- launch_ranks -> torch.distributed.run -> main -> run() with a real model, barrier + MAX
    reduction, rank-0 JSON -- can run on a 1-GPU box.  The line it prints carries a `rehearsal` key and is not a scaling number."""
    return os.environ.get("UFV_BENCH_REHEARSAL") == "1"


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks as ONE child `python -m torch.distributed.run` and return its exit
    code.  Nothing in this process has touched the GPU (torch.cuda.device_count() does not initialise it on this image) and nothing is
    exec'ed over it: the ranks are children."""
    import subprocess
    if not args.stub and not REHEARSAL():
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible on this host; refusing to report a {args.gpus}-GPU line "
                  f"from fewer devices", file=sys.stderr, flush=True)
            return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def _stub_run(args, rank, world, dist):
    """Test stand-in (tests/test_parallel_cpu.py): the real run() protocol on the CPU with a step that sleeps, slower on the last rank."""
    class Stub:
        class config:
            num_hidden_layers, num_key_value_heads, head_dim = 1, 1, 8

    def step(model, video, ids, am, cache, fs):
        time.sleep(0.01 * (1 + rank))
        return torch.zeros(1), 2399
    return run(args, rank, world, dist, torch.device("cpu"), build=lambda dev, frames: Stub(), inputs=lambda dev, frames: (None, None, None),
               step=step, sync=lambda: None, cache_factory=lambda m: None)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={world} of the launcher", file=sys.stderr, flush=True)
        return 2
    dist = None
    if args.stub:
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
        out = _stub_run(args, rank, world, dist)
    else:
        if world > 1 and REHEARSAL():
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        elif world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            torch.cuda.set_device(0)
        device = torch.device("cuda", torch.cuda.current_device())
        out = run(args, rank, world, dist, device)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
