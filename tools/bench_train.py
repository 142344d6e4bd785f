"""Training-step benchmark of the decoder (BASELINE config #4's LLM part; not the headline bench.py metric).

One step = forward with stashed activations + backward + clip + AdamW of the Qwen2-7B-dim decoder (28 layers, vocab 151748)
on ONE synthetic spliced sample of S = 2399 positions per rank (2304 visual + 95 text, the config-#2 sequence), everything
resident in HBM.  N > 1 (torch.distributed.run, one rank per GPU): ZeRO-2 exchange over RCCL -- reduce-scatter of the
fp32 gradient buckets, sharded AdamW, all-gather of the bf16 shards.  Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seq", type=int, default=2399)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--adapter-only", action="store_true",
                    help="the reference's tune_mm_mlp_adapter stage: decoder frozen (dL/dx only), STC-v35 projector trained on 32 x 24 x 24 tower tokens")
    ap.add_argument("--gradient-checkpointing", action="store_true", help="the reference's --gradient_checkpointing True: per layer only the input stream is kept")
    ap.add_argument("--micro-batches", type=int, default=1, help="samples per step and rank (gradient accumulation): the reference's per-device batch is 2 x 32 frames")
    args = ap.parse_args()
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    from ufvideo_amd.model import VideoReferQwen2Config, VideoReferQwen2ForCausalLM, QWEN2_7B
    from ufvideo_amd.train import DecoderTrainer
    cfg = VideoReferQwen2Config(**dict(QWEN2_7B, num_hidden_layers=args.layers), sam2_trunk=None)
    model = VideoReferQwen2ForCausalLM(cfg, device=dev, seed=0)
    if args.adapter_only:
        from ufvideo_amd.model.projector import STCConnectorV35

        class PCfg:
            mm_hidden_size, hidden_size = 1152, cfg.hidden_size
        model.get_model().mm_projector = STCConnectorV35(PCfg(), device=dev)
        tr = DecoderTrainer(model, lr=1e-3, weight_decay=0.0, max_grad_norm=1.0, train_projector=True, train_decoder=False)
    else:
        tr = DecoderTrainer(model, lr=1e-5, weight_decay=0.0, max_grad_norm=1.0, gradient_checkpointing=args.gradient_checkpointing)
    S, D, V = args.seq, cfg.hidden_size, cfg.vocab_size
    g = torch.Generator(device=dev).manual_seed(1237 + rank)
    emb = torch.randn(S, D, device=dev, generator=g) * 0.02
    labels = torch.randint(0, V, (S,), device=dev, generator=g)
    labels[:2304 + 14] = -100                                   # the prompt (video tokens + instruction) is not supervised
    eids = torch.full((S,), -1, device=dev, dtype=torch.int64); eids[2304:] = labels[2304:].clamp(min=0)
    feats = torch.randn(32 * 576, 1152, device=dev, generator=g).to(torch.bfloat16) if args.adapter_only else None

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    t_fb = t_opt = 0.0
    loss = None
    for it in range(args.warmup + args.steps):
        sync(); t0 = time.perf_counter()
        tr.zero_grad()
        if args.adapter_only:                       # projector (stash) -> its 2304 tokens replace the visual rows -> decoder -> projector backward
            mm, stash = tr.pgrad.forward(feats, 32, 24)
            x_in = torch.cat([mm * 0.02, emb[mm.shape[0]:]], 0)
            loss, dx = tr.forward_backward(x_in, labels, embed_ids=eids, last=True)
            grads, _ = tr.pgrad.backward(dx[:mm.shape[0]] * 0.02, stash)
            for name, gval in grads.items():
                name = "mm_projector." + name
                tr.proj_bucket.view(tr.proj_bucket.g, name).add_(gval.reshape(tr.proj_params[name].shape))
        else:
            for mb in range(args.micro_batches):
                loss, _ = tr.forward_backward(emb, labels, embed_ids=eids, last=mb == args.micro_batches - 1)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        tr.step()
        sync(); t2 = time.perf_counter()
        if it >= args.warmup:
            t_fb += t1 - t0; t_opt += t2 - t1
    dt = (t_fb + t_opt) / args.steps
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())
    H, KV, hd, I, L = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.intermediate_size, args.layers
    lin = 2.0 * S * D * ((H + 2 * KV) * hd + H * hd + 3 * I) * L + 2.0 * S * D * V          # projections + lm_head (all positions)
    att = 2.0 * 2 * H * S * S * hd / 2 * L                                                     # causal QK^T + PV
    flops = (3.0 * lin + 3.5 * att) * args.micro_batches                                       # bwd = 2x linear, 2.5x attention (a re-run forward is not counted as useful work)
    if args.adapter_only:
        flops = 2.0 * lin + 3.5 * att                                                          # no weight gradients in the decoder (projector not counted)
    if rank == 0:
        print(json.dumps({"metric": ("adapter-only training tokens/s (projector trained, decoder frozen)" if args.adapter_only else
                                     "decoder training tokens/s (fwd+bwd+AdamW), Qwen2-7B dims"), "value": round(world * S * args.micro_batches / dt, 1),
                          "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "ms_per_step": round(dt * 1e3, 1),
                          "ms_fwd_bwd": round(t_fb / args.steps * 1e3, 1), "ms_exchange_adamw": round(t_opt / args.steps * 1e3, 1),
                          "loss": round(float(loss), 4), "step_tflops": round(flops / dt / 1e12, 1), "seq_len": S, "layers": L,
                          "hbm_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1), "dtype": "bf16 (fp32 master / grads)",
                          "parallelism": f"zero2-dp{world}", "micro_batches": args.micro_batches,
                          "gradient_checkpointing": bool(args.gradient_checkpointing), "data": "synthetic"}), flush=True)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
