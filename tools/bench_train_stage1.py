"""Whole training step of the reference's adapter stage at full dims through the drop-in API (diagnostic, not bench.py's metric).

`DecoderTrainer(model, train_projector=True, train_decoder=False).train_step(**collator batch)` on the config-#2 clip (32 frames
336x336, 96-id prompt -> S = 2399): frozen SigLIP tower -> STC-v35 projector (stashed) -> splice -> frozen Qwen2-7B decoder forward
-> causal-LM loss on the answer tokens -> dL/dx back through the decoder -> projector backward -> clip + AdamW on the projector.
`--full`: the decoder is trained too (train_decoder=True)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ufvideo_amd.train import DecoderTrainer

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
full = "--full" in sys.argv
model = bench.build_model(dev)
video, ids, am = bench.synthetic_inputs(dev)
labels = ids.clone(); labels[labels < 0] = -100; labels[:, :40] = -100          # the instruction is not supervised
tr = DecoderTrainer(model, lr=1e-3 if not full else 1e-5, max_grad_norm=1.0, train_projector=True, train_decoder=full)
batch = dict(input_ids=ids, labels=labels, attention_mask=am, images=[(video, "video")])
times, losses = [], []
for it in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = tr.train_step(**batch)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t0); losses.append(float(r["loss"]))
ms = sorted(times[1:])[len(times[1:]) // 2] * 1e3
print(json.dumps({"metric": "train_step ms (%s), 32f@336px clip, S=2399" % ("projector + decoder trained" if full else "adapter stage: projector trained, tower + decoder frozen"),
                  "ms_per_step": round(ms, 1), "tokens_per_s": round(2399 / ms * 1e3, 1), "losses": [round(l, 4) for l in losses],
                  "grad_norm": round(float(r["grad_norm"]), 4), "hbm_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1), "data": "synthetic"}))
