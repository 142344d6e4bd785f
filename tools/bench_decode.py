"""Diagnostic: greedy decode latency (ms/token) after a config-#2 prefill, full UFVideo-7B dims, synthetic weights.
`--fp8`: W8A8 mode (e4m3 weights streamed by the decode step too)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
model = bench.build_model(dev)
fp8 = "--fp8" in sys.argv
if fp8:
    model.set_gemm_dtype("fp8")
video, ids, am = bench.synthetic_inputs(dev)
with torch.no_grad():
    _, am2, _, emb, _, _ = model.prepare_inputs_labels_for_multimodal(ids, am, None, None, [(video, "video")], None, None, None, None)
    for n in (8, 64):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model._greedy(emb, am2, max_new_tokens=n, eos_token_id=None)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"generate {n} tokens: {dt*1e3:.1f} ms total (incl. prefill)")
    # per-token time = (time of 8 + 128 tokens - time of 8 tokens) / 128, the prefill cancels; three repeats, all printed (one pair of runs has given figures 7 % apart
    # on the same box: the median is what the docs quote)
    def run(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model._greedy(emb, am2, max_new_tokens=n, eos_token_id=None)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    per = []
    for _ in range(3):
        t8 = run(8); t136 = run(136)
        per.append((t136 - t8) / 128)
    per.sort()
    gb = 7.6 + 1.09 if fp8 else 15.2
    print("per-token times of the three repeats (ms):", ", ".join(f"{p * 1e3:.2f}" for p in per))
    print(f"decode ({'fp8' if fp8 else 'bf16'} weights): {per[1] * 1e3:.2f} ms/token  (weights {gb:.1f} GB -> {(gb * 1e9 / per[1]) / 1e12:.2f} TB/s effective)")
