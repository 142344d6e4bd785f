"""Lab: the decode step's kernels one at a time, as the step runs them -- inside a captured HIP graph, 28 launches of ONE kind back to back, each on its own
layer's weights (15 GB are streamed per token: nothing is cache-resident in the real step and nothing may be here).  Prints microseconds per launch
(launch gaps included: that is what the step pays) and the weight-stream rate.  UFV_LAB=1 UFV_LIBRARY=<other .so> times another build for an A/B on the same box."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ufvideo_amd import ops, _lib

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
D, I, H, KV, hd, L = 3584, 18944, 28, 4, 128, 28
POS = int(os.environ.get("UFV_PROBE_POS", "2430"))
g = torch.Generator(device=dev); g.manual_seed(1)
rnd = lambda *s, sc=0.02: (torch.randn(*s, device=dev, generator=g) * sc)
side = torch.cuda.Stream()


def graph_time(body, reps=20):
    """capture body() on the side stream, replay, return ms per replay"""
    exec_ = ctypes.c_void_p(None)
    with torch.cuda.stream(side):
        body()                                                        # warm (module load) outside the capture
        side.synchronize()
        st = side.cuda_stream
        _lib.call("ufv_graph_begin", st)
        try:
            body()
        finally:
            _lib.call("ufv_graph_end", st, ctypes.byref(exec_))
        for _ in range(3):
            _lib.call("ufv_graph_launch", exec_, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(reps):
            _lib.call("ufv_graph_launch", exec_, st)
        e1.record(side)
        side.synchronize()
        _lib.call("ufv_graph_destroy", exec_)
    return e0.elapsed_time(e1) / reps


x = rnd(D, sc=1.0).contiguous()
lnw = (1 + 0.1 * torch.randn(D, device=dev, generator=g)).contiguous()
rows = {}
kinds = [("qkv   rms  N 4608  K 3584", (H + 2 * KV) * hd, D, "rms"), ("o_proj     N 3584  K 3584", D, H * hd, "resid"),
         ("gate/up rms N 37888 K 3584", 2 * I, D, "swiglu"), ("down       N 3584  K 18944", D, I, "resid")]
for name, N, K, form in kinds:
    ws = [rnd(N, K).to(torch.bfloat16) for _ in range(L)]
    a = rnd(K, sc=1.0).to(torch.bfloat16)
    bias = rnd(N, sc=1.0)
    xs = x.clone()
    out_b = torch.empty(N // 2 if form == "swiglu" else N, device=dev, dtype=torch.bfloat16)

    def body():
        for w in ws:
            if form == "rms":
                ops.gemv1(w, x=x, ln_w=lnw, bias=bias, out=out_b)
            elif form == "swiglu":
                ops.gemv1(w, x=x, ln_w=lnw, swiglu=True, out=out_b)
            else:
                ops.gemv1(w, a=a, resid=xs, out=xs)
    ms = graph_time(body)
    us = ms * 1e3 / L
    print(f"{name}: {us:7.2f} us / launch   {N * K * 2 / us / 1e6:5.2f} TB/s", flush=True)
    rows[name] = us
    del ws

# decode attention: 28 layers' caches, the fused kernel (RoPE + append + split attention + merge)
nsplit = int(os.environ.get("UFV_PROBE_SPLITS", "16"))
caches = [rnd(POS + 64, 2 * KV * hd, sc=1.0).to(torch.bfloat16) for _ in range(L)]
qkv = rnd((H + 2 * KV) * hd, sc=1.0).to(torch.bfloat16)
inv_freq = (1.0 / (1e6 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))).to(dev)
pos_dev = torch.tensor([POS], dtype=torch.int32, device=dev)
o = torch.empty(H * hd, device=dev, dtype=torch.bfloat16)
wsb = torch.zeros(_lib.load().ufv_attention_decode_fused_ws_bytes(H, hd, nsplit), device=dev, dtype=torch.uint8)


def attn_body():
    st = torch.cuda.current_stream().cuda_stream
    for c in caches:
        _lib.call("ufv_attention_decode_fused", qkv.data_ptr(), H, KV, hd, inv_freq.data_ptr(), 0, pos_dev.data_ptr(), c.data_ptr(), c.stride(0), c.shape[0],
                  o.data_ptr(), hd ** -0.5, wsb.data_ptr(), nsplit, st)


us = graph_time(attn_body) * 1e3 / L
print(f"attention (fused, pos {POS}, {nsplit} splits): {us:7.2f} us / launch", flush=True)
rows["attn"] = us
tot = sum(rows.values())
print(f"sum of the five per layer: {tot:.1f} us  -> x 28 = {tot * 28 / 1e3:.2f} ms per token (+ lm_head / norm / argmax ~0.25 ms)")
