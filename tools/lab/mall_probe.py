"""Lab (GPU box): can the decode step's weight stream be decoupled from its dependent chain through the memory-side cache (MALL, 256 MB)?
The step's GEMVs are "paced by how fast a 33-270 MB burst drains from HBM after a cold start" (LABNOTES round 3 / 4): every kernel ramps up and down alone.
(A) what a GEMV costs when its matrix is already in the MALL: the same matrix 28 x in a captured graph against 28 distinct ones;
(B) a touch chain on a SIDE branch of the graph that reads matrix j + LEAD while GEMV j runs (no dependency edge into the chain except a bound on the lead):
    the layer sequence qkv -> o -> gate/up -> down of 28 layers with distinct weights, with and without the side branch.
usage: python3 tools/lab/mall_probe.py [lead]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ufvideo_amd import ops, _lib

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
D, I, H, KV, hd, L = 3584, 18944, 28, 4, 128, 28
g = torch.Generator(device=dev); g.manual_seed(1)
rnd = lambda *s, sc=0.02: (torch.randn(*s, device=dev, generator=g) * sc)
main, side = torch.cuda.Stream(), torch.cuda.Stream()
LEAD = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NPART = int(os.environ.get("UFV_TOUCH_BLOCKS", "512"))


def graph_time(body, reps=10):
    exec_ = ctypes.c_void_p(None)
    with torch.cuda.stream(main):
        body(); torch.cuda.synchronize()
        st = main.cuda_stream
        _lib.call("ufv_graph_begin", st)
        try:
            body()
        finally:
            _lib.call("ufv_graph_end", st, ctypes.byref(exec_))
        for _ in range(3):
            _lib.call("ufv_graph_launch", exec_, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(reps):
            _lib.call("ufv_graph_launch", exec_, st)
        e1.record(main)
        main.synchronize()
        _lib.call("ufv_graph_destroy", exec_)
    return e0.elapsed_time(e1) / reps


x = rnd(D, sc=1.0).contiguous()
lnw = (1 + 0.1 * torch.randn(D, device=dev, generator=g)).contiguous()
kinds = [("qkv", (H + 2 * KV) * hd, D, "rms"), ("o", D, H * hd, "resid"), ("gu", 2 * I, D, "swiglu"), ("down", D, I, "resid")]
W = {k: [rnd(N, K).to(torch.bfloat16) for _ in range(L)] for k, N, K, f in kinds}
a_in = {k: rnd(K, sc=1.0).to(torch.bfloat16) for k, N, K, f in kinds}
xs = x.clone()
outs = {k: torch.empty(N // 2 if f == "swiglu" else N, device=dev, dtype=torch.bfloat16) for k, N, K, f in kinds}
part = torch.empty(NPART, device=dev)


def gemv(kind, w):
    f = {k: ff for k, N, K, ff in kinds}[kind]
    if f == "rms":
        ops.gemv1(w, x=x, ln_w=lnw, out=outs[kind])
    elif f == "swiglu":
        ops.gemv1(w, x=x, ln_w=lnw, swiglu=True, out=outs[kind])
    else:
        ops.gemv1(w, a=a_in[kind], resid=xs, out=xs)


def touch(w):
    _lib.call("ufv_sumsq", w.data_ptr(), w.numel() // 2, part.data_ptr(), NPART, torch.cuda.current_stream().cuda_stream)


print("(A) one kind 28 x: distinct matrices (cold) | the same matrix (MALL-warm)")
for k, N, K, f in kinds:
    cold = graph_time(lambda: [gemv(k, w) for w in W[k]]) * 1e3 / L
    warm = graph_time(lambda: [gemv(k, W[k][0]) for _ in range(L)]) * 1e3 / L
    mb = N * K * 2 / 1e6
    print(f"  {k:5s} {mb:6.1f} MB: cold {cold:6.2f} us ({mb / cold / 1e0:5.2f} GB/ms = {mb / cold:5.2f} TB/s)   warm {warm:6.2f} us ({mb / warm:5.2f} TB/s)", flush=True)
tt = graph_time(lambda: [touch(w) for w in W["gu"]]) * 1e3 / L
print(f"  touch (sumsq, {NPART} blocks) of a 271.6 MB matrix, distinct: {tt:6.2f} us = {271.6 / tt:5.2f} TB/s")

seq = [(k, W[k][l]) for l in range(L) for k in ("qkv", "o", "gu", "down")]


def chain():
    for k, w in seq:
        gemv(k, w)


def chain_prefetch():
    """side branch: touch(seq[j + LEAD]) may start once GEMV j - 1 is done (bounds the lead), and nothing waits for a touch except the end of the graph"""
    cur = torch.cuda.current_stream()
    ev0 = torch.cuda.Event(); ev0.record(cur)
    side.wait_event(ev0)
    for j, (k, w) in enumerate(seq):
        if j + LEAD < len(seq):
            with torch.cuda.stream(side):
                touch(seq[j + LEAD][1])
        gemv(k, w)
        ev = torch.cuda.Event(); ev.record(cur)
        side.wait_event(ev)
    cur.wait_stream(side)


base = graph_time(chain)
pre = graph_time(chain_prefetch)
tot_mb = sum(w.numel() * 2 for _, w in seq) / 1e6
print(f"(B) 28 layers x (qkv, o, gate/up, down), {tot_mb / 1e3:.1f} GB: chain {base:.3f} ms ({tot_mb / base / 1e3:.2f} TB/s)   with the touch branch (lead {LEAD}) {pre:.3f} ms "
      f"({tot_mb / pre / 1e3:.2f} TB/s)")
