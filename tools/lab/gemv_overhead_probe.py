"""Lab (GPU box): the fixed cost of one decode GEMV launch.  t(launch) = t0 + bytes / BW fits the step's four GEMVs with t0 = 12 us, BW = 6.6 TB/s (LABNOTES round 5):
112 launches x 12 us = 1.3 of the 3.75 ms per token.  Here: the same kernels in a captured graph of 112 dependent launches with a matrix of almost no bytes (N = 64),
with and without the fused RMSNorm prologue, and an empty kernel chain for the launch gap itself."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ufvideo_amd import ops, _lib
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
side = torch.cuda.Stream()


def graph_time(body, reps=20):
    exec_ = ctypes.c_void_p(None)
    with torch.cuda.stream(side):
        body(); side.synchronize()
        st = side.cuda_stream
        _lib.call("ufv_graph_begin", st)
        try:
            body()
        finally:
            _lib.call("ufv_graph_end", st, ctypes.byref(exec_))
        for _ in range(3): _lib.call("ufv_graph_launch", exec_, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(reps): _lib.call("ufv_graph_launch", exec_, st)
        e1.record(side); side.synchronize()
        _lib.call("ufv_graph_destroy", exec_)
    return e0.elapsed_time(e1) / reps


D = 3584
g = torch.Generator(device=dev); g.manual_seed(0)
x = torch.randn(D, device=dev, generator=g)
lnw = torch.ones(D, device=dev)
a = torch.randn(D, device=dev, generator=g).to(torch.bfloat16)
L = 112
for N in (64, 1024, 4608):
    ws = [(torch.randn(N, D, device=dev, generator=g) * 0.02).to(torch.bfloat16) for _ in range(L)]
    out = torch.empty(N, device=dev, dtype=torch.bfloat16)
    t_rms = graph_time(lambda: [ops.gemv1(w, x=x, ln_w=lnw, out=out) for w in ws]) * 1e3 / L
    t_plain = graph_time(lambda: [ops.gemv1(w, a=a, out=out) for w in ws]) * 1e3 / L
    print(f"N = {N:5d} ({N * D * 2 / 1e6:6.2f} MB): with the RMSNorm prologue {t_rms:6.2f} us / launch, bf16 row {t_plain:6.2f} us / launch")
p = torch.zeros(1, dtype=torch.int32, device=dev)
t_empty = graph_time(lambda: [_lib.call("ufv_add_int", p.data_ptr(), 1, torch.cuda.current_stream().cuda_stream) for _ in range(L)]) * 1e3 / L
print(f"one-wave kernel chain (ufv_add_int): {t_empty:6.2f} us / launch")
