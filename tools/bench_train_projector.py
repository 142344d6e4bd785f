"""Diagnostic: projector (stc_connector_v35, 1152 -> 3584) training forward + backward at config-#2 size (32 frames x 24 x 24)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ufvideo_amd.model.projector import STCConnectorV35
from ufvideo_amd.train_projector import ProjectorGrad


class Cfg:
    mm_hidden_size, hidden_size = 1152, 3584


pj = STCConnectorV35(Cfg(), device="cuda")
pg = ProjectorGrad(pj)
x = torch.randn(32 * 576, 1152, device="cuda").to(torch.bfloat16)
dout = torch.randn(2304, 3584, device="cuda")
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out, st = pg.forward(x, 32, 24)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    g, dx = pg.backward(dout, st)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iter {it}: forward (stash) {1e3 * (t1 - t0):.1f} ms, backward {1e3 * (t2 - t1):.1f} ms, {torch.cuda.max_memory_allocated() / 1e9:.1f} GB", flush=True)
