"""lab: stress loop for the causal hd-128 prefill kernel (csrc/attn_c128.inc): many launches over interleaved shapes and fresh random data,
every output compared BIT FOR BIT with the plain kernel (id 13).  A barrier / ring / wait-count slip shows up here as a mismatch or a hang
(run under `timeout`)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufvideo_amd import ops

def main(n=400):
    dev = "cuda"
    shapes = [(28, 4, 2399), (28, 4, 1217), (12, 2, 2399), (32, 8, 1000), (14, 2, 777), (8, 2, 129), (6, 2, 4703), (28, 4, 130), (16, 8, 3000), (4, 2, 64)]
    g = torch.Generator(device=dev).manual_seed(1)
    bad = 0
    t0 = time.time()
    for it in range(n):
        Hq, Hkv, S = shapes[it % len(shapes)]
        hd = 128
        W = (Hq + 2 * Hkv) * hd
        qkv = (torch.randn(S, W, device=dev, generator=g) * (0.5 + (it % 7))).to(torch.bfloat16)
        q, k, v = qkv[:, :Hq * hd], qkv[:, Hq * hd:], qkv[:, (Hq + Hkv) * hd:]
        st = ((0, W),) * 3
        a = ops.attention(q, k, v, 1, Hq, Hkv, S, S, hd, *st, causal=True, kernel=15)
        b = ops.attention(q, k, v, 1, Hq, Hkv, S, S, hd, *st, causal=True, kernel=13)
        if not torch.equal(a, b):
            bad += 1
            print(f"launch {it}: shape {Hq}/{Hkv} S={S}: {int((a != b).sum())} elements differ", flush=True)
    torch.cuda.synchronize()
    print(f"stress_c128: {n} launches, {bad} mismatching, {time.time() - t0:.1f} s")
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 400))
