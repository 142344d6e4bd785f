"""lab: soak of the generated ViT attention kernel (id 14) against the second-generation kernel (id 11), S = 576 and 729: random batch / head counts (small grids are the race-sensitive ones),
random late-tile spikes in rows of every unit and pass, several launches per data set, for a given number of seconds.  Prints the number of launches compared and of mismatching ones.
usage: python tools/lab/attn_p2_soak.py [seconds]"""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ufvideo_amd import ops
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
hd = 72
rng = random.Random(1234)
t0 = time.time()
n = bad = sets = 0
while time.time() - t0 < secs:
    S = rng.choice((576, 729))
    B, H = rng.choice(((1, 2), (1, 4), (2, 2), (2, 4), (3, 2), (1, 16), (4, 16), (32, 16), (5, 6)))
    q, k, v = torch.randn(B, S, H, hd, device="cuda"), torch.randn(B, S, H, hd, device="cuda"), torch.randn(B, S, H, hd, device="cuda")
    for _ in range(rng.randint(0, 24)):
        row, key, amp = rng.randrange(S), rng.randrange(S) if rng.random() < 0.5 else rng.randrange(S - 64, S), rng.uniform(2.0, 7.0)
        k[:, key] = q[:, row] * amp
    x = torch.empty(B * S, 3 * H * hd, device="cuda", dtype=torch.bfloat16)
    x[:, :H * hd] = q.bfloat16().reshape(B * S, H * hd); x[:, H * hd:2 * H * hd] = k.bfloat16().reshape(B * S, H * hd); x[:, 2 * H * hd:] = v.bfloat16().reshape(B * S, H * hd)
    st = ((S * 3 * H * hd, 3 * H * hd),) * 3
    ref = ops.attention(x, x[:, H * hd:], x[:, 2 * H * hd:], B, H, H, S, S, hd, *st, kernel=11).clone()
    sets += 1
    for rep in range(6):
        o = ops.attention(x, x[:, H * hd:], x[:, 2 * H * hd:], B, H, H, S, S, hd, *st, kernel=14 if rep % 2 == 0 else 0)
        n += 1
        if not torch.equal(o, ref):
            bad += 1
            d = (o != ref).nonzero()
            print(f"MISMATCH S={S} B={B} H={H} rep={rep}: {d.shape[0]} elements, rows {sorted(set((d[:, 0] % S).tolist()))[:6]}", flush=True)
torch.cuda.synchronize()
print(f"soak {time.time() - t0:.0f} s: {sets} data sets, {n} launches of the generated kernel compared with the second generation, {bad} mismatching")
