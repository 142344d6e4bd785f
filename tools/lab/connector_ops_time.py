"""Lab (GPU box): the connector's elementwise kernels at the bench clip's stage-1 shape (32 frames x 24 x 24 tokens x 3584 channels), timed in a chain that
mimics the block (the input of each kernel was written by the launch before it) and one by one over rotating buffers (cold).  Prints us per launch and a
checksum of every output so that variants (UFV_DWCONV_NO_REMAP=1) can be compared bit for bit.   usage: python3 tools/lab/connector_ops_time.py [F H W C]"""
import os
import sys
import hashlib

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufvideo_amd import ops  # noqa: E402


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def digest(t):
    return hashlib.sha1(t.detach().contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:12]


def main():
    F, H, W, C = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (32, 24, 24, 3584)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(5)
    M = F * H * W
    NB = 6                                                                  # 6 x 132 MB: more than the 256 MB of MALL
    xs = [torch.randn(M, C, device=dev, generator=g).to(torch.bfloat16) for _ in range(NB)]
    w9 = (torch.randn(9, C, device=dev, generator=g) * 0.3).to(torch.bfloat16).float()
    lw = (1 + 0.1 * torch.randn(C, device=dev, generator=g)).to(torch.bfloat16).float()
    lb = (0.1 * torch.randn(C, device=dev, generator=g)).to(torch.bfloat16).float()
    gate = torch.rand(F, C, device=dev, generator=g).to(torch.bfloat16)
    k = [0]

    def rot():
        k[0] = (k[0] + 1) % NB
        return xs[k[0]]
    print("shape", F, H, W, C, "UFV_DWCONV_NO_REMAP", os.environ.get("UFV_DWCONV_NO_REMAP"))
    y = ops.dwconv3x3_ln_silu(xs[0], w9, lw, lb, F, H, W, C, 1e-5)
    print("dwconv digest", digest(y))
    print("dwconv same-buffer  %.1f us" % timed(lambda: ops.dwconv3x3_ln_silu(xs[0], w9, lw, lb, F, H, W, C, 1e-5)))
    print("dwconv rotating     %.1f us" % timed(lambda: ops.dwconv3x3_ln_silu(rot(), w9, lw, lb, F, H, W, C, 1e-5)))
    y = ops.layernorm(xs[1], lw, lb, 1e-5, act="silu")
    print("ln+silu digest", digest(y))
    print("ln+silu same-buffer %.1f us" % timed(lambda: ops.layernorm(xs[1], lw, lb, 1e-5, act="silu")))
    print("ln+silu rotating    %.1f us" % timed(lambda: ops.layernorm(rot(), lw, lb, 1e-5, act="silu")))
    y = ops.ln_add_silu(xs[2], lw, lb, xs[3], None, None, 1e-5)
    print("ln+add+silu digest", digest(y))
    print("ln+add+silu same    %.1f us" % timed(lambda: ops.ln_add_silu(xs[2], lw, lb, xs[3], None, None, 1e-5)))
    print("ln+add+silu rotating %.1f us" % timed(lambda: ops.ln_add_silu(rot(), lw, lb, rot(), None, None, 1e-5)))
    y = ops.ln_add_silu(xs[2], lw, lb, xs[3], lw, lb, 1e-5)
    print("ln+add+silu (two norms) digest", digest(y))
    print("ln+add+silu (two norms) rotating %.1f us" % timed(lambda: ops.ln_add_silu(rot(), lw, lb, rot(), lw, lb, 1e-5)))
    print("colmean rotating    %.1f us" % timed(lambda: ops.colmean(rot(), F, H * W)))
    print("scale rotating      %.1f us" % timed(lambda: ops.scale_channels(rot(), gate, F, H * W)))

    a32 = torch.randn(F, C, device=dev, generator=g).to(torch.bfloat16)
    w1 = (torch.randn(C // 4, C, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    w2 = (torch.randn(C, C // 4, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    b1, b2 = torch.zeros(C // 4, device=dev), torch.zeros(C, device=dev)
    h = ops.gemm(a32, w1, bias=b1, act="silu")
    o = ops.gemm(h, w2, bias=b2, act="sigmoid")
    print("se digests", digest(h), digest(o), "UFV_NO_SMALL_M", os.environ.get("UFV_NO_SMALL_M"))
    ref = torch.sigmoid(torch.nn.functional.silu(a32.float() @ w1.float().T).to(torch.bfloat16).float() @ w2.float().T)
    print("se max abs err vs torch %.3e" % float((o.float() - ref).abs().max()))
    print("se1 %d x %d -> %d     %.1f us" % (F, C, C // 4, timed(lambda: ops.gemm(a32, w1, bias=b1, act="silu"))))
    print("se2 %d x %d -> %d     %.1f us" % (F, C // 4, C, timed(lambda: ops.gemm(h, w2, bias=b2, act="sigmoid"))))

    # the block's chain: ln+silu -> dwconv -> colmean -> scale (each input freshly written by the launch before)
    def chain():
        a = ops.layernorm(xs[0], lw, lb, 1e-5, act="silu")
        b = ops.dwconv3x3_ln_silu(a, w9, lw, lb, F, H, W, C, 1e-5)
        ops.colmean(b, F, H * W)
        ops.scale_channels(b, gate, F, H * W)
    print("chain ln,dwconv,colmean,scale %.1f us" % timed(chain, n=10))


if __name__ == "__main__":
    main()
