"""GPU: every HIP kernel of the C ABI against a plain PyTorch fp32 reference of the same op.
Tolerances: fp32 outputs 1e-5..2e-5 of the largest element; bf16 outputs ONE_ULP = 2^-8 (+ fp32 noise) of the largest element --
the resolution of the storage format in the max norm (a correctly rounded result cannot be closer; the per-element statement,
"never more than one ulp at the element's own magnitude, <= 2e-3 of the elements differ", is tests/test_kernel_rounding_gpu.py);
flash attention additionally rounds P to bf16 before the second MFMA: ATTN_TOL = 6e-3 against the fp32 softmax."""
import math

import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from ufvideo_amd import ops, _lib  # noqa: E402

DEV = "cuda"
ONE_ULP = 2.0 ** -8 + 1e-4
ATTN_TOL = 6e-3


def rel(a, b):
    a = a.float(); b = b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def bf(x):
    return x.to(torch.bfloat16)


def g(*shape, seed=0, scale=1.0):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=gen) * scale).to(DEV)


ACTS = {None: lambda x: x, "gelu_tanh": lambda x: torch.nn.functional.gelu(x, approximate="tanh"),
        "gelu": torch.nn.functional.gelu, "silu": torch.nn.functional.silu, "relu": torch.relu,
        "quick_gelu": lambda x: x * torch.sigmoid(1.702 * x), "sigmoid": torch.sigmoid}


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (300, 256, 192), (1000, 384, 1152), (129, 128, 640), (2399, 512, 448),
                                   (2399, 3584, 128), (2399, 4608, 64), (2500, 3584, 192)])   # last three: 160/192-row tiles
@pytest.mark.parametrize("kernel", [ops.GEMM_FAST, ops.GEMM_GENERIC])
def test_gemm_plain(M, N, K, kernel):
    a, w = bf(g(M, K, seed=1)), bf(g(N, K, seed=2, scale=0.05))
    ref = a.float() @ w.float().t()
    out = ops.gemm(a, w, kernel=kernel)
    assert rel(out, ref) <= ONE_ULP
    out32 = ops.gemm(a, w, out_dtype=torch.float32, kernel=kernel)
    assert rel(out32, ref) < 2e-5


def test_gemm_mfma_layout_asymmetric():
    # A = I (padded), asymmetric W: catches a swapped row/col map
    M = N = K = 128
    a = bf(torch.eye(M, K, device=DEV))
    w = bf((torch.arange(N, device=DEV)[:, None] * 3 + torch.arange(K, device=DEV)[None, :] * 0.5) % 17)
    out = ops.gemm(a, w, out_dtype=torch.float32, kernel=ops.GEMM_FAST)
    assert torch.equal(out, w.float().t().contiguous())


@pytest.mark.parametrize("act", list(ACTS))
def test_gemm_epilogues(act):
    M, N, K = 200, 256, 128
    a, w = bf(g(M, K, seed=3)), bf(g(N, K, seed=4, scale=0.1))
    bias, resid = g(N, seed=5), g(M, N, seed=6)
    ref = ACTS[act](a.float() @ w.float().t() + bias) + resid
    for kern in (ops.GEMM_FAST, ops.GEMM_GENERIC):
        out = ops.gemm(a, w, bias=bias, act=act, resid=resid, out_dtype=torch.float32, kernel=kern)
        assert rel(out, ref) < 1e-4, (act, kern)
    # residual broadcast with row modulo (position-embedding table)
    tab = g(50, N, seed=7)
    ref2 = a.float() @ w.float().t() + bias + tab[torch.arange(M, device=DEV) % 50]
    out2 = ops.gemm(a, w, bias=bias, resid=tab, resid_rows=50, out_dtype=torch.float32, kernel=ops.GEMM_FAST)
    assert rel(out2, ref2) < 1e-4
    # in-place residual stream update
    r = resid.clone()
    ops.gemm(a, w, bias=bias, resid=r, out=r, kernel=ops.GEMM_FAST)
    assert rel(r, a.float() @ w.float().t() + bias + resid) < 1e-4


def pack_swiglu(gate, up):
    I, K = gate.shape
    return torch.stack([gate.view(I // 16, 16, K), up.view(I // 16, 16, K)], dim=1).reshape(2 * I, K).contiguous()


@pytest.mark.parametrize("M,kern", [(300, ops.GEMM_FAST), (300, ops.GEMM_GENERIC), (1, ops.GEMM_GEMV), (5, ops.GEMM_GEMV)])
def test_gemm_swiglu(M, kern):
    I, K = 256, 192
    a = bf(g(M, K, seed=8))
    gate, up = bf(g(I, K, seed=9, scale=0.1)), bf(g(I, K, seed=10, scale=0.1))
    ref = torch.nn.functional.silu(a.float() @ gate.float().t()) * (a.float() @ up.float().t())
    out = ops.gemm(a, pack_swiglu(gate, up), swiglu=True, out_dtype=torch.float32, kernel=kern)
    assert rel(out, ref) < 1e-4


@pytest.mark.parametrize("M", [1, 3, 8, 20, 64])
def test_gemv(M):
    N, K = 1000, 1152
    a, w = bf(g(M, K, seed=11)), bf(g(N, K, seed=12, scale=0.05))
    bias = g(N, seed=13)
    ref = torch.relu(a.float() @ w.float().t() + bias)
    out = ops.gemm(a, w, bias=bias, act="relu", out_dtype=torch.float32, kernel=ops.GEMM_GEMV)
    assert rel(out, ref) < 1e-4


@pytest.mark.parametrize("M,N,K,act", [(32, 896, 3584, "silu"), (32, 3584, 896, "sigmoid"), (33, 1000, 2048, None), (9, 520, 4096, "relu"), (64, 512, 512, None),
                                         (17, 600, 2080, "gelu_tanh"), (1, 896, 3584, "silu"), (2, 3584, 896, None)])
def test_small_m_products_on_the_matrix_cores(M, N, K, act):
    """M <= 64 with 512 <= N <= 8192, K >= 512 (the connector's squeeze-excite products): gemm_small_m -- 16 columns per block, K split over 4 or 8 waves, partial
    sums added in wave order.  Covers both wave counts (K >= 2048 -> 8), ragged M (second row block partly or wholly empty), N not a multiple of 16, bf16 and fp32
    outputs, and that a row's result does not depend on how many rows travel with it (frame chunks of a video)."""
    a, w = bf(g(M, K, seed=21)), bf(g(N, K, seed=22, scale=0.03))
    bias = g(N, seed=23)
    z = a.float() @ w.float().t() + bias
    ref = ACTS[act](z)
    out = ops.gemm(a, w, bias=bias, act=act, out_dtype=torch.float32)
    assert rel(out, ref) < 1e-4
    out16 = ops.gemm(a, w, bias=bias, act=act)
    assert out16.dtype == torch.bfloat16 and torch.equal(out16, out.to(torch.bfloat16)), "bf16 output = the rounded fp32 output"
    assert torch.equal(out, ops.gemm(a, w, bias=bias, act=act, out_dtype=torch.float32)), "deterministic (fixed order of the waves' partial sums)"
    for m0, m1 in ((0, 1), (M // 2, M), (M - 1, M)):
        if m1 > m0:
            assert torch.equal(ops.gemm(a[m0:m1].contiguous(), w, bias=bias, act=act, out_dtype=torch.float32), out[m0:m1]), "rows are independent of the batch they arrive in"


@pytest.mark.parametrize("M,N,K", [(18432, 1152, 1152), (1000, 1152, 4352), (300, 256, 192), (20, 128, 64), (2399, 3584, 3584)])
def test_gemm_stream_bf16_in_place_update(M, N, K):
    """ufv_gemm_stream_bf16: x <- bf16(a w^T + bias + float(x)) in place on a bf16 stream (the opt-in reference-dtype residual stream of the tower): the same
    accumulators and element order as the fp32-residual epilogue, so the result is EXACTLY the rounded output of the fp32 form fed float(x)"""
    a, w = bf(g(M, K, seed=31)), bf(g(N, K, seed=32, scale=0.05))
    bias = g(N, seed=33)
    x = bf(g(M, N, seed=34))
    ref = ops.gemm(a, w, bias=bias, resid=x.float(), out_dtype=torch.float32).to(torch.bfloat16)
    y = x.clone()
    ops.gemm_stream_bf16(a, w, y, bias=bias)
    assert torch.equal(y, ref)


def test_gemm_strided_a_and_errors():
    buf = bf(g(300, 512, seed=14))
    a = buf[:, 128:384]
    w = bf(g(128, 256, seed=15, scale=0.1))
    assert rel(ops.gemm(a, w, out_dtype=torch.float32, kernel=ops.GEMM_FAST), a.float() @ w.float().t()) < 1e-4
    with pytest.raises(Exception):
        ops.gemm(bf(g(64, 100)), bf(g(100, 100)), kernel=ops.GEMM_FAST)     # N, K not tileable -> loud error


@pytest.mark.parametrize("D", [64, 1152, 3584])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_layernorm(D, dt):
    M = 77
    x = g(M, D, seed=16).to(dt)
    w, b = 1 + 0.1 * g(D, seed=17), 0.1 * g(D, seed=18)
    ref = torch.nn.functional.layer_norm(x.float(), (D,), w, b, 1e-6)
    assert rel(ops.layernorm(x, w, b, 1e-6, out_dtype=torch.float32), ref) < 1e-5
    assert rel(ops.layernorm(x, w, b, 1e-6), ref) <= ONE_ULP
    refs = torch.nn.functional.silu(ref)
    assert rel(ops.layernorm(x, w, b, 1e-6, act="silu", out_dtype=torch.float32), refs) < 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,D", [(4096, 1152), (18432, 1152), (5003, 144), (9216, 1280)])
def test_layernorm_pipelined_form_bit_identical(M, D, dt):
    """long inputs (M >= 4096, D <= 1280, bf16 out, no activation) run the persistent pipelined kernel (csrc/ops.hip layernorm_pipe_k: a wave walks several rows and
    requests the next one before reducing the current one); the same per-lane summation order as the plain kernel, so every row must carry the bits the plain kernel
    gives it when it is computed in a short call -- including the last rows of a ragged row count and a row-strided input"""
    x = (g(M, D + 8, seed=M + D) * 3 + 0.5).to(dt)[:, 4:4 + D]
    w, b = 1 + 0.1 * g(D, seed=17), 0.1 * g(D, seed=18)
    long = ops.layernorm(x, w, b, 1e-6)
    for lo in (0, 1234, M - 77):
        assert torch.equal(long[lo:lo + 77], ops.layernorm(x[lo:lo + 77], w, b, 1e-6)), lo
    assert rel(long, torch.nn.functional.layer_norm(x.float(), (D,), w, b, 1e-6)) <= ONE_ULP
    nob = ops.layernorm(x, w, None, 1e-6)
    assert torch.equal(nob[:50], ops.layernorm(x[:50], w, None, 1e-6))


@pytest.mark.parametrize("M,D,act", [(18432, 3584, "silu"), (1001, 3584, None), (7, 2048, "gelu_tanh"), (300, 1288, "silu"), (64, 4096, None)])
def test_layernorm_packed_row_form_bit_identical(M, D, act):
    """bf16 rows of more than 1280 elements run layernorm_packed_k (rows packed in registers, parameters from LDS: the connector's LayerNorm + SiLU, 84 -> 62 us);
    the fp32 copy of the same values runs layernorm_k<fp32>: same element -> lane assignment, same operations in the same order => the same bits."""
    x = bf(g(M, D, seed=61, scale=2.0) + 0.3)
    w, b = 1 + 0.1 * g(D, seed=62), 0.1 * g(D, seed=63)
    for out_dtype in (torch.bfloat16, torch.float32):
        y_packed = ops.layernorm(x, w, b, 1e-5, act=act, out_dtype=out_dtype)
        y_plain = ops.layernorm(x.float(), w, b, 1e-5, act=act, out_dtype=out_dtype)
        assert torch.equal(y_packed, y_plain)
    ref = ACTS[act](torch.nn.functional.layer_norm(x.float(), (D,), w, b, 1e-5))
    assert rel(ops.layernorm(x, w, b, 1e-5, act=act, out_dtype=torch.float32), ref) < 1e-5
    assert torch.equal(ops.layernorm(x, w, None, 1e-5, act=act), ops.layernorm(x.float(), w, None, 1e-5, act=act))       # no bias


def test_ln_add_silu():
    M, D = 50, 3584
    a, b = bf(g(M, D, seed=19)), bf(g(M, D, seed=20))
    wa, ba, wb, bb = 1 + 0.1 * g(D, seed=21), 0.1 * g(D, seed=22), 1 + 0.1 * g(D, seed=23), 0.1 * g(D, seed=24)
    ln = lambda x, w, b_: torch.nn.functional.layer_norm(x.float(), (D,), w, b_, 1e-5)
    assert rel(ops.ln_add_silu(a, wa, ba, b, wb, bb, 1e-5), torch.nn.functional.silu(ln(a, wa, ba) + ln(b, wb, bb))) <= ONE_ULP
    assert rel(ops.ln_add_silu(a, wa, ba, b, None, None, 1e-5), torch.nn.functional.silu(ln(a, wa, ba) + b.float())) <= ONE_ULP


@pytest.mark.parametrize("D", [64, 3584])
def test_rmsnorm(D):
    x, w = g(33, D, seed=25), 1 + 0.1 * g(D, seed=26)
    ref = w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6))
    assert rel(ops.rmsnorm(x, w, 1e-6, out_dtype=torch.float32), ref) < 1e-5


def attn_ref(q, k, v, causal, q_pos0=0):
    # q [B,Sq,Hq,hd], k/v [B,Sk,Hkv,hd]
    B, Sq, Hq, hd = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    rep = Hq // Hkv
    qf = q.float().permute(0, 2, 1, 3)
    kf = k.float().permute(0, 2, 1, 3).repeat_interleave(rep, dim=1)
    vf = v.float().permute(0, 2, 1, 3).repeat_interleave(rep, dim=1)
    s = qf @ kf.transpose(-1, -2) * hd ** -0.5
    if causal:
        qi = torch.arange(Sq, device=q.device)[:, None] + q_pos0
        kj = torch.arange(Sk, device=q.device)[None, :]
        s = s.masked_fill(kj > qi, float("-inf"))
    o = torch.softmax(s, -1) @ vf
    return o.permute(0, 2, 1, 3).reshape(B * Sq, Hq * hd)


@pytest.mark.parametrize("hd,Hq,Hkv,Sq,Sk,causal,kernel", [
    (72, 4, 4, 576, 576, False, 1), (72, 2, 2, 100, 100, False, 1), (128, 8, 2, 300, 300, True, 1),
    (128, 4, 4, 257, 257, True, 1), (64, 4, 2, 130, 130, True, 1), (128, 8, 2, 40, 140, True, 1),
    (16, 4, 2, 61, 61, True, 2), (72, 2, 2, 50, 50, False, 2), (128, 4, 2, 1, 333, True, 2), (128, 4, 2, 1, 333, True, 0),
    (80, 2, 1, 70, 70, False, 1), (96, 2, 2, 33, 200, False, 1),
    (72, 2, 2, 200, 200, True, 1), (72, 2, 1, 70, 300, True, 1), (72, 3, 3, 192, 192, False, 1), (72, 2, 2, 64, 64, False, 1),
    (72, 16, 16, 130, 130, False, 1), (72, 2, 2, 729, 729, False, 0), (72, 3, 3, 729, 729, False, 11), (72, 2, 2, 700, 650, False, 0),
    (72, 16, 16, 576, 576, False, 1), (72, 3, 3, 300, 100, False, 1), (72, 5, 5, 288, 576, False, 9),
    (72, 16, 16, 576, 576, False, 11), (72, 3, 3, 288, 300, False, 11), (72, 2, 2, 864, 70, False, 11), (72, 4, 2, 288, 288, False, 0),
    # head_dim 128 causal with the key split over two wave groups (kernel 12; 13 = the plain kernel; 0 picks by block count): odd tile counts, a
    # single tile (group 1 idle), a cached prefix (q_pos0 > 0), GQA
    (128, 8, 2, 300, 300, True, 12), (128, 4, 4, 257, 257, True, 12), (128, 8, 2, 40, 140, True, 12), (128, 4, 2, 50, 50, True, 12),
    (128, 28, 4, 1300, 1300, True, 12), (128, 28, 4, 1300, 1300, True, 13), (128, 28, 4, 700, 1500, True, 0), (128, 4, 1, 129, 640, True, 12)])
def test_attention(hd, Hq, Hkv, Sq, Sk, causal, kernel):
    B = 2
    # fused qkv buffer like the real path: [B*S, (Hq + 2 Hkv) * hd]
    S = max(Sq, Sk)
    q = bf(g(B, Sq, Hq, hd, seed=27))
    k = bf(g(B, Sk, Hkv, hd, seed=28))
    v = bf(g(B, Sk, Hkv, hd, seed=29))
    q_pos0 = Sk - Sq if causal else 0
    o = ops.attention(q, k, v, B, Hq, Hkv, Sq, Sk, hd, (Sq * Hq * hd, Hq * hd), (Sk * Hkv * hd, Hkv * hd),
                      (Sk * Hkv * hd, Hkv * hd), causal=causal, q_pos0=q_pos0, kernel=kernel)
    ref = attn_ref(q, k, v, causal, q_pos0)
    assert rel(o, ref) <= ATTN_TOL, rel(o, ref)


def test_vit72_kernel_spikes_force_rescales_and_first_tile_can_be_all_negative():
    """second-generation ViT kernel (csrc/attn_vit.inc): the running max rides in the padding of the contraction as a bf16 value --
    rows whose maximum jumps late (deferred-rescale path, several times), rows whose every early score is very negative (m < 0 after
    tile 0) and a partial last tile"""
    B, H, S, Sk, hd = 1, 2, 288, 300, 72
    q, k, v = g(B, S, H, hd, seed=40), g(B, Sk, H, hd, seed=41), g(B, Sk, H, hd, seed=42)
    k[:, 70] = q[:, 5] * 3.0; k[:, 150] = q[:, 5] * 6.0; k[:, 290] = q[:, 40] * 5.0        # growing spikes in tiles 1, 2 and 4
    k[:, :64] = -q[:, 100:164].abs().mean(dim=(1, 2), keepdim=True) * torch.sign(q[:, 100:101]) * 2.0   # tile 0: strongly negative for row 100
    q, k, v = bf(q), bf(k), bf(v)
    o = ops.attention(q, k, v, B, H, H, S, Sk, hd, (S * H * hd, H * hd), (Sk * H * hd, H * hd), (Sk * H * hd, H * hd), kernel=11)
    ref = attn_ref(q, k, v, False)
    # scores of +-50 here: q' = bf16(q * scale * log2 e) carries one more bf16 rounding than q, which moves such a score by ~0.01 and
    # the weight of a dominant key by ~1 % -- 2x ATTN_TOL for this adversarial input (ordinary inputs: test_attention, <= ATTN_TOL)
    assert torch.isfinite(o.float()).all() and rel(o, ref) <= 2 * ATTN_TOL, rel(o, ref)
    old = ops.attention(q, k, v, B, H, H, S, Sk, hd, (S * H * hd, H * hd), (Sk * H * hd, H * hd), (Sk * H * hd, H * hd), kernel=9)
    assert rel(o, old.float()) <= 2 * ATTN_TOL             # and it agrees with the first-generation kernel


def test_vit72_third_generation_bit_identical_and_rescale_paths():
    """third-generation ViT kernel (csrc/attn_vit_p2.inc, generated asm pass loop; kernel id 14 = what AUTO takes at S = 576): the same numerics as
    the second generation (kernel 11), so the outputs must be BIT-IDENTICAL -- on ordinary data, with spikes that drive the out-of-line deferred-rescale
    path in several tiles of several passes (units 0 / 1 / 2, first and later passes), with rows whose first tile is all negative, and for a head count /
    batch that leaves CUs without a block.  Shapes outside its envelope (S != 576, odd head count) are refused by id 14 and go to kernel 11 under AUTO."""
    hd = 72
    for B, H, seed in ((2, 4, 50), (1, 16, 51), (3, 2, 52)):
        S = 576
        q, k, v = g(B, S, H, hd, seed=seed), g(B, S, H, hd, seed=seed + 100), g(B, S, H, hd, seed=seed + 200)
        # growing spikes: tile 1 (key 70), tile 4 (key 300), tile 8 (key 560), seen by rows of every unit and pass (0..95, 96..191, 192..287, and the second half)
        for row, key, amp in ((5, 70, 3.0), (5, 300, 6.0), (40, 560, 5.0), (70, 130, 4.0), (100, 300, 5.0), (200, 450, 6.0), (250, 70, 4.0), (300, 520, 7.0), (500, 200, 5.0), (575, 575, 6.0),
                              # round 6: LAST-tile spikes for unit-2 rows of passes that have a successor (rows 64..95, 160..191; 352..383, 448..479): the rescale path that used
                              # to write into the next pass's in-flight Q registers (tools/gen_attn_p2.py rescale_math); frequent only with few blocks on the chip, hence the repeats below
                              (70, 565, 6.0), (170, 540, 7.0), (360, 570, 6.0), (460, 530, 7.0)):
            k[:, key] = q[:, row] * amp
        k[:, :64] = -q[:, 150:214].abs().mean(dim=(1, 2), keepdim=True) * torch.sign(q[:, 150:151]) * 2.0      # tile 0 strongly negative for row 150
        q, k, v = bf(q), bf(k), bf(v)
        st = ((S * H * hd, H * hd),) * 3
        new = ops.attention(q, k, v, B, H, H, S, S, hd, *st, kernel=14)
        old = ops.attention(q, k, v, B, H, H, S, S, hd, *st, kernel=11)
        auto = ops.attention(q, k, v, B, H, H, S, S, hd, *st)
        if not (torch.equal(new, old) and torch.equal(auto, old)):
            # seen ONCE in round 5 (1 of 7 suite runs, not reproduced in 1240 directed launches, tools/lab/attn_vit_flake*.py): say which kernel is unstable and where
            new2, old2 = ops.attention(q, k, v, B, H, H, S, S, hd, *st, kernel=14), ops.attention(q, k, v, B, H, H, S, S, hd, *st, kernel=11)
            d = (new != old).nonzero()
            raise AssertionError(dict(B=B, H=H, new_vs_old=int(d.shape[0]), first=d[:4].tolist(), last=d[-1:].tolist(), auto_vs_old=int((auto != old).sum()),
                                      kernel14_repeatable=bool(torch.equal(new2, new)), kernel11_repeatable=bool(torch.equal(old2, old)), rerun_equal=bool(torch.equal(new2, old2))))
        assert torch.isfinite(new.float()).all() and rel(new, attn_ref(q, k, v, False)) <= 3 * ATTN_TOL
        for _ in range(20):                     # the round-5 mismatch was a race: the same launch again and again (small grids are the sensitive ones)
            assert torch.equal(ops.attention(q, k, v, B, H, H, S, S, hd, *st, kernel=14), old)
    q, k, v = bf(g(1, 288, 2, hd, seed=60)), bf(g(1, 288, 2, hd, seed=61)), bf(g(1, 288, 2, hd, seed=62))
    with pytest.raises(_lib.UfvError):
        ops.attention(q, k, v, 1, 2, 2, 288, 288, hd, (288 * 2 * hd, 2 * hd), (288 * 2 * hd, 2 * hd), (288 * 2 * hd, 2 * hd), kernel=14)
    q3 = bf(g(1, 576, 3, hd, seed=63))
    o3 = ops.attention(q3, q3, q3, 1, 3, 3, 576, 576, hd, *(((576 * 3 * hd, 3 * hd),) * 3))                    # odd head count: AUTO falls back
    assert torch.equal(o3, ops.attention(q3, q3, q3, 1, 3, 3, 576, 576, hd, *(((576 * 3 * hd, 3 * hd),) * 3), kernel=11))


def test_vit72_third_generation_at_729_tokens_bit_identical_masked_tail_and_ragged_rows():
    """The released checkpoint's tower is siglip-so400m-patch14-384 (ufvideo/model/encoder.py:108): 27 x 27 = 729 tokens per frame.  The generated kernel's S = 729 body
    (tools/gen_attn_p2.py --seq 729: 12 key tiles, the last with 25 keys masked to -inf; 4 passes of 96 rows per wave, the last 39 rows of wave 1 past the end) against the
    second-generation kernel in 6-wave blocks (id 11: its own clamped rows and masked scores), BIT for bit: ordinary data; spikes that drive the deferred-rescale path in
    the first, middle and LAST (masked) tile for rows of every pass including the ragged one; a first tile that is all negative; batches that leave CUs idle; q / k / v as
    column views of one fused buffer (the tower's layout).  The rows behind the output (the next frame's) must stay untouched: the stores of rows >= 729 are dropped by
    the descriptor's range check, not by a branch."""
    from oracle import ref_cpu as O
    hd, S = 72, 729
    for B, H, seed in ((2, 4, 70), (1, 16, 71), (3, 2, 72)):
        q, k, v = g(B, S, H, hd, seed=seed), g(B, S, H, hd, seed=seed + 100), g(B, S, H, hd, seed=seed + 200)
        for row, key, amp in ((5, 70, 3.0), (5, 300, 6.0), (40, 728, 5.0), (100, 710, 4.0), (200, 450, 6.0), (383, 727, 7.0), (384, 70, 4.0), (500, 704, 5.0),
                              (671, 300, 6.0), (672, 728, 7.0), (700, 650, 5.0), (728, 728, 6.0), (728, 10, 3.0)):
            k[:, key] = q[:, row] * amp
        k[:, :64] = -q[:, 150:214].abs().mean(dim=(1, 2), keepdim=True) * torch.sign(q[:, 150:151]) * 2.0      # tile 0 strongly negative for row 150
        fused = torch.empty(B * S, 3 * H * hd, device="cuda", dtype=torch.bfloat16)
        fused[:, :H * hd] = bf(q).reshape(B * S, H * hd); fused[:, H * hd:2 * H * hd] = bf(k).reshape(B * S, H * hd); fused[:, 2 * H * hd:] = bf(v).reshape(B * S, H * hd)
        qv, kv, vv = fused, fused[:, H * hd:], fused[:, 2 * H * hd:]
        st = ((S * 3 * H * hd, 3 * H * hd),) * 3
        outs = {}
        for kern in (14, 11, 0):
            buf = torch.full((B * S + 64, H * hd), 7.0, device="cuda", dtype=torch.bfloat16)                 # 64 guard rows behind the last frame
            ops.attention(qv, kv, vv, B, H, H, S, S, hd, *st, kernel=kern, out=buf[:B * S])
            assert (buf[B * S:] == 7.0).all(), f"kernel {kern} wrote past the last row"
            outs[kern] = buf[:B * S].clone()
        assert torch.equal(outs[14], outs[11]) and torch.equal(outs[0], outs[11]), (B, H, int((outs[14] != outs[11]).sum()), (outs[14] != outs[11]).nonzero()[:4].tolist())
        for _ in range(20):                     # (a race shows up as a launch that differs from the one before it: this is how the round-5 mismatch was caught)
            assert torch.equal(ops.attention(qv, kv, vv, B, H, H, S, S, hd, *st, kernel=14), outs[11])
        qb, kb, vb = bf(q), bf(k), bf(v)
        # spikes of 7 x |q|^2 are scores of +-85 in the log2 domain: the one extra bf16 rounding of q' moves a dominant key's weight by ~1 % -- the oracle's own mirror of
        # this arithmetic is 1.2e-2 from fp32 on these inputs (ordinary inputs: test_attention at 729 tokens, <= ATTN_TOL); the tight bound is the mirror's, below
        assert torch.isfinite(outs[14].float()).all() and rel(outs[14], attn_ref(qb, kb, vb, False)) <= 3 * ATTN_TOL
        if B == 3:      # the oracle's tile-for-tile mirror of this arithmetic (what the tower's parity tests compare with), one frame
            with O.bf16_mirror():
                mir = O.attention_noncausal(qb[0].float().permute(1, 0, 2).cpu(), kb[0].float().permute(1, 0, 2).cpu(), vb[0].float().permute(1, 0, 2).cpu(), hd ** -0.5)
            got = outs[14][:S].float().reshape(S, H, hd).permute(1, 0, 2).cpu()
            assert (got - mir).abs().max() <= 2.0 ** -7 * mir.abs().max(), (got - mir).abs().max()           # the final division and bf16 rounding of O: one bf16 ulp of the largest element
    # a frame-to-frame dependence would be a bug of the ragged rows (wave 1 reads the NEXT frame's q rows unless the descriptor ends at row 728): frame 0 alone == frame 0 of the batch
    one = ops.attention(qv, kv, vv, 1, H, H, S, S, hd, *st, kernel=14)
    assert torch.equal(one, outs[14][:S])
    x3 = bf(g(1, S, 3, hd, seed=75))
    with pytest.raises(_lib.UfvError):                      # odd head count: the generated kernel refuses, AUTO takes the second generation
        ops.attention(x3, x3, x3, 1, 3, 3, S, S, hd, *(((S * 3 * hd, 3 * hd),) * 3), kernel=14)
    assert torch.equal(ops.attention(x3, x3, x3, 1, 3, 3, S, S, hd, *(((S * 3 * hd, 3 * hd),) * 3)),
                       ops.attention(x3, x3, x3, 1, 3, 3, S, S, hd, *(((S * 3 * hd, 3 * hd),) * 3), kernel=11))


def test_lab_only_attention_kernel_ids_are_refused_by_the_product_library():
    """ids 3, 4, 6, 7, 8, 10 name first-generation diagnostic forms: a product build (no -DUFV_LAB_KERNELS) must not dispatch them"""
    x = bf(g(1, 576, 4, 72, seed=76))
    for kern in (3, 4, 6, 7, 8, 10):
        with pytest.raises(_lib.UfvError, match="lab-only"):
            ops.attention(x, x, x, 1, 4, 4, 576, 576, 72, *(((576 * 4 * 72, 4 * 72),) * 3), kernel=kern)


def test_causal_hd128_prefill_kernel_bit_identical_to_the_plain_kernel():
    """the prefill kernel of csrc/attn_c128.inc (4 compute + 4 loader waves, generated instruction stream; kernel id 15 = what AUTO takes from S = 128 on):
    the arithmetic of the plain hd-128 kernel (id 13) operation for operation, so the outputs must be BIT-IDENTICAL -- for sequence lengths with every tail
    (one tile, 32-row slices that end inside a tile, a last slice with 1 valid row), group sizes 2..7, 2..8 kv heads, more items than CUs and fewer, a fused
    qkv layout (strided rows), and with spikes that drive the deferred-rescale path in middle, last (masked) and first tiles."""
    hd = 128
    for (Hq, Hkv, S, seed) in ((4, 2, 64, 70), (4, 2, 65, 71), (8, 2, 200, 72), (6, 2, 129, 73), (28, 4, 777, 74), (32, 8, 1000, 75), (14, 2, 1217, 76), (12, 4, 2399, 77), (10, 2, 96, 78), (8, 2, 4100, 80), (28, 4, 2399, 81)):
        W = (Hq + 2 * Hkv) * hd
        qkv = g(S, W, seed=seed)
        q, k = qkv[:, :Hq * hd].view(S, Hq, hd), qkv[:, Hq * hd:(Hq + Hkv) * hd].view(S, Hkv, hd)
        for row, key, amp in ((S - 1, S - 1, 3.0), (S - 1, S // 2, 5.0), (S // 2, S // 2 - 3, 4.0), (min(70, S - 1), 2, 6.0), (min(33, S - 1), min(31, S - 1), 5.0)):
            k[key, (row % Hkv)] = q[row, (row % Hkv) * (Hq // Hkv)] * amp / 4
        qkv = bf(qkv)
        qd, kd, vd = qkv[:, :Hq * hd], qkv[:, Hq * hd:], qkv[:, (Hq + Hkv) * hd:]
        st = ((0, W),) * 3
        new = ops.attention(qd, kd, vd, 1, Hq, Hkv, S, S, hd, *st, causal=True, kernel=15)
        old = ops.attention(qd, kd, vd, 1, Hq, Hkv, S, S, hd, *st, causal=True, kernel=13)
        assert torch.equal(new, old), (Hq, Hkv, S, int((new != old).sum()))
        if S >= 128:
            assert torch.equal(ops.attention(qd, kd, vd, 1, Hq, Hkv, S, S, hd, *st, causal=True), old)           # AUTO takes it
            # the training forward (same kernel + the log2-domain log-sum-exp per (head, query) written from its epilogue): same output bits, lse against fp32
            o2 = torch.empty(S, Hq * hd, device=DEV, dtype=torch.bfloat16)
            lse = torch.full((Hq, S), float("nan"), device=DEV)
            ops.attention_causal_lse(qd, kd, vd, o2, lse, S, Hq, Hkv, hd)
            assert torch.equal(o2, old)
            qf = qkv[:, :Hq * hd].float().view(S, Hq, hd)
            kf = qkv[:, Hq * hd:(Hq + Hkv) * hd].float().view(S, Hkv, hd).repeat_interleave(Hq // Hkv, 1)
            sc = torch.einsum("shd,thd->hst", qf, kf) * hd ** -0.5
            sc = sc.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool, device=DEV), 1), float("-inf"))
            assert (lse - torch.logsumexp(sc, -1) / math.log(2.0)).abs().max() < 2e-3
        ref = attn_ref(qkv[:, :Hq * hd].view(1, S, Hq, hd), qkv[:, Hq * hd:(Hq + Hkv) * hd].view(1, S, Hkv, hd), qkv[:, (Hq + Hkv) * hd:].view(1, S, Hkv, hd), True)
        assert torch.isfinite(new.float()).all() and rel(new, ref) <= 2 * ATTN_TOL
    # outside its envelope: refused by id 15, served by the other kernels under AUTO (MHA, one kv head, batch > 1, a query offset)
    x = bf(g(2, 130, 4, hd, seed=79))
    with pytest.raises(_lib.UfvError):
        ops.attention(x, x, x, 2, 4, 4, 130, 130, hd, *(((130 * 4 * hd, 4 * hd),) * 3), causal=True, kernel=15)
    o = ops.attention(x, x, x, 2, 4, 4, 130, 130, hd, *(((130 * 4 * hd, 4 * hd),) * 3), causal=True)
    assert rel(o, attn_ref(x, x, x, True)) <= ATTN_TOL


@pytest.mark.parametrize("B,H,Hkv,Sq,Sk", [(300, 4, 4, 16, 16), (300, 8, 8, 4, 16), (70, 2, 2, 16, 9), (65, 4, 2, 7, 16), (1, 2, 2, 1, 1)])
def test_attention_small_windows(B, H, Hkv, Sq, Sk):
    """csrc/attn.hip attn_win16_k (kernel id 16; what AUTO takes for head_dim 72, Sq, Sk <= 16, >= 256 (window, head) pairs): Hiera's 4 x 4 windows and their
    q-pooled form, one wave per pair, strided q / k / v views of a fused qkv buffer"""
    hd = 72
    W = (H + 2 * Hkv) * hd
    S = max(Sq, Sk)
    qkv = bf(g(B, S, W, seed=B + Sq + Sk))
    q, k, v = qkv[:, :Sq, :H * hd], qkv[:, :Sk, H * hd:(H + Hkv) * hd], qkv[:, :Sk, (H + Hkv) * hd:]
    st = (S * W, W)
    o = ops.attention(q, k, v, B, H, Hkv, Sq, Sk, hd, st, st, st, kernel=16)
    ref = attn_ref(q.reshape(B, Sq, H, hd), k.reshape(B, Sk, Hkv, hd), v.reshape(B, Sk, Hkv, hd), False)
    assert torch.isfinite(o.float()).all() and rel(o, ref) <= ATTN_TOL
    auto = ops.attention(q, k, v, B, H, Hkv, Sq, Sk, hd, st, st, st)
    if B * H >= 256:
        assert torch.equal(auto, o)
    else:
        assert rel(auto, ref) <= ATTN_TOL
    with pytest.raises(_lib.UfvError):
        ops.attention(q, k, v, B, H, Hkv, Sq, Sk, hd, st, st, st, causal=True, kernel=16)


def test_attention_spike_forces_rescale():
    # one key dominates late in the sequence -> running max jumps (online-softmax rescale path)
    B, H, S, hd = 1, 2, 256, 72
    q, k, v = g(B, S, H, hd, seed=30), g(B, S, H, hd, seed=31), g(B, S, H, hd, seed=32)
    k[:, 200] = q[:, 5] * 4.0
    q, k, v = bf(q), bf(k), bf(v)
    o = ops.attention(q, k, v, B, H, H, S, S, hd, (S * H * hd, H * hd), (S * H * hd, H * hd), (S * H * hd, H * hd), kernel=1)
    assert rel(o, attn_ref(q, k, v, False)) <= ATTN_TOL


def test_rope_kv():
    S, Hq, Hkv, hd, pos0 = 37, 4, 2, 128, 11
    qkv = bf(g(S, (Hq + 2 * Hkv) * hd, seed=33))
    inv = 1.0 / (1e6 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd)).to(DEV)
    cache = torch.zeros(64, 2 * Hkv * hd, device=DEV, dtype=torch.bfloat16)
    ref_in = qkv.float().clone()
    ops.rope_kv(qkv, S, Hq, Hkv, hd, inv, pos0, cache)
    pos = torch.arange(pos0, pos0 + S, device=DEV).float()
    ang = torch.cat([pos[:, None] * inv[None], pos[:, None] * inv[None]], -1)
    cos, sin = ang.cos()[:, None], ang.sin()[:, None]
    rot = lambda x: torch.cat([-x[..., hd // 2:], x[..., :hd // 2]], -1)
    qr = ref_in[:, :Hq * hd].view(S, Hq, hd); kr = ref_in[:, Hq * hd:(Hq + Hkv) * hd].view(S, Hkv, hd)
    assert rel(qkv[:, :Hq * hd].float().view(S, Hq, hd), qr * cos + rot(qr) * sin) <= ONE_ULP
    assert rel(cache[pos0:pos0 + S, :Hkv * hd].float().view(S, Hkv, hd), kr * cos + rot(kr) * sin) <= ONE_ULP
    assert torch.equal(cache[pos0:pos0 + S, Hkv * hd:], bf(ref_in[:, (Hq + Hkv) * hd:]))
    assert cache[:pos0].abs().sum() == 0


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_patchify(dt):
    T, C, H, P = 3, 3, 56, 14
    x = g(T, C, H, H, seed=34).to(dt)
    out = ops.patchify(x, P, 640)
    ref = torch.nn.functional.unfold(x.float(), P, stride=P).transpose(1, 2).reshape(T * 16, C * P * P)
    assert torch.equal(out[:, :588].float(), bf(ref).float())
    assert out[:, 588:].abs().sum() == 0
    # a size that is no multiple of the patch (SigLIP so400m at 384 px: 384 = 27 x 14 + 6): the stride-P convolution without padding drops the right / bottom remainder;
    # both forms of the kernel (dword runs for even widths, the scalar one for odd widths)
    for Hh, Ww in ((62, 62), (59, 61)):
        x = g(T, C, Hh, Ww, seed=35).to(dt)
        out = ops.patchify(x, P, 640)
        ref = torch.nn.functional.unfold(x.float(), P, stride=P).transpose(1, 2).reshape(T * (Hh // P) * (Ww // P), C * P * P)
        assert out.shape[0] == T * 16 and torch.equal(out[:, :588].float(), bf(ref).float()) and out[:, 588:].abs().sum() == 0


@pytest.mark.parametrize("F,H,W,C", [(3, 6, 6, 256), (2, 5, 7, 3584), (1, 24, 24, 896), (2, 3, 1, 64)])
def test_dwconv_ln_silu_and_se_pieces(F, H, W, C):
    x = bf(g(F, H, W, C, seed=35))
    w = g(C, 1, 3, 3, seed=36, scale=0.3)
    lnw, lnb = 1 + 0.1 * g(C, seed=37), 0.1 * g(C, seed=38)
    w9 = w.view(C, 9).t().contiguous()
    y = ops.dwconv3x3_ln_silu(x, w9, lnw, lnb, F, H, W, C, 1e-5)
    xc = x.float().permute(0, 3, 1, 2)
    r = torch.nn.functional.conv2d(xc, w, padding=1, groups=C).permute(0, 2, 3, 1)
    r = torch.nn.functional.silu(torch.nn.functional.layer_norm(r, (C,), lnw, lnb, 1e-5))
    assert rel(y, r) <= ONE_ULP
    m = ops.colmean(y.view(F * H * W, C), F, H * W)
    assert rel(m, y.float().view(F, H * W, C).mean(1)) <= ONE_ULP
    gate = bf(torch.sigmoid(g(F, C, seed=39)))
    y2 = y.clone().view(F * H * W, C)
    ops.scale_channels(y2, gate, F, H * W)
    assert rel(y2.view(F, H * W, C), y.float().view(F, H * W, C) * gate.float()[:, None]) <= ONE_ULP


@pytest.mark.parametrize("k,pad", [((2, 2, 2), 0), ((1, 2, 2), 1), ((2, 2, 2), 1)])
def test_conv3d_gather_matches_conv3d(k, pad):
    T, H, W, C, Co = 4, 6, 6, 16, 24
    x = bf(g(T, H, W, C, seed=40))
    w = bf(g(Co, C, *k, seed=41, scale=0.2))
    A, (To, Ho, Wo) = ops.conv3d_gather(x, T, H, W, C, k, pad)
    wp = w.permute(0, 2, 3, 4, 1).reshape(Co, -1).contiguous()
    out = A.float() @ wp.float().t()
    ref = torch.nn.functional.conv3d(x.float().permute(3, 0, 1, 2)[None], w.float(), stride=k, padding=pad)[0]
    assert rel(out, ref.permute(1, 2, 3, 0).reshape(To * Ho * Wo, Co)) < 1e-5


def test_gather_rows_and_argmax_and_convert():
    table = g(100, 64, seed=42)
    tb = bf(table)
    idx = torch.tensor([5, 99, 0, 5, 17], device=DEV)
    dst = torch.zeros(8, 64, device=DEV)
    didx = torch.tensor([7, 0, 3, 1, 2], device=DEV)
    ops.gather_rows(tb, idx, dst, didx)
    assert torch.equal(dst[didx], tb[idx].float()) and dst[4:7].abs().sum() == 0
    x = g(151748, seed=43)
    x[1234] = 50.0; x[99999] = 50.0
    assert ops.argmax(x).item() == 1234 == torch.argmax(x).item()
    assert ops.argmax(g(7, seed=44)).item() == torch.argmax(g(7, seed=44)).item()
    for n in (4097, 16384, 151747):                          # vector body + scalar tail, ties across threads, unaligned views
        y = g(n + 1, seed=45 + n)
        for view in (y[:n], y[1:]):
            assert ops.argmax(view.contiguous()).item() == torch.argmax(view).item()
            assert ops.argmax(view).item() == torch.argmax(view).item()
        z = torch.zeros(n, device=DEV); z[n - 1] = 1.0; z[n // 2] = 1.0
        assert ops.argmax(z).item() == n // 2
        z[5] = float("nan")
        assert ops.argmax(z).item() == 5
    # the 64-block form of the decode step (partials + last-arriving block merges): same answers, workspace reusable without re-zeroing
    ws = torch.zeros(_lib.load().ufv_argmax_ws_bytes(), device=DEV, dtype=torch.uint8)
    assert ops.argmax(x, ws=ws).item() == 1234
    for n in (7, 255, 4097, 16384, 151747, 151748, 262144):
        y = g(n, seed=145 + n)
        assert ops.argmax(y, ws=ws).item() == torch.argmax(y).item()
        z = torch.zeros(n, device=DEV); z[n - 1] = 1.0; z[n // 2] = 1.0
        assert ops.argmax(z, ws=ws).item() == n // 2
        z[min(5, n - 1)] = float("nan")
        assert ops.argmax(z, ws=ws).item() == min(5, n - 1)
    assert int(ws.view(torch.int32)[-1]) == 0
    # gather: the vector form (aligned rows, D % 8 == 0), few rows and many, every dtype pair
    for n_rows in (1, 5, 300):
        for sd, dd in [(torch.bfloat16, torch.float32), (torch.bfloat16, torch.bfloat16), (torch.float32, torch.float32), (torch.float16, torch.bfloat16), (torch.float32, torch.bfloat16)]:
            tab = g(400, 3584, seed=7).to(sd)
            sidx = torch.randint(0, 400, (n_rows,), device=DEV); sidx[0] = 399
            out = torch.full((n_rows, 3584), 7.0, device=DEV, dtype=dd)
            ops.gather_rows(tab, sidx, out, None)
            assert torch.equal(out, tab[sidx].to(dd)), (n_rows, sd, dd)
    for a, b_ in [(torch.float32, torch.bfloat16), (torch.float16, torch.bfloat16), (torch.bfloat16, torch.float32)]:
        t = g(1000, seed=45).to(a)
        assert torch.equal(ops.convert(t, b_), t.to(b_))


def test_mask_pool_and_preprocess():
    n, P, C, q = 3, 16, 64, 4
    feat = g(n, P, C, seed=46)
    mask = (g(q, P, seed=47) > 0).float()
    fo = torch.tensor([0, 2, 1, 2], device=DEV, dtype=torch.int32)
    out = ops.mask_pool(feat, mask, fo)
    ref = torch.stack([(feat[fo[i]] * mask[i][:, None] / (mask[i].sum() + 1e-8)).sum(0) for i in range(q)])
    assert rel(out, ref) < 1e-5
    fr = torch.randint(0, 256, (2, 8, 10, 3), dtype=torch.uint8, device=DEV)
    o = ops.preprocess_u8(fr, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
    ref = ((fr.float() / 255.0 - 0.5) / 0.5).permute(0, 3, 1, 2)
    assert rel(o, ref) <= ONE_ULP


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (512, 384, 192), (1000, 768, 1152), (2399, 1024, 448),
                                   (300, 128, 64), (4096, 3456, 256)])
def test_gemm256_bitwise_equals_128_kernel(M, N, K):
    """The 256x256 ping-pong kernel accumulates k in the same order with the same MFMA as the 128-wide kernel, so
    any difference is a staging race / wrong tile: compare bit for bit, several launches."""
    a, w = bf(g(M, K, seed=51)), bf(g(N, K, seed=52, scale=0.05))
    bias, resid = g(N, seed=53), g(M, N, seed=54)
    ref = ops.gemm(a, w, bias=bias, act="gelu_tanh", resid=resid, out_dtype=torch.float32, kernel=ops.GEMM_FAST)
    for _ in range(5):
        out = ops.gemm(a, w, bias=bias, act="gelu_tanh", resid=resid, out_dtype=torch.float32, kernel=ops.GEMM_FAST256)
        assert torch.equal(out, ref)
    refb = ops.gemm(a, w, kernel=ops.GEMM_FAST)
    assert torch.equal(ops.gemm(a, w, kernel=ops.GEMM_FAST256), refb)
    assert rel(refb, a.float() @ w.float().t()) <= ONE_ULP


PP_SHAPES = {1442: (256, 256), 1432: (224, 256), 1332: (192, 256), 1322: (160, 256), 1441: (256, 192), 1431: (224, 192), 1331: (192, 192)}


@pytest.mark.parametrize("shape", sorted(PP_SHAPES))
def test_gemm_pingpong_tile_shapes_bitwise_equal_128_kernel(shape):
    """Every tile shape of the ping-pong kernel (UFV_GEMM_PP: unequal A halves of 96 / 64 rows, a 64-column second B half, the two-phase
    schedule) accumulates k in the same order as the 128-wide kernel: bit-equal results, ragged M, several rounds of the persistent
    loop, an N that ends on the first half-tile (N % tile width == 128), every epilogue form; repeated launches catch staging races."""
    bm, bn = PP_SHAPES[shape]
    kern = ops.GEMM_FAST256 | (shape << 8)
    full, ragged = (768, 640) if bn == 256 else (1152, 512)           # N % 128 == 0 always (the reference kernel's tile); ragged: ends on the 128-column half
    for M, N, K in ((bm * 3 + 37, full, 192), (2399, ragged, 448), (max(bm, 256), 2 * bn if bn == 192 else bn, 64), (300, full * 3, 1152)):
        a, w = bf(g(M, K, seed=shape + 1)), bf(g(N, K, seed=shape + 2, scale=0.05))
        bias, resid = g(N, seed=shape + 3), g(M, N, seed=shape + 4)
        ref = ops.gemm(a, w, bias=bias, act="gelu_tanh", resid=resid, out_dtype=torch.float32, kernel=ops.GEMM_FAST)
        for _ in range(3):
            assert torch.equal(ops.gemm(a, w, bias=bias, act="gelu_tanh", resid=resid, out_dtype=torch.float32, kernel=kern), ref), (shape, M, N, K)
        assert torch.equal(ops.gemm(a, w, kernel=kern), ops.gemm(a, w, kernel=ops.GEMM_FAST))
        x1 = resid.clone()
        ops.gemm(a, w, resid=x1, out=x1, kernel=kern)                      # in-place residual stream, as the layers use it
        assert torch.equal(x1, ops.gemm(a, w, resid=resid, out_dtype=torch.float32, kernel=ops.GEMM_FAST))
    if bn == 192:                                                              # a width the shape cannot end on: refused, not mis-tiled
        with pytest.raises(_lib.UfvError, match="tile needs N"):
            ops.gemm(bf(g(256, 64, seed=1)), bf(g(192 + 64, 64, seed=2)), kernel=kern)               # 256 % 192 = 64
    if shape != 1442:
        with pytest.raises(_lib.UfvError, match="SwiGLU"):
            ops.gemm(bf(g(256, 64, seed=1)), bf(g(512, 64, seed=2)), swiglu=True, kernel=kern)


@pytest.mark.parametrize("shape", [1331, 1431, 1441])
def test_gemm_tile_shapes_with_every_cu_streaming_long_k(shape):
    """The fragment-prefetch schedule (gemm256_kernel.h FPF) reads LDS one barrier after the load step's wait; its first form let the leading wave group read
    pieces the lagging group had not waited for yet -- bit-equal on every short-K case above, wrong and varying sums only with a full grid (247 tiles) streaming a
    long K from HBM.  This is that case, at the forced shapes and their 4-part split forms: bit-equal to the 128-wide kernel / deterministic, three launches each."""
    M, N = 2399, 3584
    for K in (8192, 18944):
        a, w = bf(g(M, K, seed=M % 97)), bf(g(N, K, seed=N % 89, scale=0.05))
        resid = g(M, N, seed=5)
        want = ops.gemm(a, w, resid=resid, out_dtype=torch.float32, kernel=ops.GEMM_FAST)
        kern = ops.GEMM_FAST256 | (shape << 8)
        for _ in range(3):
            assert torch.equal(ops.gemm(a, w, resid=resid, out_dtype=torch.float32, kernel=kern), want), (shape, K)
        split = ops.GEMM_FAST256 | ((shape + 40000) << 8)
        outs = [ops.gemm(a, w, resid=resid, out_dtype=torch.float32, kernel=split) for _ in range(3)]
        assert rel(outs[0], want) < 4e-6 and torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (shape, K)


@pytest.mark.parametrize("shape,parts", [(1441, 4), (1442, 5), (1442, 2), (1432, 3), (1332, 7), (1322, 8), (1431, 6), (1331, 5)])
def test_gemm_split_k_turn_ordered_sum(shape, parts):
    """Aligned split-K of the ping-pong kernel (UFV_GEMM_PP(shape + 10000 * parts)): the parts of a tile add into the fp32 output in turn order,
    so the result is deterministic (repeated launches bit-equal) and equals the unsplit product up to the different summation tree (parts partial
    sums instead of one chain: <= 4e-6 of the output range); in place on the residual stream and with a separate residual + bias; ragged M,
    an N ending on a half tile, more items than CUs (several rounds) and fewer; refused with an activation, a bf16 output or empty parts."""
    bm, bn = PP_SHAPES[shape]
    kern = ops.GEMM_FAST256 | ((shape + 10000 * parts) << 8)
    full, ragged = (768, 640) if bn == 256 else (1152, 512)
    for M, N, K in ((2399, full, 64 * parts * 3), (bm * 2 + 5, ragged, 64 * (parts * 3 - 1)), (300, full * 3, 64 * parts * 2)):
        a, w = bf(g(M, K, seed=shape + parts)), bf(g(N, K, seed=shape + 2, scale=0.05))
        bias, resid = g(N, seed=3), g(M, N, seed=4)
        ref = ops.gemm(a, w, bias=bias, resid=resid, out_dtype=torch.float32, kernel=ops.GEMM_FAST)
        outs = []
        for _ in range(3):
            x = resid.clone()
            ops.gemm(a, w, bias=bias, resid=x, out=x, kernel=kern)                  # in place, as the decoder calls it
            outs.append(x)
        assert rel(outs[0], ref) < 4e-6, (shape, parts, M, N, K, rel(outs[0], ref))
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        y = torch.full((M, N), float("nan"), device=DEV)                              # out of place: every element is written by part 0
        ops.gemm(a, w, bias=bias, resid=resid, out=y, kernel=kern)
        assert torch.equal(y, outs[0])
        assert torch.equal(ops.gemm(a, w, out_dtype=torch.float32, kernel=kern), ops.gemm(a, w, out_dtype=torch.float32, kernel=kern))
    a, w = bf(g(512, 64 * parts * 2, seed=1)), bf(g(full, 64 * parts * 2, seed=2))
    with pytest.raises(_lib.UfvError, match="split-K"):
        ops.gemm(a, w, act="gelu", out_dtype=torch.float32, kernel=kern)
    with pytest.raises(_lib.UfvError, match="split-K"):
        ops.gemm(a, w, kernel=kern)                                                   # bf16 output
    with pytest.raises(_lib.UfvError, match="non-empty"):
        ops.gemm(bf(g(512, 64 * (parts - 1), seed=1)), bf(g(full, 64 * (parts - 1), seed=2)), out_dtype=torch.float32, kernel=kern)


def test_gemm_auto_choice_matches_every_kernel_it_can_pick():
    """GEMM_AUTO routes by a cost model (csrc/gemm.hip choose_kernel); whatever it picks at the config-#2 shapes and around them, the
    result is the 128-wide kernel's, bit for bit."""
    for M, N, K, f32 in ((2399, 3584, 512, True), (2399, 4608, 256, False), (18432, 1152, 128, True), (4703, 3584, 192, True), (2304, 3584, 128, False),
                         (1000, 1152, 1152, True), (257, 128, 64, False), (2399, 3584, 18944, True), (300, 3584, 18944, True)):
        a, w = bf(g(M, K, seed=M % 97)), bf(g(N, K, seed=N % 89, scale=0.05))
        if f32 and K >= 4096:                        # long K with few tiles: the model may take a split-K form (a different summation tree)
            resid = g(M, N, seed=5)
            got, want = ops.gemm(a, w, resid=resid, out_dtype=torch.float32), ops.gemm(a, w, resid=resid, out_dtype=torch.float32, kernel=ops.GEMM_FAST)
            assert rel(got, want) < 4e-6 and torch.equal(got, ops.gemm(a, w, resid=resid, out_dtype=torch.float32))
        elif f32:
            resid = g(M, N, seed=5)
            assert torch.equal(ops.gemm(a, w, resid=resid, out_dtype=torch.float32), ops.gemm(a, w, resid=resid, out_dtype=torch.float32, kernel=ops.GEMM_FAST))
        else:
            assert torch.equal(ops.gemm(a, w), ops.gemm(a, w, kernel=ops.GEMM_FAST))


def test_gemm256_swiglu_and_layout():
    M, I, K = 700, 512, 256
    a = bf(g(M, K, seed=55))
    gate, up = bf(g(I, K, seed=56, scale=0.1)), bf(g(I, K, seed=57, scale=0.1))
    wp = pack_swiglu(gate, up)
    ref = ops.gemm(a, wp, swiglu=True, out_dtype=torch.float32, kernel=ops.GEMM_FAST)
    assert torch.equal(ops.gemm(a, wp, swiglu=True, out_dtype=torch.float32, kernel=ops.GEMM_FAST256), ref)
    eye = bf(torch.eye(256, 256, device=DEV))
    w = bf((torch.arange(256, device=DEV)[:, None] * 3 + torch.arange(256, device=DEV)[None, :] * 0.5) % 17)
    assert torch.equal(ops.gemm(eye, w, out_dtype=torch.float32, kernel=ops.GEMM_FAST256), w.float().t().contiguous())


@pytest.mark.parametrize("M,N2,K", [(2399, 37888, 256), (2799, 37888, 192), (2399, 37888, 3584), (1100, 37888, 128), (4703, 37888, 128)])
def test_gemm256_swiglu_last_round_of_half_tile_items_bit_identical(M, N2, K):
    """Round 6 (csrc/gemm256_kernel.h HALF, opt-in by UFV_GEMM_HALF=1 -- measured at -1.2 % for +14 % traffic, so not the default): the decoder's gate/up projection can deal
    the last round of its launch as half-tile items -- M = 2399: 5 rounds of whole tiles + 2 x 52 halves of leftover tiles + the 148 tiles of the 95-row band; M = 2799
    (384 px): 6 rounds + 2 x 92 halves, no light band; M = 1100: fewer whole tiles than CUs beside a light band; M = 4703 (64 frames): the items do not fit one round, the
    launch stays as it was.  Every element is the same sum in the same order: BIT-identical to the launch of whole tiles and to the 128-wide kernel; rows behind the
    output stay untouched (a half item's A1 stores are dropped, not masked)."""
    a = bf(g(M, K, seed=90))
    wp = bf(g(N2, K, seed=91, scale=0.1))
    whole = ops.gemm(a, wp, swiglu=True, kernel=ops.GEMM_FAST256)
    out = torch.full((M + 8, N2 // 2), 3.0, device=DEV, dtype=torch.bfloat16)
    os.environ["UFV_GEMM_HALF"] = "1"
    try:
        ops.gemm(a, wp, swiglu=True, out=out[:M], kernel=ops.GEMM_FAST256)
        assert (out[M:] == 3.0).all()
        assert torch.equal(out[:M], whole)
        assert torch.equal(ops.gemm(a, wp, swiglu=True), whole)                  # AUTO
        for _ in range(3):                      # (a second and third launch: the persistent blocks' item order does not depend on what ran before)
            assert torch.equal(ops.gemm(a, wp, swiglu=True, kernel=ops.GEMM_FAST256), whole)
    finally:
        del os.environ["UFV_GEMM_HALF"]
    if K <= 256:
        assert torch.equal(whole, ops.gemm(a, wp, swiglu=True, kernel=ops.GEMM_FAST))


@pytest.mark.parametrize("M,N,K,f32", [(4096, 576, 576, False), (4900, 1728, 576, False), (2048, 320, 320, False), (3000, 960, 320, False), (8192, 192, 192, False),
                                         (4096, 576, 2304, True), (2048, 320, 1152, True)])
def test_gemm_widths_that_are_multiples_of_192_or_128_past_one(M, N, K, f32):
    """Round 6: an N that is no multiple of 128 but N % 192 in {0, 128} (Hiera-L's stage widths 192 / 320 / 576 and their q|k|v triples) is served by the ping-pong kernel's
    192-wide tile shapes (csrc/gemm.hip launch_any) instead of the generic kernel.  Same sums in the same order as the zero-padded call the model used to make (N and K padded
    to the next multiple of 128): BIT-identical on the real columns, and right against fp32."""
    a = bf(g(M, K, seed=95))
    w = bf(g(N, K, seed=96, scale=0.05))
    bias = g(N, seed=97)
    Np, Kp = -(-N // 128) * 128, -(-K // 128) * 128
    ap = torch.zeros(M, Kp, device=DEV, dtype=torch.bfloat16); ap[:, :K] = a
    wp = torch.zeros(Np, Kp, device=DEV, dtype=torch.bfloat16); wp[:N, :K] = w
    bp = torch.zeros(Np, device=DEV); bp[:N] = bias
    if f32:
        x = g(M, N, seed=98)
        xp = torch.zeros(M, Np, device=DEV); xp[:, :N] = x
        got = ops.gemm(a, w, bias=bias, resid=x.clone(), out_dtype=torch.float32)
        pad = ops.gemm(ap, wp, bias=bp, resid=xp, out_dtype=torch.float32)[:, :N]
        ref = a.float() @ w.float().t() + bias + x
        assert rel(got, ref) <= 2e-5
    else:
        got = ops.gemm(a, w, bias=bias, act="gelu")
        pad = ops.gemm(ap, wp, bias=bp, act="gelu")[:, :N]
        ref = torch.nn.functional.gelu(a.float() @ w.float().t() + bias)
        assert rel(got, ref) <= ONE_ULP
    assert torch.equal(got, pad)


def test_rope_kv_scalar_path():
    # head_dim 24 (hd/2 not a multiple of 8) takes the scalar kernel
    S, Hq, Hkv, hd, pos0 = 5, 2, 1, 24, 3
    qkv = bf(g(S, (Hq + 2 * Hkv) * hd, seed=60))
    inv = 1.0 / (1e4 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd)).to(DEV)
    cache = torch.zeros(16, 2 * Hkv * hd, device=DEV, dtype=torch.bfloat16)
    ref_in = qkv.float().clone()
    ops.rope_kv(qkv, S, Hq, Hkv, hd, inv, pos0, cache)
    pos = torch.arange(pos0, pos0 + S, device=DEV).float()
    ang = torch.cat([pos[:, None] * inv[None], pos[:, None] * inv[None]], -1)
    cos, sin = ang.cos()[:, None], ang.sin()[:, None]
    rot = lambda x: torch.cat([-x[..., hd // 2:], x[..., :hd // 2]], -1)
    qr = ref_in[:, :Hq * hd].view(S, Hq, hd)
    assert rel(qkv[:, :Hq * hd].float().view(S, Hq, hd), qr * cos + rot(qr) * sin) <= ONE_ULP
    assert torch.equal(cache[pos0:pos0 + S, Hkv * hd:], bf(ref_in[:, (Hq + Hkv) * hd:]))


@pytest.mark.parametrize("hd,Hq,Hkv,pos,nsplit", [(128, 28, 4, 2399, 16), (128, 4, 2, 4, 16), (128, 4, 4, 0, 16), (64, 8, 2, 700, 7), (64, 4, 4, 63, 1), (128, 14, 2, 300, 16)])
def test_attention_decode_fused_bit_identical_to_the_three_calls(hd, Hq, Hkv, pos, nsplit):
    """`ufv_attention_decode_fused` (RoPE of q / k, KV append, split attention, merge in one launch) == ufv_rope_kv1_dev -> ufv_attention_decode_dev (split +
    combine): the output row and the appended cache row bit for bit, position from the host and from device memory, twice in a row on the same workspace
    (the arrival counters return to zero), splits without keys (pos < nsplit), one split, GQA"""
    import ctypes
    W = (Hq + 2 * Hkv) * hd
    qkv = bf(g(1, W, seed=hd + pos))
    cache = bf(g(pos + 9, 2 * Hkv * hd, seed=pos + 1))
    inv_freq = (1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    scale = hd ** -0.5
    # the three-call sequence
    q1, c1 = qkv.clone(), cache.clone()
    pos_dev = torch.tensor([pos], dtype=torch.int32, device=DEV)
    ws1 = torch.empty(_lib.load().ufv_attention_decode_ws_bytes(1, Hq, hd, nsplit), device=DEV, dtype=torch.uint8)
    o1 = torch.empty(1, Hq * hd, device=DEV, dtype=torch.bfloat16)
    _lib.call("ufv_rope_kv1_dev", q1.data_ptr(), Hq, Hkv, hd, inv_freq.data_ptr(), pos_dev.data_ptr(), c1.data_ptr(), c1.stride(0), st)
    _lib.call("ufv_attention_decode_dev", q1.data_ptr(), 0, c1.data_ptr(), 0, c1.stride(0), c1[:, Hkv * hd:].data_ptr(), 0, c1.stride(0), o1.data_ptr(), 0, 1,
              Hq, Hkv, pos_dev.data_ptr(), c1.shape[0], hd, scale, ws1.data_ptr(), nsplit, st)
    nb = _lib.load().ufv_attention_decode_fused_ws_bytes(Hq, hd, nsplit)
    ws2 = torch.zeros(nb, device=DEV, dtype=torch.uint8)
    for use_dev in (False, True, False):
        q2, c2 = qkv.clone(), cache.clone()
        o2 = torch.full((1, Hq * hd), float("nan"), device=DEV, dtype=torch.bfloat16)
        _lib.call("ufv_attention_decode_fused", q2.data_ptr(), Hq, Hkv, hd, inv_freq.data_ptr(), pos, pos_dev.data_ptr() if use_dev else None, c2.data_ptr(),
                  c2.stride(0), c2.shape[0], o2.data_ptr(), scale, ws2.data_ptr(), nsplit, st)
        assert torch.equal(o2, o1), (use_dev, rel(o2, o1.float()))
        assert torch.equal(c2, c1) and torch.equal(q2, qkv)              # the row appended at `pos`; qkv untouched
        assert int(ws2.view(torch.int32)[-Hq:].abs().max()) == 0          # counters back at zero
    with pytest.raises(_lib.UfvError, match="head_dim"):
        _lib.call("ufv_attention_decode_fused", qkv.data_ptr(), Hq, Hkv, 72, inv_freq.data_ptr(), pos, None, cache.data_ptr(), cache.stride(0), cache.shape[0],
                  o1.data_ptr(), scale, ws2.data_ptr(), nsplit, st)


@pytest.mark.parametrize("hd,Hq,Hkv,Sk,nsplit", [(128, 28, 4, 2400, 16), (128, 4, 2, 5, 16), (16, 4, 2, 61, 3), (64, 8, 8, 700, 1)])
def test_attention_decode(hd, Hq, Hkv, Sk, nsplit):
    q = bf(g(1, 1, Hq, hd, seed=70))
    kv = bf(g(Sk + 3, 2 * Hkv * hd, seed=71))                 # cache rows [k | v]
    o = ops.attention_decode(q.view(1, Hq * hd), kv, kv[:, Hkv * hd:], Hq, Hkv, Sk, hd, kv.stride(0), kv.stride(0), nsplit=nsplit)
    k = kv[:Sk, :Hkv * hd].view(1, Sk, Hkv, hd); v = kv[:Sk, Hkv * hd:].view(1, Sk, Hkv, hd)
    assert rel(o, attn_ref(q, k, v, True, Sk - 1)) <= ATTN_TOL


@pytest.mark.parametrize("M,N,K", [(2000, 1024, 4096), (2399, 3584, 18944), (2399, 4608, 3584), (18432, 1152, 1152), (515, 768, 8192),
                                   (4096, 3456, 256), (300, 256, 2560)])
def test_gemm_streamk_vs_tile_kernel(M, N, K):
    """Stream-K work split of the 256-wide kernel: tiles are cut at arbitrary K-tile boundaries and summed through the fp32
    workspace in a fixed order -> equal to the tile kernel up to fp32 summation order, and bit-reproducible run to run."""
    a, w = bf(g(M, K, seed=61)), bf(g(N, K, seed=62, scale=0.05))
    bias, resid = g(N, seed=63), g(M, N, seed=64)
    ref = ops.gemm(a, w, bias=bias, act="gelu_tanh", resid=resid, out_dtype=torch.float32, kernel=ops.GEMM_FAST256)
    outs = [ops.gemm(a, w, bias=bias, act="gelu_tanh", resid=resid, out_dtype=torch.float32, kernel=ops.GEMM_STREAMK) for _ in range(4)]
    assert rel(outs[0], ref) < 2e-6 * (K / 64) ** 0.5 + 1e-6
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
    refb = ops.gemm(a, w, kernel=ops.GEMM_FAST256)
    ob = ops.gemm(a, w, kernel=ops.GEMM_STREAMK)
    assert rel(ob, refb) <= ONE_ULP and torch.equal(ob, ops.gemm(a, w, kernel=ops.GEMM_STREAMK))
    # in-place fp32 residual stream (the decoder's o_proj / down_proj form)
    x1, x2 = resid.clone(), resid.clone()
    ops.gemm(a, w, resid=x1, out=x1, kernel=ops.GEMM_FAST256)
    ops.gemm(a, w, resid=x2, out=x2, kernel=ops.GEMM_STREAMK)
    assert rel(x2, x1) < 2e-6 * (K / 64) ** 0.5 + 1e-6


def test_gemm_streamk_fp8_and_errors():
    a, w = bf(g(2399, 3584, seed=65)), bf(g(1280, 3584, seed=66, scale=0.05))
    W8 = ops.Fp8Weight(w)
    aq, sa = ops.quantize_fp8(a)
    ref = ops.gemm_fp8(aq, sa, W8, out_dtype=torch.float32, kernel=ops.GEMM_FAST256)
    out = ops.gemm_fp8(aq, sa, W8, out_dtype=torch.float32, kernel=ops.GEMM_STREAMK)
    assert rel(out, ref) < 2e-5
    from ufvideo_amd import _lib
    with pytest.raises(_lib.UfvError):
        ops.gemm(a, bf(g(512, 3584, seed=67)), swiglu=True, kernel=ops.GEMM_STREAMK)


@pytest.mark.parametrize("N,K", [(4608, 3584), (3584, 18944), (300, 64), (151748, 3584)])
def test_gemv1_fused_rmsnorm_bit_identical(N, K):
    """Decode GEMV: fp32 row + fused RMSNorm == ufv_rmsnorm then the bf16-row GEMV, bit for bit; and both match the
    batched GEMV kernel up to fp32 summation order."""
    w = bf(g(N, K, seed=73, scale=0.05))
    bias, resid = g(N, seed=74), g(N, seed=75)
    if K > 4096:                                   # down_proj form: bf16 row in, no norm (ufv_rmsnorm covers D <= 4096)
        h = bf(g(1, K, seed=71))
        y = ops.gemv1(w, a=h[0], bias=bias, resid=resid, out_dtype=torch.float32)
        ref = ops.gemm(h, w, bias=bias, resid=resid[None].contiguous(), out_dtype=torch.float32, kernel=ops.GEMM_GEMV)[0]
        assert rel(y, ref) < 1e-5
        return
    x = g(1, K, seed=71) * 2
    gw = 1 + 0.1 * g(K, seed=72)
    h = ops.rmsnorm(x, gw, 1e-6)
    y_sep = ops.gemv1(w, a=h[0], bias=bias, resid=resid, out_dtype=torch.float32)
    y_fused = ops.gemv1(w, x=x[0].contiguous(), ln_w=gw, eps=1e-6, bias=bias, resid=resid, out_dtype=torch.float32)
    assert torch.equal(y_sep, y_fused)
    ref = ops.gemm(h, w, bias=bias, resid=resid[None].contiguous(), out_dtype=torch.float32, kernel=ops.GEMM_GEMV)[0]
    assert rel(y_fused, ref) < 1e-5
    if N % 32 == 0:
        yb = ops.gemv1(w, x=x[0].contiguous(), ln_w=gw, eps=1e-6, swiglu=True)
        refb = ops.gemm(h, w, swiglu=True, kernel=ops.GEMM_GEMV)[0]
        assert rel(yb, refb) <= ONE_ULP


def test_device_frame_batching_matches_pillow_bit_for_bit():
    """u8 HWC frames -> Pillow-exact bicubic resize on the GPU -> normalise/layout; resize output equals PIL's byte for byte."""
    import numpy as np
    from PIL import Image
    from ufvideo_amd.mm_utils import UfvImageProcessor
    rng = np.random.default_rng(5)
    for T, H, W, S in [(3, 360, 640, 336), (2, 224, 224, 336), (2, 100, 37, 56), (1, 480, 270, 384), (2, 56, 56, 56), (1, 56, 90, 56)]:
        fr = rng.integers(0, 256, (T, H, W, 3), dtype=np.uint8)
        ref = np.stack([np.asarray(Image.fromarray(f).resize((S, S), resample=Image.BICUBIC)) for f in fr])
        got = ops.resize_bicubic_u8(torch.from_numpy(fr).to(DEV), S, S)
        assert torch.equal(got.cpu(), torch.from_numpy(ref)), (T, H, W, S)
        proc = UfvImageProcessor(size=S)
        want = proc.preprocess(list(fr))["pixel_values"]                       # the reference-style CPU path (PIL + float32)
        dev = proc.preprocess_device(fr, device=DEV)
        assert dev.shape == want.shape and dev.dtype == torch.bfloat16
        assert (dev.float().cpu() - want).abs().max() <= 2 ** -7              # bf16 rounding of values in [-1, 1]
    # process_video(device=...): same frames through the opt-in device path vs the reference-style CPU path
    from ufvideo_amd.mm_utils import process_video
    frames = rng.integers(0, 256, (5, 90, 160, 3), dtype=np.uint8)
    proc = UfvImageProcessor(size=56)
    v_cpu, f_cpu, h, w, _ = process_video(frames, proc, aspect_ratio="pad", num_frames=4, frame_idx=[1, 3])
    v_dev, f_dev, h2, w2, _ = process_video(frames, proc, aspect_ratio="pad", num_frames=4, frame_idx=[1, 3], device=DEV)
    assert (h, w) == (h2, w2) and v_dev.is_cuda and v_dev.shape == v_cpu.shape and f_dev.shape == f_cpu.shape
    assert (v_dev.float().cpu() - v_cpu).abs().max() <= 2 ** -7 and (f_dev.float().cpu() - f_cpu).abs().max() <= 2 ** -7


def test_rope_table_form_is_bit_identical():
    g = torch.Generator().manual_seed(77)
    S, Hq, Hkv, hd, pos0 = 333, 6, 2, 128, 17
    inv = (1.0 / (1e6 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))).to(DEV)
    qkv = torch.randn(S, (Hq + 2 * Hkv) * hd, generator=g).to(torch.bfloat16).to(DEV)
    a, b = qkv.clone(), qkv.clone()
    ka = torch.zeros(pos0 + S, 2 * Hkv * hd, device=DEV, dtype=torch.bfloat16); kb = ka.clone()
    ops.rope_kv(a, S, Hq, Hkv, hd, inv, pos0, ka)
    ops.rope_kv(b, S, Hq, Hkv, hd, inv, pos0, kb, table=ops.rope_table(inv, pos0, S, hd))
    assert torch.equal(a, b) and torch.equal(ka, kb)


@pytest.mark.parametrize("V,T,k,p", [(1000, 0.2, 50, 0.9), (151748, 0.7, 0, 0.8), (5000, 1.3, 20, 1.0), (300, 0.2, 0, 0.9), (4096, 1.0, 1, 0.5)])
def test_sample_top_p_vs_oracle_distribution(V, T, k, p):
    """kept set (count above the cut-off, kept mass) exact vs the oracle's HF-warper restatement; draws by inverse CDF follow it"""
    from oracle import ref_cpu as O
    g = torch.Generator().manual_seed(V + k)
    lg = torch.randn(V, generator=g) * 3
    ref = O.sampling_distribution(lg, T, k, p)
    ref_k = O.sampling_distribution(lg, T, k, 1.0)
    n = 4096
    u = torch.rand(n, generator=g)
    u[0], u[1] = 0.0, 1.0 - 2 ** -24
    rows = lg.to(DEV)[None].expand(n, V).contiguous()
    kept = torch.zeros(n, 2, device=DEV)
    toks = ops.sample_top_p(rows, T, k, p, u.to(DEV), kept=kept).cpu()
    kept = kept.cpu()
    assert int((lg >= kept[0, 1]).sum()) == int((ref > 0).sum())                       # the cut-off keeps exactly HF's set
    assert abs(float(kept[0, 0]) - float(ref_k[ref > 0].sum())) < 1e-4
    assert bool((ref[toks] > 0).all())
    exp_tok = torch.tensor([O.sample_inverse_cdf(ref, float(x)) for x in u[:256]])
    assert (toks[:256] == exp_tok).float().mean() > 0.98                                 # fp32 vs fp64 running sums at bin edges
    emp = torch.bincount(toks, minlength=V).float() / n
    assert 0.5 * float((emp - ref).abs().sum()) < 0.05 + 0.5 * float((ref > 0).sum()) / n


@pytest.mark.parametrize("S,pos0,Hq,Hkv", [(256, 0, 28, 4), (383, 5, 28, 4), (2399, 0, 28, 4), (1000, 37, 6, 1), (640, 0, 3, 2)])
def test_fused_qkv_rope_kv_append_bit_identical_to_gemm_then_rope(S, pos0, Hq, Hkv):
    """ufv_gemm_qkv_rope (round 5: q / k / v Linear + bias, RoPE and the KV-cache append in ONE launch, csrc/gemm256_kernel.h epilogue256_rope) against
    the unfused pair ufv_gemm + ufv_rope_kv_table on the same operands: torch.equal on the rotated q heads and on the cache rows, rows outside
    [pos0, pos0 + S) untouched.  Shapes: the decoder's (28 / 4 heads of 128, S = 2399 -> 13 row tiles, the last one ragged), an S one past a tile
    boundary, a position offset, and head counts that leave half a 256-column tile past N ((6 + 2) * 128 = 1024 is even; (3 + 4) * 128 = 896 is not)."""
    hd, K = 128, 3584 if Hq == 28 else 512
    g = torch.Generator().manual_seed(S + pos0)
    N = (Hq + 2 * Hkv) * hd
    a = (torch.randn(S, K, generator=g)).to(torch.bfloat16).to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    inv_freq = (1.0 / (1e6 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))).to(DEV)
    table = ops.rope_table(inv_freq, pos0, S, hd)
    rows = pos0 + S + 3
    kv0 = torch.full((rows, 2 * Hkv * hd), 7.0, device=DEV, dtype=torch.bfloat16)
    kv1 = kv0.clone()
    qkv = ops.gemm(a, w, bias=bias)
    ops.rope_kv(qkv, S, Hq, Hkv, hd, inv_freq, pos0, kv0, table=table)
    assert ops.qkv_rope_shape(2399, 28, 4, 128, 3584) == 1332           # the bench shape takes the fused kernel
    q = ops.gemm_qkv_rope(a, w, bias, Hq, Hkv, hd, table, kv1, pos0, shape=1332)
    assert torch.equal(q, qkv[:, :Hq * hd])
    assert torch.equal(kv1, kv0)
    assert bool((kv1[:pos0] == 7.0).all()) and bool((kv1[pos0 + S:] == 7.0).all())
    # no bias, and a q buffer wider than Hq * hd (the stage call hands the front of its qkv buffer)
    q2buf = torch.zeros(S, N, device=DEV, dtype=torch.bfloat16)
    qkv_nb = ops.gemm(a, w)
    kv2, kv3 = kv0.clone(), kv0.clone()
    ops.rope_kv(qkv_nb, S, Hq, Hkv, hd, inv_freq, pos0, kv2, table=table)
    ops.gemm_qkv_rope(a, w, None, Hq, Hkv, hd, table, kv3, pos0, q_out=q2buf, shape=1332)
    assert torch.equal(q2buf[:, :Hq * hd], qkv_nb[:, :Hq * hd]) and bool((q2buf[:, Hq * hd:] == 0).all()) and torch.equal(kv3, kv2)


def test_fused_qkv_rope_refuses_what_it_is_not_built_for():
    assert ops.qkv_rope_shape(2399, 28, 4, 64, 3584) == 0 and ops.qkv_rope_shape(100, 28, 4, 128, 3584) == 0 and ops.qkv_rope_shape(2399, 28, 4, 128, 3500) == 0
    a = torch.zeros(100, 512, device=DEV, dtype=torch.bfloat16)
    w = torch.zeros(8 * 128, 512, device=DEV, dtype=torch.bfloat16)
    kv = torch.zeros(100, 2 * 128, device=DEV, dtype=torch.bfloat16)
    tab = torch.zeros(100, 128, device=DEV)
    with pytest.raises(_lib.UfvError, match="no fused kernel"):
        ops.gemm_qkv_rope(a, w, None, 6, 1, 128, tab, kv, 0)
