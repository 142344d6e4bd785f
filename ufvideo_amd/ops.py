"""Thin torch-tensor wrappers over the C ABI (include/ufv.h).  torch is used only for device
memory and the current stream; every op below runs a hand-written HIP kernel from csrc/."""
import math

import torch

from . import _lib
from ._lib import ACT, DT_BF16, DT_F32, DT_F16, GEMM_AUTO, GEMM_FAST, GEMM_GENERIC, GEMM_GEMV, GEMM_FAST256, GEMM_STREAMK  # noqa: F401

_DT = {torch.bfloat16: DT_BF16, torch.float32: DT_F32, torch.float16: DT_F16}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _chk(t, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise _lib.UfvError(f"{name} must live on the GPU (the hot path has no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise _lib.UfvError(f"{name} must be {dtype}, got {t.dtype}")
    return t


def gemm(a, w, bias=None, act=None, resid=None, resid_rows=0, out=None, out_dtype=torch.bfloat16, swiglu=False,
         kernel=GEMM_AUTO):
    """out[M, N(/2)] = epilogue(a[M,K] @ w[N,K]^T).  a may be a row-strided view; w contiguous.
    w may be an `Fp8Weight`: then a is quantised per row and the fp8 MFMA path runs (same epilogues)."""
    if isinstance(w, Fp8Weight):
        aq, sa = (a.q, a.scale) if isinstance(a, QAct) else quantize_fp8(a)
        return gemm_fp8(aq, sa, w, bias=bias, act=act, resid=resid, resid_rows=resid_rows, out=out, out_dtype=out_dtype, swiglu=swiglu,
                        kernel=kernel)
    _chk(a, torch.bfloat16, "a"); _chk(w, torch.bfloat16, "w")
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1 and a.shape[1] == w.shape[1]
    M, K = a.shape
    N = w.shape[0]
    n_out = N // 2 if swiglu else N
    if out is None:
        out = torch.empty((M, n_out), device=a.device, dtype=out_dtype)
    assert out.shape == (M, n_out) and out.stride(1) == 1
    if bias is not None:
        _chk(bias, torch.float32, "bias"); assert bias.numel() >= N
    ldr = 0
    if resid is not None:
        _chk(resid, torch.float32, "resid"); ldr = resid.stride(0)
    _lib.call("ufv_gemm", a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0),
              int(out.dtype == torch.float32), M, N, K, _ptr(bias), ACT[act], _ptr(resid), ldr, resid_rows, int(swiglu),
              kernel, _stream())
    return out


def gemm_stream_bf16(a, w, x, bias=None, kernel=GEMM_AUTO):
    """x += a @ w^T + bias, x the bf16 residual stream [M, N] updated in place (one rounding of the fp32 sum per element); include/ufv.h ufv_gemm_stream_bf16"""
    _chk(a, torch.bfloat16, "a"); _chk(w, torch.bfloat16, "w"); _chk(x, torch.bfloat16, "x")
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1 and a.shape[1] == w.shape[1] and x.shape == (a.shape[0], w.shape[0]) and x.stride(1) == 1
    if bias is not None:
        _chk(bias, torch.float32, "bias")
    _lib.call("ufv_gemm_stream_bf16", a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), x.data_ptr(), x.stride(0), a.shape[0], w.shape[0], a.shape[1], _ptr(bias),
              kernel, _stream())
    return x


def gemv1(w, a=None, x=None, ln_w=None, eps=1e-6, bias=None, act=None, resid=None, swiglu=False, out=None, out_dtype=torch.bfloat16):
    """One-row GEMV (decode): a bf16 [K] row, or x fp32 [K] + RMSNorm weight (norm fused into the kernel).  w bf16 [N,K] or
    an Fp8Weight (then the row is quantised in the kernel: W8A8).  -> [N(/2)]"""
    N, K = w.shape
    n_out = N // 2 if swiglu else N
    dev = (a if a is not None else x).device
    if out is None:
        out = torch.empty((n_out,), device=dev, dtype=out_dtype)
    if isinstance(w, Fp8Weight):
        wp, ldw, ws = w.q.data_ptr(), w.q.stride(0), w.scale.data_ptr()
    else:
        wp, ldw, ws = w.data_ptr(), w.stride(0), None
    _lib.call("ufv_gemv1", _ptr(a), _ptr(x), _ptr(ln_w), float(eps), wp, ldw, ws, out.data_ptr(), int(out.dtype == torch.float32),
              N, K, _ptr(bias), ACT[act], _ptr(resid), int(swiglu), _stream())
    return out


def quantize_fp8(x, out=None, scale=None):
    """Row-wise e4m3 quantisation: x [M,K] bf16|f32 (any row pitch) -> (q uint8 [M,K], scale f32 [M]), x ~ q * scale[:, None]."""
    _chk(x, name="x")
    assert x.dim() == 2 and x.stride(1) == 1
    M, K = x.shape
    q = out if out is not None else torch.empty((M, K), device=x.device, dtype=torch.uint8)
    s = scale if scale is not None else torch.empty((M,), device=x.device, dtype=torch.float32)
    _lib.call("ufv_quantize_fp8", x.data_ptr(), _DT[x.dtype], x.stride(0), q.data_ptr(), q.stride(0), s.data_ptr(), M, K, _stream())
    return q, s


def dequantize_fp8(q, scale):
    M, K = q.shape
    out = torch.empty((M, K), device=q.device, dtype=torch.float32)
    _lib.call("ufv_dequantize_fp8", q.data_ptr(), q.stride(0), scale.data_ptr(), out.data_ptr(), out.stride(0), M, K, _stream())
    return out


class QAct:
    """A per-token-quantised activation: e4m3 codes [M, K] + fp32 row scales [M] (what `gemm` consumes with an Fp8Weight)."""

    def __init__(self, q, scale):
        self.q, self.scale = q, scale
        self.shape = tuple(q.shape)


class MxAct:
    """A block-quantised activation (round 5): e4m3 codes [M, K] + one e8m0 scale byte per (row, 32 elements), as [ceil(M / 64), ceil(K / 512), 64, 16] (include/ufv.h) -- what the MX-emitting GEMM epilogues
    write and `gemm` consumes with an Fp8Weight through ufv_gemm_fp8_mx.  `swiglu_cols`: the columns are in the SwiGLU epilogue's block order (see ufv.h)."""

    def __init__(self, q, bscale, swiglu_cols=False):
        self.q, self.bscale, self.swiglu_cols = q, bscale, swiglu_cols
        self.shape = tuple(q.shape)

    def scales_row_major(self):
        """[M, K / 32] uint8: the scale bytes in row-major block order (tests / the oracle's layout)"""
        M, K = self.q.shape
        nb, G = self.bscale.shape[:2]
        g = self.bscale.view(nb, G, 64, 4, 4).permute(0, 2, 1, 4, 3)         # [row block, row, group, K-tile in group, block in K-tile]
        return g.reshape(nb * 64, -1)[:M, :K // 32]


def mx_swiglu_perm(n, device=None):
    """physical -> logical column of the SwiGLU MX epilogue's output: inside every group of 128, physical 32 w + 16 h + c holds logical 64 h + 16 w + c"""
    p = torch.arange(n, device=device)
    g, r = p // 128, p % 128
    w, h, c = r // 32, (r % 32) // 16, r % 16
    return g * 128 + 64 * h + 16 * w + c


def quantize_mx(x):
    """x [M, K] bf16|f32 -> MxAct (ufv_quantize_mx): the stand-alone form of what the MX-emitting epilogues do"""
    _chk(x, name="x")
    M, K = x.shape
    q = torch.empty((M, K), device=x.device, dtype=torch.uint8)
    bs = torch.empty(((M + 63) // 64, (K + 511) // 512, 64, 16), device=x.device, dtype=torch.uint8)           # include/ufv.h ufv_quantize_mx
    _lib.call("ufv_quantize_mx", x.data_ptr(), _DT[x.dtype], x.stride(0), q.data_ptr(), q.stride(0), bs.data_ptr(), bs.stride(0), M, K, _stream())
    return MxAct(q, bs)


def dequantize_mx(a):
    M, K = a.q.shape
    out = torch.empty((M, K), device=a.q.device, dtype=torch.float32)
    _lib.call("ufv_dequantize_mx", a.q.data_ptr(), a.q.stride(0), a.bscale.data_ptr(), a.bscale.stride(0), out.data_ptr(), out.stride(0), M, K, _stream())
    return out


def gemm_fp8_mx(a, w, bias=None, act=None, resid=None, out=None, out_dtype=torch.bfloat16, swiglu=False, mx_out=False):
    """e4m3 GEMM with MX block scales on the activation side (ufv_gemm_fp8_mx).  a: MxAct (block-scaled input) or QAct (per-row scales); w: Fp8Weight.
    mx_out=True: the epilogue writes the result as an MxAct (codes + block scales) -- the input of the next e4m3 GEMM, no quantise launch in between."""
    M, K = a.q.shape
    N = w.shape[0]
    assert w.shape[1] == K
    n_out = N // 2 if swiglu else N
    is_mx = isinstance(a, MxAct)
    if mx_out:
        q = torch.empty((M, n_out), device=a.q.device, dtype=torch.uint8)
        bs = torch.empty(((M + 63) // 64, (n_out + 511) // 512, 64, 16), device=a.q.device, dtype=torch.uint8)
        _lib.call("ufv_gemm_fp8_mx", a.q.data_ptr(), a.q.stride(0), None if is_mx else a.scale.data_ptr(), a.bscale.data_ptr() if is_mx else None,
                  a.bscale.stride(0) if is_mx else 0, w.q.data_ptr(), w.q.stride(0), w.scale.data_ptr(), q.data_ptr(), q.stride(0), 0, bs.data_ptr(), bs.stride(0),
                  M, N, K, _ptr(bias), ACT[act], None, 0, 0, int(swiglu), _stream())
        return MxAct(q, bs, swiglu_cols=swiglu)
    assert is_mx
    if out is None:
        out = torch.empty((M, n_out), device=a.q.device, dtype=out_dtype)
    ldr = resid.stride(0) if resid is not None else 0
    rb = resid is not None and resid.dtype == torch.bfloat16            # a bf16 residual stream (updated in place as bf16): bf16 output only
    assert not rb or out.dtype == torch.bfloat16
    _lib.call("ufv_gemm_fp8_mx", a.q.data_ptr(), a.q.stride(0), None, a.bscale.data_ptr(), a.bscale.stride(0), w.q.data_ptr(), w.q.stride(0), w.scale.data_ptr(),
              out.data_ptr(), out.stride(0), int(out.dtype == torch.float32), None, 0, M, N, K, _ptr(bias), ACT[act], _ptr(resid), ldr, int(rb), int(swiglu), _stream())
    return out


class Fp8Weight:
    """A weight matrix [N, K] held as e4m3 bytes + one fp32 scale per output channel.  Passing it to `gemm` in place of a
    bf16 weight selects the W8A8 path: the activation is quantised per token on the fly."""

    def __init__(self, w, mx_swiglu_cols=False):
        """mx_swiglu_cols: the K axis (columns) is stored in the block order of the SwiGLU MX epilogue's output (mx_swiglu_perm): the weight of the GEMM that
        consumes that activation (down_proj behind a fused gate/up); per-channel scales and the product are unchanged by a permutation of K"""
        if mx_swiglu_cols:
            w = w[:, mx_swiglu_perm(w.shape[1], w.device)]
        self.q, self.scale = quantize_fp8(w.contiguous())
        self.shape = tuple(w.shape)
        self.dtype = torch.uint8
        self.mx_swiglu_cols = mx_swiglu_cols

    def stride(self, i):
        return self.q.stride(i)


def gemm_fp8(aq, a_scale, w, bias=None, act=None, resid=None, resid_rows=0, out=None, out_dtype=torch.bfloat16, swiglu=False,
             kernel=GEMM_AUTO):
    """out = epilogue((aq @ w.q^T) * a_scale[:, None] * w.scale[None, :]); aq uint8 [M,K] e4m3, w Fp8Weight [N,K]."""
    _chk(aq, torch.uint8, "aq"); _chk(a_scale, torch.float32, "a_scale")
    M, K = aq.shape
    N = w.shape[0]
    assert w.shape[1] == K and aq.stride(1) == 1
    n_out = N // 2 if swiglu else N
    if out is None:
        out = torch.empty((M, n_out), device=aq.device, dtype=out_dtype)
    assert out.shape == (M, n_out) and out.stride(1) == 1
    ldr = 0
    if resid is not None:
        _chk(resid, torch.float32, "resid"); ldr = resid.stride(0)
    if bias is not None:
        _chk(bias, torch.float32, "bias")
    _lib.call("ufv_gemm_fp8", aq.data_ptr(), aq.stride(0), a_scale.data_ptr(), w.q.data_ptr(), w.q.stride(0), w.scale.data_ptr(),
              out.data_ptr(), out.stride(0), int(out.dtype == torch.float32), M, N, K, _ptr(bias), ACT[act], _ptr(resid), ldr, resid_rows,
              int(swiglu), kernel, _stream())
    return out


def layernorm(x, w, b, eps, act=None, out=None, out_dtype=torch.bfloat16, quant=False):
    """quant=True: the bf16 result is quantised in the same pass and returned as a QAct (input of an fp8 GEMM)."""
    _chk(x, name="x"); _chk(w, torch.float32, "w")
    M, D = x.shape
    if quant:
        assert act is None
        q = torch.empty((M, D), device=x.device, dtype=torch.uint8)
        s = torch.empty((M,), device=x.device, dtype=torch.float32)
        _lib.call("ufv_layernorm_fp8", x.data_ptr(), _DT[x.dtype], x.stride(0), q.data_ptr(), q.stride(0), s.data_ptr(), w.data_ptr(),
                  _ptr(b), M, D, float(eps), _stream())
        return QAct(q, s)
    if out is None:
        out = torch.empty((M, D), device=x.device, dtype=out_dtype)
    _lib.call("ufv_layernorm", x.data_ptr(), _DT[x.dtype], x.stride(0), out.data_ptr(), int(out.dtype == torch.float32),
              out.stride(0), w.data_ptr(), _ptr(b), M, D, float(eps), ACT[act], _stream())
    return out


def ln_add_silu(a, wa, ba, b, wb=None, bb=None, eps=1e-5, out=None):
    _chk(a, torch.bfloat16, "a"); _chk(b, torch.bfloat16, "b")
    assert a.is_contiguous() and b.is_contiguous()
    M, D = a.shape
    if out is None:
        out = torch.empty_like(a)
    _lib.call("ufv_ln_add_silu", a.data_ptr(), wa.data_ptr(), ba.data_ptr(), b.data_ptr(), _ptr(wb), _ptr(bb), out.data_ptr(),
              M, D, float(eps), _stream())
    return out


def rmsnorm(x, w, eps, out=None, out_dtype=torch.bfloat16, quant=False):
    _chk(x, torch.float32, "x")
    M, D = x.shape
    if quant:
        q = torch.empty((M, D), device=x.device, dtype=torch.uint8)
        s = torch.empty((M,), device=x.device, dtype=torch.float32)
        _lib.call("ufv_rmsnorm_fp8", x.data_ptr(), x.stride(0), q.data_ptr(), q.stride(0), s.data_ptr(), w.data_ptr(), M, D, float(eps),
                  _stream())
        return QAct(q, s)
    if out is None:
        out = torch.empty((M, D), device=x.device, dtype=out_dtype)
    _lib.call("ufv_rmsnorm", x.data_ptr(), x.stride(0), out.data_ptr(), int(out.dtype == torch.float32), out.stride(0),
              w.data_ptr(), M, D, float(eps), _stream())
    return out


def attention(q, k, v, B, Hq, Hkv, Sq, Sk, hd, q_strides, k_strides, v_strides, scale=None, causal=False, q_pos0=0,
              out=None, kernel=0):
    """q/k/v: bf16 tensors (any view); *_strides = (batch stride, token stride) in elements; heads are
    contiguous blocks of hd along the last dim.  Returns o [B*Sq, Hq*hd] bf16."""
    if out is None:
        out = torch.empty((B * Sq, Hq * hd), device=q.device, dtype=torch.bfloat16)
    scale = hd ** -0.5 if scale is None else scale
    _lib.call("ufv_attention", q.data_ptr(), q_strides[0], q_strides[1], k.data_ptr(), k_strides[0], k_strides[1],
              v.data_ptr(), v_strides[0], v_strides[1], out.data_ptr(), Sq * out.stride(0), out.stride(0), B, Hq, Hkv, Sq, Sk,
              hd, float(scale), int(causal), q_pos0, kernel, _stream())
    return out


def attention_decode(q, k, v, Hq, Hkv, Sk, hd, k_ss, v_ss, scale=None, nsplit=16, out=None, ws=None):
    """One query token (q [1, Hq*hd] bf16) against Sk cached keys/values (token stride k_ss/v_ss elements)."""
    if out is None:
        out = torch.empty((1, Hq * hd), device=q.device, dtype=torch.bfloat16)
    if ws is None:
        ws = torch.empty((_lib.load().ufv_attention_decode_ws_bytes(1, Hq, hd, nsplit),), device=q.device, dtype=torch.uint8)
    scale = hd ** -0.5 if scale is None else scale
    _lib.call("ufv_attention_decode", q.data_ptr(), 0, k.data_ptr(), 0, k_ss, v.data_ptr(), 0, v_ss, out.data_ptr(), 0, 1, Hq, Hkv,
              Sk, hd, float(scale), ws.data_ptr(), nsplit, _stream())
    return out


def rope_table(inv_freq, pos0, S, hd):
    """fp32 [S, hd]: cos | sin of (pos0+s) * inv_freq, computed once per forward pass and shared by all layers"""
    _chk(inv_freq, torch.float32, "inv_freq")
    tab = torch.empty((S, hd), device=inv_freq.device, dtype=torch.float32)
    _lib.call("ufv_rope_table", inv_freq.data_ptr(), pos0, S, hd, tab.data_ptr(), _stream())
    return tab


def rope_kv(qkv, S, Hq, Hkv, hd, inv_freq, pos0, kv_cache, table=None):
    if table is not None and hd % 16 == 0:
        _chk(qkv, torch.bfloat16, "qkv"); _chk(kv_cache, torch.bfloat16, "kv_cache"); _chk(table, torch.float32, "table")
        _lib.call("ufv_rope_kv_table", qkv.data_ptr(), qkv.stride(0), S, Hq, Hkv, hd, table.data_ptr(), pos0, kv_cache.data_ptr(),
                  kv_cache.stride(0), _stream())
        return
    _chk(qkv, torch.bfloat16, "qkv"); _chk(kv_cache, torch.bfloat16, "kv_cache"); _chk(inv_freq, torch.float32, "inv_freq")
    _lib.call("ufv_rope_kv", qkv.data_ptr(), qkv.stride(0), S, Hq, Hkv, hd, inv_freq.data_ptr(), pos0, kv_cache.data_ptr(),
              kv_cache.stride(0), _stream())


def qkv_rope_shape(S, Hq, Hkv, hd, K):
    """tile shape the fused QKV + RoPE + KV-append GEMM would run, or 0: issue gemm + rope_kv (host arithmetic, include/ufv.h ufv_gemm_qkv_rope_shape)"""
    return int(_lib.load().ufv_gemm_qkv_rope_shape(S, Hq, Hkv, hd, K))


def gemm_qkv_rope(a, w, bias, Hq, Hkv, hd, table, kv_cache, pos0, q_out=None, shape=0):
    """q / k / v projection + RoPE + KV append of S = a.shape[0] prefill rows in ONE launch (ufv_gemm_qkv_rope): returns q [S, Hq * hd] (rotated); the
    rotated k heads and the v heads go to kv_cache rows pos0 .. pos0 + S - 1.  `table` = rope_table(inv_freq, pos0, S, hd).  Bit-identical to
    gemm(a, w, bias) followed by rope_kv(..., table=table)."""
    _chk(a, torch.bfloat16, "a"); _chk(w, torch.bfloat16, "w"); _chk(kv_cache, torch.bfloat16, "kv_cache"); _chk(table, torch.float32, "table")
    S, K = a.shape
    assert w.shape == ((Hq + 2 * Hkv) * hd, K) and table.shape == (S, hd) and kv_cache.shape[0] >= pos0 + S
    if q_out is None:
        q_out = torch.empty((S, Hq * hd), device=a.device, dtype=torch.bfloat16)
    _lib.call("ufv_gemm_qkv_rope", a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), bias.data_ptr() if bias is not None else None, q_out.data_ptr(),
              q_out.stride(0), kv_cache[pos0:].data_ptr(), kv_cache.stride(0), S, Hq, Hkv, hd, K, table.data_ptr(), shape, _stream())
    return q_out


def gemm_qkv_rope_fp8(a, w, bias, Hq, Hkv, hd, table, kv_cache, pos0, q_out=None):
    """gemm_qkv_rope with e4m3 operands: a QAct (codes + row scales), w Fp8Weight (ufv_gemm_qkv_rope_fp8)"""
    S, K = a.q.shape
    assert w.shape == ((Hq + 2 * Hkv) * hd, K) and table.shape == (S, hd) and kv_cache.shape[0] >= pos0 + S
    if q_out is None:
        q_out = torch.empty((S, Hq * hd), device=a.q.device, dtype=torch.bfloat16)
    _lib.call("ufv_gemm_qkv_rope_fp8", a.q.data_ptr(), a.q.stride(0), a.scale.data_ptr(), w.q.data_ptr(), w.q.stride(0), w.scale.data_ptr(), _ptr(bias), q_out.data_ptr(),
              q_out.stride(0), kv_cache[pos0:].data_ptr(), kv_cache.stride(0), S, Hq, Hkv, hd, K, table.data_ptr(), _stream())
    return q_out


def patchify(pixels, P, Kpad):
    _chk(pixels, name="pixels"); assert pixels.is_contiguous()
    T, Cc, H, W = pixels.shape
    out = torch.empty((T * (H // P) * (W // P), Kpad), device=pixels.device, dtype=torch.bfloat16)
    _lib.call("ufv_patchify", pixels.data_ptr(), _DT[pixels.dtype], out.data_ptr(), T, Cc, H, W, P, Kpad, _stream())
    return out


def dwconv3x3_ln_silu(x, w9, lnw, lnb, F, H, W, C, eps):
    _chk(x, torch.bfloat16, "x"); assert x.is_contiguous()
    y = torch.empty_like(x)
    _lib.call("ufv_dwconv3x3_ln_silu", x.data_ptr(), y.data_ptr(), w9.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), F, H, W, C,
              float(eps), _stream())
    return y


def colmean(x, F, P):
    C = x.shape[-1]
    out = torch.empty((F, C), device=x.device, dtype=torch.bfloat16)
    _lib.call("ufv_colmean", x.data_ptr(), out.data_ptr(), F, P, C, _stream())
    return out


def scale_channels(x, gate, F, P):
    _lib.call("ufv_scale_channels", x.data_ptr(), gate.data_ptr(), F, P, x.shape[-1], _stream())
    return x


def conv3d_gather(x, T, H, W, C, k, pad):
    kt, kh, kw = k
    To, Ho, Wo = (T + 2 * pad - kt) // kt + 1, (H + 2 * pad - kh) // kh + 1, (W + 2 * pad - kw) // kw + 1
    out = torch.empty((To * Ho * Wo, kt * kh * kw * C), device=x.device, dtype=torch.bfloat16)
    _lib.call("ufv_conv3d_gather", x.data_ptr(), out.data_ptr(), T, H, W, C, kt, kh, kw, pad, _stream())
    return out, (To, Ho, Wo)


def avgpool3d_silu(x, T, H, W, C, k):
    """x bf16 [T*H*W, C] token-major -> (bf16 [To*Ho*Wo, C], (To, Ho, Wo)): AvgPool3d(k) + SiLU"""
    _chk(x, torch.bfloat16, "x"); assert x.is_contiguous()
    To, Ho, Wo = T // k[0], H // k[1], W // k[2]
    out = torch.empty((To * Ho * Wo, C), device=x.device, dtype=torch.bfloat16)
    _lib.call("ufv_avgpool3d_silu", x.data_ptr(), out.data_ptr(), T, H, W, C, k[0], k[1], k[2], _stream())
    return out, (To, Ho, Wo)


def gather_rows(src, src_idx, dst, dst_idx, n=None):
    D = src.shape[-1]
    if n is None:
        n = (src_idx if src_idx is not None else dst_idx).numel() if (src_idx is not None or dst_idx is not None) else src.shape[0]
    if n == 0:
        return dst
    _lib.call("ufv_gather_rows", src.data_ptr(), _DT[src.dtype], src.stride(0), _ptr(src_idx), dst.data_ptr(), _DT[dst.dtype],
              dst.stride(0), _ptr(dst_idx), n, D, _stream())
    return dst


def mask_pool(feat, mask, frame_of):
    """feat [n, P, C] (bf16/f32), mask f32 [q, P], frame_of int32 [q] -> f32 [q, C]"""
    q, P = mask.shape
    C = feat.shape[-1]
    out = torch.empty((q, C), device=feat.device, dtype=torch.float32)
    _lib.call("ufv_mask_pool", feat.data_ptr(), _DT[feat.dtype], mask.data_ptr(), frame_of.data_ptr(), out.data_ptr(), q, P, C,
              _stream())
    return out


def im2col(pixels, ks, stride, pad, Kpad):
    _chk(pixels, name="pixels"); assert pixels.is_contiguous()
    B, Cc, H, W = pixels.shape
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    out = torch.empty((B * Ho * Wo, Kpad), device=pixels.device, dtype=torch.bfloat16)
    _lib.call("ufv_im2col", pixels.data_ptr(), _DT[pixels.dtype], out.data_ptr(), B, Cc, H, W, ks, stride, pad, Kpad, _stream())
    return out, (Ho, Wo)


def maxpool2x2(x, Bw, H, W, C, out=None):
    """x rows [Bw*H*W] (any row pitch) -> [Bw*(H/2)*(W/2), C] same dtype"""
    if out is None:
        out = torch.zeros((Bw * (H // 2) * (W // 2), x.shape[1] if x.shape[1] >= C else C), device=x.device, dtype=x.dtype)
    _lib.call("ufv_maxpool2x2", x.data_ptr(), _DT[x.dtype], x.stride(0), out.data_ptr(), out.stride(0), Bw, H, W, C, _stream())
    return out


def add_rows(src, dst, dst_idx, D=None):
    _chk(dst, torch.float32, "dst")
    n = src.shape[0]
    _lib.call("ufv_add_rows", src.data_ptr(), _DT[src.dtype], src.stride(0), dst.data_ptr(), dst.stride(0), _ptr(dst_idx), n,
              D if D is not None else src.shape[1], _stream())
    return dst


def upsample2x_add(x, prev, B, H, W, C):
    _chk(x, torch.float32, "x"); _chk(prev, torch.float32, "prev")
    _lib.call("ufv_upsample2x_add", x.data_ptr(), prev.data_ptr(), B, H, W, C, _stream())
    return x


def add_bcast(a, b=None, out=None, out_dtype=torch.bfloat16):
    """out[m] = a[m] + b[m % b.shape[0]] (b fp32 table or None); a fp32|bf16 rows of any pitch."""
    _chk(a, name="a")
    M, C = a.shape
    if out is None:
        out = torch.empty((M, C), device=a.device, dtype=out_dtype)
    if b is not None:
        _chk(b, torch.float32, "b"); assert b.shape[1] == C
    _lib.call("ufv_add_bcast", a.data_ptr(), _DT[a.dtype], a.stride(0), _ptr(b), b.stride(0) if b is not None else 0,
              b.shape[0] if b is not None else 0, out.data_ptr(), _DT[out.dtype], out.stride(0), M, C, _stream())
    return out


def sam_mask_head(up2, s0, hyper, B, h, w, C8=32):
    """up2 bf16 [B*h*w, >=4*C8], s0 bf16 [B*4*h*w, >=C8], hyper f32 [B, nm, C8] -> f32 [B, nm, 2h, 2w]"""
    _chk(up2, torch.bfloat16, "up2"); _chk(s0, torch.bfloat16, "s0"); _chk(hyper, torch.float32, "hyper")
    assert hyper.is_contiguous() and hyper.shape[0] == B and hyper.shape[2] == C8
    nm = hyper.shape[1]
    out = torch.empty((B, nm, 2 * h, 2 * w), device=up2.device, dtype=torch.float32)
    _lib.call("ufv_sam_mask_head", up2.data_ptr(), up2.stride(0), s0.data_ptr(), s0.stride(0), hyper.data_ptr(), out.data_ptr(),
              B, h, w, C8, nm, _stream())
    return out


def resize_bilinear(src, size, sel=None, sel_off=0):
    """F.interpolate(src, size, mode="bilinear", align_corners=False) for f32 [N, P, Hs, Ws].  Without `sel` every
    plane is resized ([N, P, Hd, Wd]); with sel int32 [N] only plane sel_off + sel[n] of image n ([N, 1, Hd, Wd])."""
    _chk(src, torch.float32, "src"); assert src.is_contiguous() and src.dim() == 4
    N, P, Hs, Ws = src.shape
    Hd, Wd = size
    if sel is None:
        out = torch.empty((N, P, Hd, Wd), device=src.device, dtype=torch.float32)
        _lib.call("ufv_resize_bilinear", src.data_ptr(), None, 1, 0, out.data_ptr(), N * P, Hs, Ws, Hd, Wd, _stream())
    else:
        assert sel.dtype == torch.int32 and sel.numel() == N
        out = torch.empty((N, 1, Hd, Wd), device=src.device, dtype=torch.float32)
        _lib.call("ufv_resize_bilinear", src.data_ptr(), sel.data_ptr(), P, sel_off, out.data_ptr(), N, Hs, Ws, Hd, Wd, _stream())
    return out


def argmax_rows(x):
    _chk(x, torch.float32, "x")
    out = torch.empty((x.shape[0],), device=x.device, dtype=torch.int32)
    _lib.call("ufv_argmax_rows", x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], out.data_ptr(), _stream())
    return out


def cross_entropy_rows(logits, labels, ignore_index=-100):
    """per-row -log softmax(logits)[label] (0 where label == ignore_index); logits f32 [M, V], labels int64 [M]"""
    _chk(logits, torch.float32, "logits"); _chk(labels, torch.int64, "labels")
    M, V = logits.shape
    out = torch.empty((M,), device=logits.device, dtype=torch.float32)
    _lib.call("ufv_cross_entropy_rows", logits.data_ptr(), logits.stride(0), labels.data_ptr(), M, V, ignore_index, out.data_ptr(), _stream())
    return out


def mask_loss_sums(pred, gt):
    """pred/gt f32 [n, H, W] -> f32 [n, 4] = (sum BCE-with-logits, sum sigmoid*gt, sum sigmoid, sum gt) per mask"""
    _chk(pred, torch.float32, "pred"); _chk(gt, torch.float32, "gt")
    assert pred.shape == gt.shape and pred.is_contiguous() and gt.is_contiguous()
    n = pred.shape[0]
    out = torch.zeros((n, 4), device=pred.device, dtype=torch.float32)
    _lib.call("ufv_mask_loss_sums", pred.data_ptr(), gt.data_ptr(), n, pred[0].numel() if n else 1, out.data_ptr(), _stream())
    return out


def argmax(logits, out=None, ws=None):
    """index of the maximum (ties: the lowest, NaN wins, like torch.argmax).  ws: a zero-initialised uint8 tensor of ufv_argmax_ws_bytes() bytes selects the
    64-block form the decode step uses (its counter returns to zero: the tensor can be reused)."""
    _chk(logits, torch.float32, "logits")
    if out is None:
        out = torch.empty((1,), device=logits.device, dtype=torch.int64)
    if ws is not None and logits.data_ptr() % 16 == 0:
        _lib.call("ufv_argmax_ws", logits.data_ptr(), logits.numel(), out.data_ptr(), ws.data_ptr(), _stream())
    else:
        _lib.call("ufv_argmax", logits.data_ptr(), logits.numel(), out.data_ptr(), _stream())
    return out


def preprocess_u8(frames, mean, std):
    """frames u8 [T,H,W,3] on device -> bf16 [T,3,H,W]"""
    import ctypes as C
    T, H, W, _ = frames.shape
    out = torch.empty((T, 3, H, W), device=frames.device, dtype=torch.bfloat16)
    m = (C.c_float * 3)(*mean); s = (C.c_float * 3)(*std)
    _lib.call("ufv_preprocess_u8", frames.data_ptr(), out.data_ptr(), T, H, W, C.cast(m, C.c_void_p), C.cast(s, C.c_void_p),
              _stream())
    return out


_RESIZE_TABLES = {}


def resize_bicubic_u8(frames, out_h, out_w):
    """frames u8 [T,H,W,3] on device -> u8 [T,out_h,out_w,3], bit-identical to PIL Image.resize((out_w, out_h), BICUBIC)."""
    from .mm_utils import pil_resize_coeffs
    _chk(frames, torch.uint8, "frames"); assert frames.is_contiguous() and frames.dim() == 4 and frames.shape[3] == 3
    T, H, W, _ = frames.shape
    dev = frames.device

    def tables(n_in, n_out):
        key = (n_in, n_out, str(dev))
        if key not in _RESIZE_TABLES:
            b, k = pil_resize_coeffs(n_in, n_out)
            _RESIZE_TABLES[key] = (torch.from_numpy(b).to(dev), torch.from_numpy(k).to(dev), k.shape[1])
        return _RESIZE_TABLES[key]
    bx, kx, nx = tables(W, out_w) if W != out_w else (None, None, 0)
    by, ky, ny = tables(H, out_h) if H != out_h else (None, None, 0)
    out = torch.empty((T, out_h, out_w, 3), device=dev, dtype=torch.uint8)
    tmp = torch.empty((T, H, out_w, 3), device=dev, dtype=torch.uint8) if (W != out_w and H != out_h) else None
    _lib.call("ufv_resize_bicubic_u8", frames.data_ptr(), _ptr(tmp), out.data_ptr(), T, H, W, out_h, out_w, _ptr(bx), _ptr(kx), nx,
              _ptr(by), _ptr(ky), ny, _stream())
    return out


def convert(src, dtype):
    _chk(src, name="src")
    if src.dtype == dtype:
        return src
    src = src.contiguous()
    out = torch.empty(src.shape, device=src.device, dtype=dtype)
    _lib.call("ufv_convert", src.data_ptr(), _DT[src.dtype], out.data_ptr(), _DT[dtype], src.numel(), _stream())
    return out


# ---- training step of the decoder (csrc/train.hip) --------------------------------------------------

def round_up(x, m):
    return (x + m - 1) // m * m


def transpose(x, rpad=None, out=None):
    """x bf16 [R, C] (row-strided view ok) -> bf16 [C, rpad] with out[:, R:] = 0 (rpad defaults to R rounded up to 128)."""
    _chk(x, torch.bfloat16, "x"); assert x.dim() == 2 and x.stride(1) == 1
    R, Cc = x.shape
    rpad = round_up(R, 128) if rpad is None else rpad
    if out is None:
        out = torch.empty((Cc, rpad), device=x.device, dtype=torch.bfloat16)
    assert out.shape[0] >= Cc and out.shape[1] >= rpad and out.stride(1) == 1
    _lib.call("ufv_transpose_bf16", x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), R, Cc, rpad, _stream())
    return out


_WS = {}


def _ws(dev, nbytes, tag="ws"):
    key = (str(dev), tag)
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty((int(nbytes),), device=dev, dtype=torch.uint8)
        _WS[key] = t
    return t


def rmsnorm_bwd(x, w, dy, dx, dw, eps, accumulate=True, dw_accumulate=True):
    """dx fp32 [M, D] (+)= dL/dx of y = w * x * rsqrt(mean(x^2) + eps) given dy fp32 [M, D]; dw fp32 [D] (+)= dL/dw."""
    for t, n in ((x, "x"), (dy, "dy"), (dx, "dx"), (dw, "dw"), (w, "w")):
        _chk(t, torch.float32, n)
    M, D = x.shape
    ws = _ws(x.device, _lib.load().ufv_rmsnorm_bwd_ws_bytes(D), "rms")
    _lib.call("ufv_rmsnorm_bwd", x.data_ptr(), x.stride(0), w.data_ptr(), dy.data_ptr(), dy.stride(0), dx.data_ptr(), dx.stride(0),
              int(accumulate), dw.data_ptr(), int(dw_accumulate), M, D, float(eps), ws.data_ptr(), _stream())
    return dx, dw


def colsum(x, out, accumulate=True):
    """out fp32 [C] (+)= column sums of x bf16 [R, C]"""
    _chk(x, torch.bfloat16, "x"); _chk(out, torch.float32, "out")
    R, Cc = x.shape
    ws = _ws(x.device, 32 * Cc * 4, "colsum")
    _lib.call("ufv_colsum_bf16", x.data_ptr(), x.stride(0), R, Cc, out.data_ptr(), int(accumulate), ws.data_ptr(), _stream())
    return out


def swiglu(gu, out=None):
    """gu bf16 [M, 2I] in the packed [16 gate | 16 up] layout -> act bf16 [M, I] = silu(gate) * up"""
    _chk(gu, torch.bfloat16, "gu")
    M, I2 = gu.shape
    if out is None:
        out = torch.empty((M, I2 // 2), device=gu.device, dtype=torch.bfloat16)
    _lib.call("ufv_swiglu", gu.data_ptr(), gu.stride(0), out.data_ptr(), out.stride(0), M, I2 // 2, _stream())
    return out


def swiglu_bwd(gu, dact, out=None):
    _chk(gu, torch.bfloat16, "gu"); _chk(dact, torch.bfloat16, "dact")
    M, I2 = gu.shape
    if out is None:
        out = torch.empty_like(gu)
    _lib.call("ufv_swiglu_bwd", gu.data_ptr(), gu.stride(0), dact.data_ptr(), dact.stride(0), out.data_ptr(), out.stride(0), M, I2 // 2,
              _stream())
    return out


def rope_rows(buf, col0, nheads, hd, inv_freq, pos0=0, backward=False):
    _chk(buf, torch.bfloat16, "buf"); _chk(inv_freq, torch.float32, "inv_freq")
    _lib.call("ufv_rope_rows", buf.data_ptr(), buf.stride(0), buf.shape[0], col0, nheads, hd, inv_freq.data_ptr(), pos0, int(backward),
              _stream())
    return buf


def cross_entropy_bwd(logits, labels, V, gscale, dlogits=None, ignore_index=-100):
    """logits fp32 [M, >=V], labels int64 [M] (already shifted) -> (loss fp32 [M], dlogits bf16 [M, Vpad] = gscale * dloss/dlogits)"""
    _chk(logits, torch.float32, "logits"); _chk(labels, torch.int64, "labels")
    M = logits.shape[0]
    Vp = logits.shape[1] if dlogits is None else dlogits.shape[1]
    if dlogits is None:
        dlogits = torch.empty((M, Vp), device=logits.device, dtype=torch.bfloat16)
    loss = torch.empty((M,), device=logits.device, dtype=torch.float32)
    _lib.call("ufv_cross_entropy_bwd", logits.data_ptr(), logits.stride(0), labels.data_ptr(), M, V, Vp, ignore_index, float(gscale),
              loss.data_ptr(), dlogits.data_ptr(), dlogits.stride(0), _stream())
    return loss, dlogits


def scatter_add_rows(src, idx, dst):
    _chk(src, torch.float32, "src"); _chk(idx, torch.int64, "idx"); _chk(dst, torch.float32, "dst")
    _lib.call("ufv_scatter_add_rows", src.data_ptr(), src.stride(0), idx.data_ptr(), dst.data_ptr(), dst.stride(0), src.shape[0],
              src.shape[1], _stream())
    return dst


def sumsq(x, n_partial=1024):
    """-> fp32 [n_partial] partial sums of squares of the flat fp32 tensor x (sum them for ||x||^2)"""
    _chk(x, torch.float32, "x"); assert x.is_contiguous()
    part = torch.empty((n_partial,), device=x.device, dtype=torch.float32)
    _lib.call("ufv_sumsq", x.data_ptr(), x.numel(), part.data_ptr(), n_partial, _stream())
    return part


def adamw(p, g, m, v, p_bf16, lr, beta1, beta2, eps, weight_decay, step, gscale=None):
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, torch.float32, n); assert t.is_contiguous()
    assert p.numel() == g.numel() == m.numel() == v.numel() and (p_bf16 is None or p_bf16.numel() == p.numel())
    _lib.call("ufv_adamw", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(p_bf16), p.numel(), float(lr), float(beta1),
              float(beta2), float(eps), float(weight_decay), int(step), _ptr(gscale), _stream())


def attention_bwd(q, k, v, dO, dq, dk, dv, S, Hq, Hkv, hd, scale=None):
    """Causal GQA self-attention backward.  q [S, *] (head h at column h*hd), k / v [>= round_up(S,128), *] column views of one
    buffer pitch, dO [S, *]; writes dq / dk / dv (bf16, same column conventions)."""
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (dO, "dO"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        _chk(t, torch.bfloat16, n)
    Sp = round_up(S, 128)
    assert k.shape[0] >= Sp and v.shape[0] >= Sp and k.stride(0) == v.stride(0) and dk.stride(0) == dv.stride(0)
    scale = hd ** -0.5 if scale is None else scale
    ws = _ws(q.device, _lib.load().ufv_attention_bwd_ws_bytes(S, Hq, Hkv, hd), "attn_bwd")
    _lib.call("ufv_attention_bwd", q.data_ptr(), q.stride(0), k.data_ptr(), v.data_ptr(), k.stride(0), dO.data_ptr(), dO.stride(0),
              dq.data_ptr(), dq.stride(0), dk.data_ptr(), dv.data_ptr(), dk.stride(0), S, Hq, Hkv, hd, float(scale), ws.data_ptr(),
              _stream())


def attention_causal_lse(q, k, v, out, lse, S, Hq, Hkv, hd, scale=None):
    """Training forward of causal self-attention (hd 128): out [S, Hq*hd] bf16 and lse fp32 [Hq, S] (log2 domain) for
    attention_bwd_fused.  q / k / v: row views (head h at column h*hd)."""
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (out, "out")):
        _chk(t, torch.bfloat16, n)
    _chk(lse, torch.float32, "lse"); assert lse.is_contiguous() and lse.numel() >= Hq * S
    scale = hd ** -0.5 if scale is None else scale
    _lib.call("ufv_attention_causal_lse", q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), out.data_ptr(),
              out.stride(0), Hq, Hkv, S, hd, float(scale), lse.data_ptr(), _stream())
    return out


def attention_bwd_fused(q, k, v, o, dO, lse, dq, dk, dv, S, Hq, Hkv, hd, scale=None):
    """Flash-style causal GQA attention backward (hd 128): same conventions as attention_bwd plus the forward output o and lse."""
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (o, "o"), (dO, "dO"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        _chk(t, torch.bfloat16, n)
    _chk(lse, torch.float32, "lse")
    assert k.stride(0) == v.stride(0) and dk.stride(0) == dv.stride(0) and k.shape[0] >= S and v.shape[0] >= S
    scale = hd ** -0.5 if scale is None else scale
    ws = _ws(q.device, _lib.load().ufv_attention_bwd_fused_ws_bytes(S, Hq), "attn_bwd_fused")
    _lib.call("ufv_attention_bwd_fused", q.data_ptr(), q.stride(0), k.data_ptr(), v.data_ptr(), k.stride(0), o.data_ptr(), o.stride(0),
              dO.data_ptr(), dO.stride(0), lse.data_ptr(), dq.data_ptr(), dq.stride(0), dk.data_ptr(), dv.data_ptr(), dk.stride(0), S, Hq,
              Hkv, hd, float(scale), ws.data_ptr(), _stream())


def convert_into(src, dst):
    """dst[...] = src converted to dst.dtype (both contiguous, same number of elements)"""
    _chk(src, name="src"); _chk(dst, name="dst")
    assert src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
    _lib.call("ufv_convert", src.data_ptr(), _DT[src.dtype], dst.data_ptr(), _DT[dst.dtype], src.numel(), _stream())
    return dst


def gemm_splitk(a, w, out, nsplit, accumulate=False):
    """out [M, N] (+)= a [M, K] @ w [N, K]^T with the K range split over up to `nsplit` blocks per tile; out fp32 (accumulate
    allowed) or bf16"""
    _chk(a, torch.bfloat16, "a"); _chk(w, torch.bfloat16, "w"); _chk(out, name="out")
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and out.shape == (M, N) and a.stride(1) == 1 and w.stride(1) == 1 and out.stride(1) == 1
    ws = _ws(a.device, nsplit * M * N * 4, "splitk")
    _lib.call("ufv_gemm_splitk", a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0),
              int(out.dtype == torch.float32), int(accumulate), M, N, K, nsplit, ws.data_ptr(), _stream())
    return out


def sample_top_p(logits, temperature, top_k, top_p, u, out=None, kept=None):
    """logits fp32 [M, V] (or [V]) -> int64 [M] tokens drawn after temperature / top-k / top-p filtering; u fp32 [M] uniforms in [0,1)"""
    _chk(logits, torch.float32, "logits"); _chk(u, torch.float32, "u")
    lg = logits.view(1, -1) if logits.dim() == 1 else logits
    M, V = lg.shape
    if out is None:
        out = torch.empty((M,), device=lg.device, dtype=torch.int64)
    _lib.call("ufv_sample_top_p", lg.data_ptr(), lg.stride(0), M, V, float(temperature), int(top_k), float(top_p), u.data_ptr(),
              out.data_ptr(), _ptr(kept), _stream())
    return out


# ---- projector backward (csrc/train_proj.hip) -----------------------------------------------------------------------------

def act_fwd(pre, act):
    _chk(pre, torch.bfloat16, "pre"); assert pre.is_contiguous()
    out = torch.empty_like(pre)
    _lib.call("ufv_act", pre.data_ptr(), out.data_ptr(), pre.numel(), ACT[act], _stream())
    return out


def act_bwd(pre, dout, act):
    _chk(pre, torch.bfloat16, "pre"); _chk(dout, torch.bfloat16, "dout"); assert pre.is_contiguous() and dout.is_contiguous()
    out = torch.empty_like(pre)
    _lib.call("ufv_act_bwd", pre.data_ptr(), dout.data_ptr(), out.data_ptr(), pre.numel(), ACT[act], _stream())
    return out


def add_bf16(a, b):
    _chk(a, torch.bfloat16, "a"); _chk(b, torch.bfloat16, "b"); assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
    out = torch.empty_like(a)
    _lib.call("ufv_add_bf16", a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream())
    return out


def layernorm_bwd(x, w, b, dout, dw, db, eps, act=None):
    """x, dout bf16 [M, C] -> dx bf16; dw, db fp32 [C] are added to"""
    _chk(x, torch.bfloat16, "x"); _chk(dout, torch.bfloat16, "dout"); _chk(dw, torch.float32, "dw"); _chk(db, torch.float32, "db")
    M, Cc = x.shape
    dx = torch.empty((M, Cc), device=x.device, dtype=torch.bfloat16)
    ws = _ws(x.device, _lib.load().ufv_layernorm_bwd_ws_bytes(Cc), "ln_bwd")
    _lib.call("ufv_layernorm_bwd", x.data_ptr(), x.stride(0), w.data_ptr(), b.data_ptr(), dout.data_ptr(), dout.stride(0), dx.data_ptr(),
              dx.stride(0), dw.data_ptr(), db.data_ptr(), M, Cc, float(eps), ACT[act], ws.data_ptr(), _stream())
    return dx


def ln_add_silu_g(z, wa, ba, s, wb, bb, dout, eps):
    for t_, n in ((z, "z"), (s, "s"), (dout, "dout")):
        _chk(t_, torch.bfloat16, n); assert t_.is_contiguous()
    M, Cc = z.shape
    g = torch.empty_like(z)
    _lib.call("ufv_ln_add_silu_g", z.data_ptr(), wa.data_ptr(), ba.data_ptr(), s.data_ptr(), _ptr(wb), _ptr(bb), dout.data_ptr(), g.data_ptr(),
              M, Cc, float(eps), _stream())
    return g


def dwconv3x3(x, w9, F, H, W, flip=False):
    _chk(x, torch.bfloat16, "x"); _chk(w9, torch.float32, "w9"); assert x.is_contiguous()
    y = torch.empty_like(x)
    _lib.call("ufv_dwconv3x3", x.data_ptr(), y.data_ptr(), w9.data_ptr(), F, H, W, x.shape[-1], int(flip), _stream())
    return y


def dwconv3x3_dw(x, dy, dw9, F, H, W):
    _chk(x, torch.bfloat16, "x"); _chk(dy, torch.bfloat16, "dy"); _chk(dw9, torch.float32, "dw9")
    Cc = x.shape[-1]
    ws = _ws(x.device, _lib.load().ufv_dwconv3x3_dw_ws_bytes(Cc), "dw9")
    _lib.call("ufv_dwconv3x3_dw", x.data_ptr(), dy.data_ptr(), dw9.data_ptr(), F, H, W, Cc, ws.data_ptr(), _stream())
    return dw9


def prod_colsum(a, b, F, P):
    _chk(a, torch.bfloat16, "a"); _chk(b, torch.bfloat16, "b")
    Cc = a.shape[-1]
    out = torch.empty((F, Cc), device=a.device, dtype=torch.float32)
    _lib.call("ufv_prod_colsum", a.data_ptr(), b.data_ptr(), F, P, Cc, out.data_ptr(), _stream())
    return out


def scale_add_bcast(a, g, s, k, F, P):
    _chk(a, torch.bfloat16, "a"); _chk(g, torch.bfloat16, "g")
    out = torch.empty_like(a)
    _lib.call("ufv_scale_add_bcast", a.data_ptr(), g.data_ptr(), _ptr(s), float(k), out.data_ptr(), F, P, a.shape[-1], _stream())
    return out


def conv3d_scatter(dA, T, H, W, C, k, pad=0):
    _chk(dA, torch.bfloat16, "dA"); assert dA.is_contiguous()
    dx = torch.empty((T * H * W, C), device=dA.device, dtype=torch.bfloat16)
    _lib.call("ufv_conv3d_scatter", dA.data_ptr(), dx.data_ptr(), T, H, W, C, k[0], k[1], k[2], pad, _stream())
    return dx


def avgpool3d(x, T, H, W, C, k):
    """AvgPool3d(k) without the activation: x bf16 [T*H*W, C] -> (bf16 [To*Ho*Wo, C], (To, Ho, Wo))"""
    _chk(x, torch.bfloat16, "x"); assert x.is_contiguous()
    To, Ho, Wo = T // k[0], H // k[1], W // k[2]
    out = torch.empty((To * Ho * Wo, C), device=x.device, dtype=torch.bfloat16)
    _lib.call("ufv_avgpool3d", x.data_ptr(), out.data_ptr(), T, H, W, C, k[0], k[1], k[2], _stream())
    return out, (To, Ho, Wo)


def avgpool3d_bwd(dy, T, H, W, C, k):
    _chk(dy, torch.bfloat16, "dy"); assert dy.is_contiguous()
    dx = torch.empty((T * H * W, C), device=dy.device, dtype=torch.bfloat16)
    _lib.call("ufv_avgpool3d_bwd", dy.data_ptr(), dx.data_ptr(), T, H, W, C, k[0], k[1], k[2], _stream())
    return dx


# ---- [SEG] mask-loss backward (csrc/seg_train.hip) ---------------------------------------------------------------------------------
def small_attn_fwd(q, k, v, B, H, Nq, Nk, hd, scale=None):
    """q [B*Nq, H*hd], k / v [B*Nk, H*hd] bf16 contiguous -> (o bf16 [B*Nq, H*hd], lse fp32 [B, H, Nq])"""
    for t_, n in ((q, "q"), (k, "k"), (v, "v")):
        _chk(t_, torch.bfloat16, n); assert t_.is_contiguous()
    o = torch.empty_like(q)
    lse = torch.empty((B, H, Nq), device=q.device, dtype=torch.float32)
    _lib.call("ufv_small_attn_fwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, Nq, Nk, hd,
              float(hd ** -0.5 if scale is None else scale), _stream())
    return o, lse


def small_attn_bwd(q, k, v, o, dO, lse, B, H, Nq, Nk, hd, scale=None):
    for t_, n in ((q, "q"), (k, "k"), (v, "v"), (o, "o"), (dO, "dO")):
        _chk(t_, torch.bfloat16, n); assert t_.is_contiguous()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty_like(lse)
    _lib.call("ufv_small_attn_bwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), dO.data_ptr(), lse.data_ptr(), delta.data_ptr(),
              dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, H, Nq, Nk, hd, float(hd ** -0.5 if scale is None else scale), _stream())
    return dq, dk, dv


def mask_dot_fwd(up, h, B, P):
    """up bf16 [B*P, C], h fp32 [B, C] -> fp32 [B, P]"""
    _chk(up, torch.bfloat16, "up"); _chk(h, torch.float32, "h"); assert up.is_contiguous() and h.is_contiguous()
    out = torch.empty((B, P), device=up.device, dtype=torch.float32)
    _lib.call("ufv_mask_dot_fwd", up.data_ptr(), h.data_ptr(), out.data_ptr(), B, P, up.shape[1], _stream())
    return out


def mask_dot_bwd(up, h, dm, B, P):
    """-> (d_up bf16 [B*P, C], d_h fp32 [B, C])"""
    _chk(dm, torch.float32, "dm"); assert dm.is_contiguous()
    C = up.shape[1]
    dup = torch.empty_like(up)
    dh = torch.empty_like(h)
    ws = _ws(up.device, _lib.load().ufv_mask_dot_bwd_ws_bytes(B, C), "mask_dot")
    _lib.call("ufv_mask_dot_bwd", up.data_ptr(), h.data_ptr(), dm.data_ptr(), dup.data_ptr(), dh.data_ptr(), ws.data_ptr(), B, P, C, _stream())
    return dup, dh


def resize_bilinear_bwd(dout, in_hw):
    """dout fp32 [N, 1, Hd, Wd] (gradient of resize_bilinear's output) -> fp32 [N, 1, Hs, Ws]"""
    _chk(dout, torch.float32, "dout"); assert dout.is_contiguous()
    N = dout.shape[0] * dout.shape[1]
    din = torch.empty((dout.shape[0], dout.shape[1], in_hw[0], in_hw[1]), device=dout.device, dtype=torch.float32)
    _lib.call("ufv_resize_bilinear_bwd", dout.data_ptr(), din.data_ptr(), N, in_hw[0], in_hw[1], dout.shape[-2], dout.shape[-1], _stream())
    return din


def mask_loss_bwd(x, t, coef, cb):
    """x, t fp32 [N, h, w]; coef fp32 [N, 2] -> dx = cb (sigmoid(x) - t) + sigmoid'(x) (coef[n,0] t + coef[n,1])"""
    _chk(x, torch.float32, "x"); _chk(t, torch.float32, "t"); _chk(coef, torch.float32, "coef")
    assert x.is_contiguous() and t.is_contiguous() and coef.is_contiguous()
    dx = torch.empty_like(x)
    _lib.call("ufv_mask_loss_bwd", x.data_ptr(), t.data_ptr(), coef.data_ptr(), float(cb), dx.data_ptr(), x.shape[0], x.shape[1] * x.shape[2], _stream())
    return dx
