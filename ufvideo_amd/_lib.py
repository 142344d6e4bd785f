"""ctypes binding of include/ufv.h (libufv_hip.so).  The product path has no CPU fallback: if the
library is missing or a call fails, we raise."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product loads the in-tree build, always.  Lab A/B runs of two builds on one box (tools/lab/ab_bench.sh) may name another one with UFV_LIBRARY -- honoured ONLY beside
# UFV_LAB=1, and announced on stderr, so that no environment variable can silently swap the library under a product, test or bench run.
LIB_PATH = os.path.join(_HERE, "libufv_hip.so")
if os.environ.get("UFV_LIBRARY"):
    import sys as _sys
    if os.environ.get("UFV_LAB") == "1":
        LIB_PATH = os.environ["UFV_LIBRARY"]
        print(f"ufvideo_amd: LAB RUN -- loading {LIB_PATH} instead of the in-tree library (UFV_LAB=1 UFV_LIBRARY=...)", file=_sys.stderr, flush=True)
    else:
        print("ufvideo_amd: UFV_LIBRARY is set but UFV_LAB=1 is not: ignored, the in-tree library is loaded", file=_sys.stderr, flush=True)

ACT = {None: 0, "none": 0, "gelu_pytorch_tanh": 1, "gelu_tanh": 1, "gelu": 2, "gelu_erf": 2, "silu": 3, "relu": 4,
       "quick_gelu": 5, "sigmoid": 6}
DT_BF16, DT_F32, DT_F16 = 0, 1, 2
GEMM_AUTO, GEMM_FAST, GEMM_GENERIC, GEMM_GEMV, GEMM_FAST256, GEMM_STREAMK = 0, 1, 2, 3, 4, 5

_p, _i, _f, _l = C.c_void_p, C.c_int, C.c_float, C.c_int64

# name -> argtypes; this table is also what tests/test_host_cpu.py (test_c_abi_exports_every_declared_symbol) checks against include/ufv.h
SIGNATURES = {
    "ufv_small_attn_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p],
    "ufv_small_attn_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p],
    "ufv_mask_dot_fwd": [_p, _p, _p, _i, _i, _i, _p],
    "ufv_mask_dot_bwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "ufv_resize_bilinear_bwd": [_p, _p, _i, _i, _i, _i, _i, _p],
    "ufv_mask_loss_bwd": [_p, _p, _p, _f, _p, _i, _l, _p],
    "ufv_gemm": [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _i, _i, _i, _p],
    "ufv_gemm_stream_bf16": [_p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _i, _p],
    "ufv_layernorm": [_p, _i, _i, _p, _i, _i, _p, _p, _i, _i, _f, _i, _p],
    "ufv_ln_add_silu": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p],
    "ufv_rmsnorm": [_p, _i, _p, _i, _i, _p, _i, _i, _f, _p],
    "ufv_attention": [_p, _l, _l, _p, _l, _l, _p, _l, _l, _p, _l, _l, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _p],
    "ufv_rope_kv": [_p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p],
    "ufv_rope_table": [_p, _i, _i, _i, _p, _p],
    "ufv_rope_kv_table": [_p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p],
    "ufv_quantize_mx": [_p, _i, _l, _p, _l, _p, _l, _i, _i, _p],
    "ufv_dequantize_mx": [_p, _l, _p, _l, _p, _l, _i, _i, _p],
    "ufv_gemm_fp8_mx": [_p, _i, _p, _p, _i, _p, _i, _p, _p, _i, _i, _p, _i, _i, _i, _i, _p, _i, _p, _i, _i, _i, _p],
    "ufv_gemm_qkv_rope_fp8": [_p, _i, _p, _p, _i, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "ufv_gemm_qkv_rope": [_p, _i, _p, _i, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p],
    "ufv_patchify": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "ufv_dwconv3x3_ln_silu": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "ufv_colmean": [_p, _p, _i, _i, _i, _p],
    "ufv_scale_channels": [_p, _p, _i, _i, _i, _p],
    "ufv_conv3d_gather": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "ufv_gather_rows": [_p, _i, _l, _p, _p, _i, _l, _p, _i, _i, _p],
    "ufv_mask_pool": [_p, _i, _p, _p, _p, _i, _i, _i, _p],
    "ufv_argmax": [_p, _i, _p, _p],
    "ufv_argmax_ws": [_p, _i, _p, _p, _p],
    "ufv_preprocess_u8": [_p, _p, _i, _i, _i, _p, _p, _p],
    "ufv_convert": [_p, _i, _p, _i, _l, _p],
    "ufv_im2col": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "ufv_maxpool2x2": [_p, _i, _l, _p, _l, _i, _i, _i, _i, _p],
    "ufv_add_rows": [_p, _i, _l, _p, _l, _p, _i, _i, _p],
    "ufv_upsample2x_add": [_p, _p, _i, _i, _i, _i, _p],
    "ufv_quantize_fp8": [_p, _i, _l, _p, _l, _p, _i, _i, _p],
    "ufv_layernorm_fp8": [_p, _i, _i, _p, _l, _p, _p, _p, _i, _i, _f, _p],
    "ufv_rmsnorm_fp8": [_p, _i, _p, _l, _p, _p, _i, _i, _f, _p],
    "ufv_dequantize_fp8": [_p, _l, _p, _p, _l, _i, _i, _p],
    "ufv_gemm_fp8": [_p, _i, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _i, _i, _i, _p],
    "ufv_cross_entropy_rows": [_p, _l, _p, _i, _i, _l, _p, _p],
    "ufv_mask_loss_sums": [_p, _p, _i, _l, _p, _p],
    "ufv_gemv1": [_p, _p, _p, _f, _p, _i, _p, _p, _i, _i, _i, _p, _i, _p, _i, _p],
    "ufv_resize_bicubic_u8": [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _p, _i, _p],
    "ufv_avgpool3d_silu": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "ufv_add_bcast": [_p, _i, _l, _p, _l, _i, _p, _i, _l, _l, _i, _p],
    "ufv_sam_mask_head": [_p, _l, _p, _l, _p, _p, _i, _i, _i, _i, _i, _p],
    "ufv_resize_bilinear": [_p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _p],
    "ufv_argmax_rows": [_p, _l, _i, _i, _p, _p],
    "ufv_attention_decode": [_p, _l, _p, _l, _l, _p, _l, _l, _p, _l, _i, _i, _i, _i, _i, _f, _p, _i, _p],
    "ufv_qwen2_decode_step": [_p, _p, _i, _p, _l, _p, _p, _p, _p],
    "ufv_qwen2_prefill": [_p, _p, _i, _i, _p, _l, _p, _p, _p, _p],
    "ufv_gemm_timing": [_i],
    "ufv_gemm_prepare": [],
    "ufv_gemm_clear_error": [],
    "ufv_vit_forward": [_p, _p, _i, _i, _i, _i, _i, _p, _p, _l, _p],
    "ufv_stc_forward": [_p, _p, _i, _i, _i, _p, _p, _l, _p],
    "ufv_qwen2_decode_step_dev": [_p, _p, _p, _p, _l, _p, _p, _p, _p],
    "ufv_rope_kv1_dev": [_p, _i, _i, _i, _p, _p, _p, _i, _p],
    "ufv_attention_decode_dev": [_p, _l, _p, _l, _l, _p, _l, _l, _p, _l, _i, _i, _i, _p, _i, _i, _f, _p, _i, _p],
    "ufv_add_int": [_p, _i, _p],
    "ufv_attention_decode_fused": [_p, _i, _i, _i, _p, _i, _p, _p, _i, _i, _p, _f, _p, _i, _p],
    "ufv_graph_begin": [_p],
    "ufv_graph_end": [_p, _p],
    "ufv_graph_launch": [_p, _p],
    "ufv_graph_destroy": [_p],
    "ufv_act": [_p, _p, _l, _i, _p],
    "ufv_act_bwd": [_p, _p, _p, _l, _i, _p],
    "ufv_add_bf16": [_p, _p, _p, _l, _p],
    "ufv_layernorm_bwd": [_p, _l, _p, _p, _p, _l, _p, _l, _p, _p, _i, _i, _f, _i, _p, _p],
    "ufv_ln_add_silu_g": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p],
    "ufv_dwconv3x3": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "ufv_dwconv3x3_dw": [_p, _p, _p, _i, _i, _i, _i, _p, _p],
    "ufv_prod_colsum": [_p, _p, _i, _i, _i, _p, _p],
    "ufv_scale_add_bcast": [_p, _p, _p, _f, _p, _i, _i, _i, _p],
    "ufv_conv3d_scatter": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "ufv_avgpool3d": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "ufv_avgpool3d_bwd": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "ufv_sample_top_p": [_p, _l, _i, _i, _f, _i, _f, _p, _p, _p, _p],
    "ufv_transpose_bf16": [_p, _l, _p, _l, _i, _i, _i, _p],
    "ufv_rmsnorm_bwd": [_p, _i, _p, _p, _i, _p, _i, _i, _p, _i, _i, _i, _f, _p, _p],
    "ufv_colsum_bf16": [_p, _l, _i, _i, _p, _i, _p, _p],
    "ufv_swiglu": [_p, _l, _p, _l, _i, _i, _p],
    "ufv_swiglu_bwd": [_p, _l, _p, _l, _p, _l, _i, _i, _p],
    "ufv_rope_rows": [_p, _l, _i, _i, _i, _i, _p, _i, _i, _p],
    "ufv_cross_entropy_bwd": [_p, _l, _p, _i, _i, _i, _l, _f, _p, _p, _l, _p],
    "ufv_scatter_add_rows": [_p, _l, _p, _p, _l, _i, _i, _p],
    "ufv_sumsq": [_p, _l, _p, _i, _p],
    "ufv_adamw": [_p, _p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _i, _p, _p],
    "ufv_gemm_splitk": [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p],
    "ufv_attention_bwd": [_p, _l, _p, _p, _l, _p, _l, _p, _l, _p, _p, _l, _i, _i, _i, _i, _f, _p, _p],
    "ufv_attention_causal_lse": [_p, _l, _p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _f, _p, _p],
    "ufv_attention_bwd_fused": [_p, _l, _p, _p, _l, _p, _l, _p, _l, _p, _p, _l, _p, _p, _l, _i, _i, _i, _i, _f, _p, _p],
}
# entry points that return a size instead of a status
SIZE_FUNCS = {"ufv_argmax_ws_bytes": ([], _l), "ufv_attention_decode_ws_bytes": ([_i, _i, _i, _i], _i), "ufv_qwen2_decode_ws_bytes": ([_p], _l), "ufv_attention_decode_fused_ws_bytes": ([_i, _i, _i], _l), "ufv_gemm_timing_read": ([_p, _p, _i], _i), "ufv_gemm_choice": ([_i, _i, _i, _i, _i, _i], _i), "ufv_gemm_set_splitk": ([_i], _i), "ufv_gemm_qkv_rope_shape": ([_i, _i, _i, _i, _i], _i), "ufv_gemm_error_state": ([], _i),
              "ufv_qwen2_prefill_ws_bytes": ([_p, _i], _l), "ufv_vit_forward_ws_bytes": ([_p, _i], _l), "ufv_stc_forward_ws_bytes": ([_p, _i, _i], _l),
              "ufv_rmsnorm_bwd_ws_bytes": ([_i], _l), "ufv_attention_bwd_ws_bytes": ([_i, _i, _i, _i], _l), "ufv_attention_bwd_fused_ws_bytes": ([_i, _i], _l),
              "ufv_layernorm_bwd_ws_bytes": ([_i], _l), "ufv_dwconv3x3_dw_ws_bytes": ([_i], _l), "ufv_mask_dot_bwd_ws_bytes": ([_i, _i], _l)}


class Qwen2Layer(C.Structure):
    _fields_ = [("wqkv", _p), ("bqkv", _p), ("wo", _p), ("wgu", _p), ("wd", _p), ("ln1", _p), ("ln2", _p), ("kv_cache", _p),
                ("wqkv8", _p), ("sqkv", _p), ("wo8", _p), ("so", _p), ("wgu8", _p), ("sgu", _p), ("wd8", _p), ("sd", _p)]


class Qwen2Model(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_layers", "d", "n_q", "n_kv", "hd", "d_ff", "vocab", "ldkv", "max_len", "attn_splits")] + \
               [("eps", _f), ("inv_freq", _p), ("norm", _p), ("embed", _p), ("lm_head", _p), ("layers", C.POINTER(Qwen2Layer))]



class VitLayer(C.Structure):
    _fields_ = [(n, _p) for n in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "wqkv", "bqkv", "wo", "bo", "w1", "b1", "w2", "b2")]


class VitModel(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_layers", "d", "n_heads", "d_ff_pad", "patch", "channels", "kpad", "n_patches", "act")] + \
               [("eps", _f), ("patch_w", _p), ("patch_b", _p), ("pos", _p), ("layers", C.POINTER(VitLayer))]



class StcBlock(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("c_in", "c_out", "se_rd", "_pad")] + \
               [(n, _p) for n in ("w1", "n1_w", "n1_b", "w9", "n2_w", "n2_b", "se1_w", "se1_b", "se2_w", "se2_b", "w3", "n3_w", "n3_b", "ds_w", "ds_nw", "ds_nb")]


class StcModel(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("depth", "mlp_depth", "kt", "kh", "kw", "pad", "avgpool", "c_in", "c_hid", "_pad")] + \
               [("eps", _f), ("_padf", _f), ("s1", C.POINTER(StcBlock)), ("s2", C.POINTER(StcBlock)), ("samp_w", _p), ("samp_b", _p),
                ("readout_w", C.POINTER(_p)), ("readout_b", C.POINTER(_p))]

ABI_VERSION = 3          # include/ufv.h UFV_ABI_VERSION
_lib = None


class UfvError(RuntimeError):
    pass


def load():
    """Load libufv_hip.so (built in-tree by `make` / __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UfvError(f"{LIB_PATH} not found: build the HIP extension first (make, or __graft_entry__.build()). "
                       "There is no CPU fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    lib.ufv_last_error.restype = C.c_char_p
    lib.ufv_last_error.argtypes = []
    lib.ufv_abi_version.restype = _i
    lib.ufv_abi_version.argtypes = []
    if lib.ufv_abi_version() != ABI_VERSION:
        raise UfvError(f"{LIB_PATH} has ABI version {lib.ufv_abi_version()}, this binding was written for {ABI_VERSION}: rebuild (make)")
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _i
    for name, (args, res) in SIZE_FUNCS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise UfvError(f"{name} failed ({rc}): {lib.ufv_last_error().decode()}")
