/* ufv.h — C ABI of the MI355X (gfx950) hot-path library `libufv_hip.so`.
 *
 * The reference (Heven-Pan/UFVideo) has no FFI: its hot path is Python calling torch /
 * transformers / timm modules.  This header is the boundary a maintainer binds instead of those
 * modules; each entry point names the reference arithmetic it replaces (file:line relative to
 * the reference tree, or the pinned third-party module the reference calls there).
 *
 * Conventions
 *  - plain C: raw device pointers + sizes; no C++/torch types cross the boundary.
 *  - all buffers are caller-owned device memory on the current HIP device; every call is enqueued on
 *    `stream` (a hipStream_t; NULL = the default stream) and returns immediately; nothing is allocated,
 *    freed or synchronised inside, with ONE exception: the split-K form of ufv_gemm (what
 *    UFV_GEMM_AUTO takes for few-tile / long-K fp32 products, and the opt-in UFV_GEMM_STREAMK) keeps a
 *    per-device record inside the library -- a 4 MiB ring of turn flags and a pinned error word,
 *    allocated under a mutex on the first such launch on a device or by ufv_gemm_prepare() (call it
 *    before stream capture).  Every launch gets a private slice of the ring and its own ticket base,
 *    so split-K GEMMs may be in flight on several streams and devices at once.
 *  - return value: 0 = ok, <0 = error (UFV_E*); `ufv_last_error()` returns a thread-local message.
 *  - bf16 = bfloat16 storage (uint16_t bit pattern), activations bf16, accumulation fp32, the
 *    residual streams fp32.  "ld*" arguments are row pitches in ELEMENTS.
 *  - thread-safe: the only process-wide state is that per-device record (mutex-guarded), the
 *    split-K on/off switch (an atomic, ufv_gemm_set_splitk) and the measurement aid ufv_gemm_timing
 *    (not thread-safe, off by default); the error string is thread-local.
 *  - forward progress of the split-K form: a tile's K parts add into the output in turn order and a
 *    part waits for its predecessor's flag -- BOUNDED: if a predecessor is not scheduled within ~2^21
 *    polls (seconds; only possible when the launch's <= 256 blocks cannot all become resident beside
 *    another long-running kernel) the waiter sets the device's error word and goes on; that launch's
 *    tile is wrong, nothing hangs, and the next SPLIT-K launch on the device returns UFV_EHIP (the
 *    word is read where the flag ring is handed out; unsplit GEMMs never wait on a turn and are not
 *    gated).  ufv_gemm_error_state reads the word at any time; the kernels set it WHEN THEY RUN, so a
 *    caller that needs the guarantee per step synchronises the stream first and then reads it
 *    (ufvideo_amd.train.DecoderTrainer.step does, before the update touches the fp32 masters, and with
 *    several ranks it all-reduces the flag so that every rank refuses the step together).  A negative
 *    return is UFV_EHIP (no current device), not a time-out.  ufv_gemm_clear_error resets the word.
 */
#ifndef UFV_H_
#define UFV_H_

#include <stdint.h>

/* Every entry point below carries UFV_API; the library is built with -fvisibility=hidden, so these are ALL of its dynamic symbols
 * (tests/test_host_cpu.py::test_c_abi_exports_every_declared_symbol compares `nm -D` with this header, both ways). */
#ifndef UFV_API
#define UFV_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* 2: ufv_conv3d_scatter gained `pad`; ufv_qwen2_decode_step needs a zero-filled workspace; ufv_gemm_prepare / _set_splitk / _error_state /
 *    _clear_error added.
 * 3 (round 5): ufv_gemm_qkv_rope / _fp8 / _shape, ufv_quantize_mx / ufv_dequantize_mx / ufv_gemm_fp8_mx (with `resid_bf16`), ufv_gemm_stream_bf16 added; no existing
 *    entry changed its signature.  A binder checks ufv_abi_version() == UFV_ABI_VERSION at load (ufvideo_amd/_lib.py does). */
#define UFV_ABI_VERSION 3

/* activation ids (ufv_gemm `act`, ufv_layernorm `act`) */
#define UFV_ACT_NONE 0
#define UFV_ACT_GELU_TANH 1  /* HF ACT2FN["gelu_pytorch_tanh"] (SigLIP MLP) */
#define UFV_ACT_GELU_ERF 2   /* nn.GELU() (projector readout, region encoder) */
#define UFV_ACT_SILU 3
#define UFV_ACT_RELU 4
#define UFV_ACT_QUICK_GELU 5 /* CLIP MLP */
#define UFV_ACT_SIGMOID 6

/* kernel selector for ufv_gemm */
#define UFV_GEMM_AUTO 0
#define UFV_GEMM_FAST 1    /* 128x128x64 MFMA tile kernel; errors out if the shape does not qualify */
#define UFV_GEMM_GENERIC 2 /* one-thread-per-output kernel, any shape */
#define UFV_GEMM_GEMV 3    /* weight-streaming kernel for M <= 64 */
#define UFV_GEMM_FAST256 4 /* 8-wave ping-pong MFMA kernel, 256x256x64 tile (large M) */
#define UFV_GEMM_PP(shape) (4 | ((shape) << 8)) /* the ping-pong kernel at a named tile shape, 1000 + MA0 MA1 NB1 as decimal digits:
                                                1442 = 256x256 (= UFV_GEMM_FAST256), 1432 = 224x256, 1332 = 192x256, 1322 = 160x256,
                                                1441 = 256x192, 1431 = 224x192, 1331 = 192x192; N % tile width must be 0 or 128.
                                                + 10000 * parts: the aligned split-K form (fp32 output, no activation): the parts of a tile add into the
                                                output in turn order, e.g. 41441 = 256x192 tiles, K in 4 parts */
#define UFV_GEMM_STREAMK 5 /* the same kernel with stream-K work split: every CU gets the same number of K-tile iterations */

/* dtype ids for inputs that may arrive in several formats */
#define UFV_DT_BF16 0
#define UFV_DT_F32 1
#define UFV_DT_F16 2

UFV_API const char* ufv_last_error(void);
UFV_API int ufv_abi_version(void);

/* C[M,N] = epilogue(A[M,K] * W[N,K]^T): every nn.Linear / 1x1 conv / patch-embed / Conv3d-as-GEMM
 * on the path (torch F.linear under HF modeling_siglip.py:267-270,310-322; modeling_qwen2.py:35-48,
 * 176-235; projector.py:125-130,153-184).  A, W bf16.  out = act(acc + bias[n]) + resid[m(,% resid_rows)][n];
 * `out_f32` selects fp32 vs bf16 C.  `swiglu`: W rows packed [16 gate | 16 up] alternating, C has N/2
 * columns = silu(gate)*up (modeling_qwen2.py:47). */
UFV_API int ufv_gemm(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int out_f32, int M, int N, int K,
             const float* bias, int act, const float* resid, int ldr, int resid_rows, int swiglu, int kernel,
             void* stream);

/* x[M,N] = bf16(A[M,K] * W[N,K]^T + bias[n] + float(x[M,N])), x bf16 updated in place: the residual add of a bf16 module on its bf16 hidden states
 * (modeling_siglip.py:371-383 `hidden_states = residual + hidden_states` with the tower loaded in bfloat16, as videorefer_arch.py builds it).  One rounding
 * per element, of the fp32 sum -- the torch bf16 add rounds the projection first; the stream's storage points are the reference's. */
UFV_API int ufv_gemm_stream_bf16(const void* A, int lda, const void* W, int ldw, void* x, int ldx, int M, int N, int K, const float* bias, int kernel,
                         void* stream);

/* y = act(LayerNorm(x) * w + b) per row (nn.LayerNorm: modeling_siglip.py:329-331; timm LayerNorm2d in
 * NHWC).  x fp32 or bf16 (x_dtype), y bf16 (or fp32 when y_f32). */
UFV_API int ufv_layernorm(const void* x, int x_dtype, int ldx, void* y, int y_f32, int ldy, const float* w, const float* b,
                  int M, int D, float eps, int act, void* stream);

/* out = silu(LN_a(a) + (LN_b(b) | b)) — tail of a timm RegNet bottleneck: conv3 norm + shortcut
 * (downsample norm when wb != NULL) + act3.  a, b, out bf16 [M, D]. */
UFV_API int ufv_ln_add_silu(const void* a, const float* wa, const float* ba, const void* b, const float* wb, const float* bb,
                    void* out, int M, int D, float eps, void* stream);

/* Qwen2RMSNorm (modeling_qwen2.py:238-254): y = w * (x * rsqrt(mean(x^2) + eps)); x fp32, y bf16 (fp32 if y_f32). */
UFV_API int ufv_rmsnorm(const float* x, int ldx, void* y, int y_f32, int ldy, const float* w, int M, int D, float eps,
                void* stream);

/* softmax(q k^T * scale [+ causal]) v, flash-style, fp32 softmax (modeling_siglip.py:227-247,
 * modeling_qwen2.py:150-172).  q/k/v/o bf16; element (b, s, h, d) of X lives at
 * X + b*x_bs + s*x_ss + h*hd + d.  Hq % Hkv == 0 (GQA).  causal: key j visible to query i iff
 * j <= q_pos0 + i.  kernel: 0 auto, 1 MFMA kernel (hd in {64,72,80,96,128}), 2 generic,
 * 5 few-keys kernel (Sk <= 64, hd 16|32, non-causal: SAM2 image->token cross attention); diagnostic ids that name ONE kernel and fail
 * with UFV_EUNSUPPORTED outside its envelope: 11 / 14 the second / third generation hd-72 kernels (SigLIP), 12 / 13 the hd-128 kernel
 * with / without the key split over two wave groups, 15 the causal hd-128 prefill kernel of csrc/attn_c128.inc (B = 1, Sq = Sk >= 64,
 * q_pos0 = 0, >= 2 kv heads, >= 2 q heads per kv head; what auto takes from Sq = 128 on, bit-identical to 13), 16 the small-window
 * kernel (non-causal head_dim 72, Sq, Sk <= 16: Hiera's 4 x 4 windows; auto from 256 (window, head) pairs on). */
UFV_API int ufv_attention(const void* q, int64_t q_bs, int64_t q_ss, const void* k, int64_t k_bs, int64_t k_ss, const void* v,
                  int64_t v_bs, int64_t v_ss, void* o, int64_t o_bs, int64_t o_ss, int B, int Hq, int Hkv, int Sq,
                  int Sk, int hd, float scale, int causal, int q_pos0, int kernel, void* stream);

/* Decode-step attention (one query token against the KV cache; modeling_qwen2.py:150-172 with q_len 1): keys are
 * split over `nsplit` blocks per head and merged; `ws` = ufv_attention_decode_ws_bytes(...) bytes of scratch.
 * q element (b, h, d) at q + b*q_bs + h*hd + d; k/v as in ufv_attention; o at o + b*o_bs + h*hd + d. */
UFV_API int ufv_attention_decode_ws_bytes(int B, int Hq, int hd, int nsplit);
UFV_API int ufv_attention_decode(const void* q, int64_t q_bs, const void* k, int64_t k_bs, int64_t k_ss, const void* v, int64_t v_bs,
                         int64_t v_ss, void* o, int64_t o_bs, int B, int Hq, int Hkv, int Sk, int hd, float scale, void* ws,
                         int nsplit, void* stream);

/* RoPE rotate-half (modeling_qwen2.py:105-135) applied in place to the q columns of a fused qkv
 * buffer [S, ldqkv] and, for k, written together with v into the KV cache rows pos0..pos0+S-1
 * (cache row = [Hkv*hd k | Hkv*hd v]).  angle = (pos0+s) * inv_freq[i] in fp32. */
UFV_API int ufv_rope_kv(void* qkv, int ldqkv, int S, int Hq, int Hkv, int hd, const float* inv_freq, int pos0, void* kv_cache,
                int ldkv, void* stream);
/* the same with cos / sin of all S positions precomputed once per forward pass (table fp32 [S, hd]: cos | sin per row, row s =
 * position pos0+s; bit-identical to ufv_rope_kv): every layer re-uses the table instead of 16 sincos per thread */
UFV_API int ufv_rope_table(const float* inv_freq, int pos0, int S, int hd, float* table, void* stream);
UFV_API int ufv_rope_kv_table(void* qkv, int ldqkv, int S, int Hq, int Hkv, int hd, const float* table, int pos0, void* kv_cache, int ldkv,
                      void* stream);
/* The three calls above as ONE launch for the prefill (round 5): q / k / v Linear + bias, rotate-half RoPE of the q and k heads with the cos | sin table of
 * ufv_rope_table, append of the k / v rows to the KV cache (modeling_qwen2.py:176-205 under videorefer_qwen2.py:187-197).  A, W, bias as for ufv_gemm with
 * W = [q | k | v] rows ((Hq + 2 Hkv) * hd, K); q_out bf16 [S, ldq] receives the rotated q heads (Hq * hd columns); row s of the cache, kv_row0 + s * ldkv
 * (bf16, [Hkv * hd k | Hkv * hd v]), receives the rotated k heads and the v heads -- kv_row0 is the cache row of the call's FIRST position, which is also
 * row 0 of `rope_table` (f32 [S, hd]).  Bit-identical to ufv_gemm (bf16 output) followed by ufv_rope_kv_table: the same operations in the same order, the
 * rounding to bf16 between them included.  Built for head_dim 128 and S >= 256 at the tile shape 1332 (192 x 256); ufv_gemm_qkv_rope_shape returns the shape it
 * would run (host arithmetic only) or 0 when the caller should issue the unfused pair (also when the environment variable UFV_NO_FUSED_ROPE is set);
 * shape = 0 lets the call choose. */
UFV_API int ufv_gemm_qkv_rope_shape(int S, int Hq, int Hkv, int hd, int K);
/* the W8A8 form (e4m3 codes + fp32 row scales of A, per-channel scales of W, as ufv_gemm_fp8): same epilogue behind the de-quantising scale; head_dim 128, S >= 256 */
UFV_API int ufv_gemm_qkv_rope_fp8(const void* A, int lda, const float* a_scale, const void* W, int ldw, const float* w_scale, const float* bias, void* q_out, int ldq,
                          void* kv_row0, int ldkv, int S, int Hq, int Hkv, int hd, int K, const float* rope_table, void* stream);
UFV_API int ufv_gemm_qkv_rope(const void* A, int lda, const void* W, int ldw, const float* bias, void* q_out, int ldq, void* kv_row0, int ldkv, int S, int Hq,
                      int Hkv, int hd, int K, const float* rope_table, int shape, void* stream);

/* Conv2d(k=s=P, valid) as im2col: pixels [T,C,H,W] (dtype id) -> bf16 [T*(H/P)*(W/P), Kpad],
 * k = c*P*P + py*P + px, zero-filled to Kpad (modeling_siglip.py:124-130,178-179). */
UFV_API int ufv_patchify(const void* pixels, int dtype, void* out, int T, int C, int H, int W, int P, int Kpad, void* stream);

/* depthwise 3x3 (pad 1) -> LayerNorm over C -> SiLU, NHWC bf16 [F,H,W,C]; w9 fp32 [9][C]
 * (timm ConvNormAct with groups=C inside RegNet Bottleneck; projector.py:153-184). */
UFV_API int ufv_dwconv3x3_ln_silu(const void* x, void* y, const float* w9, const float* lnw, const float* lnb, int F, int H, int W,
                          int C, float eps, void* stream);

/* out[f, c] = mean over P rows of x[f*P + p, c]  (SE squeeze; x bf16 [F*P, C], out bf16 [F, C]) */
UFV_API int ufv_colmean(const void* x, void* out, int F, int P, int C, void* stream);

/* x[f*P + p, c] *= gate[f, c]   (SE excite; bf16 in place) */
UFV_API int ufv_scale_channels(void* x, const void* gate, int F, int P, int C, void* stream);

/* nn.AvgPool3d(k) + nn.SiLU sampler of STPConnector / SpatialPool (projector.py:218-222,247-250): x bf16 [T,H,W,C] token-major
 * -> bf16 [T/kt, H/kh, W/kw, C] (floor, no padding). */
UFV_API int ufv_avgpool3d_silu(const void* x, void* out, int T, int H, int W, int C, int kt, int kh, int kw, void* stream);
/* Conv3d patch gather (kernel = stride = (kt,kh,kw), zero padding `pad` on all three dims):
 * x bf16 [T,H,W,C] -> out bf16 [To*Ho*Wo, kt*kh*kw*C], k = ((dt*kh+dh)*kw+dw)*C + c (projector.py:164-172,229-237) */
UFV_API int ufv_conv3d_gather(const void* x, void* out, int T, int H, int W, int C, int kt, int kh, int kw, int pad, void* stream);

/* dst[dst_idx[i]] = src[src_idx[i]] for rows of `D` elements; src dtype bf16 or f32, dst fp32 or bf16
 * (embed_tokens gather + splice, videorefer_arch.py:280-316; dst_idx/src_idx int64, NULL = identity) */
UFV_API int ufv_gather_rows(const void* src, int src_dtype, int64_t ld_src, const int64_t* src_idx, void* dst, int dst_dtype,
                    int64_t ld_dst, const int64_t* dst_idx, int n, int D, void* stream);

/* masked mean pooling (layer.py:135-152): feat bf16/f32 [n, P, C] (token-major), mask f32 [q, P] already
 * resized+binarised; out f32 [q, C]; frame index of mask i = frame_of[i] */
UFV_API int ufv_mask_pool(const void* feat, int feat_dtype, const float* mask, const int32_t* frame_of, float* out, int q, int P,
                  int C, void* stream);

/* ---- SAM2 image-encoder pieces (ufvideo/model/sam2.py) -------------------------------------------------------- */
/* general Conv2d im2col (Hiera PatchEmbed 7x7/s4/p3, sam2.py:954-984): pixels [B,C,H,W] -> bf16 [B*Ho*Wo, Kpad] */
UFV_API int ufv_im2col(const void* pixels, int dtype, void* out, int B, int C, int H, int W, int ks, int stride, int pad, int Kpad,
               void* stream);
/* MaxPool2d(2,2) over a token grid (Hiera q-pooling / pooled shortcut, sam2.py:986-998): rows [Bw*H*W] x C, f32 or bf16 */
UFV_API int ufv_maxpool2x2(const void* x, int dtype, int64_t ldx, void* out, int64_t ldo, int Bw, int H, int W, int C, void* stream);
/* dst[dst_idx[i]] += src[i] (window un-partition + residual add, sam2.py:927-951,1126); dst f32, idx<0 = padding */
UFV_API int ufv_add_rows(const void* src, int src_dtype, int64_t ld_src, float* dst, int64_t ld_dst, const int64_t* dst_idx, int n, int D,
                 void* stream);
/* FPN top-down path, nearest x2 (sam2.py:885-896): x[b,y,x,:] += prev[b,y/2,x/2,:], f32 NHWC */
UFV_API int ufv_upsample2x_add(float* x, const float* prev, int B, int H, int W, int C, void* stream);

/* greedy sampling: out[0] = argmax(logits[0..N)) with torch.argmax tie-breaking (lowest index) */
UFV_API int ufv_argmax(const float* logits, int N, int64_t* out, void* stream);
/* the same result from 64 blocks (the decode step's form): `ws` = ufv_argmax_ws_bytes() bytes whose LAST int is zero before the first call (it returns to zero) */
UFV_API int64_t ufv_argmax_ws_bytes(void);
UFV_API int ufv_argmax_ws(const float* logits, int N, int64_t* out, void* ws, void* stream);

/* frame batching tail of process_video (mm_utils.py:284,291): u8 HWC frames -> (x/255 - mean)/std -> bf16 NCHW */
UFV_API int ufv_preprocess_u8(const uint8_t* frames, void* out, int T, int H, int W, const float* mean3, const float* std3,
                      void* stream);

/* Single-row GEMV of the decode step: y[N(/2)] = epilogue(h . W^T) with h = the bf16 row `a`, or -- when `x` (fp32 row) is
 * given instead -- bf16(RMSNorm(x) * ln_w) computed in the kernel's prologue bit-identically to ufv_rmsnorm, which removes
 * the separate norm launch (Qwen2RMSNorm + q/k/v or gate/up Linear, modeling_qwen2.py:35-48,150-172,238-254).
 * Epilogue as ufv_gemm (bias -> act -> fp32 residual [N], or SwiGLU); K <= 21845, weights streamed once.
 * w_scale != NULL: W holds e4m3 bytes with one scale per row and the row h is quantised in the prologue exactly as
 * ufv_quantize_fp8 would (W8A8, the decode side of config #5a). */
UFV_API int ufv_gemv1(const void* a, const float* x, const float* ln_w, float eps, const void* W, int ldw, const float* w_scale, void* C,
              int out_f32, int N, int K, const float* bias, int act, const float* resid, int swiglu, void* stream);

/* Resize step of the frame batching (mm_utils.py:269-295 -> HF SiglipImageProcessor -> PIL Image.resize(BICUBIC)): Pillow's
 * 8-bit separable resample reproduced bit for bit on uint8 HWC frames [T,H,W,3] -> [T,Ho,Wo,3].  bounds_* int32 [out,2] = (first
 * input index, tap count), coeff_* int32 [out, ksize] fixed point 2^22 (host: ufvideo_amd.mm_utils.pil_resize_coeffs);
 * tmp = [T,H,Wo,3] scratch when both passes run. */
UFV_API int ufv_resize_bicubic_u8(const uint8_t* frames, uint8_t* tmp, uint8_t* out, int T, int H, int W, int Ho, int Wo,
                          const int32_t* bounds_x, const int32_t* coeff_x, int ksize_x, const int32_t* bounds_y, const int32_t* coeff_y,
                          int ksize_y, void* stream);

/* ---- one-call greedy decode step (replaces HF GenerationMixin's per-token Qwen2ForCausalLM.forward under
 * videorefer_qwen2.py:414-426).  All pointers are device memory prepared by the caller (packed weights as for the
 * op-level calls: wqkv = [q|k|v] rows, wgu = gate/up rows interleaved in blocks of 16). */
typedef struct {
    const void* wqkv;   /* bf16 [(Hq+2Hkv)*hd, d] */
    const float* bqkv;  /* f32  [(Hq+2Hkv)*hd] */
    const void* wo;     /* bf16 [d, Hq*hd] */
    const void* wgu;    /* bf16 [2*d_ff, d] (SwiGLU packing) */
    const void* wd;     /* bf16 [d, d_ff] */
    const float* ln1;   /* f32 [d] input_layernorm */
    const float* ln2;   /* f32 [d] post_attention_layernorm */
    void* kv_cache;     /* bf16 [max_len, ldkv]: row = [Hkv*hd k | Hkv*hd v] */
    /* optional W8A8 decode (all four NULL = bf16): e4m3 weights in the same layouts + one fp32 scale per weight row */
    const void* wqkv8; const float* sqkv; const void* wo8; const float* so; const void* wgu8; const float* sgu; const void* wd8; const float* sd;
} ufv_qwen2_layer;

typedef struct {
    int32_t n_layers, d, n_q, n_kv, hd, d_ff, vocab, ldkv, max_len, attn_splits;
    float eps;
    const float* inv_freq; /* f32 [hd/2] */
    const float* norm;     /* f32 [d] */
    const void* embed;     /* bf16 [vocab, d] */
    const void* lm_head;   /* bf16 [vocab, d] */
    const ufv_qwen2_layer* layers;
} ufv_qwen2_model;

/* the workspace must be ZERO-filled once before the first step (it holds the arrival counters of the fused decode attention) */
UFV_API int64_t ufv_qwen2_decode_ws_bytes(const ufv_qwen2_model* m);
/* token_dev: previous token id (device int64[1]); pos: its position (= number of cached tokens); writes logits f32
 * [vocab], optional hidden_out f32 [d] (final-norm hidden state), next_token_dev = argmax. */
UFV_API int ufv_qwen2_decode_step(const ufv_qwen2_model* m, const int64_t* token_dev, int pos, void* ws, int64_t ws_bytes, float* logits,
                          float* hidden_out, int64_t* next_token_dev, void* stream);
/* ---- whole-stage calls: the layer loops as ONE call each; compositions of the op-level entry points above on one
 * stream, bit-identical to issuing that sequence from the host.
 * Prefill of S > 0 positions pos0 .. pos0+S-1 of one sequence (HF Qwen2Model.forward under videorefer_qwen2.py:154-196):
 * x f32 [S, d] holds inputs_embeds on entry and the last layer's output on exit (in place); every layer's K / V rows are written
 * to its kv_cache (which must hold pos0 + S rows); hidden_layers (optional) f32 [n_layers-1, S, d] = the stream after each layer but
 * the last (HF output_hidden_states[1:-1]); normed (optional) f32 [S, d] = final RMSNorm of all rows (= hidden_states[-1]);
 * logits_last (optional) f32 [vocab] = lm_head of the last position. */
UFV_API int64_t ufv_qwen2_prefill_ws_bytes(const ufv_qwen2_model* m, int S);
UFV_API int ufv_qwen2_prefill(const ufv_qwen2_model* m, float* x, int S, int pos0, void* ws, int64_t ws_bytes, float* hidden_layers, float* normed,
                      float* logits_last, void* stream);

/* SigLIP vision tower (HF SiglipVisionTransformer embeddings + the first n_layers encoder layers, as the reference's
 * `hidden_states[select_layer]` asks; encoder.py:96-146): packed weights as for the op-level calls -- q|k|v rows fused, the MLP
 * width zero-padded to a multiple of 128, the patch kernel flattened to [d, kpad] with k = c*P*P + py*P + px. */
typedef struct {
    const float* ln1_w; const float* ln1_b; const float* ln2_w; const float* ln2_b;
    const void* wqkv; const float* bqkv;   /* bf16 [3d, d],        f32 [3d] */
    const void* wo; const float* bo;       /* bf16 [d, d],         f32 [d] */
    const void* w1; const float* b1;       /* bf16 [d_ff_pad, d],  f32 [d_ff_pad] (zero rows past d_ff) */
    const void* w2; const float* b2;       /* bf16 [d, d_ff_pad],  f32 [d] */
} ufv_vit_layer;

typedef struct {
    int32_t n_layers, d, n_heads, d_ff_pad, patch, channels, kpad, n_patches, act;
    float eps;
    const void* patch_w;    /* bf16 [d, kpad] */
    const float* patch_b;   /* f32 [d] */
    const float* pos;       /* f32 [n_patches, d] position embedding */
    const ufv_vit_layer* layers;
} ufv_vit_model;

UFV_API int64_t ufv_vit_forward_ws_bytes(const ufv_vit_model* m, int T);
/* pixels [T, channels, H, W] of dtype id `dtype` -> x f32 [T * n_patches, d]: the residual stream after n_layers layers */
UFV_API int ufv_vit_forward(const ufv_vit_model* m, const void* pixels, int dtype, int T, int H, int W, int n_layers, float* x, void* ws,
                    int64_t ws_bytes, void* stream);

/* STC connector = `temporal_aggregator` for the stc_connector / stc_connector_v35 / stp_connector / spatial_conv / spatial_pool projectors
 * (projector.py:133-250): RegStage x depth -> Conv3d(kernel = stride = (kt,kh,kw), padding pad) + SiLU | AvgPool3d + SiLU -> RegStage x depth ->
 * readout MLP (Linear, GELU, ..., Linear).  One RegNet bottleneck (timm Bottleneck, bottle_ratio 1, group width 1 = depthwise, SE):
 * conv1 1x1 -> LN -> SiLU -> depthwise 3x3 -> LN -> SiLU -> SE -> conv3 1x1 -> LN -> (+ shortcut [conv 1x1 -> LN when c_in != c_out]) -> SiLU. */
typedef struct {
    int32_t c_in, c_out, se_rd, _pad;
    const void* w1; const float* n1_w; const float* n1_b;                    /* bf16 [c_out, c_in] */
    const float* w9; const float* n2_w; const float* n2_b;                   /* f32 [9, c_out] depthwise taps */
    const void* se1_w; const float* se1_b; const void* se2_w; const float* se2_b;   /* bf16 [se_rd, c_out], bf16 [c_out, se_rd] */
    const void* w3; const float* n3_w; const float* n3_b;                    /* bf16 [c_out, c_out] */
    const void* ds_w; const float* ds_nw; const float* ds_nb;                /* shortcut conv + norm, NULL when c_in == c_out */
} ufv_stc_block;

typedef struct {
    int32_t depth, mlp_depth, kt, kh, kw, pad, avgpool, c_in, c_hid, _pad;
    float eps, _padf;
    const ufv_stc_block* s1; const ufv_stc_block* s2;                        /* depth blocks each */
    const void* samp_w; const float* samp_b;                                 /* bf16 [c_hid, kt*kh*kw*C], k = ((dt*kh+dh)*kw+dw)*C + c; NULL with avgpool */
    const void* const* readout_w; const float* const* readout_b;            /* mlp_depth entries: bf16 [c_hid, c_hid], f32 [c_hid] */
} ufv_stc_model;

UFV_API int64_t ufv_stc_forward_ws_bytes(const ufv_stc_model* m, int T, int HW);
/* x [T*HW*HW, c_in] token-major features of ONE video (dtype id x_dtype) -> out f32 [To*Ho*Wo, c_hid] */
UFV_API int ufv_stc_forward(const ufv_stc_model* m, const void* x, int x_dtype, int T, int HW, float* out, void* ws, int64_t ws_bytes, void* stream);

/* the same with the position in device memory (*pos_dev, incremented by the step): every launch argument is identical from
 * token to token, so the step can be recorded once into a HIP graph and replayed (ufv_graph_*) */
UFV_API int ufv_qwen2_decode_step_dev(const ufv_qwen2_model* m, const int64_t* token_dev, int* pos_dev, void* ws, int64_t ws_bytes, float* logits,
                              float* hidden_out, int64_t* next_token_dev, void* stream);
/* pieces of it: RoPE + KV append of ONE token at position *pos_dev; decode attention over *pos_dev + 1 <= max_keys keys; *p += v */
UFV_API int ufv_rope_kv1_dev(void* qkv, int Hq, int Hkv, int hd, const float* inv_freq, const int* pos_dev, void* kv_cache, int ldkv, void* stream);
UFV_API int ufv_attention_decode_dev(const void* q, int64_t q_bs, const void* k, int64_t k_bs, int64_t k_ss, const void* v, int64_t v_bs,
                             int64_t v_ss, void* o, int64_t o_bs, int B, int Hq, int Hkv, const int* pos_dev, int max_keys, int hd,
                             float scale, void* ws, int nsplit, void* stream);
UFV_API int ufv_add_int(int* p, int v, void* stream);
/* the three of them in ONE launch (RoPE of the new token's q / k, KV-cache append at `pos`, split attention over pos + 1 keys, merge): qkv bf16
 * [(Hq + 2 Hkv) * hd] = the raw projections of the new token (left untouched), kv_cache row = [Hkv*hd k | Hkv*hd v]; pos from the host or
 * *pos_dev; o bf16 [Hq * hd].  ws: ufv_attention_decode_fused_ws_bytes(...) bytes whose trailing Hq ints (arrival counters) are ZERO before the
 * first call -- they return to zero by themselves.  head_dim 64 / 128.  Bit-identical to the three calls. */
UFV_API int64_t ufv_attention_decode_fused_ws_bytes(int Hq, int hd, int nsplit);
UFV_API int ufv_attention_decode_fused(const void* qkv, int Hq, int Hkv, int hd, const float* inv_freq, int pos, const int* pos_dev, void* kv_cache, int ldkv,
                               int max_keys, void* o, float scale, void* ws, int nsplit, void* stream);
/* HIP-graph capture of a launch-bound call sequence: begin; ufv_* calls on `stream` (a created stream, not the legacy default
 * one) are recorded instead of executed; end returns an executable graph; launch replays it on a stream */
UFV_API int ufv_graph_begin(void* stream);
UFV_API int ufv_graph_end(void* stream, void** exec_out);
UFV_API int ufv_graph_launch(void* exec, void* stream);
UFV_API int ufv_graph_destroy(void* exec);

/* ---- W8A8 fp8 GEMM path (SURVEY §8f row 1 / BASELINE config #5a; not in the reference, which runs bf16/fp16) ----
 * OCP e4m3 operands with one fp32 scale per row: x[m,k] ~ q[m,k] * scale[m], scale = max|x[m,:]| / 448,
 * q = rne_e4m3(x * (1/scale)). */
UFV_API int ufv_quantize_fp8(const void* x, int x_dtype, int64_t ldx, void* q, int64_t ldq, float* scale, int M, int K, void* stream);
/* ufv_layernorm / ufv_rmsnorm with the bf16 result quantised in the same pass (bit-identical to norm followed by
 * ufv_quantize_fp8): q e4m3 [M, D], scale f32 [M]. */
UFV_API int ufv_layernorm_fp8(const void* x, int x_dtype, int ldx, void* q, int64_t ldq, float* scale, const float* w, const float* b, int M,
                      int D, float eps, void* stream);
UFV_API int ufv_rmsnorm_fp8(const float* x, int ldx, void* q, int64_t ldq, float* scale, const float* w, int M, int D, float eps, void* stream);
/* MX-style block quantisation (round 5; OCP microscaling blocks as v_mfma_scale_f32_16x16x128_f8f6f4 consumes them): x [M, K] -> e4m3 codes q [M, K] + one e8m0
 * scale byte per (row, 32 consecutive elements): scale = 2^(byte - 127) = amax / 448 of the block rounded UP to a power of two (no element saturates),
 * code = rne_e4m3(x / scale).  The scale bytes are stored as bscale [ceil(M / 64)][ceil(K / 512)][64 rows][16] (ldb = bytes per 64-row block >= 1024 ceil(K / 512)):
 * per (row, group of four 128-element K-tiles) 16 bytes, byte 4 f + t = the scale of block f (0..3) of K-tile t (0..3) of the group.  64 rows of a group are one
 * contiguous KiB (one LDS-DMA piece per wave and four K-tiles), the groups of a row block follow each other along K (the GEMM's scale stream stays inside a few
 * pages), and a lane of the scaled MFMA finds the four K-tiles of its block f in one dword.  K % 128 == 0; the buffer holds ceil(M / 64) * ldb bytes.  The e4m3 GEMM epilogues emit the same
 * format (ufv_gemm_fp8_mx). */
UFV_API int ufv_quantize_mx(const void* x, int x_dtype, int64_t ldx, void* q, int64_t ldq, void* bscale, int64_t ldb, int M, int K, void* stream);
UFV_API int ufv_dequantize_mx(const void* q, int64_t ldq, const void* bscale, int64_t ldb, float* out, int64_t ldo, int M, int K, void* stream);
/* ufv_gemm_fp8 with block scales on the activation side: exactly one of a_scale (fp32 per row) / a_bscale (e8m0 per row and 32 K-elements, pitch ld_abs bytes);
 * out_bscale != NULL: C receives e4m3 codes (ldc = byte pitch) and out_bscale [M, ld_obs] their block scales, written by the GEMM's epilogue -- the A operand of the
 * next e4m3 GEMM without a quantise launch in between (modeling_siglip.py:310-322 fc1 -> fc2, modeling_qwen2.py:35-48 gate/up -> down).  With `swiglu` the 32-column
 * blocks of an output row are stored in a fixed permutation inside every group of 128 columns (physical column 32 w + 16 h + c holds logical column 64 h + 16 w + c,
 * w < 4, h < 2, c < 16): the consumer's weight carries the same permutation on its K axis (ufvideo_amd.ops.Fp8Weight(..., mx_swiglu_cols=True)).
 * M >= 256, N % 128 == 0 (N % 256 == 0 for an MX output), K % 128 == 0. */
UFV_API int ufv_gemm_fp8_mx(const void* A, int lda, const float* a_scale, const void* a_bscale, int ld_abs, const void* W, int ldw, const float* w_scale, void* C, int ldc,
                    int out_f32, void* out_bscale, int ld_obs, int M, int N, int K, const float* bias, int act, const void* resid, int ldr, int resid_bf16, int swiglu,
                    void* stream);          /* resid: fp32 [M, ldr], or bf16 when resid_bf16 (bf16 outputs only: the in-place update of a bf16 residual stream) */
UFV_API int ufv_dequantize_fp8(const void* q, int64_t ldq, const float* scale, float* out, int64_t ldo, int M, int K, void* stream);
/* ufv_gemm with e4m3 A [M,K] (a_scale [M]) and W [N,K] (w_scale [N]):  C = epilogue((Aq Wq^T) * a_scale[m] * w_scale[n]).
 * v_mfma_f32_16x16x128_f8f6f4 tiles (N % 128 == 0, K % 128 == 0) or the fp8 GEMV (M <= 64, K % 16 == 0); same epilogues. */
UFV_API int ufv_gemm_fp8(const void* A, int lda, const float* a_scale, const void* W, int ldw, const float* w_scale, void* C, int ldc,
                 int out_f32, int M, int N, int K, const float* bias, int act, const float* resid, int ldr, int resid_rows,
                 int swiglu, int kernel, void* stream);

/* ---- training-loss forward values (no backward) ----
 * loss[i] = logsumexp(logits[i,:]) - logits[i, labels[i]], 0 where labels[i] == ignore_index: the per-token terms of the
 * causal-LM cross entropy inside HF Qwen2ForCausalLM.forward (used at videorefer_qwen2.py:198-215); labels already shifted. */
UFV_API int ufv_cross_entropy_rows(const float* logits, int64_t ld, const int64_t* labels, int M, int V, int64_t ignore_index, float* loss,
                           void* stream);
/* sums[n] = {sum bce_with_logits(pred, gt), sum sigmoid(pred)*gt, sum sigmoid(pred), sum gt} over the HW elements of mask n:
 * the reductions of sigmoid_ce_loss / dice_loss (videorefer_qwen2.py:34-77). */
UFV_API int ufv_mask_loss_sums(const float* pred, const float* gt, int n_masks, int64_t HW, float* sums, void* stream);

/* ---- SAM2 prompt/mask heads (sam2.py MaskDecoder.predict_masks :2094-2174, _forward_sam_heads :3276-3452) ---- */
/* out[m,:] = a[m,:] + b[m % b_rows,:] (b may be NULL = plain convert); a/out f32|bf16, b f32.  The `queries + query_pe`
 * / `keys + key_pe` adds of TwoWayAttentionBlock (:1384-1412) and the no_mem_embed / no_mask_embed broadcasts. */
UFV_API int ufv_add_bcast(const void* a, int a_dtype, int64_t lda, const float* b, int64_t ldb, int b_rows, void* out, int out_dtype,
                  int64_t ldo, int64_t M, int C, void* stream);
/* masks[b,i,Y,X] = sum_c hyper[b,i,c] * gelu(up2[b,Y/2,X/2,((Y&1)*2+(X&1))*C8+c] + s0[b,Y,X,c]): pixel shuffle of the second
 * ConvTranspose2d (computed as a GEMM, 4 taps side by side), + feat_s0, GELU, `hyper_in @ upscaled` (:2150-2162).
 * up2 bf16 [B*h*w, ld_up], s0 bf16 [B*2h*2w, ld_s0], hyper f32 [B,nm,C8], out f32 [B,nm,2h,2w]. */
UFV_API int ufv_sam_mask_head(const void* up2, int64_t ld_up, const void* s0, int64_t ld_s0, const float* hyper, float* out, int B, int h,
                      int w, int C8, int nm, void* stream);
/* F.interpolate(mode="bilinear", align_corners=False) on f32 planes [Hs,Ws] -> [Hd,Wd]; with sel != NULL image n reads
 * plane n*planes_per + sel_off + sel[n] (best-IoU mask pick of _forward_sam_heads :3409-3421 fused into the resize). */
UFV_API int ufv_resize_bilinear(const float* src, const int32_t* sel, int planes_per, int sel_off, float* dst, int N, int Hs, int Ws, int Hd,
                        int Wd, void* stream);
/* out[m] = argmax_j x[m, j], j < N (torch.argmax tie-breaking) */
UFV_API int ufv_argmax_rows(const float* x, int64_t ld, int M, int N, int32_t* out, void* stream);

/* ---- projector backward (ufvideo/model/projector.py:133-238 under torch autograd; timm RegStage bottlenecks) ---- */
/* out = act(pre) and dpre = dout * act'(pre) on flat bf16 arrays (n % 8 == 0): the un-fused activations of the training forward */
UFV_API int ufv_act(const void* pre, void* out, int64_t n, int act, void* stream);
UFV_API int ufv_act_bwd(const void* pre, const void* dout, void* dpre, int64_t n, int act, void* stream);
UFV_API int ufv_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);
/* row LayerNorm (+ optional activation) backward: x, dout, dx bf16 [M, C]; dw, db fp32 [C] are ADDED to; ws = ufv_layernorm_bwd_ws_bytes(C) */
UFV_API int64_t ufv_layernorm_bwd_ws_bytes(int C);
UFV_API int ufv_layernorm_bwd(const void* x, int64_t ldx, const float* w, const float* b, const void* dout, int64_t ldd, void* dx, int64_t lddx,
                      float* dw, float* db, int M, int C, float eps, int act, void* ws, void* stream);
/* g = dout * silu'(LN_a(z) + (LN_b(s) or s when wb == NULL)): the gradient entering both branches of a bottleneck's output */
UFV_API int ufv_ln_add_silu_g(const void* z, const float* wa, const float* ba, const void* s, const float* wb, const float* bb, const void* dout,
                      void* g, int M, int C, float eps, void* stream);
/* depthwise 3x3 conv, padding 1, token-major NHWC bf16, w9 fp32 [9, C]; flip = 1: taps mirrored = gradient with respect to the input */
UFV_API int ufv_dwconv3x3(const void* x, void* y, const float* w9, int F, int H, int W, int C, int flip, void* stream);
/* dw9[tap][c] += sum over pixels of dy * shifted x; ws = ufv_dwconv3x3_dw_ws_bytes(C) */
UFV_API int64_t ufv_dwconv3x3_dw_ws_bytes(int C);
UFV_API int ufv_dwconv3x3_dw(const void* x, const void* dy, float* dw9, int F, int H, int W, int C, void* ws, void* stream);
/* out[f][c] = sum_p a[f,p,c] * b[f,p,c] (SE gate gradient); out[f,p,c] = a[f,p,c] * g[f,c] + s[f,c] * k (s may be NULL) */
UFV_API int ufv_prod_colsum(const void* a, const void* b, int F, int P, int C, float* out, void* stream);
UFV_API int ufv_scale_add_bcast(const void* a, const void* g, const float* s, float k, void* out, int F, int P, int C, void* stream);
/* inverse of ufv_conv3d_gather (stride = kernel, zero padding `pad`): dx [T*H*W, C] from dA [To*Ho*Wo, kt*kh*kw*C]; pixels past the last window get 0 */
UFV_API int ufv_conv3d_scatter(const void* dA, void* dx, int T, int H, int W, int C, int kt, int kh, int kw, int pad, void* stream);
/* AvgPool3d sampler in training: the pooled value without the activation (ufv_avgpool3d_silu fuses SiLU), and its backward
 * dx [T*H*W, C] = dy [To*Ho*Wo, C] / (kt*kh*kw) on the pixels inside a window */
UFV_API int ufv_avgpool3d(const void* x, void* out, int T, int H, int W, int C, int kt, int kh, int kw, void* stream);
UFV_API int ufv_avgpool3d_bwd(const void* dy, void* dx, int T, int H, int W, int C, int kt, int kh, int kw, void* stream);

/* generate(do_sample=True) (ufvideo/__init__.py:113-127 -> HF TemperatureLogitsWarper / TopKLogitsWarper / TopPLogitsWarper +
 * multinomial): for each of the M rows of logits f32 [M, ld], out[m] = a token drawn from softmax(logits / temperature)
 * restricted to the top_k (0 = off) most likely tokens and then to the smallest set reaching mass top_p; u[m] in [0,1) is the
 * caller's uniform variate (inverse CDF in vocabulary order).  kept_out (optional, f32 [M,2]) = {kept mass / top-k mass, cut-off logit}. */
UFV_API int ufv_sample_top_p(const float* logits, int64_t ld, int M, int V, float temperature, int top_k, float top_p, const float* u, int64_t* out,
                     float* kept_out, void* stream);

/* elementwise convert between bf16 / f32 / f16 (n elements) */
UFV_API int ufv_convert(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);

/* ---- training step of the decoder: backward + optimizer ----
 * The reference gets these from torch autograd over HF Qwen2 (modeling_qwen2.py) and from DeepSpeed ZeRO-2 + AdamW
 * (train.py:749, scripts/zero2.json).  The contractions (dX = dY W, dW = dY^T X, attention's five products) run on ufv_gemm;
 * the entry points below are what surrounds them. */
/* out[c][r] = in[r][c], r < R, c < C; out columns R..Rpad-1 are zero-filled (K padding of the consuming NT GEMM) */
UFV_API int ufv_transpose_bf16(const void* in, int64_t ldi, void* out, int64_t ldo, int R, int C, int Rpad, void* stream);
/* Qwen2RMSNorm backward (modeling_qwen2.py:238-254): dx (+)= d/dx, dw (+)= d/dw; x, dy, dx fp32 [M, D]; ws = ufv_rmsnorm_bwd_ws_bytes(D) */
UFV_API int64_t ufv_rmsnorm_bwd_ws_bytes(int D);
UFV_API int ufv_rmsnorm_bwd(const float* x, int ldx, const float* w, const float* dy, int lddy, float* dx, int lddx, int accumulate, float* dw,
                    int dw_accumulate, int M, int D, float eps, void* ws, void* stream);
/* out[j] (+)= sum_r x[r][j], x bf16 [R, C] (bias gradients); ws = 32*C floats */
UFV_API int ufv_colsum_bf16(const void* x, int64_t ld, int R, int C, float* out, int accumulate, void* ws, void* stream);
/* Qwen2MLP gate (modeling_qwen2.py:47) on the packed gate/up layout of ufv_gemm's swiglu mode, unfused (training keeps the
 * pre-activations): act[M, I] = silu(g) * u, and its backward dgu[M, 2I] from dact[M, I] */
UFV_API int ufv_swiglu(const void* gu, int64_t ldgu, void* act, int64_t lda, int M, int I, void* stream);
UFV_API int ufv_swiglu_bwd(const void* gu, int64_t ldgu, const void* dact, int64_t ldd, void* dgu, int64_t ldo, int M, int I, void* stream);
/* rotate-half RoPE (modeling_qwen2.py:113-135) in place on `nheads` heads at column col0 of buf bf16 [S, ld]; backward != 0
 * applies the transposed rotation (the gradient of the forward one) */
UFV_API int ufv_rope_rows(void* buf, int64_t ld, int S, int col0, int nheads, int hd, const float* inv_freq, int pos0, int backward, void* stream);
/* causal-LM cross entropy, forward + backward in one pass: loss[i] as ufv_cross_entropy_rows; dlogits bf16 [M, ldd] =
 * gscale * (softmax - onehot), zero for ignored rows and for columns V..Vpad-1 */
UFV_API int ufv_cross_entropy_bwd(const float* logits, int64_t ld, const int64_t* labels, int M, int V, int Vpad, int64_t ignore_index,
                          float gscale, float* loss, void* dlogits, int64_t ldd, void* stream);
/* dst[idx[r], :] += src[r, :] (fp32; idx < 0 skipped): gradient of the embed_tokens gather of the splice (videorefer_arch.py:239-370) */
UFV_API int ufv_scatter_add_rows(const float* src, int64_t lds, const int64_t* idx, float* dst, int64_t ldd, int R, int D, void* stream);
/* partial[b] = sum of squares of block b's slice of x[n] (torch.nn.utils.clip_grad_norm_) */
UFV_API int ufv_sumsq(const float* x, int64_t n, float* partial, int n_partial, void* stream);
/* torch.optim.AdamW step `step` (1-based) on fp32 p/m/v with gradient g * (*gscale if given); p_bf16 (may be NULL) receives the
 * rounded working copy */
UFV_API int ufv_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1, float beta2, float eps,
              float weight_decay, int step, const float* gscale, void* stream);
/* causal GQA self-attention backward (eager Qwen2 attention, modeling_qwen2.py:150-172, differentiated): q bf16 [S, ldq]
 * (head h at column h*hd, RoPE applied), k / v bf16 [>= round_up(S,128) rows, ldkv] (kv-head g at column g*hd), dO bf16 [S, lddo]
 * -> dq [S, lddq], dk / dv [S, lddkv] bf16.  ws = ufv_attention_bwd_ws_bytes(S, Hq, Hkv, hd). */
UFV_API int64_t ufv_attention_bwd_ws_bytes(int S, int Hq, int Hkv, int hd);
/* Measurement aid (bench.py's roofline entry): enable = n > 0: every n-th ufv_gemm / ufv_gemm_fp8 launch with the SwiGLU epilogue and M > 64 -- the
 * decoder's gate/up projection, the dominant kernel -- is bracketed by a HIP event pair on its own stream, whether it is issued op by op or
 * inside a stage call (a bracket idles the stream for ~11 us: bench.py samples every 7th launch, 4 of a step's 28); 0 = off.  ufv_gemm_timing_read waits for the recorded launches, returns their count, copies up to `cap` durations (ms) and
 * shapes (M, N, K per launch) and forgets them.  Not thread-safe; off by default. */
UFV_API int ufv_gemm_timing(int enable);
/* which kernel UFV_GEMM_AUTO takes for a bf16 GEMM of this shape (host arithmetic, no launch): 0 = the 128-wide / small-shape kernels, else the
 * UFV_GEMM_PP shape code (+ 10000 * parts for the split-K form; act_none = no activation in the epilogue) */
UFV_API int ufv_gemm_choice(int M, int N, int K, int out_f32, int swiglu, int act_none);
UFV_API int ufv_gemm_timing_read(float* ms, int32_t* mnk, int cap);
/* Split-K housekeeping (see Conventions).  ufv_gemm_prepare: create the current device's flag ring now (allocates; returns 0 / UFV_EHIP).
 * ufv_gemm_set_splitk(0 | 1): whether UFV_GEMM_AUTO may take the split-K form; returns the previous setting (initial value 1, or 0 when the
 * environment variable UFV_GEMM_NO_SPLITK is set at load time).  The split and unsplit forms sum K in different orders: fp32 outputs differ in
 * the last bits (<= 4e-6 of the largest element).  ufv_gemm_error_state: 0, or the code a timed-out turn wait left on the current device;
 * ufv_gemm_clear_error resets it. */
UFV_API int ufv_gemm_prepare(void);
UFV_API int ufv_gemm_set_splitk(int enable);
UFV_API int ufv_gemm_error_state(void);
UFV_API int ufv_gemm_clear_error(void);

/* C[M,N] (+)= A[M,K] * W[N,K]^T with K split over up to nsplit blocks per output tile (thin outputs over a long K: dV = P^T dO,
 * dK = dS^T Q); partial tiles go to ws (fp32 [nsplit][M][N]) and are summed in order; C fp32 (optionally accumulated) or bf16 */
UFV_API int ufv_gemm_splitk(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int out_f32, int accumulate, int M, int N, int K,
                    int nsplit, void* ws, void* stream);
UFV_API int ufv_attention_bwd(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* dO, int64_t lddo, void* dq,
                      int64_t lddq, void* dk, void* dv, int64_t lddkv, int S, int Hq, int Hkv, int hd, float scale, void* ws, void* stream);
/* Fused (flash-style) form of the same backward for hd == 128: nothing of size S x S is written.  ufv_attention_causal_lse is the
 * training forward (same bits as ufv_attention with causal = 1, batch 1, q_pos0 = 0; *_ss = token strides in elements) that also
 * stores lse fp32 [Hq, S] = log2-domain log-sum-exp of the scaled scores; ufv_attention_bwd_fused takes it with the forward output
 * o bf16 [S, ldo].  k / v need only S rows here.  ws = ufv_attention_bwd_fused_ws_bytes(S, Hq). */
UFV_API int ufv_attention_causal_lse(const void* q, int64_t q_ss, const void* k, int64_t k_ss, const void* v, int64_t v_ss, void* o, int64_t o_ss,
                             int Hq, int Hkv, int S, int hd, float scale, float* lse, void* stream);
UFV_API int64_t ufv_attention_bwd_fused_ws_bytes(int S, int Hq);
UFV_API int ufv_attention_bwd_fused(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* o, int64_t ldo,
                            const void* dO, int64_t lddo, const float* lse, void* dq, int64_t lddq, void* dk, void* dv, int64_t lddkv,
                            int S, int Hq, int Hkv, int hd, float scale, void* ws, void* stream);

/* ---- [SEG] mask-loss backward (videorefer_qwen2.py:34-77,279-338; sam2.py _forward_sam_heads :3276-3452 under
 * autograd).  The mask decoder's attention is between a handful of prompt tokens and h*w image tokens: q [B, Nq, H*hd], k / v
 * [B, Nk, H*hd] bf16 contiguous, head_dim 16 or 32; lse fp32 [B, H, Nq] = log sum exp of the scaled scores. */
UFV_API int ufv_small_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq, int Nk, int hd, float scale,
                       void* stream);
/* dq / dk / dv bf16 (overwritten); delta fp32 [B, H, Nq] is scratch (dO . O) */
UFV_API int ufv_small_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse, float* delta, void* dq,
                       void* dk, void* dv, int B, int H, int Nq, int Nk, int hd, float scale, void* stream);
/* the selected mask: out[b, p] = sum_c up[b * P + p, c] * h[b, c]  (up bf16 [B*P, C], C <= 32; h, out fp32) and its gradients */
UFV_API int ufv_mask_dot_fwd(const void* up, const float* h, float* out, int B, int P, int C, void* stream);
UFV_API int64_t ufv_mask_dot_bwd_ws_bytes(int B, int C);
UFV_API int ufv_mask_dot_bwd(const void* up, const float* h, const float* dm, void* dup, float* dh, void* ws, int B, int P, int C, void* stream);
/* backward of ufv_resize_bilinear (no plane selection): din [N, Hs, Ws] = gather of dout [N, Hd, Wd] through the same 2x2 stencils */
UFV_API int ufv_resize_bilinear_bwd(const float* dout, float* din, int N, int Hs, int Ws, int Hd, int Wd, void* stream);
/* dx = cb * (sigmoid(x) - t) + sigmoid(x) (1 - sigmoid(x)) (coef[2n] * t + coef[2n + 1]): BCE-with-logits (mean) + DICE of mask n */
UFV_API int ufv_mask_loss_bwd(const float* x, const float* t, const float* coef, float cb, float* dx, int N, int64_t HW, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UFV_H_ */
