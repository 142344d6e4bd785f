// Whole-stage entry points (ufv_vit_forward, ufv_llm_prefill): the layer loops of the SigLIP tower and of the Qwen2 prefill
// as single C calls, so that a reference-side binder needs ONE call per stage instead of re-writing the loops (~330 / ~230 op-level
// calls per clip).  Pure composition of the op-level entry points of include/ufv.h on one stream: no allocation, no synchronisation,
// results bit-identical to the same sequence issued from the host (tests/test_stages_gpu.py).
#include "common.h"
#include "../../include/ufv.h"

#define UFV_TRY(expr)            \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != UFV_OK) return rc_; \
    } while (0)

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- Qwen2 prefill (modeling_qwen2.py Qwen2Model.forward over S > 1 positions of one sequence; videorefer_qwen2.py:154-196) ----------
extern "C" int64_t ufv_qwen2_prefill_ws_bytes(const ufv_qwen2_model* m, int S) {
    if (!m || S <= 0) return -1;
    const size_t qkv = (size_t)(m->n_q + 2 * m->n_kv) * m->hd;
    size_t b = 0;
    b += align256(2 * (size_t)S * m->d);                 // h
    b += align256(2 * (size_t)S * qkv);                  // qkv
    b += align256(2 * (size_t)S * m->n_q * m->hd);       // o
    b += align256(2 * (size_t)S * m->d_ff);              // act
    b += align256(sizeof(float) * (size_t)S * m->hd);    // RoPE cos | sin table
    b += align256(sizeof(float) * m->d);                 // final-norm row
    b += align256(2 * (size_t)m->d);                     // its bf16 copy
    return (int64_t)b;
}

extern "C" int ufv_qwen2_prefill(const ufv_qwen2_model* m, float* x, int S, int pos0, void* ws, int64_t ws_bytes, float* hidden_layers,
                                 float* normed, float* logits_last, void* stream) {
    UFV_REQUIRE(m && x && ws && S > 0 && pos0 >= 0, "ufv_qwen2_prefill: bad arguments");
    UFV_REQUIRE(ws_bytes >= ufv_qwen2_prefill_ws_bytes(m, S), "ufv_qwen2_prefill: workspace too small");
    UFV_REQUIRE(pos0 + S <= m->max_len, "ufv_qwen2_prefill: positions %d..%d outside the KV cache (max_len %d)", pos0, pos0 + S - 1, m->max_len);
    const int D = m->d, H = m->n_q, KV = m->n_kv, hd = m->hd, I = m->d_ff;
    const int qkv_n = (H + 2 * KV) * hd;
    UFV_REQUIRE(hd % 16 == 0, "ufv_qwen2_prefill: head_dim %d (needs a multiple of 16)", hd);
    char* p = reinterpret_cast<char*>(ws);
    void* h = p; p += align256(2 * (size_t)S * D);
    char* qkv = p; p += align256(2 * (size_t)S * qkv_n);
    void* o = p; p += align256(2 * (size_t)S * H * hd);
    void* act = p; p += align256(2 * (size_t)S * I);
    float* table = reinterpret_cast<float*>(p); p += align256(sizeof(float) * (size_t)S * hd);
    float* nrow = reinterpret_cast<float*>(p); p += align256(sizeof(float) * D);
    void* nrow_bf = p;
    const float scale = 1.0f / sqrtf((float)hd);
    UFV_TRY(ufv_rope_table(m->inv_freq, pos0, S, hd, table, stream));
    // round 5: q / k / v projection, RoPE and the KV append as ONE launch where the fused kernel is built (head_dim 128, S >= 256; bit-identical to the three
    // calls of the else branch: tests/test_stages_gpu.py); q then lives compactly in the front of the qkv buffer, [S, H * hd]
    const int fused = ufv_gemm_qkv_rope_shape(S, H, KV, hd, D);
    for (int l = 0; l < m->n_layers; ++l) {
        const ufv_qwen2_layer& L = m->layers[l];
        char* kv = reinterpret_cast<char*>(L.kv_cache);
        UFV_TRY(ufv_rmsnorm(x, D, h, 0, D, L.ln1, S, D, m->eps, stream));
        int q_ss = qkv_n;
        if (fused) {
            q_ss = H * hd;
            UFV_TRY(ufv_gemm_qkv_rope(h, D, L.wqkv, D, L.bqkv, qkv, q_ss, kv + 2 * (size_t)pos0 * m->ldkv, m->ldkv, S, H, KV, hd, D, table, fused, stream));
        } else {
            UFV_TRY(ufv_gemm(h, D, L.wqkv, D, qkv, qkv_n, 0, S, qkv_n, D, L.bqkv, UFV_ACT_NONE, nullptr, 0, 0, 0, UFV_GEMM_AUTO, stream));
            UFV_TRY(ufv_rope_kv_table(qkv, qkv_n, S, H, KV, hd, table, pos0, kv, m->ldkv, stream));
        }
        UFV_TRY(ufv_attention(qkv, 0, q_ss, kv, 0, m->ldkv, kv + 2 * (size_t)KV * hd, 0, m->ldkv, o, 0, (int64_t)H * hd, 1, H, KV, S, pos0 + S, hd,
                              scale, 1, pos0, 0, stream));
        UFV_TRY(ufv_gemm(o, H * hd, L.wo, H * hd, x, D, 1, S, D, H * hd, nullptr, UFV_ACT_NONE, x, D, 0, 0, UFV_GEMM_AUTO, stream));
        UFV_TRY(ufv_rmsnorm(x, D, h, 0, D, L.ln2, S, D, m->eps, stream));
        UFV_TRY(ufv_gemm(h, D, L.wgu, D, act, I, 0, S, 2 * I, D, nullptr, UFV_ACT_NONE, nullptr, 0, 0, 1, UFV_GEMM_AUTO, stream));
        UFV_TRY(ufv_gemm(act, I, L.wd, I, x, D, 1, S, D, I, nullptr, UFV_ACT_NONE, x, D, 0, 0, UFV_GEMM_AUTO, stream));
        if (hidden_layers && l < m->n_layers - 1)      // HF output_hidden_states: the stream after every layer but the last (which is normed)
            UFV_TRY(ufv_convert(x, UFV_DT_F32, hidden_layers + (size_t)l * S * D, UFV_DT_F32, (int64_t)S * D, stream));
    }
    if (normed) UFV_TRY(ufv_rmsnorm(x, D, normed, 1, D, m->norm, S, D, m->eps, stream));       // HF: hidden_states[-1] = norm(last layer), all rows
    if (logits_last) {
        const float* last = normed ? normed + (size_t)(S - 1) * D : nrow;
        if (!normed) UFV_TRY(ufv_rmsnorm(x + (size_t)(S - 1) * D, D, nrow, 1, D, m->norm, 1, D, m->eps, stream));
        UFV_TRY(ufv_convert(last, UFV_DT_F32, nrow_bf, UFV_DT_BF16, D, stream));
        UFV_TRY(ufv_gemm(nrow_bf, D, m->lm_head, D, logits_last, m->vocab, 1, 1, m->vocab, D, nullptr, UFV_ACT_NONE, nullptr, 0, 0, 0, UFV_GEMM_AUTO, stream));
    }
    return UFV_OK;
}

// ---- SigLIP tower (modeling_siglip.py SiglipVisionTransformer without the head; encoder.py:96-146 of the reference) -----------------
extern "C" int64_t ufv_vit_forward_ws_bytes(const ufv_vit_model* m, int T) {
    if (!m || T <= 0) return -1;
    const size_t M = (size_t)T * m->n_patches;
    size_t b = 0;
    b += align256(2 * M * m->kpad);          // im2col rows
    b += align256(2 * M * m->d);             // h
    b += align256(2 * M * 3 * m->d);         // qkv
    b += align256(2 * M * m->d);             // o
    b += align256(2 * M * m->d_ff_pad);      // ff
    return (int64_t)b;
}

extern "C" int ufv_vit_forward(const ufv_vit_model* m, const void* pixels, int dtype, int T, int H, int W, int n_layers, float* x,
                               void* ws, int64_t ws_bytes, void* stream) {
    UFV_REQUIRE(m && pixels && x && ws && T > 0, "ufv_vit_forward: bad arguments");
    UFV_REQUIRE(n_layers >= 0 && n_layers <= m->n_layers, "ufv_vit_forward: %d layers asked of a %d-layer tower", n_layers, m->n_layers);
    UFV_REQUIRE(H >= m->patch && W >= m->patch && (H / m->patch) * (W / m->patch) == m->n_patches,      // floor: a stride-P convolution without padding (384 px -> 27 x 27)
                "ufv_vit_forward: tower built for %d patches, image %dx%d gives %d", m->n_patches, H, W, (H / m->patch) * (W / m->patch));
    UFV_REQUIRE(ws_bytes >= ufv_vit_forward_ws_bytes(m, T), "ufv_vit_forward: workspace too small");
    const int D = m->d, Hh = m->n_heads, hd = D / Hh, S = m->n_patches, Ip = m->d_ff_pad;
    const int M = T * S;
    char* p = reinterpret_cast<char*>(ws);
    void* cols = p; p += align256(2 * (size_t)M * m->kpad);
    void* h = p; p += align256(2 * (size_t)M * D);
    char* qkv = p; p += align256(2 * (size_t)M * 3 * D);
    void* o = p; p += align256(2 * (size_t)M * D);
    void* ff = p;
    const float scale = 1.0f / sqrtf((float)hd);
    UFV_TRY(ufv_patchify(pixels, dtype, cols, T, m->channels, H, W, m->patch, m->kpad, stream));
    // patch embedding + bias + position embedding (a broadcast residual table of n_patches rows)
    UFV_TRY(ufv_gemm(cols, m->kpad, m->patch_w, m->kpad, x, D, 1, M, D, m->kpad, m->patch_b, UFV_ACT_NONE, m->pos, D, S, 0, UFV_GEMM_AUTO, stream));
    const int64_t bs = (int64_t)S * 3 * D, ss = 3 * D;
    for (int l = 0; l < n_layers; ++l) {
        const ufv_vit_layer& L = m->layers[l];
        UFV_TRY(ufv_layernorm(x, UFV_DT_F32, D, h, 0, D, L.ln1_w, L.ln1_b, M, D, m->eps, UFV_ACT_NONE, stream));
        UFV_TRY(ufv_gemm(h, D, L.wqkv, D, qkv, 3 * D, 0, M, 3 * D, D, L.bqkv, UFV_ACT_NONE, nullptr, 0, 0, 0, UFV_GEMM_AUTO, stream));
        UFV_TRY(ufv_attention(qkv, bs, ss, qkv + 2 * (size_t)D, bs, ss, qkv + 4 * (size_t)D, bs, ss, o, (int64_t)S * D, D, T, Hh, Hh, S, S, hd, scale, 0, 0,
                              0, stream));
        UFV_TRY(ufv_gemm(o, D, L.wo, D, x, D, 1, M, D, D, L.bo, UFV_ACT_NONE, x, D, 0, 0, UFV_GEMM_AUTO, stream));
        UFV_TRY(ufv_layernorm(x, UFV_DT_F32, D, h, 0, D, L.ln2_w, L.ln2_b, M, D, m->eps, UFV_ACT_NONE, stream));
        UFV_TRY(ufv_gemm(h, D, L.w1, D, ff, Ip, 0, M, Ip, D, L.b1, m->act, nullptr, 0, 0, 0, UFV_GEMM_AUTO, stream));
        UFV_TRY(ufv_gemm(ff, Ip, L.w2, Ip, x, D, 1, M, D, Ip, L.b2, UFV_ACT_NONE, x, D, 0, 0, UFV_GEMM_AUTO, stream));
    }
    return UFV_OK;
}

// ---- STC connector (projector.py:133-250: RegStage x depth -> Conv3d | AvgPool3d sampler -> RegStage x depth -> readout MLP) -------------
// token-major bf16 [frames * H * W, C] throughout; one video per call
static size_t stc_rows_out(const ufv_stc_model* m, int T, int HW, int* To, int* Ho, int* Wo) {
    if (m->avgpool) { *To = T / m->kt; *Ho = HW / m->kh; *Wo = HW / m->kw; }
    else { *To = (T + 2 * m->pad - m->kt) / m->kt + 1; *Ho = (HW + 2 * m->pad - m->kh) / m->kh + 1; *Wo = (HW + 2 * m->pad - m->kw) / m->kw + 1; }
    return (size_t)*To * *Ho * *Wo;
}

static size_t stc_buf_elems(const ufv_stc_model* m, int T, int HW) {
    int To, Ho, Wo;
    const size_t M1 = (size_t)T * HW * HW, M2 = stc_rows_out(m, T, HW, &To, &Ho, &Wo);
    const size_t cmax = m->c_in > m->c_hid ? m->c_in : m->c_hid;
    size_t e = M1 * cmax;
    const size_t g = M2 * (size_t)m->kt * m->kh * m->kw * m->c_hid;        // the Conv3d patch matrix
    if (!m->avgpool && g > e) e = g;
    if (M2 * cmax > e) e = M2 * cmax;
    return e;
}

extern "C" int64_t ufv_stc_forward_ws_bytes(const ufv_stc_model* m, int T, int HW) {
    if (!m || T <= 0 || HW <= 0) return -1;
    const size_t fmax = (size_t)T;                                          // frames of the larger stage
    return (int64_t)(4 * align256(2 * stc_buf_elems(m, T, HW)) + 3 * align256(2 * fmax * m->c_hid));
}

static int stc_block(const ufv_stc_model* m, const ufv_stc_block& b, char*& X, char*& A, char*& B, char*& C, char* s0, char* s1, char* s2, int F, int H, int W,
                     void* stream) {
    const int P = H * W, M = F * P, Ci = b.c_in, Co = b.c_out;
    UFV_TRY(ufv_gemm(X, Ci, b.w1, Ci, A, Co, 0, M, Co, Ci, nullptr, UFV_ACT_NONE, nullptr, 0, 0, 0, UFV_GEMM_AUTO, stream));
    UFV_TRY(ufv_layernorm(A, UFV_DT_BF16, Co, B, 0, Co, b.n1_w, b.n1_b, M, Co, m->eps, UFV_ACT_SILU, stream));
    UFV_TRY(ufv_dwconv3x3_ln_silu(B, A, b.w9, b.n2_w, b.n2_b, F, H, W, Co, m->eps, stream));
    UFV_TRY(ufv_colmean(A, s0, F, P, Co, stream));
    UFV_TRY(ufv_gemm(s0, Co, b.se1_w, Co, s1, b.se_rd, 0, F, b.se_rd, Co, b.se1_b, UFV_ACT_SILU, nullptr, 0, 0, 0, UFV_GEMM_AUTO, stream));
    UFV_TRY(ufv_gemm(s1, b.se_rd, b.se2_w, b.se_rd, s2, Co, 0, F, Co, b.se_rd, b.se2_b, UFV_ACT_SIGMOID, nullptr, 0, 0, 0, UFV_GEMM_AUTO, stream));
    UFV_TRY(ufv_scale_channels(A, s2, F, P, Co, stream));
    UFV_TRY(ufv_gemm(A, Co, b.w3, Co, B, Co, 0, M, Co, Co, nullptr, UFV_ACT_NONE, nullptr, 0, 0, 0, UFV_GEMM_AUTO, stream));
    if (b.ds_w) {
        UFV_TRY(ufv_gemm(X, Ci, b.ds_w, Ci, C, Co, 0, M, Co, Ci, nullptr, UFV_ACT_NONE, nullptr, 0, 0, 0, UFV_GEMM_AUTO, stream));
        UFV_TRY(ufv_ln_add_silu(B, b.n3_w, b.n3_b, C, b.ds_nw, b.ds_nb, A, M, Co, m->eps, stream));
    } else {
        UFV_TRY(ufv_ln_add_silu(B, b.n3_w, b.n3_b, X, nullptr, nullptr, A, M, Co, m->eps, stream));
    }
    char* t = X; X = A; A = t;               // the block's output becomes the next input
    return UFV_OK;
}

extern "C" int ufv_stc_forward(const ufv_stc_model* m, const void* x, int x_dtype, int T, int HW, float* out, void* ws, int64_t ws_bytes, void* stream) {
    UFV_REQUIRE(m && x && out && ws && T > 0 && HW > 0, "ufv_stc_forward: bad arguments");
    UFV_REQUIRE(ws_bytes >= ufv_stc_forward_ws_bytes(m, T, HW), "ufv_stc_forward: workspace too small");
    UFV_REQUIRE(m->mlp_depth >= 1 && (m->depth == 0 || (m->s1 && m->s2)), "ufv_stc_forward: incomplete model description");
    const size_t be = align256(2 * stc_buf_elems(m, T, HW));
    char* p = reinterpret_cast<char*>(ws);
    char *X = p, *A = p + be, *B = p + 2 * be, *C = p + 3 * be;
    char* s0 = p + 4 * be; char* s1 = s0 + align256(2 * (size_t)T * m->c_hid); char* s2 = s1 + align256(2 * (size_t)T * m->c_hid);
    const size_t M1 = (size_t)T * HW * HW;
    UFV_TRY(ufv_convert(x, x_dtype, X, UFV_DT_BF16, (int64_t)(M1 * m->c_in), stream));
    for (int i = 0; i < m->depth; ++i) UFV_TRY(stc_block(m, m->s1[i], X, A, B, C, s0, s1, s2, T, HW, HW, stream));
    const int Cm = m->depth ? m->c_hid : m->c_in;
    int To, Ho, Wo;
    const size_t M2 = stc_rows_out(m, T, HW, &To, &Ho, &Wo);
    UFV_REQUIRE(M2 > 0, "ufv_stc_forward: %d frames of %dx%d give no output token", T, HW, HW);
    if (m->avgpool) {
        UFV_TRY(ufv_avgpool3d_silu(X, A, T, HW, HW, Cm, m->kt, m->kh, m->kw, stream));
        char* t = X; X = A; A = t;
    } else {
        const int Kc = m->kt * m->kh * m->kw * Cm;
        UFV_TRY(ufv_conv3d_gather(X, A, T, HW, HW, Cm, m->kt, m->kh, m->kw, m->pad, stream));
        UFV_TRY(ufv_gemm(A, Kc, m->samp_w, Kc, X, m->c_hid, 0, (int)M2, m->c_hid, Kc, m->samp_b, UFV_ACT_SILU, nullptr, 0, 0, 0, UFV_GEMM_AUTO, stream));
    }
    for (int i = 0; i < m->depth; ++i) UFV_TRY(stc_block(m, m->s2[i], X, A, B, C, s0, s1, s2, To, Ho, Wo, stream));
    for (int i = 0; i < m->mlp_depth; ++i) {
        const bool last = i == m->mlp_depth - 1;
        UFV_TRY(ufv_gemm(X, m->c_hid, m->readout_w[i], m->c_hid, last ? (void*)out : (void*)A, m->c_hid, last ? 1 : 0, (int)M2, m->c_hid, m->c_hid, m->readout_b[i],
                         last ? UFV_ACT_NONE : UFV_ACT_GELU_ERF, nullptr, 0, 0, 0, UFV_GEMM_AUTO, stream));
        char* t = X; X = A; A = t;
    }
    return UFV_OK;
}
