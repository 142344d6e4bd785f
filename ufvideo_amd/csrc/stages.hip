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
    for (int l = 0; l < m->n_layers; ++l) {
        const ufv_qwen2_layer& L = m->layers[l];
        char* kv = reinterpret_cast<char*>(L.kv_cache);
        UFV_TRY(ufv_rmsnorm(x, D, h, 0, D, L.ln1, S, D, m->eps, stream));
        UFV_TRY(ufv_gemm(h, D, L.wqkv, D, qkv, qkv_n, 0, S, qkv_n, D, L.bqkv, UFV_ACT_NONE, nullptr, 0, 0, 0, UFV_GEMM_AUTO, stream));
        UFV_TRY(ufv_rope_kv_table(qkv, qkv_n, S, H, KV, hd, table, pos0, kv, m->ldkv, stream));
        UFV_TRY(ufv_attention(qkv, 0, qkv_n, kv, 0, m->ldkv, kv + 2 * (size_t)KV * hd, 0, m->ldkv, o, 0, (int64_t)H * hd, 1, H, KV, S, pos0 + S, hd,
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
    UFV_REQUIRE(H % m->patch == 0 && W % m->patch == 0 && (H / m->patch) * (W / m->patch) == m->n_patches,
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
