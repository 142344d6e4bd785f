// One greedy decode step of the Qwen2-style decoder as a single C call: embedding lookup of the previous token ->
// N x {[RMSNorm+]fused QKV GEMV(+bias), RoPE + KV-cache append, split-key decode attention, o_proj(+residual),
// [RMSNorm+]gate/up GEMV with SwiGLU, down(+residual)} -> final RMSNorm -> lm_head GEMV -> argmax.  Pure composition of the
// entry points in include/ufv.h on one stream (no allocation, no synchronisation): it exists so that the host pays
// one FFI call per token instead of ~340.
#include "common.h"
#include "../../include/ufv.h"

#define UFV_TRY(expr)            \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != UFV_OK) return rc_; \
    } while (0)

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" int64_t ufv_qwen2_decode_ws_bytes(const ufv_qwen2_model* m) {
    if (!m) return -1;
    const size_t qkv = (size_t)(m->n_q + 2 * m->n_kv) * m->hd;
    size_t b = 0;
    b += align256(sizeof(float) * m->d);            // x
    b += align256(sizeof(float) * m->d);            // normed (fp32, last layer hidden state)
    b += align256(2 * (size_t)m->d);                // h
    b += align256(2 * qkv);                         // qkv
    b += align256(2 * (size_t)m->n_q * m->hd);      // o
    b += align256(2 * (size_t)m->d_ff);             // act
    b += align256(ufv_attention_decode_ws_bytes(1, m->n_q, m->hd, m->attn_splits));
    return (int64_t)b;
}

extern "C" int ufv_qwen2_decode_step(const ufv_qwen2_model* m, const int64_t* token_dev, int pos, void* ws, int64_t ws_bytes,
                                     float* logits, float* hidden_out, int64_t* next_token_dev, void* stream) {
    UFV_REQUIRE(m && token_dev && ws && logits && next_token_dev, "ufv_qwen2_decode_step: null argument");
    UFV_REQUIRE(ws_bytes >= ufv_qwen2_decode_ws_bytes(m), "ufv_qwen2_decode_step: workspace too small");
    UFV_REQUIRE(pos >= 0 && pos < m->max_len, "ufv_qwen2_decode_step: position %d outside the KV cache (max_len %d)", pos, m->max_len);
    const int D = m->d, H = m->n_q, KV = m->n_kv, hd = m->hd, I = m->d_ff;
    const int qkv_n = (H + 2 * KV) * hd;
    char* p = reinterpret_cast<char*>(ws);
    float* x = reinterpret_cast<float*>(p); p += align256(sizeof(float) * D);
    float* normed = reinterpret_cast<float*>(p); p += align256(sizeof(float) * D);
    void* h = p; p += align256(2 * (size_t)D);
    void* qkv = p; p += align256(2 * (size_t)qkv_n);
    void* o = p; p += align256(2 * (size_t)H * hd);
    void* act = p; p += align256(2 * (size_t)I);
    void* aws = p;
    const float scale = 1.0f / sqrtf((float)hd);

    UFV_TRY(ufv_gather_rows(m->embed, UFV_DT_BF16, D, token_dev, x, UFV_DT_F32, D, nullptr, 1, D, stream));
    for (int l = 0; l < m->n_layers; ++l) {
        const ufv_qwen2_layer& L = m->layers[l];
        char* kv = reinterpret_cast<char*>(L.kv_cache);
        // RMSNorm is fused into the GEMV that consumes it (bit-identical h, one launch less per norm)
        const bool q8 = L.wqkv8 && L.wo8 && L.wgu8 && L.wd8;      // W8A8 decode: e4m3 weights, rows quantised inside the GEMV
        UFV_TRY(ufv_gemv1(nullptr, x, L.ln1, m->eps, q8 ? L.wqkv8 : L.wqkv, D, q8 ? L.sqkv : nullptr, qkv, 0, qkv_n, D, L.bqkv, UFV_ACT_NONE,
                          nullptr, 0, stream));
        UFV_TRY(ufv_rope_kv(qkv, qkv_n, 1, H, KV, hd, m->inv_freq, pos, kv, m->ldkv, stream));
        UFV_TRY(ufv_attention_decode(qkv, 0, kv, 0, m->ldkv, kv + 2 * (size_t)KV * hd, 0, m->ldkv, o, 0, 1, H, KV, pos + 1, hd, scale,
                                     aws, m->attn_splits, stream));
        UFV_TRY(ufv_gemv1(o, nullptr, nullptr, 0.f, q8 ? L.wo8 : L.wo, H * hd, q8 ? L.so : nullptr, x, 1, D, H * hd, nullptr, UFV_ACT_NONE, x, 0,
                          stream));
        UFV_TRY(ufv_gemv1(nullptr, x, L.ln2, m->eps, q8 ? L.wgu8 : L.wgu, D, q8 ? L.sgu : nullptr, act, 0, 2 * I, D, nullptr, UFV_ACT_NONE,
                          nullptr, 1, stream));
        UFV_TRY(ufv_gemv1(act, nullptr, nullptr, 0.f, q8 ? L.wd8 : L.wd, I, q8 ? L.sd : nullptr, x, 1, D, I, nullptr, UFV_ACT_NONE, x, 0,
                          stream));
    }
    UFV_TRY(ufv_rmsnorm(x, D, normed, 1, D, m->norm, 1, D, m->eps, stream));
    if (hidden_out) UFV_TRY(ufv_convert(normed, UFV_DT_F32, hidden_out, UFV_DT_F32, D, stream));
    UFV_TRY(ufv_convert(normed, UFV_DT_F32, h, UFV_DT_BF16, D, stream));
    UFV_TRY(ufv_gemv1(h, nullptr, nullptr, 0.f, m->lm_head, D, nullptr, logits, 1, m->vocab, D, nullptr, UFV_ACT_NONE, nullptr, 0, stream));
    UFV_TRY(ufv_argmax(logits, m->vocab, next_token_dev, stream));
    return UFV_OK;
}
