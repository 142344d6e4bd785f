// One greedy decode step of the Qwen2-style decoder as a single C call: embedding lookup of the previous token ->
// N x {[RMSNorm+]fused QKV GEMV(+bias), RoPE + KV-cache append, split-key decode attention, o_proj(+residual),
// [RMSNorm+]gate/up GEMV with SwiGLU, down(+residual)} -> final RMSNorm -> lm_head GEMV -> argmax.  Pure composition of the
// entry points in include/ufv.h on one stream (no allocation, no synchronisation): it exists so that the host pays
// one FFI call per token instead of ~340.
#include "common.h"
#include <cstdlib>
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
    const int64_t fused = (m->hd == 64 || m->hd == 128) ? ufv_attention_decode_fused_ws_bytes(m->n_q, m->hd, m->attn_splits) : 0;
    const int64_t plain = ufv_attention_decode_ws_bytes(1, m->n_q, m->hd, m->attn_splits);
    b += align256((size_t)(fused > plain ? fused : plain));
    b += align256((size_t)ufv_argmax_ws_bytes());    // arg-max partials + arrival counter (zero before the first step, like the attention counters)
    return (int64_t)b;
}

// pos_dev == nullptr: position `pos` from the host (launch arguments change every token).  pos_dev != nullptr: the position is
// read from device memory by the two kernels that need it and incremented at the end of the step, so every launch argument is
// the same for every token and the whole step can be captured into a HIP graph once (ufv_graph_*) and replayed.
static int decode_step_impl(const ufv_qwen2_model* m, const int64_t* token_dev, int pos, int* pos_dev, void* ws, int64_t ws_bytes,
                            float* logits, float* hidden_out, int64_t* next_token_dev, void* stream) {
    UFV_REQUIRE(m && token_dev && ws && logits && next_token_dev, "ufv_qwen2_decode_step: null argument");
    UFV_REQUIRE(ws_bytes >= ufv_qwen2_decode_ws_bytes(m), "ufv_qwen2_decode_step: workspace too small");
    UFV_REQUIRE(pos_dev || (pos >= 0 && pos < m->max_len), "ufv_qwen2_decode_step: position %d outside the KV cache (max_len %d)", pos, m->max_len);
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
    {
        const int64_t fused_b = (hd == 64 || hd == 128) ? ufv_attention_decode_fused_ws_bytes(H, hd, m->attn_splits) : 0;
        const int64_t plain_b = ufv_attention_decode_ws_bytes(1, H, hd, m->attn_splits);
        p += align256((size_t)(fused_b > plain_b ? fused_b : plain_b));
    }
    void* amws = p;
    const float scale = 1.0f / sqrtf((float)hd);
    // the workspace's arrival counters must be zero before the first step (the caller zero-fills `ws` once); UFV_DECODE_FUSED_ATTN=0: three launches
    static const bool fused_env = getenv("UFV_DECODE_FUSED_ATTN") == nullptr || getenv("UFV_DECODE_FUSED_ATTN")[0] != '0';
    const bool fused_attn = fused_env && (hd == 64 || hd == 128) && (m->ldkv % 8 == 0);

    UFV_TRY(ufv_gather_rows(m->embed, UFV_DT_BF16, D, token_dev, x, UFV_DT_F32, D, nullptr, 1, D, stream));
    for (int l = 0; l < m->n_layers; ++l) {
        const ufv_qwen2_layer& L = m->layers[l];
        char* kv = reinterpret_cast<char*>(L.kv_cache);
        // RMSNorm is fused into the GEMV that consumes it (bit-identical h, one launch less per norm)
        const bool q8 = L.wqkv8 && L.wo8 && L.wgu8 && L.wd8;      // W8A8 decode: e4m3 weights, rows quantised inside the GEMV
        UFV_TRY(ufv_gemv1(nullptr, x, L.ln1, m->eps, q8 ? L.wqkv8 : L.wqkv, D, q8 ? L.sqkv : nullptr, qkv, 0, qkv_n, D, L.bqkv, UFV_ACT_NONE,
                          nullptr, 0, stream));
        if (fused_attn) {
            // RoPE + KV append + split attention + merge in ONE launch (csrc/attn.hip attn_decode_fused; bit-identical to the three calls below)
            UFV_TRY(ufv_attention_decode_fused(qkv, H, KV, hd, m->inv_freq, pos, pos_dev, kv, m->ldkv, m->max_len, o, scale, aws, m->attn_splits, stream));
        } else if (pos_dev) {
            UFV_TRY(ufv_rope_kv1_dev(qkv, H, KV, hd, m->inv_freq, pos_dev, kv, m->ldkv, stream));
            UFV_TRY(ufv_attention_decode_dev(qkv, 0, kv, 0, m->ldkv, kv + 2 * (size_t)KV * hd, 0, m->ldkv, o, 0, 1, H, KV, pos_dev, m->max_len,
                                             hd, scale, aws, m->attn_splits, stream));
        } else {
            UFV_TRY(ufv_rope_kv(qkv, qkv_n, 1, H, KV, hd, m->inv_freq, pos, kv, m->ldkv, stream));
            UFV_TRY(ufv_attention_decode(qkv, 0, kv, 0, m->ldkv, kv + 2 * (size_t)KV * hd, 0, m->ldkv, o, 0, 1, H, KV, pos + 1, hd, scale,
                                         aws, m->attn_splits, stream));
        }
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
    if (((uintptr_t)logits & 15) == 0) UFV_TRY(ufv_argmax_ws(logits, m->vocab, next_token_dev, amws, stream));
    else UFV_TRY(ufv_argmax(logits, m->vocab, next_token_dev, stream));
    if (pos_dev) UFV_TRY(ufv_add_int(pos_dev, 1, stream));
    return UFV_OK;
}

extern "C" int ufv_qwen2_decode_step(const ufv_qwen2_model* m, const int64_t* token_dev, int pos, void* ws, int64_t ws_bytes,
                                     float* logits, float* hidden_out, int64_t* next_token_dev, void* stream) {
    return decode_step_impl(m, token_dev, pos, nullptr, ws, ws_bytes, logits, hidden_out, next_token_dev, stream);
}

extern "C" int ufv_qwen2_decode_step_dev(const ufv_qwen2_model* m, const int64_t* token_dev, int* pos_dev, void* ws, int64_t ws_bytes,
                                         float* logits, float* hidden_out, int64_t* next_token_dev, void* stream) {
    UFV_REQUIRE(pos_dev, "ufv_qwen2_decode_step_dev: null position");
    return decode_step_impl(m, token_dev, 0, pos_dev, ws, ws_bytes, logits, hidden_out, next_token_dev, stream);
}

// ---- HIP graph capture of a launch-bound sequence (the ~200 dependent launches of one decode step) --------------------
// begin -> any ufv_* calls on `stream` (recorded, not executed) -> end returns an executable graph; launch replays it.
// `stream` must not be the legacy default stream.  Nothing may synchronise or allocate between begin and end.
extern "C" int ufv_graph_begin(void* stream) {
    UFV_REQUIRE(stream, "ufv_graph_begin: the legacy default stream cannot be captured; use a created stream");
    if (hipStreamBeginCapture(reinterpret_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal) != hipSuccess) {
        ufv_set_error("ufv_graph_begin: %s", hipGetErrorString(hipGetLastError()));
        return UFV_EHIP;
    }
    return UFV_OK;
}

extern "C" int ufv_graph_end(void* stream, void** exec_out) {
    UFV_REQUIRE(stream && exec_out, "ufv_graph_end: bad arguments");
    hipGraph_t g = nullptr;
    if (hipStreamEndCapture(reinterpret_cast<hipStream_t>(stream), &g) != hipSuccess || !g) {
        ufv_set_error("ufv_graph_end: capture failed: %s", hipGetErrorString(hipGetLastError()));
        return UFV_EHIP;
    }
    hipGraphExec_t ex = nullptr;
    const hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess || !ex) {
        ufv_set_error("ufv_graph_end: instantiate failed: %s", hipGetErrorString(e));
        return UFV_EHIP;
    }
    *exec_out = ex;
    return UFV_OK;
}

extern "C" int ufv_graph_launch(void* exec, void* stream) {
    UFV_REQUIRE(exec, "ufv_graph_launch: null graph");
    if (hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(exec), reinterpret_cast<hipStream_t>(stream)) != hipSuccess) {
        ufv_set_error("ufv_graph_launch: %s", hipGetErrorString(hipGetLastError()));
        return UFV_EHIP;
    }
    return UFV_OK;
}

extern "C" int ufv_graph_destroy(void* exec) {
    if (exec) (void)hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(exec));
    return UFV_OK;
}
