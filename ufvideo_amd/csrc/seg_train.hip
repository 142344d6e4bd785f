// Backward-pass kernels of the [SEG] mask losses (videorefer_qwen2.py:34-77 dice / sigmoid-CE, :279-338 loss
// assembly; sam2.py `_forward_sam_heads` under torch autograd in the reference).  The mask decoder is a small model (4 M
// parameters, 9 prompt tokens against h*w image tokens): its GEMMs, norms and activations reuse the training kernels of train.hip /
// train_proj.hip; what is new here is
//   * attention between FEW tokens and MANY tokens in both directions, forward (with the log-sum-exp) and backward:
//       item = one (batch, head, query) [forward, dQ] or one (batch, head, key) [dK, dV]; an item is served by L lanes (L = 64: a
//       wave strides over the long side, L = 1: one thread when the other side is a handful of tokens), partial results are
//       combined with wave shuffles -- no atomics, fixed summation order;
//   * the selected-mask product  masks[b, p] = sum_c up[b, p, c] * h[b, c]  and its two gradients;
//   * F.interpolate(bilinear, align_corners=False) backward as a GATHER (each input pixel sums the output pixels that read it);
//   * d(BCE-with-logits + DICE) / d(logit).
// HBM-bound elementwise / small-reduction kernels; fp32 statistics; bf16 activations like the rest of the training path.
#include "common.h"
#include "../../include/ufv.h"

namespace {

#define ST(s) reinterpret_cast<hipStream_t>(s)
inline int grid_for(int64_t n, int per = 256) { return (int)((n + per - 1) / per > 1048576 ? 1048576 : (n + per - 1) / per); }

template <int L>
__device__ __forceinline__ float item_sum(float v) {
    if (L == 64) return wave_sum(v);
    return v;
}
template <int L>
__device__ __forceinline__ float item_max(float v) {
    if (L == 64) return wave_max(v);
    return v;
}

// ---- forward: o[b, i, h] = softmax_j(q_i . k_j * scale) v_j ; lse[b, h, i] = log sum_j exp(s_ij)  (natural log, scaled scores)
template <int HD, int L>
__global__ __launch_bounds__(256) void small_attn_fwd_k(const bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ v,
                                                        bf16* __restrict__ o, float* __restrict__ lse, int B, int H, int Nq, int Nk, float scale) {
    const int64_t item = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / L;
    const int sub = threadIdx.x % L;
    if (item >= (int64_t)B * H * Nq) return;
    const int i = (int)(item % Nq), h = (int)((item / Nq) % H), b = (int)(item / ((int64_t)Nq * H));
    const int W = H * HD;
    float qv[HD];
    const bf16* qp = q + ((int64_t)b * Nq + i) * W + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) qv[d] = (float)qp[d] * scale;
    float m = -INFINITY, s = 0.f, acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
    for (int j = sub; j < Nk; j += L) {
        const bf16* kp = k + ((int64_t)b * Nk + j) * W + h * HD;
        const bf16* vp = v + ((int64_t)b * Nk + j) * W + h * HD;
        float x = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) x += qv[d] * (float)kp[d];
        const float mn = fmaxf(m, x), a = __expf(m - mn), p = __expf(x - mn);
        s = s * a + p;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = acc[d] * a + p * (float)vp[d];
        m = mn;
    }
    const float M = item_max<L>(m);
    const float a = (m == -INFINITY) ? 0.f : __expf(m - M);
    const float S = item_sum<L>(s * a);
    const float inv = 1.f / S;
    bf16* op = o + ((int64_t)b * Nq + i) * W + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        const float t = item_sum<L>(acc[d] * a) * inv;
        if (sub == 0) op[d] = (bf16)t;
    }
    if (sub == 0) lse[((int64_t)b * H + h) * Nq + i] = M + __logf(S);
}

// ---- dQ (and delta = dO . O, kept for the dK / dV pass)
template <int HD, int L>
__global__ __launch_bounds__(256) void small_attn_dq_k(const bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ v,
                                                       const bf16* __restrict__ o, const bf16* __restrict__ dO, const float* __restrict__ lse,
                                                       bf16* __restrict__ dq, float* __restrict__ delta, int B, int H, int Nq, int Nk, float scale) {
    const int64_t item = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / L;
    const int sub = threadIdx.x % L;
    if (item >= (int64_t)B * H * Nq) return;
    const int i = (int)(item % Nq), h = (int)((item / Nq) % H), b = (int)(item / ((int64_t)Nq * H));
    const int W = H * HD;
    const int64_t ro = ((int64_t)b * Nq + i) * W + h * HD;
    float qv[HD], dov[HD], acc[HD];
    float dl = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        qv[d] = (float)q[ro + d] * scale; dov[d] = (float)dO[ro + d]; acc[d] = 0.f;
        dl += dov[d] * (float)o[ro + d];
    }
    const float ls = lse[((int64_t)b * H + h) * Nq + i];
    for (int j = sub; j < Nk; j += L) {
        const bf16* kp = k + ((int64_t)b * Nk + j) * W + h * HD;
        const bf16* vp = v + ((int64_t)b * Nk + j) * W + h * HD;
        float x = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) { x += qv[d] * (float)kp[d]; dp += dov[d] * (float)vp[d]; }
        const float ds = __expf(x - ls) * (dp - dl);
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] += ds * (float)kp[d];
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        const float t = item_sum<L>(acc[d]) * scale;
        if (sub == 0) dq[ro + d] = (bf16)t;
    }
    if (sub == 0) delta[((int64_t)b * H + h) * Nq + i] = dl;
}

// ---- dK, dV: item = (b, h, key j), the L lanes stride over the queries
template <int HD, int L>
__global__ __launch_bounds__(256) void small_attn_dkv_k(const bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ v,
                                                        const bf16* __restrict__ dO, const float* __restrict__ lse, const float* __restrict__ delta,
                                                        bf16* __restrict__ dk, bf16* __restrict__ dv, int B, int H, int Nq, int Nk, float scale) {
    const int64_t item = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / L;
    const int sub = threadIdx.x % L;
    if (item >= (int64_t)B * H * Nk) return;
    const int j = (int)(item % Nk), h = (int)((item / Nk) % H), b = (int)(item / ((int64_t)Nk * H));
    const int W = H * HD;
    const int64_t ko = ((int64_t)b * Nk + j) * W + h * HD;
    float kv[HD], vv[HD], ak[HD], av[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { kv[d] = (float)k[ko + d]; vv[d] = (float)v[ko + d]; ak[d] = 0.f; av[d] = 0.f; }
    for (int i = sub; i < Nq; i += L) {
        const int64_t ro = ((int64_t)b * Nq + i) * W + h * HD;
        float x = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) { x += (float)q[ro + d] * kv[d]; dp += (float)dO[ro + d] * vv[d]; }
        const int64_t so = ((int64_t)b * H + h) * Nq + i;
        const float p = __expf(x * scale - lse[so]);
        const float ds = p * (dp - delta[so]) * scale;
#pragma unroll
        for (int d = 0; d < HD; ++d) { av[d] += p * (float)dO[ro + d]; ak[d] += ds * (float)q[ro + d]; }
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        const float tk = item_sum<L>(ak[d]), tv = item_sum<L>(av[d]);
        if (sub == 0) { dk[ko + d] = (bf16)tk; dv[ko + d] = (bf16)tv; }
    }
}

// ---- masks[b, p] = sum_c up[b * P + p, c] * h[b, c]
__global__ __launch_bounds__(256) void mask_dot_fwd_k(const bf16* __restrict__ up, const float* __restrict__ hs, float* __restrict__ out, int B, int P, int C) {
    const int64_t n = (int64_t)B * P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / P);
        const bf16* u = up + i * C;
        const float* hh = hs + (int64_t)b * C;
        float acc = 0.f;
        for (int c = 0; c < C; ++c) acc += (float)u[c] * hh[c];
        out[i] = acc;
    }
}
// d_up[b*P + p, c] = dm[b, p] * h[b, c];   partial[b, blk, c] = sum over the block's pixels of dm * up  (reduced by the second kernel)
__global__ __launch_bounds__(256) void mask_dot_bwd_k(const bf16* __restrict__ up, const float* __restrict__ hs, const float* __restrict__ dm,
                                                      bf16* __restrict__ dup, float* __restrict__ partial, int B, int P, int C, int nblk) {
    __shared__ float red[16];
    const int b = blockIdx.y, blk = blockIdx.x;
    const int per = (P + nblk - 1) / nblk, p0 = blk * per, p1 = min(P, p0 + per);
    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0.f;
    const float* hh = hs + (int64_t)b * C;
    for (int p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
        const int64_t r = (int64_t)b * P + p;
        const float g = dm[r];
#pragma unroll
        for (int c = 0; c < 32; ++c)
            if (c < C) {
                acc[c] += g * (float)up[r * C + c];
                dup[r * C + c] = (bf16)(g * hh[c]);
            }
    }
#pragma unroll
    for (int c = 0; c < 32; ++c)
        if (c < C) {
            const float t = block_sum(acc[c], red);
            if (threadIdx.x == 0) partial[((int64_t)b * nblk + blk) * C + c] = t;
        }
}
__global__ void mask_dot_reduce_k(const float* __restrict__ partial, float* __restrict__ dh, int B, int C, int nblk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i % C;
    float t = 0.f;
    for (int k = 0; k < nblk; ++k) t += partial[((int64_t)b * nblk + k) * C + c];
    dh[i] = t;
}

// ---- bilinear (align_corners=False) backward: din[n, y, x] = sum over the output pixels whose 2x2 stencil contains (y, x)
__device__ __forceinline__ void taps(int d, float s, int n_in, int& i0, int& i1, float& l) {
    const float f = fmaxf(s * (d + 0.5f) - 0.5f, 0.f);
    i0 = min((int)f, n_in - 1);
    i1 = i0 + (i0 < n_in - 1);
    l = f - i0;
}
__global__ __launch_bounds__(256) void resize_bilinear_bwd_k(const float* __restrict__ dout, float* __restrict__ din, int N, int Hs, int Ws, int Hd, int Wd) {
    const int64_t total = (int64_t)N * Hs * Ws;
    const float sy = (float)Hs / (float)Hd, sx = (float)Ws / (float)Wd;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Ws), y = (int)((i / Ws) % Hs), n = (int)(i / ((int64_t)Ws * Hs));
        // output rows whose source coordinate lies in (y - 1, y + 1): d in ((y - 0.5) / sy - 0.5, (y + 1.5) / sy - 0.5); the border
        // rows also collect the clamped coordinates
        int dy0 = (int)floorf((y - 0.5f) / sy - 0.5f) - 1, dy1 = (int)ceilf((y + 1.5f) / sy - 0.5f) + 1;
        int dx0 = (int)floorf((x - 0.5f) / sx - 0.5f) - 1, dx1 = (int)ceilf((x + 1.5f) / sx - 0.5f) + 1;
        if (y == 0) dy0 = 0;
        if (x == 0) dx0 = 0;
        if (y == Hs - 1) dy1 = Hd - 1;
        if (x == Ws - 1) dx1 = Wd - 1;
        dy0 = max(dy0, 0); dx0 = max(dx0, 0); dy1 = min(dy1, Hd - 1); dx1 = min(dx1, Wd - 1);
        const float* g = dout + (int64_t)n * Hd * Wd;
        float acc = 0.f;
        for (int dy = dy0; dy <= dy1; ++dy) {
            int a0, a1; float ly;
            taps(dy, sy, Hs, a0, a1, ly);
            const float wy = (a0 == y ? 1.f - ly : 0.f) + (a1 == y ? ly : 0.f);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int dx = dx0; dx <= dx1; ++dx) {
                int b0, b1; float lx;
                taps(dx, sx, Ws, b0, b1, lx);
                const float wx = (b0 == x ? 1.f - lx : 0.f) + (b1 == x ? lx : 0.f);
                if (wx != 0.f) row += wx * g[(int64_t)dy * Wd + dx];
            }
            acc += wy * row;
        }
        din[i] = acc;
    }
}

// ---- d(loss)/d(logit): cb * (sigmoid(x) - t) + sigmoid(x) (1 - sigmoid(x)) (a[n] * t + b[n])
__global__ __launch_bounds__(256) void mask_loss_bwd_k(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ coef,
                                                       float cb, float* __restrict__ dx, int N, int64_t HW) {
    const int64_t total = (int64_t)N * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / HW);
        const float s = 1.f / (1.f + __expf(-x[i])), tt = t[i];
        dx[i] = cb * (s - tt) + s * (1.f - s) * (coef[2 * n] * tt + coef[2 * n + 1]);
    }
}

}  // namespace

#define SA_LAUNCH(KERN, HD_, L_, items, ...)                                                                                   \
    do {                                                                                                                       \
        const int64_t thr = (int64_t)(items) * (L_);                                                                           \
        hipLaunchKernelGGL((KERN<HD_, L_>), dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, ST(stream), __VA_ARGS__);        \
    } while (0)
#define SA_DISPATCH(KERN, items, wide, ...)                                               \
    do {                                                                                  \
        if (hd == 16) { if (wide) SA_LAUNCH(KERN, 16, 64, items, __VA_ARGS__); else SA_LAUNCH(KERN, 16, 1, items, __VA_ARGS__); } \
        else { if (wide) SA_LAUNCH(KERN, 32, 64, items, __VA_ARGS__); else SA_LAUNCH(KERN, 32, 1, items, __VA_ARGS__); }         \
    } while (0)

extern "C" int ufv_small_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq, int Nk, int hd,
                                  float scale, void* stream) {
    UFV_REQUIRE(q && k && v && o && lse && B > 0 && H > 0 && Nq > 0 && Nk > 0 && (hd == 16 || hd == 32), "ufv_small_attn_fwd: bad arguments (head_dim 16 or 32)");
    const bool wide = Nk > 32;                               // many keys: a wave per query strides over them
    SA_DISPATCH(small_attn_fwd_k, (int64_t)B * H * Nq, wide, (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, lse, B, H, Nq, Nk, scale);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_small_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse, float* delta,
                                  void* dq, void* dk, void* dv, int B, int H, int Nq, int Nk, int hd, float scale, void* stream) {
    UFV_REQUIRE(q && k && v && o && dO && lse && delta && dq && dk && dv && B > 0 && H > 0 && Nq > 0 && Nk > 0 && (hd == 16 || hd == 32),
                "ufv_small_attn_bwd: bad arguments (head_dim 16 or 32)");
    const bool wide_q = Nk > 32, wide_k = Nq > 32;
    SA_DISPATCH(small_attn_dq_k, (int64_t)B * H * Nq, wide_q, (const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)o, (const bf16*)dO, lse,
                (bf16*)dq, delta, B, H, Nq, Nk, scale);
    SA_DISPATCH(small_attn_dkv_k, (int64_t)B * H * Nk, wide_k, (const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)dO, lse, delta, (bf16*)dk,
                (bf16*)dv, B, H, Nq, Nk, scale);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_mask_dot_fwd(const void* up, const float* h, float* out, int B, int P, int C, void* stream) {
    UFV_REQUIRE(up && h && out && B > 0 && P > 0 && C > 0 && C <= 32, "ufv_mask_dot_fwd: bad arguments (C <= 32)");
    hipLaunchKernelGGL(mask_dot_fwd_k, dim3(grid_for((int64_t)B * P)), dim3(256), 0, ST(stream), (const bf16*)up, h, out, B, P, C);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int64_t ufv_mask_dot_bwd_ws_bytes(int B, int C) { return (int64_t)B * 64 * C * sizeof(float); }

extern "C" int ufv_mask_dot_bwd(const void* up, const float* h, const float* dm, void* dup, float* dh, void* ws, int B, int P, int C, void* stream) {
    UFV_REQUIRE(up && h && dm && dup && dh && ws && B > 0 && P > 0 && C > 0 && C <= 32, "ufv_mask_dot_bwd: bad arguments (C <= 32)");
    const int nblk = 64;
    hipLaunchKernelGGL(mask_dot_bwd_k, dim3(nblk, B), dim3(256), 0, ST(stream), (const bf16*)up, h, dm, (bf16*)dup, (float*)ws, B, P, C, nblk);
    hipLaunchKernelGGL(mask_dot_reduce_k, dim3(cdiv(B * C, 64)), dim3(64), 0, ST(stream), (const float*)ws, dh, B, C, nblk);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_resize_bilinear_bwd(const float* dout, float* din, int N, int Hs, int Ws, int Hd, int Wd, void* stream) {
    UFV_REQUIRE(dout && din && N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "ufv_resize_bilinear_bwd: bad arguments");
    hipLaunchKernelGGL(resize_bilinear_bwd_k, dim3(grid_for((int64_t)N * Hs * Ws)), dim3(256), 0, ST(stream), dout, din, N, Hs, Ws, Hd, Wd);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_mask_loss_bwd(const float* x, const float* t, const float* coef, float cb, float* dx, int N, int64_t HW, void* stream) {
    UFV_REQUIRE(x && t && coef && dx && N > 0 && HW > 0, "ufv_mask_loss_bwd: bad arguments");
    hipLaunchKernelGGL(mask_loss_bwd_k, dim3(grid_for((int64_t)N * HW)), dim3(256), 0, ST(stream), x, t, coef, cb, dx, N, HW);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}
