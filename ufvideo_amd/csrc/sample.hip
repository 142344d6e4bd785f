// Token sampling for generate(do_sample=True): temperature -> top-k -> top-p -> multinomial, the chain HF's GenerationMixin
// builds from the kwargs the reference forwards (ufvideo/__init__.py:113-127: do_sample, temperature, top_p;
// transformers TemperatureLogitsWarper / TopKLogitsWarper / TopPLogitsWarper + torch.multinomial).
//
// One block per row of logits, no sort: both cut-offs are found by bisection on the order-preserving integer image of the
// float logits (32 passes each over the row, which stays in L2):
//   top-k : t_k = the k-th largest logit  = max { t : #{x >= t} >= k }                  (ties at t_k are kept, as HF does)
//   top-p : t_p = max { t >= t_k : sum_{x >= t} softmax_T(x) >= top_p * Z },  Z over the top-k survivors
//           = the smallest set of most probable tokens whose mass reaches top_p (HF's ascending-cumsum rule); elements
//           exactly tied with the cut-off value are all kept (HF keeps them in sort order until the mass is reached).
//   draw  : the first index (in vocabulary order) whose running kept mass exceeds u * kept_mass, u in [0,1) from the caller.
#include "common.h"
#include "../../include/ufv.h"

namespace {

__device__ __forceinline__ uint32_t f2ord(float f) {            // monotone float -> uint32
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float ord2f(uint32_t o) {             // inverse of f2ord
    return __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
}

__device__ __forceinline__ float block_sum1024(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i];
    return t;
}

__global__ __launch_bounds__(1024) void sample_top_p_k(const float* __restrict__ logits, int64_t ld, int V, float inv_temp, int top_k,
                                                       float top_p, const float* __restrict__ u, int64_t* __restrict__ out,
                                                       float* __restrict__ kept_out) {
    __shared__ float red[16];
    __shared__ float scan[1024];
    __shared__ int sel[2];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* x = logits + (int64_t)row * ld;
    // ---- max
    float mx = -INFINITY;
    for (int j = tid; j < V; j += 1024) mx = fmaxf(mx, x[j]);
    mx = wave_max(mx);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, red[i]);
    // ---- top-k threshold (ordered-int bisection): largest t with count(x >= t) >= k
    uint32_t tk = 0;                                              // 0 = below every finite float: keeps everything
    if (top_k > 0 && top_k < V) {
        uint32_t lo = 0, hi = f2ord(mx);                          // invariant: count(>= lo) >= k
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo + 1) >> 1);
            float c = 0.f;
            for (int j = tid; j < V; j += 1024) c += (f2ord(x[j]) >= mid) ? 1.f : 0.f;
            c = block_sum1024(c, red);
            if (c >= (float)top_k) lo = mid; else hi = mid - 1;
        }
        tk = lo;
    }
    // ---- mass of the survivors, then the top-p threshold
    float z = 0.f;
    for (int j = tid; j < V; j += 1024) z += (f2ord(x[j]) >= tk) ? __expf((x[j] - mx) * inv_temp) : 0.f;
    z = block_sum1024(z, red);
    uint32_t tp = tk;
    if (top_p < 1.0f) {
        const float need = top_p * z;
        uint32_t lo = tk, hi = f2ord(mx);                         // invariant: mass(>= lo) >= need
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo + 1) >> 1);
            float m = 0.f;
            for (int j = tid; j < V; j += 1024) m += (f2ord(x[j]) >= mid) ? __expf((x[j] - mx) * inv_temp) : 0.f;
            m = block_sum1024(m, red);
            if (m >= need) lo = mid; else hi = mid - 1;
        }
        tp = lo;
    }
    // ---- inverse CDF in vocabulary order: thread t owns the contiguous chunk [t*per, (t+1)*per)
    const int per = (V + 1023) / 1024, j0 = tid * per, j1 = min(V, j0 + per);
    float mine = 0.f;
    for (int j = j0; j < j1; ++j) mine += (f2ord(x[j]) >= tp) ? __expf((x[j] - mx) * inv_temp) : 0.f;
    scan[tid] = mine;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {                    // inclusive Hillis-Steele scan
        const float add = tid >= off ? scan[tid - off] : 0.f;
        __syncthreads();
        scan[tid] += add;
        __syncthreads();
    }
    const float total = scan[1023];
    if (kept_out && tid == 0) { kept_out[2 * row] = total / z; kept_out[2 * row + 1] = ord2f(tp); }
    const float target = fminf(u[row], 0.99999994f) * total;      // < total
    // owner = the first thread that holds kept mass and whose inclusive prefix exceeds the target (threads without mass are
    // skipped explicitly: the fp32 scan may differ from thread to thread in the last bit); fallback = the last thread with mass
    if (tid == 0) { sel[0] = 0x7fffffff; sel[1] = -1; }
    __syncthreads();
    if (mine > 0.f) {
        if (scan[tid] > target) atomicMin(&sel[0], tid);
        atomicMax(&sel[1], tid);
    }
    __syncthreads();
    const int owner = sel[0] != 0x7fffffff ? sel[0] : sel[1];
    if (tid == owner) {
        float run = scan[tid] - mine;
        int pick = -1, last = -1;
        for (int j = j0; j < j1; ++j)
            if (f2ord(x[j]) >= tp) {
                run += __expf((x[j] - mx) * inv_temp);
                last = j;
                if (run > target) { pick = j; break; }
            }
        out[row] = pick >= 0 ? pick : last;                       // rounding between the scan and the walk: the chunk's last kept token
    }
}

}  // namespace

extern "C" int ufv_sample_top_p(const float* logits, int64_t ld, int M, int V, float temperature, int top_k, float top_p, const float* u,
                                int64_t* out, float* kept_out, void* stream) {
    UFV_REQUIRE(logits && u && out && M > 0 && V > 0 && temperature > 0.f && top_p > 0.f && top_p <= 1.0f && top_k >= 0,
                "ufv_sample_top_p: needs temperature > 0, 0 < top_p <= 1, top_k >= 0 (temperature=%g top_p=%g top_k=%d)", (double)temperature,
                (double)top_p, top_k);
    hipLaunchKernelGGL(sample_top_p_k, dim3(M), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), logits, ld, V, 1.0f / temperature, top_k,
                       top_p, u, out, kept_out);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}
