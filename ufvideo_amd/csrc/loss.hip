// Forward values of the training losses (reference videorefer_qwen2.py:34-77 dice_loss / sigmoid_ce_loss and the causal-LM
// cross entropy HF computes inside Qwen2ForCausalLM.forward, used at :198-215).  HBM-bound reductions; no backward.
#include "common.h"
#include "../../include/ufv.h"

namespace {

#define ST(s) reinterpret_cast<hipStream_t>(s)

// loss[i] = logsumexp(logits[i, :]) - logits[i, label[i]]   (0 when label[i] == ignore_index); one block per row
__global__ __launch_bounds__(256) void cross_entropy_rows_k(const float* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                            int V, int64_t ignore_index, float* __restrict__ loss) {
    __shared__ float red[16];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int64_t lab = labels[row];
    if (lab == ignore_index) {
        if (tid == 0) loss[row] = 0.f;
        return;
    }
    const float* x = logits + (int64_t)row * ld;
    float mx = -INFINITY;
    for (int j = tid; j < V; j += 256) mx = fmaxf(mx, x[j]);
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int j = tid; j < V; j += 256) s += expf(x[j] - mx);
    s = block_sum(s, red + 4);
    if (tid == 0) loss[row] = logf(s) + mx - x[lab];
}

// per mask n over HW elements: sums[n] = { sum bce_with_logits(x, t), sum sigmoid(x) * t, sum sigmoid(x), sum t }
__global__ __launch_bounds__(256) void mask_loss_sums_k(const float* __restrict__ pred, const float* __restrict__ gt, int64_t HW,
                                                        float* __restrict__ sums) {
    __shared__ float red[16];
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* x = pred + (int64_t)n * HW;
    const float* t = gt + (int64_t)n * HW;
    float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
    for (int64_t i = tid; i < HW; i += 256) {
        const float xv = x[i], tv = t[i];
        a += fmaxf(xv, 0.f) - xv * tv + log1pf(expf(-fabsf(xv)));
        const float sg = 1.0f / (1.0f + expf(-xv));
        b += sg * tv; c += sg; d += tv;
    }
    a = block_sum(a, red); b = block_sum(b, red); c = block_sum(c, red); d = block_sum(d, red);
    if (tid == 0) { sums[n * 4 + 0] = a; sums[n * 4 + 1] = b; sums[n * 4 + 2] = c; sums[n * 4 + 3] = d; }
}

}  // namespace

extern "C" int ufv_cross_entropy_rows(const float* logits, int64_t ld, const int64_t* labels, int M, int V, int64_t ignore_index,
                                      float* loss, void* stream) {
    UFV_REQUIRE(logits && labels && loss && M > 0 && V > 0, "ufv_cross_entropy_rows: bad arguments");
    hipLaunchKernelGGL(cross_entropy_rows_k, dim3(M), dim3(256), 0, ST(stream), logits, ld, labels, V, ignore_index, loss);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_mask_loss_sums(const float* pred, const float* gt, int n_masks, int64_t HW, float* sums, void* stream) {
    if (n_masks == 0) return UFV_OK;
    UFV_REQUIRE(pred && gt && sums && n_masks > 0 && HW > 0, "ufv_mask_loss_sums: bad arguments");
    hipLaunchKernelGGL(mask_loss_sums_k, dim3(n_masks), dim3(256), 0, ST(stream), pred, gt, HW, sums);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}
