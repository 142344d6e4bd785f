// Fused QKV projection + RoPE + KV-cache append: the ROPE instantiations of the ping-pong GEMM (gemm256_kernel.h, epilogue256_rope), one build job.
#include "gemm256_kernel.h"

// the same with e4m3 operands (per-row-scaled activations, per-channel-scaled weights: the accumulators are de-quantised before the epilogue, as in every e4m3 kernel)
int ufv_launch_pp_rope_fp8(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, int shape, hipStream_t st) {
    if (shape == 1332) return launch_pp<false, false, true, 3, 3, 2, true, false, true>(A, W, e, M, N, K, lda, ldw, st);
    ufv_set_error("ufv_gemm_qkv_rope: no fused e4m3 kernel at tile shape %d", shape);
    return UFV_EINVAL;
}

int ufv_launch_pp_rope(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, int shape, hipStream_t st) {
    // 192 x 256: a 256-column shape with an even number of accumulator rows.  (256 x 256 was built and dropped: 128 accumulator registers + the 64 cos / sin
    // registers of a lane's 8 rows spill -- 80 bytes of scratch, whose loads inside the epilogue also break the store count of the relaxed waits; the CPU
    // ISA test caught it.)
    switch (shape) {
        case 1332: return launch_pp<false, false, false, 3, 3, 2, true, false, true>(A, W, e, M, N, K, lda, ldw, st);      // 192 x 256
    }
    ufv_set_error("ufv_gemm_qkv_rope: no fused kernel at tile shape %d", shape);
    return UFV_EINVAL;
}
