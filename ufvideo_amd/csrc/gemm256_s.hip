// Split-K instantiations of the ping-pong GEMM (bf16 operands, fp32 output accumulated in turn order; gemm256_kernel.h KSPL)
#include "gemm256_kernel.h"

int ufv_launch_pp_split(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, int shape, int ksplit, hipStream_t st) {
    switch (shape) {
        case 0: case 1442: return launch_pp<true, false, false, 4, 4, 2, true, true>(A, W, e, M, N, K, lda, ldw, st, ksplit);
        case 1432: return launch_pp<true, false, false, 4, 3, 2, true, true>(A, W, e, M, N, K, lda, ldw, st, ksplit);
        case 1332: return launch_pp<true, false, false, 3, 3, 2, true, true>(A, W, e, M, N, K, lda, ldw, st, ksplit);
        case 1322: return launch_pp<true, false, false, 3, 2, 2, true, true>(A, W, e, M, N, K, lda, ldw, st, ksplit);
        case 1441: return launch_pp<true, false, false, 4, 4, 1, true, true>(A, W, e, M, N, K, lda, ldw, st, ksplit);
        case 1431: return launch_pp<true, false, false, 4, 3, 1, true, true>(A, W, e, M, N, K, lda, ldw, st, ksplit);
        case 1331: return launch_pp<true, false, false, 3, 3, 1, true, true>(A, W, e, M, N, K, lda, ldw, st, ksplit);
    }
    ufv_set_error("ufv_gemm: unknown ping-pong tile shape %d", shape);
    return UFV_EINVAL;
}
