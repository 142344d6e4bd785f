// e4m3-operand instantiations of the ping-pong GEMM at the named tile shapes (gemm256_kernel.h)
#include "gemm256_kernel.h"

int ufv_launch_pp_shape_fp8(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, bool out_f32, int shape, hipStream_t st) {
    return out_f32 ? launch_pp_shape<true, true>(A, W, e, M, N, K, lda, ldw, shape, st) : launch_pp_shape<false, true>(A, W, e, M, N, K, lda, ldw, shape, st);
}
