// bf16-operand instantiations of the ping-pong GEMM at the named tile shapes (gemm256_kernel.h)
#include "gemm256_kernel.h"

int ufv_launch_pp_shape_bf16(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, bool out_f32, int shape, hipStream_t st) {
    return out_f32 ? launch_pp_shape<true, false>(A, W, e, M, N, K, lda, ldw, shape, st) : launch_pp_shape<false, false>(A, W, e, M, N, K, lda, ldw, shape, st);
}
