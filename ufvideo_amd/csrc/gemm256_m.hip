// MX instantiations of the e4m3 ping-pong GEMM, part 1: block-scaled A operand (MX = 1) at the named tile shapes, fp32 / bf16 output (gemm256_kernel.h MXA)
#include "gemm256_kernel.h"

template <bool F>
static int launch_mxa(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, int shape, hipStream_t st) {
    switch (shape) {
        // (1442 / 1432 with a block-scaled A operand spill: not built)
        case 1332: return launch_pp<F, false, true, 3, 3, 2, true, false, false, 1>(A, W, e, M, N, K, lda, ldw, st);
        case 1322: return launch_pp<F, false, true, 3, 2, 2, true, false, false, 1>(A, W, e, M, N, K, lda, ldw, st);
        case 1441: return launch_pp<F, false, true, 4, 4, 1, true, false, false, 1>(A, W, e, M, N, K, lda, ldw, st);
        case 1431: return launch_pp<F, false, true, 4, 3, 1, true, false, false, 1>(A, W, e, M, N, K, lda, ldw, st);
        case 1331: return launch_pp<F, false, true, 3, 3, 1, true, false, false, 1>(A, W, e, M, N, K, lda, ldw, st);
    }
    ufv_set_error("ufv_gemm_fp8_mx: unknown ping-pong tile shape %d", shape);
    return UFV_EINVAL;
}

int ufv_launch_pp_mx2(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, bool swiglu, int mx, hipStream_t st);

int ufv_launch_pp_mx(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, bool out_f32, bool swiglu, int shape, int mx, hipStream_t st) {
    if (mx & 2) return ufv_launch_pp_mx2(A, W, e, M, N, K, lda, ldw, swiglu, mx, st);
    if (swiglu) {
        ufv_set_error("ufv_gemm_fp8_mx: the SwiGLU epilogue with a block-scaled A operand is built in its MX-emitting form only");
        return UFV_EUNSUPPORTED;
    }
    return out_f32 ? launch_mxa<true>(A, W, e, M, N, K, lda, ldw, shape, st) : launch_mxa<false>(A, W, e, M, N, K, lda, ldw, shape, st);
}
