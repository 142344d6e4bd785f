// MX instantiations of the e4m3 ping-pong GEMM, part 2: the 256 x 256 kernels whose epilogue EMITS e4m3 codes + block scales (MX & 2), with a per-row-scaled A operand
// (MX = 2), plain / activation and SwiGLU forms (gemm256_kernel.h epilogue256_wide_mx, epilogue256_swiglu_mx)
#include "gemm256_kernel.h"

int ufv_launch_pp_mx2(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, bool swiglu, int mx, hipStream_t st) {
    if (mx == 2) return swiglu ? launch_pp<false, true, true, 4, 4, 2, true, false, false, 2>(A, W, e, M, N, K, lda, ldw, st)
                               : launch_pp<false, false, true, 4, 4, 2, true, false, false, 2>(A, W, e, M, N, K, lda, ldw, st);
    // (MX = 3, block-scaled A in AND MX out, was built and dropped: 256 registers + 120-136 bytes of scratch at the 256 x 256 tile -- the producers take their
    //  A operand with per-row scales from the fused norm kernels)
    ufv_set_error("ufv_gemm_fp8_mx: bad MX mode %d", mx);
    return UFV_EINVAL;
}
