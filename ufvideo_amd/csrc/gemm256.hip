// The ping-pong GEMM's 256x256 tile (two phases per K-tile; SwiGLU and plain epilogues), its stream-K variant, and the dispatch to the
// other tile shapes.  Kernel: gemm256_kernel.h.
#include "gemm256_kernel.h"
#include <cstdlib>

template <bool F, bool S, bool Q>
static int launch256_t(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, bool streamk, int shape, hipStream_t st) {
    // shape: 0 / 1442 = the 256x256 tile; other codes (1000 + MA0 MA1 NB1, plain epilogues only) are built in gemm256_b.hip / gemm256_q.hip;
    // + 10000 * parts = the split-K form of that shape (gemm256_s.hip)
    if (shape >= 10000) {
        if (!F || S || Q) {
            ufv_set_error("ufv_gemm: split-K is built for bf16 operands with an fp32 output (no SwiGLU)");
            return UFV_EUNSUPPORTED;
        }
        return ufv_launch_pp_split(A, W, e, M, N, K, lda, ldw, shape % 10000, shape / 10000, st);
    }
    if (shape != 0 && shape != 1442) {
        if constexpr (S) {
            ufv_set_error("ufv_gemm: the SwiGLU epilogue is built for the 256x256 tile only");
            return UFV_EUNSUPPORTED;
        } else {
            return Q ? ufv_launch_pp_shape_fp8(A, W, e, M, N, K, lda, ldw, F, shape, st) : ufv_launch_pp_shape_bf16(A, W, e, M, N, K, lda, ldw, F, shape, st);
        }
    }
    static const bool four_phase = getenv("UFV_GEMM_4PHASE") != nullptr;      // A/B switch (diagnostics): the first-generation schedule
    if (!streamk && four_phase) return launch_pp<F, S, Q, 4, 4, 2, false>(A, W, e, M, N, K, lda, ldw, st);
    // the SwiGLU form (the decoder's gate/up projection: 37888 columns, a ragged last row band at every prompt length) CAN deal its last round as half-tile items (HALF:
    // opt-in by UFV_GEMM_HALF=1; measured, not the default -- gemm256_kernel.h launch_pp)
    if (!streamk) return launch_pp<F, S, Q, 4, 4, 2, true, false, false, 0, S>(A, W, e, M, N, K, lda, ldw, st);
    if constexpr (S) {
        ufv_set_error("ufv_gemm: the stream-K split is not built for the SwiGLU epilogue (its tile counts are large anyway)");
        return UFV_EUNSUPPORTED;
    } else {
        UFV_ONCE_PER_DEVICE(
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_256<F, S, Q, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM256);
        );
        const int n_cu = pp_n_cu();
        const int tiles = cdiv(M, 256) * cdiv(N, 256);
        const int nk = K / (Q ? 128 : 64);
        if ((long long)tiles * nk < n_cu) return launch_pp<F, S, Q, 4, 4, 2, true>(A, W, e, M, N, K, lda, ldw, st);
        StreamK sk = {nullptr, nullptr, 0, pp_group(cdiv(M, 256)), 1, nullptr};
        const int rc = ufv_streamk_acquire(n_cu, &sk.ws, &sk.flags, &sk.epoch, &sk.err);     // one stream-K launch at a time per device (opt-in form)
        if (rc != UFV_OK) return rc;
        hipLaunchKernelGGL((gemm_nt_256<F, S, Q, true>), dim3(n_cu), dim3(512), SMEM256, st, A, W, e, M, N, K, lda, ldw, sk);
        UFV_CHECK_LAUNCH();
        return UFV_OK;
    }
}

template <bool Q>
static int launch256_q(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, bool out_f32, bool swiglu,
                       bool streamk, int shape, hipStream_t st) {
    if (out_f32) return swiglu ? launch256_t<true, true, Q>(A, W, e, M, N, K, lda, ldw, streamk, shape, st) : launch256_t<true, false, Q>(A, W, e, M, N, K, lda, ldw, streamk, shape, st);
    return swiglu ? launch256_t<false, true, Q>(A, W, e, M, N, K, lda, ldw, streamk, shape, st) : launch256_t<false, false, Q>(A, W, e, M, N, K, lda, ldw, streamk, shape, st);
}

int ufv_launch_gemm256(const void* A, const void* W, const Epi& e, int M, int N, int K, int lda, int ldw, bool out_f32,
                       bool swiglu, bool fp8, bool streamk, int shape, hipStream_t st) {
    return fp8 ? launch256_q<true>(A, W, e, M, N, K, lda, ldw, out_f32, swiglu, streamk, shape, st)
               : launch256_q<false>(A, W, e, M, N, K, lda, ldw, out_f32, swiglu, streamk, shape, st);
}
