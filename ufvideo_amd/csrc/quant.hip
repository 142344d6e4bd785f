// fp8 (OCP e4m3) row quantisation for the W8A8 GEMM path (BASELINE config #5a):
//   scale[m] = max|x[m,:]| / 448,   q[m,k] = rne_e4m3(x[m,k] / scale[m]),   x ~ q * scale.
// Used for activations (per token, on the fly) and for weights (per output channel, once at pack time).
#include "common.h"
#include "../../include/ufv.h"

namespace {

#define ST(s) reinterpret_cast<hipStream_t>(s)
constexpr float E4M3_MAX = 448.0f;

template <int DT> __device__ __forceinline__ f32x4 ld4(const void* p, int64_t i);
template <> __device__ __forceinline__ f32x4 ld4<UFV_DT_F32>(const void* p, int64_t i) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + i);
}
template <> __device__ __forceinline__ f32x4 ld4<UFV_DT_BF16>(const void* p, int64_t i) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p) + i);
    return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

// one block per row; two passes over the row (the second one hits L2)
template <int DT>
__global__ __launch_bounds__(256) void quantize_fp8_rows_k(const void* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q, int64_t ldq,
                                                           float* __restrict__ scale, int K) {
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const int64_t base = (int64_t)m * ldx;
    float amax = 0.f;
    for (int k = tid * 4; k < K; k += 1024) {
        const f32x4 v = ld4<DT>(x, base + k);
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    amax = wave_max(amax);
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float s = amax > 0.f ? amax / E4M3_MAX : 1.0f;
    if (tid == 0) scale[m] = s;
    for (int k = tid * 4; k < K; k += 1024) {
        const f32x4 v = ld4<DT>(x, base + k);
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] / s, v[1] / s, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] / s, v[3] / s, w, true);
        *reinterpret_cast<int*>(q + (int64_t)m * ldq + k) = w;
    }
}

__global__ void dequantize_fp8_k(const uint8_t* __restrict__ q, int64_t ldq, const float* __restrict__ scale, float* __restrict__ out,
                                 int64_t ldo, int M, int K) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * K) return;
    const int m = (int)(i / K), k = (int)(i - (int64_t)m * K);
    const int byte = q[(int64_t)m * ldq + k];
    out[(int64_t)m * ldo + k] = __builtin_amdgcn_cvt_f32_fp8(byte, 0) * scale[m];
}

}  // namespace

extern "C" int ufv_quantize_fp8(const void* x, int x_dtype, int64_t ldx, void* q, int64_t ldq, float* scale, int M, int K, void* stream) {
    UFV_REQUIRE(x && q && scale && M > 0 && K > 0, "ufv_quantize_fp8: bad arguments");
    UFV_REQUIRE(K % 4 == 0 && ldx % 4 == 0 && ldq % 4 == 0 && (uintptr_t)q % 4 == 0, "ufv_quantize_fp8: K and row pitches must be multiples of 4");
    if (x_dtype == UFV_DT_F32) hipLaunchKernelGGL((quantize_fp8_rows_k<UFV_DT_F32>), dim3(M), dim3(256), 0, ST(stream), x, ldx, (uint8_t*)q, ldq, scale, K);
    else if (x_dtype == UFV_DT_BF16) hipLaunchKernelGGL((quantize_fp8_rows_k<UFV_DT_BF16>), dim3(M), dim3(256), 0, ST(stream), x, ldx, (uint8_t*)q, ldq, scale, K);
    else { ufv_set_error("ufv_quantize_fp8: unsupported dtype %d", x_dtype); return UFV_EINVAL; }
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_dequantize_fp8(const void* q, int64_t ldq, const float* scale, float* out, int64_t ldo, int M, int K, void* stream) {
    UFV_REQUIRE(q && scale && out && M > 0 && K > 0, "ufv_dequantize_fp8: bad arguments");
    const int64_t n = (int64_t)M * K;
    hipLaunchKernelGGL(dequantize_fp8_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ST(stream), (const uint8_t*)q, ldq, scale, out, ldo, M, K);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}
