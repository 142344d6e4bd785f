// fp8 (OCP e4m3) row quantisation for the W8A8 GEMM path (BASELINE config #5a):
//   scale[m] = max|x[m,:]| / 448,   q[m,k] = rne_e4m3(x[m,k] * (1 / scale[m])),   x ~ q * scale.
// (the reciprocal is one correctly-rounded division per row; per element a multiply, which the CPU restatement mirrors)
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
    const float inv = 1.0f / s;
    if (tid == 0) scale[m] = s;
    for (int k = tid * 4; k < K; k += 1024) {
        const f32x4 v = ld4<DT>(x, base + k);
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, w, true);
        *reinterpret_cast<int*>(q + (int64_t)m * ldq + k) = w;
    }
}

// K <= 8192, bf16 input: a wave takes R rows at once (R x CPL = 16 sixteen-byte loads per lane issued before any use) and
// keeps them in registers: one pass over HBM with enough loads in flight for short rows too.
template <int R, int CPL>
__global__ __launch_bounds__(256) void quantize_fp8_wave_k(const bf16* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q, int64_t ldq,
                                                           float* __restrict__ scale, int M, int K) {
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R, lane = threadIdx.x & 63;
    if (row0 >= M) return;
    const int nc = K >> 3;
    bf16x8 v[R][CPL];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bf16* xr = x + (int64_t)min(row0 + r, M - 1) * ldx;
#pragma unroll
        for (int i = 0; i < CPL; ++i)
            if (lane + 64 * i < nc) v[r][i] = *reinterpret_cast<const bf16x8*>(xr + 8 * (lane + 64 * i));
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (row0 + r >= M) break;
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i)
            if (lane + 64 * i < nc)
#pragma unroll
                for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf((float)v[r][i][j]));
        amax = wave_max(amax);
        const float s = amax > 0.f ? amax / E4M3_MAX : 1.0f;
        const float inv = 1.0f / s;
        if (lane == 0) scale[row0 + r] = s;
#pragma unroll
        for (int i = 0; i < CPL; ++i)
            if (lane + 64 * i < nc) {
                int w0 = 0, w1 = 0;
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[r][i][0] * inv, (float)v[r][i][1] * inv, w0, false);
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[r][i][2] * inv, (float)v[r][i][3] * inv, w0, true);
                w1 = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[r][i][4] * inv, (float)v[r][i][5] * inv, w1, false);
                w1 = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[r][i][6] * inv, (float)v[r][i][7] * inv, w1, true);
                int2 o; o.x = w0; o.y = w1;
                *reinterpret_cast<int2*>(q + (int64_t)(row0 + r) * ldq + 8 * (lane + 64 * i)) = o;
            }
    }
}

// LayerNorm / RMSNorm with the output quantised in the same pass: y = bf16(norm(x) * w (+ b)) exactly as the unfused norm
// kernel writes it, then the row scale and e4m3 codes exactly as quantize_fp8 derives them -> bit-identical to norm + quantize.
template <int XDT, bool RMS>
__global__ __launch_bounds__(256) void norm_fp8_k(const void* __restrict__ x, int ldx, uint8_t* __restrict__ q, int64_t ldq,
                                                  float* __restrict__ scale, const float* __restrict__ w, const float* __restrict__ b, int M,
                                                  int D, float eps) {
    constexpr int MAXV = 16;                         // D <= 4096
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int nv = D >> 2;
    f32x4 v[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nv) {
            v[i] = ld4<XDT>(x, (int64_t)row * ldx + 4 * (lane + 64 * i));
            if (RMS) sum += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
            else sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];            // layernorm_k's order: the fused form rounds exactly as norm + quantise
        }
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(wave_sum(sum) / D + eps);
    } else {
        mean = wave_sum(sum) / D;
        float qq = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (lane + 64 * i < nv)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[i][j] - mean;
                    qq += d * d;
                }
        rstd = rsqrtf(wave_sum(qq) / D + eps);
    }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nv) {
            const int c = 4 * (lane + 64 * i);
            const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
            const f32x4 bb = b ? *reinterpret_cast<const f32x4*>(b + c) : f32x4{0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = RMS ? v[i][j] * rstd * ww[j] : (v[i][j] - mean) * rstd * ww[j] + bb[j];
                v[i][j] = (float)(bf16)y;                       // the value the bf16 norm output would hold
                amax = fmaxf(amax, fabsf(v[i][j]));
            }
        }
    amax = wave_max(amax);
    const float s = amax > 0.f ? amax / E4M3_MAX : 1.0f;
    const float inv = 1.0f / s;
    if (lane == 0) scale[row] = s;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nv) {
            int w0 = 0;
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv, v[i][1] * inv, w0, false);
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv, v[i][3] * inv, w0, true);
            *reinterpret_cast<int*>(q + (int64_t)row * ldq + 4 * (lane + 64 * i)) = w0;
        }
}

// Pipelined form of norm_fp8_k<XDT, false> for long inputs (the ViT's 18 432 x 1152 stream under set_gemm_dtype("fp8")): persistent blocks, a wave walks rows
// g, g + G, ... and requests the next row before it reduces, normalises and quantises the current one (csrc/ops.hip layernorm_pipe_k); weight / bias staged once
// per block in LDS.  The same arithmetic in the same order per row: bit-identical.
template <int XDT, int NV>
__global__ __launch_bounds__(256) void layernorm_fp8_pipe_k(const void* __restrict__ x, int ldx, uint8_t* __restrict__ q, int64_t ldq, float* __restrict__ scale,
                                                            const float* __restrict__ w, const float* __restrict__ b, int M, int D, float eps) {
    extern __shared__ __attribute__((aligned(16))) float lnq_wb[];
    const int lane = threadIdx.x & 63, nv = D >> 2;
    for (int i = threadIdx.x; i < nv; i += 256) {
        reinterpret_cast<f32x4*>(lnq_wb)[i] = reinterpret_cast<const f32x4*>(w)[i];
        reinterpret_cast<f32x4*>(lnq_wb + D)[i] = b ? reinterpret_cast<const f32x4*>(b)[i] : f32x4{0, 0, 0, 0};
    }
    __syncthreads();
    const int G = gridDim.x * 4;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x4 v[NV], nx[NV];
    if (row < M) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < nv) v[i] = ld4<XDT>(x, (int64_t)row * ldx + 4 * (lane + 64 * i));
    }
    while (row < M) {
        const int nrow = row + G;
        if (nrow < M) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (lane + 64 * i < nv) nx[i] = ld4<XDT>(x, (int64_t)nrow * ldx + 4 * (lane + 64 * i));
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < nv) sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];            // layernorm_k's order: the fused form rounds exactly as norm + quantise
        const float mean = wave_sum(sum) / D;
        float qq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < nv)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[i][j] - mean;
                    qq += d * d;
                }
        const float rstd = rsqrtf(wave_sum(qq) / D + eps);
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < nv) {
                const int c = 4 * (lane + 64 * i);
                const f32x4 ww = *reinterpret_cast<const f32x4*>(lnq_wb + c);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(lnq_wb + D + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float y = (v[i][j] - mean) * rstd * ww[j] + bb[j];
                    v[i][j] = (float)(bf16)y;
                    amax = fmaxf(amax, fabsf(v[i][j]));
                }
            }
        amax = wave_max(amax);
        const float s = amax > 0.f ? amax / E4M3_MAX : 1.0f;
        const float inv = 1.0f / s;
        if (lane == 0) scale[row] = s;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < nv) {
                int w0 = 0;
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv, v[i][1] * inv, w0, false);
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv, v[i][3] * inv, w0, true);
                *reinterpret_cast<int*>(q + (int64_t)row * ldq + 4 * (lane + 64 * i)) = w0;
            }
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = nx[i];
        row = nrow;
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

// MX block quantisation: one thread per (row, 32-element block): 32 values -> amax -> scale byte -> 32 codes (8 dword stores)
template <int DT>
__global__ __launch_bounds__(256) void quantize_mx_k(const void* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q, int64_t ldq, uint8_t* __restrict__ bs,
                                                     int64_t ldb, int M, int nb) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)M * nb) return;
    const int m = (int)(id / nb), b = (int)(id % nb);
    f32x4 v[8];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = ld4<DT>(x, (int64_t)m * ldx + b * 32 + 4 * i);
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[i][0]), fabsf(v[i][1]))), fmaxf(fabsf(v[i][2]), fabsf(v[i][3])));
    }
    const unsigned e = mx_scale_byte(amax);
    const float inv = __builtin_bit_cast(float, (254u - e) << 23);
    bs[(int64_t)(m >> 6) * ldb + (int64_t)(b >> 4) * 1024 + (m & 63) * 16 + (b & 3) * 4 + ((b >> 2) & 3)] = (uint8_t)e;      // [M / 64][K / 512][64][16], byte 4 (block in K-tile) + (K-tile in group)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv, v[i][1] * inv, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv, v[i][3] * inv, w, true);
        *reinterpret_cast<int*>(q + (int64_t)m * ldq + b * 32 + 4 * i) = w;
    }
}

__global__ __launch_bounds__(256) void dequantize_mx_k(const uint8_t* __restrict__ q, int64_t ldq, const uint8_t* __restrict__ bs, int64_t ldb, float* __restrict__ out,
                                                       int64_t ldo, int M, int K) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)M * (K >> 2)) return;
    const int m = (int)(id / (K >> 2)), k = (int)(id % (K >> 2)) * 4;
    const int w = *reinterpret_cast<const int*>(q + (int64_t)m * ldq + k);
    const float sc = __builtin_bit_cast(float, (unsigned)bs[(int64_t)(m >> 6) * ldb + (int64_t)(k >> 9) * 1024 + (m & 63) * 16 + ((k >> 5) & 3) * 4 + ((k >> 7) & 3)] << 23);
    f32x4 o = {__builtin_amdgcn_cvt_f32_fp8(w, 0) * sc, __builtin_amdgcn_cvt_f32_fp8(w, 1) * sc, __builtin_amdgcn_cvt_f32_fp8(w, 2) * sc, __builtin_amdgcn_cvt_f32_fp8(w, 3) * sc};
    *reinterpret_cast<f32x4*>(out + (int64_t)m * ldo + k) = o;
}

}  // namespace

extern "C" int ufv_quantize_mx(const void* x, int x_dtype, int64_t ldx, void* q, int64_t ldq, void* bscale, int64_t ldb, int M, int K, void* stream) {
    UFV_REQUIRE(x && q && bscale && M > 0 && K > 0 && K % 128 == 0 && ldx % 4 == 0 && ldq % 4 == 0 && ldb % 1024 == 0 && ldb >= 1024 * ((K + 511) / 512) && (uintptr_t)q % 4 == 0 && (uintptr_t)x % 8 == 0,
                "ufv_quantize_mx: K %% 128 == 0, row pitches multiples of 4, ldb = bytes per 64-row block >= 1024 ceil(K / 512) (K=%d)", K);
    // the fp32 instantiation reads 16 bytes per lane (four floats), the bf16 one 8: an fp32 row view must start on a 16-byte boundary
    UFV_REQUIRE(x_dtype != UFV_DT_F32 || (uintptr_t)x % 16 == 0, "ufv_quantize_mx: an fp32 input must be 16-byte aligned (sliced views: pass a column offset that is a multiple of 4 floats)");
    const int nb = K / 32;
    const int64_t total = (int64_t)M * nb;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (x_dtype == UFV_DT_F32) hipLaunchKernelGGL((quantize_mx_k<UFV_DT_F32>), grid, dim3(256), 0, ST(stream), x, ldx, (uint8_t*)q, ldq, (uint8_t*)bscale, ldb, M, nb);
    else if (x_dtype == UFV_DT_BF16) hipLaunchKernelGGL((quantize_mx_k<UFV_DT_BF16>), grid, dim3(256), 0, ST(stream), x, ldx, (uint8_t*)q, ldq, (uint8_t*)bscale, ldb, M, nb);
    else { ufv_set_error("ufv_quantize_mx: unsupported dtype %d", x_dtype); return UFV_EINVAL; }
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_dequantize_mx(const void* q, int64_t ldq, const void* bscale, int64_t ldb, float* out, int64_t ldo, int M, int K, void* stream) {
    UFV_REQUIRE(q && bscale && out && M > 0 && K > 0 && K % 128 == 0 && ldq % 4 == 0 && ldo % 4 == 0 && (uintptr_t)q % 4 == 0 && (uintptr_t)out % 16 == 0, "ufv_dequantize_mx: bad arguments");
    UFV_REQUIRE(ldb % 1024 == 0 && ldb >= 1024 * ((K + 511) / 512), "ufv_dequantize_mx: ldb = bytes per 64-row block of scales >= 1024 ceil(K / 512) (K=%d, ldb=%lld)", K, (long long)ldb);
    const int64_t total = (int64_t)M * (K / 4);
    hipLaunchKernelGGL(dequantize_mx_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ST(stream), (const uint8_t*)q, ldq, (const uint8_t*)bscale, ldb, out, ldo, M, K);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_quantize_fp8(const void* x, int x_dtype, int64_t ldx, void* q, int64_t ldq, float* scale, int M, int K, void* stream) {
    UFV_REQUIRE(x && q && scale && M > 0 && K > 0, "ufv_quantize_fp8: bad arguments");
    UFV_REQUIRE(K % 4 == 0 && ldx % 4 == 0 && ldq % 4 == 0 && (uintptr_t)q % 4 == 0, "ufv_quantize_fp8: K and row pitches must be multiples of 4");
    if (x_dtype == UFV_DT_BF16 && K % 8 == 0 && K <= 8192 && ldx % 8 == 0 && ldq % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)q % 8 == 0)
    {
        if (K <= 2048) hipLaunchKernelGGL((quantize_fp8_wave_k<4, 4>), dim3((M + 15) / 16), dim3(256), 0, ST(stream), (const bf16*)x, ldx, (uint8_t*)q, ldq, scale, M, K);
        else if (K <= 4096) hipLaunchKernelGGL((quantize_fp8_wave_k<2, 8>), dim3((M + 7) / 8), dim3(256), 0, ST(stream), (const bf16*)x, ldx, (uint8_t*)q, ldq, scale, M, K);
        else hipLaunchKernelGGL((quantize_fp8_wave_k<1, 16>), dim3((M + 3) / 4), dim3(256), 0, ST(stream), (const bf16*)x, ldx, (uint8_t*)q, ldq, scale, M, K);
    }
    else if (x_dtype == UFV_DT_F32) hipLaunchKernelGGL((quantize_fp8_rows_k<UFV_DT_F32>), dim3(M), dim3(256), 0, ST(stream), x, ldx, (uint8_t*)q, ldq, scale, K);
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

extern "C" int ufv_layernorm_fp8(const void* x, int x_dtype, int ldx, void* q, int64_t ldq, float* scale, const float* w, const float* b,
                                 int M, int D, float eps, void* stream) {
    UFV_REQUIRE(x && q && scale && w && M > 0 && D > 0, "ufv_layernorm_fp8: bad arguments");
    UFV_REQUIRE(D % 4 == 0 && D <= 4096 && ldx % 4 == 0 && ldq % 4 == 0, "ufv_layernorm_fp8: D=%d must be a multiple of 4 and <= 4096", D);
    dim3 g((M + 3) / 4), blk(256);
    if (D <= 1280 && M >= 4096 && (x_dtype == UFV_DT_F32 || x_dtype == UFV_DT_BF16)) {          // long inputs: the pipelined persistent form (as ufv_layernorm)
        const int blocks = M / 16 < 1024 ? M / 16 : 1024;
        if (x_dtype == UFV_DT_F32)
            hipLaunchKernelGGL((layernorm_fp8_pipe_k<UFV_DT_F32, 5>), dim3(blocks), blk, 2 * D * sizeof(float), ST(stream), x, ldx, (uint8_t*)q, ldq, scale, w, b, M, D, eps);
        else
            hipLaunchKernelGGL((layernorm_fp8_pipe_k<UFV_DT_BF16, 5>), dim3(blocks), blk, 2 * D * sizeof(float), ST(stream), x, ldx, (uint8_t*)q, ldq, scale, w, b, M, D, eps);
        UFV_CHECK_LAUNCH();
        return UFV_OK;
    }
    if (x_dtype == UFV_DT_F32) hipLaunchKernelGGL((norm_fp8_k<UFV_DT_F32, false>), g, blk, 0, ST(stream), x, ldx, (uint8_t*)q, ldq, scale, w, b, M, D, eps);
    else if (x_dtype == UFV_DT_BF16) hipLaunchKernelGGL((norm_fp8_k<UFV_DT_BF16, false>), g, blk, 0, ST(stream), x, ldx, (uint8_t*)q, ldq, scale, w, b, M, D, eps);
    else { ufv_set_error("ufv_layernorm_fp8: unsupported input dtype %d", x_dtype); return UFV_EINVAL; }
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_rmsnorm_fp8(const float* x, int ldx, void* q, int64_t ldq, float* scale, const float* w, int M, int D, float eps,
                               void* stream) {
    UFV_REQUIRE(x && q && scale && w && M > 0 && D > 0, "ufv_rmsnorm_fp8: bad arguments");
    UFV_REQUIRE(D % 4 == 0 && D <= 4096 && ldx % 4 == 0 && ldq % 4 == 0, "ufv_rmsnorm_fp8: D=%d must be a multiple of 4 and <= 4096", D);
    hipLaunchKernelGGL((norm_fp8_k<UFV_DT_F32, true>), dim3((M + 3) / 4), dim3(256), 0, ST(stream), x, ldx, (uint8_t*)q, ldq, scale, w,
                       (const float*)nullptr, M, D, eps);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}
