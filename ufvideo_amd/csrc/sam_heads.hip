// Small kernels of the SAM2 prompt/mask heads (reference ufvideo/model/sam2.py: MaskDecoder.predict_masks :2094-2174,
// _forward_sam_heads :3276-3452, _get_orig_video_res_output :3770-3790).  All HBM-bound, token-major (NHWC) layouts.
#include "common.h"
#include "../../include/ufv.h"

namespace {

#define ST(s) reinterpret_cast<hipStream_t>(s)

template <int DT> __device__ __forceinline__ f32x4 load4(const void* p, int64_t i);
template <> __device__ __forceinline__ f32x4 load4<UFV_DT_F32>(const void* p, int64_t i) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + i);
}
template <> __device__ __forceinline__ f32x4 load4<UFV_DT_BF16>(const void* p, int64_t i) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p) + i);
    return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <int DT> __device__ __forceinline__ void store4(void* p, int64_t i, f32x4 v);
template <> __device__ __forceinline__ void store4<UFV_DT_F32>(void* p, int64_t i, f32x4 v) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p) + i) = v;
}
template <> __device__ __forceinline__ void store4<UFV_DT_BF16>(void* p, int64_t i, f32x4 v) {
    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p) + i) = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
}

// out[m, :] = a[m, :] + b[m % b_rows, :]   (4 channels per thread)
template <int ADT, int ODT>
__global__ __launch_bounds__(256) void add_bcast_k(const void* a, int64_t lda, const float* b, int64_t ldb, int b_rows, void* out,
                                                   int64_t ldo, int64_t M, int C4) {
    const int64_t total = M * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / C4;
        const int c = (int)(i - m * C4) * 4;
        f32x4 v = load4<ADT>(a, m * lda + c);
        if (b) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(b + (m % b_rows) * ldb + c);
            v += w;
        }
        store4<ODT>(out, m * ldo + c, v);
    }
}

// masks[b,i,Y,X] = sum_c hyper[b,i,c] * gelu(up2[b, Y/2, X/2, ((Y&1)*2 + (X&1))*C8 + c] + s0[b,Y,X,c]):
// the pixel shuffle of the second ConvTranspose2d(k2,s2) (run as a GEMM with the 4 taps side by side in the row), the
// high-res skip, the GELU and the hypernetwork dot products in one pass; one thread per output pixel.
template <int C8, int NM>
__global__ __launch_bounds__(256) void sam_mask_head_k(const bf16* __restrict__ up2, int64_t ld_up, const bf16* __restrict__ s0,
                                                       int64_t ld_s0, const float* __restrict__ hyper, float* __restrict__ out,
                                                       int h, int w) {
    __shared__ float hy[NM * C8];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < NM * C8; i += 256) hy[i] = hyper[(int64_t)b * NM * C8 + i];
    __syncthreads();
    const int H = 2 * h, W = 2 * w;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int Y = p / W, X = p - Y * W;
    const bf16* u = up2 + ((int64_t)b * h * w + (int64_t)(Y >> 1) * w + (X >> 1)) * ld_up + ((Y & 1) * 2 + (X & 1)) * C8;
    const bf16* s = s0 + ((int64_t)b * H * W + p) * ld_s0;
    float acc[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i] = 0.f;
#pragma unroll
    for (int c = 0; c < C8; c += 8) {
        const bf16x8 uv = *reinterpret_cast<const bf16x8*>(u + c);
        const bf16x8 sv = *reinterpret_cast<const bf16x8*>(s + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float g = act_apply_t<ACT_GELU_ERF>((float)uv[j] + (float)sv[j]);
#pragma unroll
            for (int i = 0; i < NM; ++i) acc[i] += hy[i * C8 + c + j] * g;
        }
    }
#pragma unroll
    for (int i = 0; i < NM; ++i) out[((int64_t)b * NM + i) * H * W + p] = acc[i];
}

// torch upsample_bilinear2d, align_corners=False: src = max(0, (dst + 0.5) * in/out - 0.5)
__global__ __launch_bounds__(256) void resize_bilinear_k(const float* __restrict__ src, const int32_t* __restrict__ sel, int planes_per,
                                                         int sel_off, float* __restrict__ dst, int N, int Hs, int Ws, int Hd, int Wd) {
    const int64_t total = (int64_t)N * Hd * Wd;
    const float sy = (float)Hs / (float)Hd, sx = (float)Ws / (float)Wd;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wd), y = (int)((i / Wd) % Hd), n = (int)(i / ((int64_t)Wd * Hd));
        const float fy = fmaxf(sy * (y + 0.5f) - 0.5f, 0.f), fx = fmaxf(sx * (x + 0.5f) - 0.5f, 0.f);
        const int y0 = min((int)fy, Hs - 1), x0 = min((int)fx, Ws - 1);
        const int y1 = y0 + (y0 < Hs - 1), x1 = x0 + (x0 < Ws - 1);
        const float ly = fy - y0, lx = fx - x0;
        const int64_t plane = sel ? (int64_t)n * planes_per + sel_off + sel[n] : n;
        const float* s = src + plane * Hs * Ws;
        const float v = (1.f - ly) * ((1.f - lx) * s[(int64_t)y0 * Ws + x0] + lx * s[(int64_t)y0 * Ws + x1]) +
                        ly * ((1.f - lx) * s[(int64_t)y1 * Ws + x0] + lx * s[(int64_t)y1 * Ws + x1]);
        dst[i] = v;
    }
}

// row-wise argmax over a few columns (torch.argmax tie-breaking: lowest index)
__global__ void argmax_rows_k(const float* __restrict__ x, int64_t ld, int M, int N, int32_t* __restrict__ out) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float* r = x + (int64_t)m * ld;
    int bi = 0;
    float bv = r[0];
    for (int j = 1; j < N; ++j)
        if (r[j] > bv) { bv = r[j]; bi = j; }
    out[m] = bi;
}

inline int grid_for(int64_t n) { return (int)((n + 255) / 256 > 262144 ? 262144 : (n + 255) / 256); }

}  // namespace

extern "C" int ufv_add_bcast(const void* a, int a_dtype, int64_t lda, const float* b, int64_t ldb, int b_rows, void* out, int out_dtype,
                             int64_t ldo, int64_t M, int C, void* stream) {
    if (M == 0) return UFV_OK;
    UFV_REQUIRE(a && out && M > 0 && C > 0 && C % 4 == 0 && lda % 4 == 0 && ldo % 4 == 0 && (!b || (b_rows > 0 && ldb % 4 == 0)),
                "ufv_add_bcast: bad arguments (C and row pitches must be multiples of 4)");
    dim3 g(grid_for(M * (C / 4))), blk(256);
#define AB(AD, OD) hipLaunchKernelGGL((add_bcast_k<AD, OD>), g, blk, 0, ST(stream), a, lda, b, ldb, b_rows, out, ldo, M, C / 4)
    if (a_dtype == UFV_DT_F32 && out_dtype == UFV_DT_F32) AB(UFV_DT_F32, UFV_DT_F32);
    else if (a_dtype == UFV_DT_F32 && out_dtype == UFV_DT_BF16) AB(UFV_DT_F32, UFV_DT_BF16);
    else if (a_dtype == UFV_DT_BF16 && out_dtype == UFV_DT_F32) AB(UFV_DT_BF16, UFV_DT_F32);
    else if (a_dtype == UFV_DT_BF16 && out_dtype == UFV_DT_BF16) AB(UFV_DT_BF16, UFV_DT_BF16);
    else { ufv_set_error("ufv_add_bcast: unsupported dtypes %d -> %d", a_dtype, out_dtype); return UFV_EINVAL; }
#undef AB
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_sam_mask_head(const void* up2, int64_t ld_up, const void* s0, int64_t ld_s0, const float* hyper, float* out, int B,
                                 int h, int w, int C8, int nm, void* stream) {
    UFV_REQUIRE(up2 && s0 && hyper && out && B > 0 && h > 0 && w > 0, "ufv_sam_mask_head: bad arguments");
    UFV_REQUIRE(C8 == 32 && nm == 4, "ufv_sam_mask_head: built for SAM2's 32 upscaled channels x 4 mask tokens (got %d x %d)", C8, nm);
    UFV_REQUIRE(ld_up % 8 == 0 && ld_s0 % 8 == 0 && (uintptr_t)up2 % 16 == 0 && (uintptr_t)s0 % 16 == 0,
                "ufv_sam_mask_head: rows must be 16-byte aligned");
    hipLaunchKernelGGL((sam_mask_head_k<32, 4>), dim3(cdiv(4 * h * w, 256), B), dim3(256), 0, ST(stream), (const bf16*)up2, ld_up,
                       (const bf16*)s0, ld_s0, hyper, out, h, w);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_resize_bilinear(const float* src, const int32_t* sel, int planes_per, int sel_off, float* dst, int N, int Hs, int Ws,
                                   int Hd, int Wd, void* stream) {
    UFV_REQUIRE(src && dst && N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "ufv_resize_bilinear: bad arguments");
    hipLaunchKernelGGL(resize_bilinear_k, dim3(grid_for((int64_t)N * Hd * Wd)), dim3(256), 0, ST(stream), src, sel, planes_per, sel_off,
                       dst, N, Hs, Ws, Hd, Wd);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_argmax_rows(const float* x, int64_t ld, int M, int N, int32_t* out, void* stream) {
    UFV_REQUIRE(x && out && M > 0 && N > 0, "ufv_argmax_rows: bad arguments");
    hipLaunchKernelGGL(argmax_rows_k, dim3(cdiv(M, 64)), dim3(64), 0, ST(stream), x, ld, M, N, out);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}
