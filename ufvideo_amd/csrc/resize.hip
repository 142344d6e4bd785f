// Device-side frame batching, resize step: Pillow's 8-bit bicubic resample -- what
// `Image.resize((S, S), resample=BICUBIC)` does under the HF SiglipImageProcessor the reference calls (mm_utils.py:269-295) --
// reproduced bit for bit: separable, horizontal pass first, each pass a fixed-point (2^22) dot product of <= ksize uint8 taps
// with host-precomputed integer coefficients, + 2^21, >> 22, clamped to [0, 255].  HBM-bound byte work, HWC uint8 frames.
#include "common.h"
#include "../../include/ufv.h"

namespace {

#define ST(s) reinterpret_cast<hipStream_t>(s)
constexpr int PREC = 22;

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PREC;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// out[t, y, xo, c] = clip8(2^21 + sum_i in[t, y, x0(xo) + i, c] * k[xo, i]);  one thread per output pixel (3 channels)
__global__ __launch_bounds__(256) void resize_h_k(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const int* __restrict__ bounds,
                                                  const int* __restrict__ kk, int ksize, int T, int H, int W, int Wo) {
    const int64_t total = (int64_t)T * H * Wo;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int xo = (int)(id % Wo);
        const int64_t row = id / Wo;                            // t * H + y
        const int x0 = bounds[2 * xo], n = bounds[2 * xo + 1];
        const uint8_t* p = in + (row * W + x0) * 3;
        const int* k = kk + xo * ksize;
        int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
        for (int i = 0; i < n; ++i) {
            const int c = k[i];
            s0 += p[3 * i] * c; s1 += p[3 * i + 1] * c; s2 += p[3 * i + 2] * c;
        }
        uint8_t* o = out + id * 3;
        o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
    }
}

// out[t, yo, x, c] = clip8(2^21 + sum_i in[t, y0(yo) + i, x, c] * k[yo, i]);  one thread per output byte (coalesced rows)
__global__ __launch_bounds__(256) void resize_v_k(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const int* __restrict__ bounds,
                                                  const int* __restrict__ kk, int ksize, int T, int H, int Ho, int W3) {
    const int64_t total = (int64_t)T * Ho * W3;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        const int xb = (int)(id % W3);
        const int yo = (int)((id / W3) % Ho);
        const int t = (int)(id / ((int64_t)W3 * Ho));
        const int y0 = bounds[2 * yo], n = bounds[2 * yo + 1];
        const uint8_t* p = in + ((int64_t)t * H + y0) * W3 + xb;
        const int* k = kk + yo * ksize;
        int s = 1 << (PREC - 1);
        for (int i = 0; i < n; ++i) s += p[(int64_t)i * W3] * k[i];
        out[id] = clip8(s);
    }
}

inline int grid_for(int64_t n) { return (int)((n + 255) / 256 > 1048576 ? 1048576 : (n + 255) / 256); }

}  // namespace

extern "C" int ufv_resize_bicubic_u8(const uint8_t* frames, uint8_t* tmp, uint8_t* out, int T, int H, int W, int Ho, int Wo,
                                     const int32_t* bounds_x, const int32_t* coeff_x, int ksize_x, const int32_t* bounds_y,
                                     const int32_t* coeff_y, int ksize_y, void* stream) {
    UFV_REQUIRE(frames && out && T > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "ufv_resize_bicubic_u8: bad arguments");
    const bool do_h = W != Wo, do_v = H != Ho;
    UFV_REQUIRE(!do_h || (bounds_x && coeff_x && ksize_x > 0), "ufv_resize_bicubic_u8: horizontal pass needs its coefficient tables");
    UFV_REQUIRE(!do_v || (bounds_y && coeff_y && ksize_y > 0), "ufv_resize_bicubic_u8: vertical pass needs its coefficient tables");
    UFV_REQUIRE(!(do_h && do_v) || tmp, "ufv_resize_bicubic_u8: a two-pass resize needs the [T, H, Wo, 3] scratch buffer");
    if (!do_h && !do_v) {
        if (hipMemcpyAsync(out, frames, (size_t)T * H * W * 3, hipMemcpyDeviceToDevice, ST(stream)) != hipSuccess) return UFV_EHIP;
        return UFV_OK;
    }
    const uint8_t* src = frames;
    if (do_h) {
        uint8_t* dst = do_v ? tmp : out;
        hipLaunchKernelGGL(resize_h_k, dim3(grid_for((int64_t)T * H * Wo)), dim3(256), 0, ST(stream), src, dst, bounds_x, coeff_x, ksize_x, T,
                           H, W, Wo);
        src = dst;
    }
    if (do_v)
        hipLaunchKernelGGL(resize_v_k, dim3(grid_for((int64_t)T * Ho * Wo * 3)), dim3(256), 0, ST(stream), src, out, bounds_y, coeff_y,
                           ksize_y, T, H, Ho, Wo * 3);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}
