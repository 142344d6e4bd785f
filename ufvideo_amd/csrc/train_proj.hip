// Backward kernels of the multimodal projector (STC connector: timm RegStage bottlenecks + Conv3d sampler + readout MLP;
// reference ufvideo/model/projector.py:133-238, differentiated by torch autograd in the reference's training step).
// Token-major bf16 activations [pixels, C] as in the forward path; the 1x1 convolutions / Conv3d / Linear layers go through
// the NT GEMMs (ufvideo_amd/train_projector.py), this file holds the rest.  All HBM-bound.
#include "common.h"
#include "../../include/ufv.h"

namespace {

#define ST(s) reinterpret_cast<hipStream_t>(s)

inline int grid_for(int64_t n, int per_block = 256) {
    const int64_t g = (n + per_block - 1) / per_block;
    return (int)(g < 16384 ? (g > 0 ? g : 1) : 16384);
}

__device__ __forceinline__ float act_grad(float x, int act) {
    switch (act) {
        case ACT_SILU: { const float s = sigmoid_fast(x); return s * (1.0f + x * (1.0f - s)); }
        case ACT_SIGMOID: { const float s = sigmoid_fast(x); return s * (1.0f - s); }
        case ACT_GELU_ERF: return 0.5f * (1.0f + erff(x * 0.7071067811865476f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
        case ACT_RELU: return x > 0.f ? 1.f : 0.f;
        default: return 1.f;
    }
}

// out = act(pre)  /  dpre = dout * act'(pre), 8 elements per thread
template <bool BWD>
__global__ __launch_bounds__(256) void act_k(const bf16* __restrict__ pre, const bf16* __restrict__ dout, bf16* __restrict__ out, int64_t n8,
                                             int act) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const bf16x8 p = reinterpret_cast<const bf16x8*>(pre)[i];
        bf16x8 o;
        if (BWD) {
            const bf16x8 d = reinterpret_cast<const bf16x8*>(dout)[i];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)d[j] * act_grad((float)p[j], act));
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)act_apply((float)p[j], act);
        }
        reinterpret_cast<bf16x8*>(out)[i] = o;
    }
}

__global__ __launch_bounds__(256) void add_bf16_k(const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ out, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const bf16x8 x = reinterpret_cast<const bf16x8*>(a)[i], y = reinterpret_cast<const bf16x8*>(b)[i];
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)x[j] + (float)y[j]);
        reinterpret_cast<bf16x8*>(out)[i] = o;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Row LayerNorm (+ optional activation) backward, one wave per row, grid-stride over rows.
//   xh = (x - mean) * rstd;  y = xh * w + b;  out = act(y)
//   g = dout * act'(y);  dw += sum_rows g * xh;  db += sum_rows g;  dx = rstd * (g*w - mean(g*w) - xh * mean(g*w*xh))
// The row and its incoming gradient are loaded once and stay in registers as bf16 (64 VGPRs at C = 4096) beside the per-wave
// dw / db partials (128 VGPRs), which are written once per wave and reduced in two ordered stages (deterministic, no atomics).
// Used for narrow rows (C <= 1024) and the activations the block form is not instantiated for.
// ---------------------------------------------------------------------------------------------------------
template <int MAXI>
__global__ __launch_bounds__(256) void layernorm_bwd_k(const bf16* __restrict__ x, int64_t ldx, const float* __restrict__ w, const float* __restrict__ b,
                                                       const bf16* __restrict__ dout, int64_t ldd, bf16* __restrict__ dx, int64_t lddx,
                                                       float* __restrict__ part, int M, int C, float eps, int act) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int nch = C >> 3;
    float aw[MAXI][8], ab[MAXI][8];
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { aw[i][j] = 0.f; ab[i][j] = 0.f; }
    for (int row = wid; row < M; row += nw) {
        const bf16* xr = x + (int64_t)row * ldx;
        const bf16* dr = dout + (int64_t)row * ldd;
        bf16x8 xv[MAXI], dv[MAXI];                                   // the row and its incoming gradient stay in registers (bf16)
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                xv[i] = *reinterpret_cast<const bf16x8*>(xr + ch * 8);
                dv[i] = *reinterpret_cast<const bf16x8*>(dr + ch * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) s += (float)xv[i][j];
            }
        }
        const float mean = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
            if (lane + 64 * i < nch) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = (float)xv[i][j] - mean; q += d * d; }
            }
        const float rstd = rsqrtf(wave_sum(q) / C + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + ch * 8), w1 = *reinterpret_cast<const f32x4*>(w + ch * 8 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(b + ch * 8), b1 = *reinterpret_cast<const f32x4*>(b + ch * 8 + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float ww = j < 4 ? w0[j & 3] : w1[j & 3], bb = j < 4 ? b0[j & 3] : b1[j & 3];
                    const float xh = ((float)xv[i][j] - mean) * rstd;
                    const float g = (float)dv[i][j] * (act ? act_grad(xh * ww + bb, act) : 1.f);
                    aw[i][j] += g * xh; ab[i][j] += g;
                    s1 += g * ww; s2 += g * ww * xh;
                    dv[i][j] = (bf16)g;                               // g replaces dout (bf16-rounded; dx below uses the same value)
                }
            }
        }
        s1 = wave_sum(s1) / C; s2 = wave_sum(s2) / C;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + ch * 8), w1 = *reinterpret_cast<const f32x4*>(w + ch * 8 + 4);
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float ww = j < 4 ? w0[j & 3] : w1[j & 3];
                    const float xh = ((float)xv[i][j] - mean) * rstd;
                    o[j] = (bf16)(rstd * ((float)dv[i][j] * ww - s1 - xh * s2));
                }
                *reinterpret_cast<bf16x8*>(dx + (int64_t)row * lddx + ch * 8) = o;
            }
        }
    }
    float* pw = part + (int64_t)wid * 2 * C;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { pw[ch * 8 + j] = aw[i][j]; pw[C + ch * 8 + j] = ab[i][j]; }
        }
    }
}

// Wide rows (C > 1024, the connector's 3584-channel stages: 18432 x 3584 per tensor): one 256-thread block per row.  A thread owns
// the same MAXI channel chunks in every row, so the affine parameters and its dw / db partials live in registers (48 VGPRs against
// the 190 of the wave-per-row form, which left 2 waves per SIMD to hide three dependent row reductions); the reductions go through
// LDS slots that alternate between consecutive rows (one barrier each).  One partial row per block.
template <int MAXI, int ACT>
__global__ __launch_bounds__(256) void layernorm_bwd_blk_k(const bf16* __restrict__ x, int64_t ldx, const float* __restrict__ w, const float* __restrict__ b,
                                                           const bf16* __restrict__ dout, int64_t ldd, bf16* __restrict__ dx, int64_t lddx,
                                                           float* __restrict__ part, int M, int C, float eps) {
    __shared__ float red[2][4][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nch = C >> 3;
    float aw[MAXI][8], ab[MAXI][8], ww[MAXI][8], bb[MAXI][8];
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int ch = tid + 256 * i;
#pragma unroll
        for (int j = 0; j < 8; ++j) { aw[i][j] = 0.f; ab[i][j] = 0.f; ww[i][j] = 0.f; bb[i][j] = 0.f; }
        if (ch < nch) {
            *reinterpret_cast<f32x4*>(ww[i]) = *reinterpret_cast<const f32x4*>(w + ch * 8); *reinterpret_cast<f32x4*>(ww[i] + 4) = *reinterpret_cast<const f32x4*>(w + ch * 8 + 4);
            *reinterpret_cast<f32x4*>(bb[i]) = *reinterpret_cast<const f32x4*>(b + ch * 8); *reinterpret_cast<f32x4*>(bb[i] + 4) = *reinterpret_cast<const f32x4*>(b + ch * 8 + 4);
        }
    }
    auto block_sum1 = [&](float v, int buf, int slot) {
        v = wave_sum(v);
        if (lane == 0) red[buf][slot][wv] = v;
        __syncthreads();
        return (red[buf][slot][0] + red[buf][slot][1]) + (red[buf][slot][2] + red[buf][slot][3]);
    };
    int it = 0;
    for (int row = blockIdx.x; row < M; row += gridDim.x, it ^= 1) {
        const bf16* xr = x + (int64_t)row * ldx;
        const bf16* dr = dout + (int64_t)row * ldd;
        bf16x8 xv[MAXI], dv[MAXI];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int ch = tid + 256 * i;
            if (ch < nch) {
                xv[i] = *reinterpret_cast<const bf16x8*>(xr + ch * 8);
                dv[i] = *reinterpret_cast<const bf16x8*>(dr + ch * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) s += (float)xv[i][j];
            }
        }
        const float mean = block_sum1(s, it, 0) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
            if (tid + 256 * i < nch) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = (float)xv[i][j] - mean; q += d * d; }
            }
        const float rstd = rsqrtf(block_sum1(q, it, 1) / C + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
            if (tid + 256 * i < nch) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = ((float)xv[i][j] - mean) * rstd;
                    const float g = (float)dv[i][j] * (ACT ? act_grad(xh * ww[i][j] + bb[i][j], ACT) : 1.f);
                    aw[i][j] += g * xh; ab[i][j] += g;
                    s1 += g * ww[i][j]; s2 += g * ww[i][j] * xh;
                    dv[i][j] = (bf16)g;                               // g replaces dout (bf16-rounded; dx below uses the same value)
                }
            }
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (lane == 0) { red[it][2][wv] = s1; red[it][3][wv] = s2; }
        __syncthreads();
        s1 = ((red[it][2][0] + red[it][2][1]) + (red[it][2][2] + red[it][2][3])) / C;
        s2 = ((red[it][3][0] + red[it][3][1]) + (red[it][3][2] + red[it][3][3])) / C;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int ch = tid + 256 * i;
            if (ch < nch) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = ((float)xv[i][j] - mean) * rstd;
                    o[j] = (bf16)(rstd * ((float)dv[i][j] * ww[i][j] - s1 - xh * s2));
                }
                *reinterpret_cast<bf16x8*>(dx + (int64_t)row * lddx + ch * 8) = o;
            }
        }
    }
    float* pw = part + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int ch = tid + 256 * i;
        if (ch < nch) {
            *reinterpret_cast<f32x4*>(pw + ch * 8) = *reinterpret_cast<const f32x4*>(aw[i]); *reinterpret_cast<f32x4*>(pw + ch * 8 + 4) = *reinterpret_cast<const f32x4*>(aw[i] + 4);
            *reinterpret_cast<f32x4*>(pw + C + ch * 8) = *reinterpret_cast<const f32x4*>(ab[i]); *reinterpret_cast<f32x4*>(pw + C + ch * 8 + 4) = *reinterpret_cast<const f32x4*>(ab[i] + 4);
        }
    }
}

// stage 1: tmp[slice][c] = sum of the slice's rows of the [R, 2C] partial image;  stage 2 (gridDim.y == 1, out0/out1 given):
// adds the slices into dw (columns < C) and db (columns >= C)
__global__ __launch_bounds__(256) void colsum2_f32_k(const float* __restrict__ x, int64_t ld, int R, int C, float* __restrict__ tmp,
                                                     float* __restrict__ out0, float* __restrict__ out1) {
    const int c = blockIdx.x * 256 + threadIdx.x;          // column of the [R, 2C] image
    if (c >= 2 * C) return;
    const int per = (R + gridDim.y - 1) / gridDim.y, r0 = blockIdx.y * per, r1 = min(R, r0 + per);
    float s0 = 0.f, s1 = 0.f;
    int r = r0;
    for (; r + 1 < r1; r += 2) { s0 += x[(int64_t)r * ld + c]; s1 += x[(int64_t)(r + 1) * ld + c]; }
    if (r < r1) s0 += x[(int64_t)r * ld + c];
    const float s = s0 + s1;
    if (tmp) tmp[(int64_t)blockIdx.y * 2 * C + c] = s;
    else if (c < C) out0[c] += s;
    else out1[c - C] += s;
}

// g = dout * silu'(LN_a(z) + (LN_b(s) | s))   (the gradient entering both branches of a bottleneck's output); one wave per row
template <int MAXI>
__global__ __launch_bounds__(256) void ln_add_silu_g_k(const bf16* __restrict__ z, const float* __restrict__ wa, const float* __restrict__ ba,
                                                       const bf16* __restrict__ s, const float* __restrict__ wb, const float* __restrict__ bb,
                                                       const bf16* __restrict__ dout, bf16* __restrict__ g, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nch = C >> 3;
    const bf16* zr = z + (int64_t)row * C;
    const bf16* sr = s + (int64_t)row * C;
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(zr + ch * 8), u = *reinterpret_cast<const bf16x8*>(sr + ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { sa += (float)v[j]; sb += (float)u[j]; }
        }
    }
    const float ma = wave_sum(sa) / C, mb = wave_sum(sb) / C;
    float qa = 0.f, qb = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(zr + ch * 8), u = *reinterpret_cast<const bf16x8*>(sr + ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float da = (float)v[j] - ma, db = (float)u[j] - mb;
                qa += da * da; qb += db * db;
            }
        }
    }
    const float ra = rsqrtf(wave_sum(qa) / C + eps), rb = rsqrtf(wave_sum(qb) / C + eps);
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(zr + ch * 8), u = *reinterpret_cast<const bf16x8*>(sr + ch * 8);
            const bf16x8 d = *reinterpret_cast<const bf16x8*>(dout + (int64_t)row * C + ch * 8);
            bf16x8 o;
            float pw[8], pb[8], qw[8], qb[8];                          // affine parameters as 16-byte loads (per-element loads are 32-byte-strided gathers)
            *reinterpret_cast<f32x4*>(pw) = *reinterpret_cast<const f32x4*>(wa + ch * 8); *reinterpret_cast<f32x4*>(pw + 4) = *reinterpret_cast<const f32x4*>(wa + ch * 8 + 4);
            *reinterpret_cast<f32x4*>(pb) = *reinterpret_cast<const f32x4*>(ba + ch * 8); *reinterpret_cast<f32x4*>(pb + 4) = *reinterpret_cast<const f32x4*>(ba + ch * 8 + 4);
            if (wb) {
                *reinterpret_cast<f32x4*>(qw) = *reinterpret_cast<const f32x4*>(wb + ch * 8); *reinterpret_cast<f32x4*>(qw + 4) = *reinterpret_cast<const f32x4*>(wb + ch * 8 + 4);
                *reinterpret_cast<f32x4*>(qb) = *reinterpret_cast<const f32x4*>(bb + ch * 8); *reinterpret_cast<f32x4*>(qb + 4) = *reinterpret_cast<const f32x4*>(bb + ch * 8 + 4);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = ((float)v[j] - ma) * ra * pw[j] + pb[j];
                t += wb ? ((float)u[j] - mb) * rb * qw[j] + qb[j] : (float)u[j];
                o[j] = (bf16)((float)d[j] * act_grad(t, ACT_SILU));
            }
            *reinterpret_cast<bf16x8*>(g + (int64_t)row * C + ch * 8) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Depthwise 3x3 convolution, padding 1, NHWC token-major: y[f,h,w,c] = sum_{dy,dx} x[f,h+dy-1,w+dx-1,c] * w9[dy*3+dx][c].
// flip = 1 uses tap (2-dy, 2-dx): the gradient with respect to the input.  One thread per (pixel, 8 channels).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv3x3_k(const bf16* __restrict__ x, bf16* __restrict__ y, const float* __restrict__ w9, int F, int H,
                                                   int W, int C, int flip) {
    const int cv = C >> 3;
    const int total = F * H * W * cv;                            // < 2^31 (checked by the launcher): 32-bit index arithmetic
    for (int id = blockIdx.x * 256 + threadIdx.x; id < total; id += gridDim.x * 256) {
        const int c8 = id % cv;
        int p = id / cv;
        const int px = p % W; p /= W;
        const int py = p % H;
        const int f = p / H;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = py + dy - 1;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = px + dx - 1;
                if (xx < 0 || xx >= W) continue;
                const int tap = flip ? (2 - dy) * 3 + (2 - dx) : dy * 3 + dx;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (((int64_t)f * H + yy) * W + xx) * C + c8 * 8);
                const float* wk = w9 + tap * C + c8 * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)v[j] * wk[j];
            }
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)acc[j];
        *reinterpret_cast<bf16x8*>(y + (int64_t)id * 8) = o;
    }
}

// dw9[tap][c] partials: a block owns 32 channel chunks of 8 (16-byte loads) and one slice of the pixels; its 8 pixel lanes stride
// through the slice, and their 9 x 8 sums per chunk are added in lane order through LDS (deterministic).  The first version
// (one thread per channel, 2-byte loads, 320 blocks) took 1.06 ms on the 18432 x 1152 stage.
__global__ __launch_bounds__(256) void dwconv3x3_dw_k(const bf16* __restrict__ x, const bf16* __restrict__ dy, float* __restrict__ part, int F, int H,
                                                      int W, int C) {
    __shared__ float red[8][32][9];
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int cv = C >> 3, c8 = blockIdx.x * 32 + cl;
    const int nslice = gridDim.y;
    const int64_t NP = (int64_t)F * H * W, per = (NP + nslice - 1) / nslice;
    const int64_t p0 = (int64_t)blockIdx.y * per, p1 = min(NP, p0 + per);
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    if (c8 < cv) {
        for (int64_t p = p0 + pl; p < p1; p += 8) {
            const int px = p % W, py = (p / W) % H;
            const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + p * C + c8 * 8);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (p + (int64_t)(t / 3 - 1) * W + (t % 3 - 1)) * C + c8 * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[t][j] += (float)g[j] * (float)v[j];
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[pl][cl][j] = acc[t][j];
        __syncthreads();
        if (c8 < cv) {                                               // thread (cl, pl) adds channel pl of chunk cl over the 8 pixel lanes
            float sum = red[0][cl][pl];
#pragma unroll
            for (int k = 1; k < 8; ++k) sum += red[k][cl][pl];
            part[((int64_t)blockIdx.y * 9 + t) * C + c8 * 8 + pl] = sum;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void dw_reduce_k(const float* __restrict__ part, int nslice, int n, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < nslice; ++k) s += part[(int64_t)k * n + i];
    out[i] += s;
}

// out[f][c] = sum_p a[f,p,c] * b[f,p,c]   (gradient of the SE gate): block = 32 channel chunks of 8 x 8 pixel lanes, lanes added in
// order through LDS (one thread per channel with 2-byte loads took 175 us on the 32 x 576 x 1152 stage)
__global__ __launch_bounds__(256) void prod_colsum_k(const bf16* __restrict__ a, const bf16* __restrict__ b, int F, int P, int C, float* __restrict__ out) {
    __shared__ float red[8][32][9];
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int cv = C >> 3, c8 = blockIdx.x * 32 + cl, f = blockIdx.y;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c8 < cv) {
        const bf16* ap = a + (int64_t)f * P * C + c8 * 8;
        const bf16* bp = b + (int64_t)f * P * C + c8 * 8;
        for (int p = pl; p < P; p += 8) {
            const bf16x8 u = *reinterpret_cast<const bf16x8*>(ap + (int64_t)p * C), v = *reinterpret_cast<const bf16x8*>(bp + (int64_t)p * C);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)u[j] * (float)v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[pl][cl][j] = acc[j];
    __syncthreads();
    if (c8 < cv) {
        float sum = red[0][cl][pl];
#pragma unroll
        for (int k = 1; k < 8; ++k) sum += red[k][cl][pl];
        out[(int64_t)f * C + c8 * 8 + pl] = sum;
    }
}

// out[f,p,c] = a[f,p,c] * g[f,c] + s[f,c] * k      (da2 = dy3 * gate + d(colmean) / P)
__global__ __launch_bounds__(256) void scale_add_bcast_k(const bf16* __restrict__ a, const bf16* __restrict__ g, const float* __restrict__ s, float k,
                                                         bf16* __restrict__ out, int F, int P, int C) {
    const int cv = C >> 3;
    const int total = F * P * cv;                                // < 2^31 (checked by the launcher)
    for (int id = blockIdx.x * 256 + threadIdx.x; id < total; id += gridDim.x * 256) {
        const int c8 = id % cv;
        const int f = id / (P * cv);
        const bf16x8 v = reinterpret_cast<const bf16x8*>(a)[id];
        const bf16x8 gv = *reinterpret_cast<const bf16x8*>(g + (int64_t)f * C + c8 * 8);
        float sv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (s) {
            *reinterpret_cast<f32x4*>(sv) = *reinterpret_cast<const f32x4*>(s + (int64_t)f * C + c8 * 8);
            *reinterpret_cast<f32x4*>(sv + 4) = *reinterpret_cast<const f32x4*>(s + (int64_t)f * C + c8 * 8 + 4);
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)v[j] * (float)gv[j] + sv[j] * k);
        reinterpret_cast<bf16x8*>(out)[id] = o;
    }
}

// inverse of conv3d_gather for padding 0 and stride = kernel (non-overlapping windows): dx[t,y,x,:] = dA[(to,ho,wo),(dt,dh,dw),:];
// positions outside every window (odd trailing rows) get zeros
// POOL: dA is [To*Ho*Wo, C] (gradient of an average-pooled value, shared by the window's taps and scaled by 1 / taps) instead of the Conv3d
// patch matrix [To*Ho*Wo, taps*C]
template <bool POOL>
__global__ __launch_bounds__(256) void conv3d_scatter_k(const bf16* __restrict__ dA, bf16* __restrict__ dx, int T, int H, int W, int C, int kt, int kh,
                                                        int kw, int pad, int To, int Ho, int Wo) {
    const int cv = C >> 3;
    const int64_t total = (int64_t)T * H * W * cv;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int c8 = id % cv;
        int64_t p = id / cv;
        const int x = p % W; p /= W;
        const int y = p % H;
        const int t = (int)(p / H);
        // kernel = stride: a pixel (shifted by the zero padding) lies in exactly one window, or past the last one
        const int tp = t + pad, yp = y + pad, xp = x + pad;
        const int to = tp / kt, ho = yp / kh, wo = xp / kw;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (to < To && ho < Ho && wo < Wo) {
            const int64_t r = ((int64_t)to * Ho + ho) * Wo + wo;
            if (POOL) {
                const bf16x8 d = *reinterpret_cast<const bf16x8*>(dA + r * C + c8 * 8);
                const float inv = 1.0f / (kt * kh * kw);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)d[j] * inv);
            } else {
                const int tap = ((tp % kt) * kh + (yp % kh)) * kw + (xp % kw);
                v = *reinterpret_cast<const bf16x8*>(dA + (r * (kt * kh * kw) + tap) * C + c8 * 8);
            }
        }
        reinterpret_cast<bf16x8*>(dx)[id] = v;
    }
}

}  // namespace

extern "C" int ufv_act(const void* pre, void* out, int64_t n, int act, void* stream) {
    UFV_REQUIRE(pre && out && n > 0 && n % 8 == 0, "ufv_act: n must be a positive multiple of 8");
    hipLaunchKernelGGL((act_k<false>), dim3(grid_for(n / 8)), dim3(256), 0, ST(stream), (const bf16*)pre, nullptr, (bf16*)out, n / 8, act);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_act_bwd(const void* pre, const void* dout, void* dpre, int64_t n, int act, void* stream) {
    UFV_REQUIRE(pre && dout && dpre && n > 0 && n % 8 == 0, "ufv_act_bwd: n must be a positive multiple of 8");
    hipLaunchKernelGGL((act_k<true>), dim3(grid_for(n / 8)), dim3(256), 0, ST(stream), (const bf16*)pre, (const bf16*)dout, (bf16*)dpre, n / 8, act);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
    UFV_REQUIRE(a && b && out && n > 0 && n % 8 == 0, "ufv_add_bf16: n must be a positive multiple of 8");
    hipLaunchKernelGGL(add_bf16_k, dim3(grid_for(n / 8)), dim3(256), 0, ST(stream), (const bf16*)a, (const bf16*)b, (bf16*)out, n / 8);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int64_t ufv_layernorm_bwd_ws_bytes(int C) { return (int64_t)(1024 + 32) * 2 * C * sizeof(float); }

extern "C" int ufv_layernorm_bwd(const void* x, int64_t ldx, const float* w, const float* b, const void* dout, int64_t ldd, void* dx,
                                 int64_t lddx, float* dw, float* db, int M, int C, float eps, int act, void* ws, void* stream) {
    UFV_REQUIRE(x && w && b && dout && dx && dw && db && ws && M > 0 && C > 0 && C % 8 == 0 && C <= 4096 && ldx % 8 == 0 && ldd % 8 == 0 &&
                lddx % 8 == 0, "ufv_layernorm_bwd: C=%d must be a multiple of 8 and <= 4096", C);
    float* part = reinterpret_cast<float*>(ws);
    const int nch = C / 8;
    int blocks, R;
    if (nch > 128 && (act == ACT_SILU || act == ACT_NONE)) {        // block per row, equal row counts per block, <= 1024 partial rows
        blocks = R = cdiv(M, cdiv(M, 1024));
#define LBB(MI, A) hipLaunchKernelGGL((layernorm_bwd_blk_k<MI, A>), dim3(blocks), dim3(256), 0, ST(stream), (const bf16*)x, ldx, w, b, \
                                      (const bf16*)dout, ldd, (bf16*)dx, lddx, part, M, C, eps)
        if (nch <= 256) { if (act) LBB(1, ACT_SILU); else LBB(1, ACT_NONE); }
        else { if (act) LBB(2, ACT_SILU); else LBB(2, ACT_NONE); }
#undef LBB
    } else {
        blocks = M < 1024 ? cdiv(M, 4) : 256;                       // 4 waves per block, one partial row per wave
        R = blocks * 4;
#define LB(MI) hipLaunchKernelGGL((layernorm_bwd_k<MI>), dim3(blocks), dim3(256), 0, ST(stream), (const bf16*)x, ldx, w, b, (const bf16*)dout, ldd, \
                                  (bf16*)dx, lddx, part, M, C, eps, act)
        if (nch <= 64) LB(1); else if (nch <= 128) LB(2); else if (nch <= 256) LB(4); else LB(8);
#undef LB
    }
    UFV_CHECK_LAUNCH();
    float* tmp = part + (int64_t)1024 * 2 * C;
    const int slices = R >= 64 ? 32 : 1;
    if (slices > 1) {
        hipLaunchKernelGGL(colsum2_f32_k, dim3(cdiv(2 * C, 256), slices), dim3(256), 0, ST(stream), part, (int64_t)2 * C, R, C, tmp, nullptr, nullptr);
        UFV_CHECK_LAUNCH();
        hipLaunchKernelGGL(colsum2_f32_k, dim3(cdiv(2 * C, 256), 1), dim3(256), 0, ST(stream), tmp, (int64_t)2 * C, slices, C, nullptr, dw, db);
    } else {
        hipLaunchKernelGGL(colsum2_f32_k, dim3(cdiv(2 * C, 256), 1), dim3(256), 0, ST(stream), part, (int64_t)2 * C, R, C, nullptr, dw, db);
    }
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_ln_add_silu_g(const void* z, const float* wa, const float* ba, const void* s, const float* wb, const float* bb,
                                 const void* dout, void* g, int M, int C, float eps, void* stream) {
    UFV_REQUIRE(z && wa && ba && s && dout && g && M > 0 && C % 8 == 0 && C <= 4096 && ((wb != nullptr) == (bb != nullptr)),
                "ufv_ln_add_silu_g: C=%d must be a multiple of 8 and <= 4096", C);
    const int nch = C / 8;
#define LG(MI) hipLaunchKernelGGL((ln_add_silu_g_k<MI>), dim3(cdiv(M, 4)), dim3(256), 0, ST(stream), (const bf16*)z, wa, ba, (const bf16*)s, wb, bb, \
                                  (const bf16*)dout, (bf16*)g, M, C, eps)
    if (nch <= 64) LG(1); else if (nch <= 128) LG(2); else if (nch <= 256) LG(4); else LG(8);
#undef LG
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_dwconv3x3(const void* x, void* y, const float* w9, int F, int H, int W, int C, int flip, void* stream) {
    UFV_REQUIRE(x && y && w9 && F > 0 && H > 0 && W > 0 && C % 8 == 0, "ufv_dwconv3x3: C must be a multiple of 8");
    UFV_REQUIRE((int64_t)F * H * W * (C / 8) < (int64_t)1 << 31, "ufv_dwconv3x3: tensor too large for 32-bit indexing");
    hipLaunchKernelGGL(dwconv3x3_k, dim3(grid_for((int64_t)F * H * W * (C / 8))), dim3(256), 0, ST(stream), (const bf16*)x, (bf16*)y, w9, F, H, W,
                       C, flip);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int64_t ufv_dwconv3x3_dw_ws_bytes(int C) { return (int64_t)128 * 9 * C * sizeof(float); }

extern "C" int ufv_dwconv3x3_dw(const void* x, const void* dy, float* dw9, int F, int H, int W, int C, void* ws, void* stream) {
    UFV_REQUIRE(x && dy && dw9 && ws && F > 0 && C > 0, "ufv_dwconv3x3_dw: bad arguments");
    const int64_t NP = (int64_t)F * H * W;
    UFV_REQUIRE(C % 8 == 0, "ufv_dwconv3x3_dw: C=%d must be a multiple of 8", C);
    const int nslice = (int)(NP / 128 < 1 ? 1 : (NP / 128 > 128 ? 128 : NP / 128));      // >= 16 pixels per pixel lane
    float* part = reinterpret_cast<float*>(ws);
    hipLaunchKernelGGL(dwconv3x3_dw_k, dim3(cdiv(C / 8, 32), nslice), dim3(256), 0, ST(stream), (const bf16*)x, (const bf16*)dy, part, F, H, W, C);
    UFV_CHECK_LAUNCH();
    hipLaunchKernelGGL(dw_reduce_k, dim3(cdiv(9 * C, 256)), dim3(256), 0, ST(stream), part, nslice, 9 * C, dw9);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_prod_colsum(const void* a, const void* b, int F, int P, int C, float* out, void* stream) {
    UFV_REQUIRE(a && b && out && F > 0 && P > 0 && C > 0, "ufv_prod_colsum: bad arguments");
    UFV_REQUIRE(C % 8 == 0, "ufv_prod_colsum: C=%d must be a multiple of 8", C);
    hipLaunchKernelGGL(prod_colsum_k, dim3(cdiv(C / 8, 32), F), dim3(256), 0, ST(stream), (const bf16*)a, (const bf16*)b, F, P, C, out);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_scale_add_bcast(const void* a, const void* g, const float* s, float k, void* out, int F, int P, int C, void* stream) {
    UFV_REQUIRE(a && g && out && F > 0 && P > 0 && C % 8 == 0, "ufv_scale_add_bcast: C must be a multiple of 8");
    UFV_REQUIRE((int64_t)F * P * (C / 8) < (int64_t)1 << 31, "ufv_scale_add_bcast: tensor too large for 32-bit indexing");
    hipLaunchKernelGGL(scale_add_bcast_k, dim3(grid_for((int64_t)F * P * C / 8)), dim3(256), 0, ST(stream), (const bf16*)a, (const bf16*)g, s, k,
                       (bf16*)out, F, P, C);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_conv3d_scatter(const void* dA, void* dx, int T, int H, int W, int C, int kt, int kh, int kw, int pad, void* stream) {
    UFV_REQUIRE(dA && dx && C % 8 == 0 && kt > 0 && kh > 0 && kw > 0 && pad >= 0, "ufv_conv3d_scatter: C must be a multiple of 8");
    const int To = (T + 2 * pad - kt) / kt + 1, Ho = (H + 2 * pad - kh) / kh + 1, Wo = (W + 2 * pad - kw) / kw + 1;
    UFV_REQUIRE(To > 0 && Ho > 0 && Wo > 0, "ufv_conv3d_scatter: empty output");
    hipLaunchKernelGGL(conv3d_scatter_k<false>, dim3(grid_for((int64_t)T * H * W * (C / 8))), dim3(256), 0, ST(stream), (const bf16*)dA, (bf16*)dx, T, H,
                       W, C, kt, kh, kw, pad, To, Ho, Wo);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_avgpool3d_bwd(const void* dy, void* dx, int T, int H, int W, int C, int kt, int kh, int kw, void* stream) {
    UFV_REQUIRE(dy && dx && C % 8 == 0 && kt > 0 && kh > 0 && kw > 0, "ufv_avgpool3d_bwd: C must be a multiple of 8");
    const int To = T / kt, Ho = H / kh, Wo = W / kw;
    UFV_REQUIRE(To > 0 && Ho > 0 && Wo > 0, "ufv_avgpool3d_bwd: empty output");
    hipLaunchKernelGGL(conv3d_scatter_k<true>, dim3(grid_for((int64_t)T * H * W * (C / 8))), dim3(256), 0, ST(stream), (const bf16*)dy, (bf16*)dx, T, H,
                       W, C, kt, kh, kw, 0, To, Ho, Wo);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}
