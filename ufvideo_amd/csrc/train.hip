// Training step of the Qwen2 decoder: the backward kernels and the optimizer
// update that the reference gets from torch autograd + DeepSpeed ZeRO-2 / AdamW (videorefer_qwen2.py:198-352 builds the
// loss; train.py:749 + scripts/zero2.json run backward and the update).
//
// The big contractions of the backward pass (dX = dY W, dW = dY^T X, and the five products of attention's backward) run on
// the same MFMA GEMM kernels as the forward (ufv_gemm, NT form); this file holds what surrounds them:
//   * transposes (an NT GEMM wants both operands K-contiguous, so dY^T / X^T / W^T are materialised: HBM-bound, LDS-tiled)
//   * RMSNorm / SwiGLU / RoPE backward, the causal softmax backward of attention, cross-entropy forward+backward
//   * column sums (bias gradients, dw partials), row scatter-add (embedding gradient), sum of squares (gradient norm)
//   * AdamW on fp32 master weights with the bf16 working copy written in the same pass
// All HBM-bound: one wave per row or 16-byte lanes, fp32 arithmetic.
#include "common.h"
#include "../../include/ufv.h"

namespace {

#define ST(s) reinterpret_cast<hipStream_t>(s)

inline int grid_for(int64_t n, int per_block = 256) {
    const int64_t g = (n + per_block - 1) / per_block;
    return (int)(g < 16384 ? (g > 0 ? g : 1) : 16384);
}

// ---------------------------------------------------------------------------------------------------------
// out[c][r] = in[r][c] for r < R, c < C; out columns R..Rpad-1 are written as zeros (the K padding of the GEMM
// that consumes it).  64x64 tiles through LDS (pitch 66 elements: conflict-free both ways), 256 threads.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_bf16_k(const bf16* __restrict__ in, int64_t ldi, bf16* __restrict__ out, int64_t ldo,
                                                        int R, int C, int Rpad) {
    __shared__ bf16 tile[64][66];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // 4 rows per pass
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        tile[ty + 4 * i][tx] = (r < R && c < C) ? in[(int64_t)r * ldi + c] : (bf16)0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;
        if (c < C && r < Rpad) out[(int64_t)c * ldo + r] = tile[tx][ty + 4 * i];
    }
}

// Same transpose with 16-byte global accesses (rows 16-byte aligned, C and Rpad multiples of 8): a wave reads 8 rows x 128 B and
// writes 8 rows x 128 B; the 64x64 tile sits in LDS as 32-bit words (2 columns each) at an odd word pitch.
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__global__ __launch_bounds__(256) void transpose_bf16_v8_k(const bf16* __restrict__ in, int64_t ldi, bf16* __restrict__ out, int64_t ldo,
                                                           int R, int C, int Rpad) {
    __shared__ uint32_t tile[64][33];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int hi = threadIdx.x >> 3, lo = threadIdx.x & 7;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = hi + 32 * p, c = c0 + 8 * lo;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < R && c < C) v = *reinterpret_cast<const u32x4*>(in + (int64_t)(r0 + r) * ldi + c);
#pragma unroll
        for (int w = 0; w < 4; ++w) tile[r][4 * lo + w] = v[w];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int c = hi + 32 * p, r = r0 + 8 * lo;
        if (c0 + c < C && r < Rpad) {
            const int sh = (c & 1) * 16;
            uint32_t o[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t a = (tile[8 * lo + 2 * w][c >> 1] >> sh) & 0xffffu, b = (tile[8 * lo + 2 * w + 1][c >> 1] >> sh) & 0xffffu;
                o[w] = a | (b << 16);
            }
            *reinterpret_cast<u32x4*>(out + (int64_t)(c0 + c) * ldo + r) = u32x4{o[0], o[1], o[2], o[3]};
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// RMSNorm backward.  y = w * (x * r), r = rsqrt(mean(x^2) + eps)  (modeling_qwen2.py:238-254)
//   dx[row] (+)= r * (w*dy) - x * r^3/D * sum_j(w_j dy_j x_j);   dw_part[wave][j] = sum over this wave's rows of dy_j x_j r
// One wave per row, grid-stride over rows; every wave keeps its dw partial in registers and writes one row of
// dw_part [nwaves, D] at the end (reduced by colsum_f32_k: deterministic, no atomics).
// ---------------------------------------------------------------------------------------------------------
template <int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_bwd_k(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                     const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx,
                                                     float* __restrict__ dw_part, int M, int D, float eps, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int nv = D >> 2;
    f32x4 dwa[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) dwa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int row = wid; row < M; row += nw) {
        f32x4 xv[MAXV], gv[MAXV];
        float q = 0.f, dot = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (lane + 64 * i < nv) {
                const int c = 4 * (lane + 64 * i);
                xv[i] = *reinterpret_cast<const f32x4*>(x + (int64_t)row * ldx + c);
                gv[i] = *reinterpret_cast<const f32x4*>(dy + (int64_t)row * lddy + c);
                const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    q += xv[i][j] * xv[i][j];
                    dot += ww[j] * gv[i][j] * xv[i][j];
                }
            }
        q = wave_sum(q); dot = wave_sum(dot);
        const float r = rsqrtf(q / D + eps);
        const float k = dot * r * r * r / D;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (lane + 64 * i < nv) {
                const int c = 4 * (lane + 64 * i);
                const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = r * ww[j] * gv[i][j] - xv[i][j] * k;
                    dwa[i][j] += gv[i][j] * xv[i][j] * r;
                }
                float* dp = dx + (int64_t)row * lddx + c;
                if (accumulate) {
                    const f32x4 old = *reinterpret_cast<const f32x4*>(dp);
                    o += old;
                }
                *reinterpret_cast<f32x4*>(dp) = o;
            }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < nv) *reinterpret_cast<f32x4*>(dw_part + (int64_t)wid * D + 4 * (lane + 64 * i)) = dwa[i];
}

// Wide rows (D > 1024): one 256-thread block per row, a thread holds MAXV 16-byte chunks of x and dy (32 VGPRs instead of the
// 128 the wave-per-row form needs at D = 3584, so 8 waves per SIMD stay resident); q and dot are reduced through LDS together.
// dw_part has one row per block.
template <int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_bwd_blk_k(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                         const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx,
                                                         float* __restrict__ dw_part, int M, int D, float eps, int accumulate) {
    __shared__ float red[2][2][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nv = D >> 2;
    f32x4 dwa[MAXV], wv4[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        dwa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        wv4[i] = tid + 256 * i < nv ? *reinterpret_cast<const f32x4*>(w + 4 * (tid + 256 * i)) : dwa[i];
    }
    int it = 0;
    for (int row = blockIdx.x; row < M; row += gridDim.x, it ^= 1) {
        f32x4 xv[MAXV], gv[MAXV];
        float q = 0.f, dot = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (tid + 256 * i < nv) {
                const int c = 4 * (tid + 256 * i);
                xv[i] = *reinterpret_cast<const f32x4*>(x + (int64_t)row * ldx + c);
                gv[i] = *reinterpret_cast<const f32x4*>(dy + (int64_t)row * lddy + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    q += xv[i][j] * xv[i][j];
                    dot += wv4[i][j] * gv[i][j] * xv[i][j];
                }
            }
        q = wave_sum(q); dot = wave_sum(dot);
        if (lane == 0) { red[it][0][wv] = q; red[it][1][wv] = dot; }       // two buffers: one barrier per row
        __syncthreads();
        q = (red[it][0][0] + red[it][0][1]) + (red[it][0][2] + red[it][0][3]);
        dot = (red[it][1][0] + red[it][1][1]) + (red[it][1][2] + red[it][1][3]);
        const float r = rsqrtf(q / D + eps);
        const float k = dot * r * r * r / D;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (tid + 256 * i < nv) {
                const int c = 4 * (tid + 256 * i);
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = r * wv4[i][j] * gv[i][j] - xv[i][j] * k;
                    dwa[i][j] += gv[i][j] * xv[i][j] * r;
                }
                float* dp = dx + (int64_t)row * lddx + c;
                if (accumulate) {
                    const f32x4 old = *reinterpret_cast<const f32x4*>(dp);
                    o += old;
                }
                *reinterpret_cast<f32x4*>(dp) = o;
            }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (tid + 256 * i < nv) *reinterpret_cast<f32x4*>(dw_part + (int64_t)blockIdx.x * D + 4 * (tid + 256 * i)) = dwa[i];
}

// out[j] (+)= sum_r x[r][j]  (fp32 [R, C] -> [C]).  A block owns 32 columns (one 128-byte line per row); its 8 row groups each
// take every 8th row and the 8 partial sums are added in a fixed order through LDS (deterministic, no atomics).  One thread per
// column over all rows left the chip almost idle (C / 256 blocks) and took 57 us for the 1024 x 3584 RMSNorm partials.
__global__ __launch_bounds__(256) void colsum_f32_k(const float* __restrict__ x, int64_t ld, int R, int C, float* __restrict__ out,
                                                    int accumulate) {
    __shared__ float red[8][32];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
        int r = rg;
        for (; r + 24 < R; r += 32) {
            s0 += x[(int64_t)r * ld + c]; s1 += x[(int64_t)(r + 8) * ld + c];
            s2 += x[(int64_t)(r + 16) * ld + c]; s3 += x[(int64_t)(r + 24) * ld + c];
        }
        for (; r < R; r += 8) s0 += x[(int64_t)r * ld + c];
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && c < C) {
        float s = red[0][cl];
#pragma unroll
        for (int i = 1; i < 8; ++i) s += red[i][cl];
        out[c] = accumulate ? out[c] + s : s;
    }
}

// bias gradient: out[j] (+)= sum_r dy[r][j] for bf16 dy [R, C]: grid (C/256, RS) partials into part[RS][C], then colsum_f32_k
__global__ __launch_bounds__(256) void colsum_bf16_part_k(const bf16* __restrict__ x, int64_t ld, int R, int C, float* __restrict__ part) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int rs = gridDim.y, per = (R + rs - 1) / rs;
    const int r0 = blockIdx.y * per, r1 = min(R, r0 + per);
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += (float)x[(int64_t)r * ld + c];
    part[(int64_t)blockIdx.y * C + c] = s;
}

// ---------------------------------------------------------------------------------------------------------
// SwiGLU on the packed gate/up layout (rows of `gu` = [16 gate | 16 up] alternating, see pack_swiglu):
//   forward  act[m][16b + j] = silu(g) * u                       (modeling_qwen2.py:47)
//   backward dgu: dg = dact * u * s * (1 + g * (1 - s)),  du = dact * g * s,   s = sigmoid(g)
// 8 elements per thread.
// ---------------------------------------------------------------------------------------------------------
template <bool BWD>
__global__ __launch_bounds__(256) void swiglu_k(const bf16* __restrict__ gu, int64_t ldgu, const bf16* __restrict__ dact, int64_t ldd,
                                                bf16* __restrict__ out, int64_t ldo, int M, int I) {
    const int cpr = I >> 3;                                        // 8-element chunks per row of act
    const int64_t total = (int64_t)M * cpr;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int m = id / cpr, ch = id % cpr;
        const int blk = ch >> 1, half = ch & 1;                    // 16-wide block, which 8 of it
        const bf16* gp = gu + (int64_t)m * ldgu + blk * 32 + half * 8;
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(gp), u = *reinterpret_cast<const bf16x8*>(gp + 16);
        if (!BWD) {
            bf16x8 a;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float gf = (float)g[j];
                a[j] = (bf16)swiglu_f(gf, (float)u[j]);
            }
            *reinterpret_cast<bf16x8*>(out + (int64_t)m * ldo + ch * 8) = a;
        } else {
            const bf16x8 d = *reinterpret_cast<const bf16x8*>(dact + (int64_t)m * ldd + ch * 8);
            bf16x8 dg, du;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float gf = (float)g[j], uf = (float)u[j], df = (float)d[j];
                const float s = sigmoid_fast(gf);
                dg[j] = (bf16)(df * uf * s * (1.0f + gf * (1.0f - s)));
                du[j] = (bf16)(df * gf * s);
            }
            bf16* op = out + (int64_t)m * ldo + blk * 32 + half * 8;
            *reinterpret_cast<bf16x8*>(op) = dg;
            *reinterpret_cast<bf16x8*>(op + 16) = du;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Rotate-half RoPE on `nheads` heads starting at column col0 of buf [S, ld], in place, angle sign * pos * inv_freq:
// sign = +1 is the forward rotation (modeling_qwen2.py:113-135), sign = -1 its transpose = the backward pass.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_rows_k(bf16* buf, int64_t ld, int S, int col0, int nheads, int hd, const float* inv_freq,
                                                   int pos0, float sign) {
    const int half = hd >> 1;
    const int64_t total = (int64_t)S * nheads * half;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int i = id % half;
        const int hh = (id / half) % nheads;
        const int s = id / ((int64_t)half * nheads);
        bf16* p = buf + (int64_t)s * ld + col0 + hh * hd;
        const float ang = (float)(pos0 + s) * inv_freq[i];
        const float c = cosf(ang), sn = sign * sinf(ang);
        const float x1 = (float)p[i], x2 = (float)p[half + i];
        p[i] = (bf16)(x1 * c - x2 * sn);
        p[half + i] = (bf16)(x2 * c + x1 * sn);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Attention backward, softmax part, one block per stacked query row i (causal: keys j <= i / rep, j < Sk):
//   p_j = softmax_j(scale * sc_ij);  delta = sum_j p_j dp_ij;  ds_ij = scale * p_j * (dp_ij - delta)
// writes P (bf16, the operand of dV = P^T dO) and dS (bf16, the operand of dQ = dS K and dK = dS^T Q); every column up to
// ldp (the padded key count) is written, masked ones as exact zeros.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_softmax_bwd_k(const float* __restrict__ sc, const float* __restrict__ dp, int64_t ld,
                                                          bf16* __restrict__ P, bf16* __restrict__ dS, int64_t ldp, int Sk, int Skpad,
                                                          float scale, int rep) {
    __shared__ float red[16];
    const int i = blockIdx.x, tid = threadIdx.x;
    const int nvalid = min(Sk, i / rep + 1);                      // row i = query position i / rep (rep stacked heads per position)
    const float* s = sc + (int64_t)i * ld;
    const float* d = dp + (int64_t)i * ld;
    float mx = -INFINITY;
    for (int j = tid; j < nvalid; j += 256) mx = fmaxf(mx, s[j]);
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale;
    float sum = 0.f, dot = 0.f;
    for (int j = tid; j < nvalid; j += 256) {
        const float e = __expf(s[j] * scale - mx);
        sum += e; dot += e * d[j];
    }
    sum = block_sum(sum, red + 4);
    dot = block_sum(dot, red + 8);
    const float inv = 1.0f / sum, delta = dot * inv;
    bf16* pr = P + (int64_t)i * ldp;
    bf16* dr = dS + (int64_t)i * ldp;
    for (int j = tid; j < Skpad; j += 256) {
        if (j < nvalid) {
            const float p = __expf(s[j] * scale - mx) * inv;
            pr[j] = (bf16)p;
            dr[j] = (bf16)(scale * p * (d[j] - delta));
        } else {
            pr[j] = (bf16)0.f; dr[j] = (bf16)0.f;
        }
    }
}

// fp32 [R, C] (pitch ldi) -> bf16 [R, C] (pitch ldo)
__global__ __launch_bounds__(256) void cvt_rows_k(const float* __restrict__ in, int64_t ldi, bf16* __restrict__ out, int64_t ldo, int R, int C) {
    const int64_t total = (int64_t)R * C;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int r = id / C, c = id % C;
        out[(int64_t)r * ldo + c] = (bf16)in[(int64_t)r * ldi + c];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Cross entropy forward + backward in one pass over the logits row (HF Qwen2ForCausalLM loss, labels already shifted):
//   loss[i] = logsumexp(x_i) - x_i[label];  dlogits[i][j] = gscale * (softmax(x_i)_j - [j == label])   (0 for ignored rows)
// dlogits is bf16 [M, ldd]; columns V..Vpad-1 are written as zeros.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cross_entropy_bwd_k(const float* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                           int V, int Vpad, int64_t ignore_index, float gscale, float* __restrict__ loss,
                                                           bf16* __restrict__ dl, int64_t ldd) {
    __shared__ float red[16];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int64_t lab = labels[row];
    bf16* o = dl + (int64_t)row * ldd;
    if (lab == ignore_index) {
        if (tid == 0) loss[row] = 0.f;
        for (int j = tid; j < Vpad; j += 256) o[j] = (bf16)0.f;
        return;
    }
    const float* x = logits + (int64_t)row * ld;
    float mx = -INFINITY;
    for (int j = tid; j < V; j += 256) mx = fmaxf(mx, x[j]);
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int j = tid; j < V; j += 256) s += expf(x[j] - mx);
    s = block_sum(s, red + 4);
    if (tid == 0) loss[row] = logf(s) + mx - x[lab];
    const float inv = gscale / s;
    for (int j = tid; j < Vpad; j += 256) {
        float g = 0.f;
        if (j < V) g = expf(x[j] - mx) * inv - (j == lab ? gscale : 0.f);
        o[j] = (bf16)g;
    }
}

// dst[idx[r]][:] += src[r][:] for r < R (fp32, atomic: rows of idx may repeat; idx < 0 rows are skipped)
__global__ __launch_bounds__(256) void scatter_add_rows_k(const float* __restrict__ src, int64_t lds, const int64_t* __restrict__ idx,
                                                          float* __restrict__ dst, int64_t ldd, int R, int D) {
    const int64_t total = (int64_t)R * D;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int r = id / D, c = id % D;
        const int64_t t = idx[r];
        if (t >= 0) atomicAdd(dst + t * ldd + c, src[(int64_t)r * lds + c]);
    }
}

// part[b] = sum of squares of this block's grid-stride slice (gradient-norm clipping: torch.nn.utils.clip_grad_norm_);
// x is 16-byte aligned, 4 floats per lane per step, scalar tail
__global__ __launch_bounds__(256) void sumsq_k(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
    __shared__ float red[16];
    float s = 0.f;
    const int64_t n4 = n >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = x4[i];
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += x[i] * x[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------------------
// AdamW (torch.optim.AdamW, decoupled weight decay) on fp32 master weights; the bf16 working copy is written in the same
// pass.  g is scaled by *gscale_ptr (the clipping coefficient, computed on the device) when given.
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adamw_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, bf16* __restrict__ pb, int64_t n, float lr, float b1, float b2,
                                               float eps, float wd, float bc1, float bc2, const float* __restrict__ gscale_ptr) {
    const float gs = gscale_ptr ? *gscale_ptr : 1.0f;
    const float step = lr / bc1, rs2 = rsqrtf(bc2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * gs;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        pi -= step * mi / (sqrtf(vi) * rs2 + eps);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (pb) pb[i] = (bf16)pi;
    }
}

}  // namespace

extern "C" int ufv_transpose_bf16(const void* in, int64_t ldi, void* out, int64_t ldo, int R, int C, int Rpad, void* stream) {
    UFV_REQUIRE(in && out && R > 0 && C > 0 && Rpad >= R && ldo >= Rpad && ldi >= C, "ufv_transpose_bf16: bad arguments (R=%d C=%d Rpad=%d)", R, C, Rpad);
    const bool vec = ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0) && ldi % 8 == 0 && ldo % 8 == 0 && C % 8 == 0 && Rpad % 8 == 0;
    if (vec)
        hipLaunchKernelGGL(transpose_bf16_v8_k, dim3(cdiv(Rpad, 64), cdiv(C, 64)), dim3(256), 0, ST(stream), (const bf16*)in, ldi, (bf16*)out,
                           ldo, R, C, Rpad);
    else
        hipLaunchKernelGGL(transpose_bf16_k, dim3(cdiv(Rpad, 64), cdiv(C, 64)), dim3(256), 0, ST(stream), (const bf16*)in, ldi, (bf16*)out, ldo,
                           R, C, Rpad);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int64_t ufv_rmsnorm_bwd_ws_bytes(int D) { return (int64_t)1024 * D * sizeof(float); }

extern "C" int ufv_rmsnorm_bwd(const float* x, int ldx, const float* w, const float* dy, int lddy, float* dx, int lddx, int accumulate,
                               float* dw, int dw_accumulate, int M, int D, float eps, void* ws, void* stream) {
    UFV_REQUIRE(x && w && dy && dx && dw && ws && M > 0 && D > 0, "ufv_rmsnorm_bwd: bad arguments");
    UFV_REQUIRE(D % 4 == 0 && D <= 64 * 4 * 16 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0, "ufv_rmsnorm_bwd: D=%d must be a multiple of 4 and <= 4096", D);
    float* part = reinterpret_cast<float*>(ws);
    const int nv = D / 4;
    if (nv > 256) {                                                // block per row; equal row counts per block, <= 1024 partial rows
        const int nb = cdiv(M, cdiv(M, 1024));
        hipLaunchKernelGGL((rmsnorm_bwd_blk_k<4>), dim3(nb), dim3(256), 0, ST(stream), x, ldx, w, dy, lddy, dx, lddx, part, M, D, eps, accumulate);
        UFV_CHECK_LAUNCH();
        hipLaunchKernelGGL(colsum_f32_k, dim3(cdiv(D, 32)), dim3(256), 0, ST(stream), part, (int64_t)D, nb, D, dw, dw_accumulate);
        UFV_CHECK_LAUNCH();
        return UFV_OK;
    }
    const int blocks = M < 1024 ? cdiv(M, 4) : 256;               // 4 waves per block -> at most 1024 partial rows
#define RB(MV) hipLaunchKernelGGL((rmsnorm_bwd_k<MV>), dim3(blocks), dim3(256), 0, ST(stream), x, ldx, w, dy, lddy, dx, lddx, part, M, D, eps, accumulate)
    if (nv <= 64) RB(1); else RB(4);
#undef RB
    UFV_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_f32_k, dim3(cdiv(D, 32)), dim3(256), 0, ST(stream), part, (int64_t)D, blocks * 4, D, dw, dw_accumulate);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_colsum_bf16(const void* x, int64_t ld, int R, int C, float* out, int accumulate, void* ws, void* stream) {
    UFV_REQUIRE(x && out && ws && R > 0 && C > 0, "ufv_colsum_bf16: bad arguments");
    const int rs = R >= 64 ? 32 : 1;
    float* part = reinterpret_cast<float*>(ws);                    // [32][C] fp32
    hipLaunchKernelGGL(colsum_bf16_part_k, dim3(cdiv(C, 256), rs), dim3(256), 0, ST(stream), (const bf16*)x, ld, R, C, part);
    UFV_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_f32_k, dim3(cdiv(C, 32)), dim3(256), 0, ST(stream), part, (int64_t)C, rs, C, out, accumulate);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_swiglu(const void* gu, int64_t ldgu, void* act, int64_t lda, int M, int I, void* stream) {
    UFV_REQUIRE(gu && act && M > 0 && I > 0 && I % 16 == 0 && ldgu % 8 == 0 && lda % 8 == 0, "ufv_swiglu: I=%d must be a multiple of 16", I);
    hipLaunchKernelGGL((swiglu_k<false>), dim3(grid_for((int64_t)M * I / 8)), dim3(256), 0, ST(stream), (const bf16*)gu, ldgu, nullptr, 0,
                       (bf16*)act, lda, M, I);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_swiglu_bwd(const void* gu, int64_t ldgu, const void* dact, int64_t ldd, void* dgu, int64_t ldo, int M, int I,
                              void* stream) {
    UFV_REQUIRE(gu && dact && dgu && M > 0 && I > 0 && I % 16 == 0 && ldgu % 8 == 0 && ldd % 8 == 0 && ldo % 8 == 0,
                "ufv_swiglu_bwd: I=%d must be a multiple of 16", I);
    hipLaunchKernelGGL((swiglu_k<true>), dim3(grid_for((int64_t)M * I / 8)), dim3(256), 0, ST(stream), (const bf16*)gu, ldgu,
                       (const bf16*)dact, ldd, (bf16*)dgu, ldo, M, I);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_rope_rows(void* buf, int64_t ld, int S, int col0, int nheads, int hd, const float* inv_freq, int pos0, int backward,
                             void* stream) {
    UFV_REQUIRE(buf && inv_freq && S > 0 && nheads > 0 && hd % 2 == 0, "ufv_rope_rows: bad arguments");
    hipLaunchKernelGGL(rope_rows_k, dim3(grid_for((int64_t)S * nheads * (hd / 2))), dim3(256), 0, ST(stream), (bf16*)buf, ld, S, col0,
                       nheads, hd, inv_freq, pos0, backward ? -1.0f : 1.0f);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_cross_entropy_bwd(const float* logits, int64_t ld, const int64_t* labels, int M, int V, int Vpad, int64_t ignore_index,
                                     float gscale, float* loss, void* dlogits, int64_t ldd, void* stream) {
    UFV_REQUIRE(logits && labels && loss && dlogits && M > 0 && V > 0 && Vpad >= V && ldd >= Vpad, "ufv_cross_entropy_bwd: bad arguments");
    hipLaunchKernelGGL(cross_entropy_bwd_k, dim3(M), dim3(256), 0, ST(stream), logits, ld, labels, V, Vpad, ignore_index, gscale, loss,
                       (bf16*)dlogits, ldd);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_scatter_add_rows(const float* src, int64_t lds, const int64_t* idx, float* dst, int64_t ldd, int R, int D, void* stream) {
    if (R == 0) return UFV_OK;
    UFV_REQUIRE(src && idx && dst && R > 0 && D > 0, "ufv_scatter_add_rows: bad arguments");
    hipLaunchKernelGGL(scatter_add_rows_k, dim3(grid_for((int64_t)R * D)), dim3(256), 0, ST(stream), src, lds, idx, dst, ldd, R, D);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_sumsq(const float* x, int64_t n, float* partial, int n_partial, void* stream) {
    UFV_REQUIRE(x && partial && n > 0 && n_partial > 0 && n_partial <= 16384 && (uintptr_t)x % 16 == 0, "ufv_sumsq: bad arguments (x must be 16-byte aligned)");
    hipLaunchKernelGGL(sumsq_k, dim3(n_partial), dim3(256), 0, ST(stream), x, n, partial);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

extern "C" int ufv_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int step, const float* gscale, void* stream) {
    if (n == 0) return UFV_OK;
    UFV_REQUIRE(p && g && m && v && n > 0 && step >= 1, "ufv_adamw: bad arguments");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_k, dim3(grid_for(n)), dim3(256), 0, ST(stream), p, g, m, v, (bf16*)p_bf16, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2, gscale);
    UFV_CHECK_LAUNCH();
    return UFV_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Causal self-attention backward (GQA), materialised on the MFMA GEMMs (HF Qwen2Attention eager math,
// modeling_qwen2.py:150-172, differentiated).  The `rep` = Hq/Hkv query heads of kv-group g are processed TOGETHER:
// their rows are gathered into Qg / dOg [S*rep, hd] (row s*rep + j = token s of head g*rep + j), so that
//   sc = Qg k_g^T, dp = dOg v_g^T                            (fp32 [S*rep, Sp], one GEMM each)
//   P, dS = softmax backward, query position = row / rep     (bf16 [S*rep, Sp])
//   dQg = dS k_g                                             (one GEMM, scattered back to the head columns)
//   dv_g = P^T dOg,  dk_g = dS^T Qg                          (K = S*rep: split-K GEMMs with fp32 atomic accumulation)
// Sp = S rounded up to 128 (the fast GEMMs' N / K granularity); k / v must have at least Sp readable rows.
// ---------------------------------------------------------------------------------------------------------
static inline int64_t rup(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

namespace {
// dst[r][c] = src[(r / rep)][ (r % rep) * C + c ]  (gather = 1: head columns -> stacked rows) or the inverse scatter
__global__ __launch_bounds__(256) void head_rows_k(const bf16* __restrict__ src, int64_t lds, bf16* __restrict__ dst, int64_t ldd, int S,
                                                   int rep, int C, int gather) {
    const int cpr = C >> 3;
    const int64_t total = (int64_t)S * rep * cpr;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int ch = id % cpr;
        const int64_t r = id / cpr;
        const int s = r / rep, j = r % rep;
        if (gather)
            *reinterpret_cast<bf16x8*>(dst + r * ldd + ch * 8) = *reinterpret_cast<const bf16x8*>(src + (int64_t)s * lds + j * C + ch * 8);
        else
            *reinterpret_cast<bf16x8*>(dst + (int64_t)s * ldd + j * C + ch * 8) = *reinterpret_cast<const bf16x8*>(src + r * lds + ch * 8);
    }
}
}  // namespace


extern "C" int64_t ufv_attention_bwd_ws_bytes(int S, int Hq, int Hkv, int hd) {
    const int64_t rep = Hq / (Hkv > 0 ? Hkv : 1), M = (int64_t)S * rep, Sp = rup(S, 128), Mp = rup(M, 128);
    int64_t b = 0;
    b += 2 * M * Sp * 4;                 // sc, dp
    b += 2 * M * Sp * 2;                 // P, dS
    b += 2 * Sp * Mp * 2;                // P^T, dS^T
    b += (int64_t)hd * Sp * 2;           // k^T
    b += 3 * M * hd * 2;                 // Qg, dOg, dQg
    b += 2 * (int64_t)hd * Mp * 2;       // Qg^T, dOg^T
    b += 16 * Sp * hd * 4;               // split-K partial tiles (up to 16 slices)
    return b + 16 * 256 + 4096;
}

extern "C" int ufv_attention_bwd(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const void* dO, int64_t lddo,
                                 void* dq, int64_t lddq, void* dk, void* dv, int64_t lddkv, int S, int Hq, int Hkv, int hd, float scale,
                                 void* ws, void* stream) {
    UFV_REQUIRE(q && k && v && dO && dq && dk && dv && ws && S > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && hd > 0 && hd % 8 == 0,
                "ufv_attention_bwd: bad arguments (S=%d Hq=%d Hkv=%d hd=%d)", S, Hq, Hkv, hd);
    const int rep = Hq / Hkv;
    const int Sp = (int)rup(S, 128);
    const int64_t M64 = (int64_t)S * rep;
    UFV_REQUIRE(M64 < (1 << 30), "ufv_attention_bwd: S * heads-per-group too large");
    const int M = (int)M64, Mp = (int)rup(M, 128);
    char* w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    auto take = [&](int64_t bytes) { char* p = w; w += rup(bytes, 256); return p; };
    float* sc = (float*)take((int64_t)M * Sp * 4);
    float* dp = (float*)take((int64_t)M * Sp * 4);
    bf16* P = (bf16*)take((int64_t)M * Sp * 2);
    bf16* dS = (bf16*)take((int64_t)M * Sp * 2);
    bf16* PT = (bf16*)take((int64_t)Sp * Mp * 2);
    bf16* dST = (bf16*)take((int64_t)Sp * Mp * 2);
    bf16* kT = (bf16*)take((int64_t)hd * Sp * 2);
    bf16* Qg = (bf16*)take((int64_t)M * hd * 2);
    bf16* dOg = (bf16*)take((int64_t)M * hd * 2);
    bf16* dQg = (bf16*)take((int64_t)M * hd * 2);
    bf16* QgT = (bf16*)take((int64_t)hd * Mp * 2);
    bf16* dOgT = (bf16*)take((int64_t)hd * Mp * 2);
    float* part = (float*)take((int64_t)16 * Sp * hd * 4);
    const bf16* qb = (const bf16*)q; const bf16* kb = (const bf16*)k; const bf16* vb = (const bf16*)v; const bf16* dob = (const bf16*)dO;
    // K of the two split products = M: give every output tile ~16 K-tiles' worth of work per block
    const int nsplit = Mp / 64 >= 64 ? 16 : (Mp / 64 >= 16 ? 4 : 1);
    hipStream_t st = ST(stream);
    int rc;
#define TRY(call) do { rc = (call); if (rc != UFV_OK) return rc; } while (0)
    for (int g = 0; g < Hkv; ++g) {
        const bf16* kg = kb + (int64_t)g * hd;
        const bf16* vg = vb + (int64_t)g * hd;
        hipLaunchKernelGGL(head_rows_k, dim3(grid_for((int64_t)M * hd / 8)), dim3(256), 0, st, qb + (int64_t)g * rep * hd, ldq, Qg, (int64_t)hd, S, rep, hd, 1);
        UFV_CHECK_LAUNCH();
        hipLaunchKernelGGL(head_rows_k, dim3(grid_for((int64_t)M * hd / 8)), dim3(256), 0, st, dob + (int64_t)g * rep * hd, lddo, dOg, (int64_t)hd, S, rep, hd, 1);
        UFV_CHECK_LAUNCH();
        TRY(ufv_transpose_bf16(kg, ldkv, kT, Sp, S, hd, Sp, stream));
        TRY(ufv_gemm(Qg, hd, kg, (int)ldkv, sc, Sp, 1, M, Sp, hd, nullptr, 0, nullptr, 0, 0, 0, UFV_GEMM_AUTO, stream));
        TRY(ufv_gemm(dOg, hd, vg, (int)ldkv, dp, Sp, 1, M, Sp, hd, nullptr, 0, nullptr, 0, 0, 0, UFV_GEMM_AUTO, stream));
        hipLaunchKernelGGL(attn_softmax_bwd_k, dim3(M), dim3(256), 0, st, sc, dp, (int64_t)Sp, P, dS, (int64_t)Sp, S, Sp, scale, rep);
        UFV_CHECK_LAUNCH();
        TRY(ufv_gemm(dS, Sp, kT, Sp, dQg, hd, 0, M, hd, Sp, nullptr, 0, nullptr, 0, 0, 0, UFV_GEMM_AUTO, stream));
        hipLaunchKernelGGL(head_rows_k, dim3(grid_for((int64_t)M * hd / 8)), dim3(256), 0, st, dQg, (int64_t)hd, (bf16*)dq + (int64_t)g * rep * hd, lddq, S, rep, hd, 0);
        UFV_CHECK_LAUNCH();
        TRY(ufv_transpose_bf16(P, Sp, PT, Mp, M, Sp, Mp, stream));         // all Sp columns (the pad columns are zeros): 16-byte path
        TRY(ufv_transpose_bf16(dS, Sp, dST, Mp, M, Sp, Mp, stream));
        TRY(ufv_transpose_bf16(Qg, hd, QgT, Mp, M, hd, Mp, stream));
        TRY(ufv_transpose_bf16(dOg, hd, dOgT, Mp, M, hd, Mp, stream));
        TRY(ufv_gemm_splitk(PT, Mp, dOgT, Mp, (bf16*)dv + (int64_t)g * hd, (int)lddkv, 0, 0, S, hd, Mp, nsplit, part, stream));
        TRY(ufv_gemm_splitk(dST, Mp, QgT, Mp, (bf16*)dk + (int64_t)g * hd, (int)lddkv, 0, 0, S, hd, Mp, nsplit, part, stream));
    }
#undef TRY
    return UFV_OK;
}
