// Diagnostic (not product): dumps the lane mapping of ds_read_b64_tr_b16 and both bf16 MFMA shapes.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(float* out) {
    __shared__ __attribute__((aligned(16))) bf16 t[16 * 64];   // [16 rows][64 cols], value = row*64+col (exact in bf16 up to 256: use row*16+col%16.. keep small)
    for (int i = threadIdx.x; i < 16 * 64; i += 64) t[i] = (bf16)(float)((i / 64) * 16 + (i % 64) % 16 + ((i % 64) / 16) * 0);
    __syncthreads();
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    // block = rows 4*(g>>1)..+3, cols 16*(g&1)..+15 ; lane 4q+p supplies row q, cols 4p..4p+3
    const bf16* p = t + (4 * (g >> 1) + (i >> 2)) * 64 + 16 * (g & 1) + 4 * (i & 3);
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (float)v[j];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4 * 4);
    k<<<1, 64>>>(d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // expected under the guide's semantics: lane (g,i) element q = T[4*(g>>1)+q][16*(g&1)+i] = (4*(g>>1)+q)*16 + i
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) {
        int g = l >> 4, i = l & 15; float e = (4 * (g >> 1) + q) * 16 + i;
        if (h[l * 4 + q] != e) { if (bad < 16) printf("lane %d elem %d got %g expected %g\n", l, q, h[l * 4 + q], e); ++bad; }
    }
    printf("tr16 probe: %d mismatches\n", bad);
    return bad != 0;
}
