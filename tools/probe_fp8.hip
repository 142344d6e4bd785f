// Diagnostic (not product): operand lane map of v_mfma_f32_16x16x128_f8f6f4 with e4m3 (OCP) operands, exact integer data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const uint8_t* A, const uint8_t* B, float* out, int hyp) {   // A [16][128], B [128][16] (as Bt [16][128])
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    union { i32x8 v; uint8_t b[32]; } a, b;
    for (int j = 0; j < 32; ++j) {
        int kk = hyp == 0 ? 32 * g + j : (j < 16 ? 16 * g + j : 64 + 16 * g + (j - 16));
        a.b[j] = A[r * 128 + kk];
        b.b[j] = B[r * 128 + kk];
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a.v, b.v, c, 0, 0, 0, 0, 0, 0);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = c[j];
}
static uint8_t enc(int v) {   // OCP e4m3fn of small integers
    static const uint8_t t[5] = {0x00, 0x38, 0x40, 0x44, 0x48};
    return v < 0 ? (t[-v] | 0x80) : t[v];
}
int main() {
    uint8_t hA[16 * 128], hB[16 * 128]; int iA[16][128], iB[128][16];
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 128; ++kk) { iA[i][kk] = ((i * 7 + kk * 3) % 5) - 2; hA[i * 128 + kk] = enc(iA[i][kk]); }
    for (int kk = 0; kk < 128; ++kk) for (int j = 0; j < 16; ++j) { iB[kk][j] = ((kk * 5 + j * 11) % 7) - 3; hB[j * 128 + kk] = enc(iB[kk][j]); }
    uint8_t *dA, *dB; float* d;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&d, 256 * 4);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    int ok_any = 0;
    for (int hyp = 0; hyp < 2; ++hyp) {
        k<<<1, 64>>>(dA, dB, d, hyp);
        float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad_n = 0, bad_t = 0;     // n: C[row=(lane>>4)*4+reg][col=lane&15] = sum_k A[row][k] B[k][col];  t: transposed roles
        for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) {
            int row = (l >> 4) * 4 + q, col = l & 15; long e = 0, et = 0;
            for (int kk = 0; kk < 128; ++kk) { e += iA[row][kk] * iB[kk][col]; et += iA[col][kk] * iB[kk][row]; }
            if (h[l * 4 + q] != (float)e) ++bad_n;
            if (h[l * 4 + q] != (float)et) ++bad_t;
        }
        printf("hyp %d (k = %s): C[row=4*(lane>>4)+reg][col=lane&15] = A(first operand rows) x B(second operand cols): %d mismatches; swapped roles: %d mismatches; sample c[0..3] lane0 = %g %g %g %g\n",
               hyp, hyp == 0 ? "32*(lane>>4)+j" : "two 64-wide halves", bad_n, bad_t, h[0], h[1], h[2], h[3]);
        if (bad_n == 0 || bad_t == 0) ok_any = 1;
    }
    return !ok_any;
}
