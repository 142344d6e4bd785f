// Lab: what a global_store_dwordx4 costs by ADDRESS PATTERN, every CU storing at once (the GEMM / attention epilogue situation).  One 512-thread block per CU,
// each wave issues 16 stores of 1 KB per "tile" (128 KB per block and tile, as the 256x256 bf16 GEMM epilogue does) for T tiles; the rows of one store
// instruction are  P rows x (1024 / P) bytes  with row pitch `ld` bytes:  P = 16 (the MFMA fragment layout: 16 rows x 64 B), 4, 2, 1 (one contiguous KB).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/store_probe.hip -o tools/lab/store_probe && tools/lab/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int P>
__global__ __launch_bounds__(512) void store_k(char* out, long ld, int tiles, long tile_stride, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = 64 / P;                         // lanes per row
    const int row = lane / LPR, chunk = lane % LPR;
    i32x4 v = {lane, wave, (int)blockIdx.x, 7};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        // block tile = 256 rows x 512 B; wave w owns rows [32 w, 32 w + 32) in 16 instructions
        char* base = out + ((long)blockIdx.x * tiles + t) * tile_stride;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // instruction i covers P rows x (1024 / P) bytes:  rows (i * P + row) within the wave's 16 * P row slots of width 1024 / P
            const long r = (long)wave * 16 * P + i * P + row;
            char* p = base + (r % 256) * ld + (r / 256) * (1024 / P) + chunk * 16;
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int blocks = 256, tiles = 8;
    const long ld = 6912;                               // row pitch in bytes (the ViT qkv output: 3456 bf16)
    const long tile_stride = 256 * ld;                  // consecutive tiles of a block: the next 256 rows
    const size_t bytes = (size_t)blocks * tiles * tile_stride + (1 << 20);
    char* out; unsigned long long* cyc;
    if (hipMalloc(&out, bytes) != hipSuccess || hipMalloc(&cyc, blocks * 8) != hipSuccess) return 1;
    (void)hipMemset(out, 0, bytes);
    auto run = [&](int P, const char* name) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float best = 1e9f; double cmean = 0;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0);
            if (P == 16) hipLaunchKernelGGL(store_k<16>, dim3(blocks), dim3(512), 0, 0, out, ld, tiles, tile_stride, cyc);
            if (P == 4) hipLaunchKernelGGL(store_k<4>, dim3(blocks), dim3(512), 0, 0, out, ld, tiles, tile_stride, cyc);
            if (P == 2) hipLaunchKernelGGL(store_k<2>, dim3(blocks), dim3(512), 0, 0, out, ld, tiles, tile_stride, cyc);
            if (P == 1) hipLaunchKernelGGL(store_k<1>, dim3(blocks), dim3(512), 0, 0, out, ld, tiles, tile_stride, cyc);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
            unsigned long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            cmean = 0; for (int i = 0; i < blocks; ++i) cmean += (double)h[i] / blocks;
        }
        const double mb = (double)blocks * tiles * 128.0 * 1024 / 1e6;
        printf("%-28s %7.1f us  %6.2f TB/s  | %8.0f cycles per block = %6.0f per 128 KB tile = %5.1f per store instruction per CU (%4.1f B/clk/CU)\n", name, best * 1e3,
               mb / (best * 1e3) , cmean, cmean / tiles, cmean / tiles / 128.0, 131072.0 * tiles / cmean);
    };
    run(16, "16 rows x 64 B (MFMA layout)");
    run(4, "4 rows x 256 B");
    run(2, "2 rows x 512 B");
    run(1, "1 row x 1024 B (contiguous)");
    return 0;
}
