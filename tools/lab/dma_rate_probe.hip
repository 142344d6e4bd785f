// Lab: how many bytes per clock one CU moves from L2 into LDS with `global_load_lds_dwordx4` (the GEMM's operand path), against plain `global_load_dwordx4` into
// registers -- the number that decides whether the 192-row tile shapes (72-96 flop per staged byte) are bound by the operand path rather than by MFMA.
// One 512-thread block per CU (8 waves, like the GEMM); every wave requests `pieces` x 1 KB per round from a 64 KB window that all rounds re-read (L2 / L1 resident
// after the first pass, like the K-loop's panels shared by neighbouring CUs), ROUNDS rounds, counted waits that keep 8 requests in flight per wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/dma_rate_probe.hip -o tools/lab/dma_rate_probe && tools/lab/dma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) int i32x4;
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int MODE>   // 0: LDS-DMA 16 B per lane; 1: register loads 16 B per lane; 2: LDS-DMA 4 B per lane
__global__ __launch_bounds__(512) void rate_k(const char* src, long cu_stride, int rounds, int window, unsigned long long* cyc, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = src + (long)blockIdx.x * cu_stride;
    i32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // piece = 8 rows x 128 B (the GEMM's staging piece): lane -> (row lane >> 3, 16-byte chunk lane & 7); rows 256 B apart as in a K-major panel
            const int piece = (r * 8 + i) * 8 + wave;
            const unsigned off = ((unsigned)(piece * 8 + (lane >> 3)) * 256u + (lane & 7) * 16u) & (unsigned)(window - 1);      // window: a power of two
            if (MODE == 0) __builtin_amdgcn_global_load_lds(GLB_PTR(base + off), LDS_PTR(smem + wave * 8192 + i * 1024), 16, 0, 0);
            else if (MODE == 2) __builtin_amdgcn_global_load_lds(GLB_PTR(base + off), LDS_PTR(smem + wave * 8192 + i * 1024), 4, 0, 0);
            else {
                i32x4 v;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(base + off) : "memory");
                asm volatile("s_waitcnt vmcnt(7)" : "+v"(v)::"memory");
                acc += v;
            }
        }
        if (MODE != 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc[0] == 0x12345678) sink[0] = acc[1];
}

int main() {
    const int blocks = 256, rounds = 400;
    const long cu_stride = 1 << 20;
    char* src; unsigned long long* cyc; int* sink;
    if (hipMalloc(&src, (size_t)blocks * cu_stride) != hipSuccess || hipMalloc(&cyc, blocks * 8) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    (void)hipMemset(src, 1, (size_t)blocks * cu_stride);
    auto run = [&](int mode, int window, const char* name) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rate_k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rate_k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rate_k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float best = 1e9f; double cmean = 0;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(rate_k<0>, dim3(blocks), dim3(512), 65536, 0, src, cu_stride, rounds, window, cyc, sink);
            if (mode == 1) hipLaunchKernelGGL(rate_k<1>, dim3(blocks), dim3(512), 65536, 0, src, cu_stride, rounds, window, cyc, sink);
            if (mode == 2) hipLaunchKernelGGL(rate_k<2>, dim3(blocks), dim3(512), 65536, 0, src, cu_stride, rounds, window, cyc, sink);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
            unsigned long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            cmean = 0; for (int i = 0; i < blocks; ++i) cmean += (double)h[i] / blocks;
        }
        const double per_instr = mode == 2 ? 256.0 : 1024.0;
        const double bytes = (double)rounds * 8 * 8 * per_instr;             // per CU
        printf("%-34s window %4d KB: %8.1f us  %7.0f ticks(100 MHz) per CU -> %6.1f B / ns / CU = %5.2f TB/s chip-wide; %5.1f ns per wave-instruction per CU\n", name,
               window >> 10, best * 1e3, cmean, bytes / (best * 1e6), bytes * blocks / (best * 1e-3) / 1e12, best * 1e6 / (rounds * 64.0));
        fflush(stdout);
    };
    for (int window : {65536, 1 << 20}) {
        run(0, window, "LDS-DMA dwordx4 (16 B / lane)");
        run(2, window, "LDS-DMA dword   ( 4 B / lane)");
        run(1, window, "global_load_dwordx4 -> VGPR");
    }
    return 0;
}
