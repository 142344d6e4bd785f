// Lab: what a grid-wide barrier costs on MI355X (256 CUs in 8 XCDs, one L2 per XCD) and what a persistent weight-streaming kernel reaches with barriers between
// its phases -- the two numbers that decide whether the decode step should be ONE kernel instead of ~140 launches per token.
//   (1) barrier only: B blocks per CU x 256 CUs, N barriers back to back (arrive: agent-scope atomic add; wait: one thread polls with an L2-bypassing load)
//   (2) phases of a GEMV-like stream (each wave reads its share of `bytes` of weights, 16 x 1 KB requests in flight) separated by a barrier and a small
//       coherent exchange (every block publishes 64 B, reads the whole 14 KB vector), with and without requesting the next phase's first buffers BEFORE the barrier
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/grid_barrier_probe.hip -o tools/lab/grid_barrier_probe && tools/lab/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ unsigned load_sc(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}

// every thread's stores must be acknowledged before the block arrives; returns false on a timeout (never hangs the box)
__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned target, int* failed) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        ok = 1;
        while (load_sc(bar) < target) {
            if (++spins > 2000000) { ok = 0; *failed = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return ok != 0;
}

__global__ __launch_bounds__(512) void barrier_only_k(unsigned* bar, int n, int* failed, unsigned long long* cyc) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 1; i <= n; ++i)
        if (!grid_barrier(bar, (unsigned)i * gridDim.x, failed)) return;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// phases: each phase streams `rows` rows of `rowbytes` bytes (row r belongs to global wave r % total_waves), then publishes + barrier + reads the vector
template <bool PREFETCH>
__global__ __launch_bounds__(512) void phases_k(const char* W, long phase_stride, int rows, int rowbytes, int phases, float* vec, unsigned* bar, int* failed,
                                                float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int gw = blockIdx.x * nw + wave, total = gridDim.x * nw;
    __shared__ float xs[3584];
    float acc = 0.f;
    i32x4 buf[16];
    int have = 0;                                              // a prefetched first batch is in `buf`
    for (int ph = 0; ph < phases; ++ph) {
        const char* Wp = W + (long)ph * phase_stride;
        for (int r = gw; r < rows; r += total) {
            const char* row = Wp + (long)r * rowbytes;
            for (int k0 = 0; k0 < rowbytes; k0 += 16384) {
                if (!(have && r == gw && k0 == 0)) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int k = k0 + i * 1024 + lane * 16;
                        buf[i] = *reinterpret_cast<const i32x4*>(row + (k < rowbytes ? k : lane * 16));
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (k0 + i * 1024 + lane * 16 < rowbytes) acc += __int_as_float(buf[i][0] ^ buf[i][1]) * xs[(lane + i) & 1023] + __int_as_float(buf[i][2] & buf[i][3]);
            }
        }
        have = 0;
        if (PREFETCH && ph + 1 < phases && gw < rows) {        // the next phase's first batch goes out before the barrier
            const char* row = W + (long)(ph + 1) * phase_stride + (long)gw * rowbytes;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = i * 1024 + lane * 16;
                buf[i] = *reinterpret_cast<const i32x4*>(row + (k < rowbytes ? k : lane * 16));
            }
            have = 1;
        }
        // publish 16 floats per block, barrier, read the whole vector coherently
        if (threadIdx.x < 14) __hip_atomic_store(vec + blockIdx.x * 14 + threadIdx.x, acc + ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!grid_barrier(bar, (unsigned)(ph + 1) * gridDim.x, failed)) return;
        for (int i = threadIdx.x; i < 3584; i += blockDim.x) {
            float v;
            asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(vec + i) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory");
            xs[i] = v;
        }
        __syncthreads();
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    int* failed; unsigned* bar; unsigned long long* cyc; float* vec; float* sink;
    (void)hipMalloc(&failed, 4); (void)hipMalloc(&bar, 4); (void)hipMalloc(&cyc, 8 * 1024); (void)hipMalloc(&vec, 4 * 4096); (void)hipMalloc(&sink, 4);
    (void)hipMemset(failed, 0, 4); (void)hipMemset(vec, 0, 4 * 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int bpc : {1, 2}) {
        for (int threads : {256, 512}) {
            const int n = 2000, blocks = 256 * bpc;
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipMemset(bar, 0, 4);
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(barrier_only_k, dim3(blocks), dim3(threads), 0, 0, bar, n, failed, cyc);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            int f; (void)hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost);
            printf("barrier only: %d blocks x %d threads: %.2f us per barrier%s\n", blocks, threads, best * 1e3 / n, f ? "  (TIMED OUT)" : "");
            if (f) return 1;
        }
    }
    // phases: the decode step's weight matrices in sequence -- per layer qkv 33 MB, o 25.7 MB, gate/up 271.6 MB, down 135.8 MB; 8 layers = 3.7 GB
    struct Ph { int rows, rowbytes; };
    const Ph layer[4] = {{4608, 7168}, {3584, 7168}, {37888, 7168}, {3584, 37888}};
    for (int which = 0; which < 4; ++which) {
        const int phases = 28;
        const long stride = (((long)layer[which].rows * layer[which].rowbytes) + 4095) / 4096 * 4096;
        char* W;
        if (hipMalloc(&W, stride * phases + (1 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
        (void)hipMemset(W, 1, stride * phases);
        for (int pre = 0; pre < 2; ++pre)
            for (int threads : {256, 512}) {
                for (int bpc : {1, 2}) {
                    if (threads == 512 && bpc == 2) continue;
                    float best = 1e9f;
                    for (int rep = 0; rep < 3; ++rep) {
                        (void)hipMemset(bar, 0, 4);
                        (void)hipEventRecord(e0);
                        if (pre) hipLaunchKernelGGL(phases_k<true>, dim3(256 * bpc), dim3(threads), 0, 0, W, stride, layer[which].rows, layer[which].rowbytes, phases, vec, bar, failed, sink);
                        else hipLaunchKernelGGL(phases_k<false>, dim3(256 * bpc), dim3(threads), 0, 0, W, stride, layer[which].rows, layer[which].rowbytes, phases, vec, bar, failed, sink);
                        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                        if (ms < best) best = ms;
                    }
                    int f; (void)hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost);
                    const double mb = (double)layer[which].rows * layer[which].rowbytes / 1e6;
                    printf("phase %d rows x %d B (%.1f MB)  %s  %d blocks x %d threads: %.2f us per phase  %.2f TB/s%s\n", layer[which].rows, layer[which].rowbytes, mb,
                           pre ? "prefetch  " : "no prefetch", 256 * bpc, threads, best * 1e3 / phases, mb / (best * 1e3 / phases), f ? "  (TIMED OUT)" : "");
                    if (f) return 1;
                }
            }
        (void)hipFree(W);
    }
    return 0;
}
