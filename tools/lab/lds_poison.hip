// Lab: fill the LDS of every CU with a bit pattern (NaNs, infinities) so that a kernel which reads LDS it never wrote shows it.  hipcc --offload-arch=gfx950 -shared -fPIC
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void lds_poison_k(unsigned pattern, unsigned* sink) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) lds[i] = pattern;
    __syncthreads();
    if (sink && lds[(blockIdx.x * 97) % (160 * 256)] == 12345u) sink[0] = 1;      // keep the stores
    for (volatile int spin = 0; spin < 2000; ++spin) {}                            // stay resident long enough that every CU takes a block
}
extern "C" int lds_poison(unsigned pattern, void* sink, void* stream) {
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_poison_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
    hipLaunchKernelGGL(lds_poison_k, dim3(1024), dim3(256), 160 * 1024, (hipStream_t)stream, pattern, (unsigned*)sink);
    return (int)hipGetLastError();
}
