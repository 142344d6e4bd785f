// Stand-in for the HIP runtime, for `make asan` only: the library's HOST code (argument checks, the cost model behind UFV_GEMM_AUTO, the split-K flag ring
// and its error word, the whole-stage C calls' workspace carving and launch sequences) compiled with -fsanitize=address,undefined runs on a CPU-only machine
// against these functions.  Device memory is host memory (so ASAN sees every byte the host code touches through it), a kernel launch is counted and dropped.
// NOT part of the product: libufv_hip.so links the real libamdhip64 and never sees this file.
#include <hip/hip_runtime_api.h>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>

namespace {
std::atomic<long> g_launches{0};
std::atomic<long> g_allocs{0};
thread_local dim3 t_grid, t_block;
thread_local size_t t_shmem;
thread_local hipStream_t t_stream;
unsigned g_last_grid[3], g_last_block[3];
std::mutex g_mu;
int g_device_ok = 1;
}  // namespace

extern "C" {
long ufv_stub_launches(void) { return g_launches.load(); }
long ufv_stub_live_allocations(void) { return g_allocs.load(); }
void ufv_stub_last_launch(unsigned* grid, unsigned* block) { std::lock_guard<std::mutex> g(g_mu); for (int i = 0; i < 3; ++i) { grid[i] = g_last_grid[i]; block[i] = g_last_block[i]; } }
void ufv_stub_set_device_ok(int ok) { g_device_ok = ok; }

hipError_t hipGetDevice(int* d) { if (!g_device_ok) return hipErrorNoDevice; *d = 0; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) { std::memset(p, 0, sizeof(*p)); p->multiProcessorCount = 256; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = std::calloc(1, n ? n : 1); if (*p) ++g_allocs; return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipFree(void* p) { if (p) { std::free(p); --g_allocs; } return hipSuccess; }
hipError_t hipMemset(void* p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "hip_stub"; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*) { return hipErrorNotSupported; }
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, hipGraphNode_t*, char*, size_t) { return hipErrorNotSupported; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }

hipError_t hipLaunchKernel(const void*, dim3 grid, dim3 block, void**, size_t, hipStream_t) {
    ++g_launches;
    std::lock_guard<std::mutex> g(g_mu);
    g_last_grid[0] = grid.x; g_last_grid[1] = grid.y; g_last_grid[2] = grid.z; g_last_block[0] = block.x; g_last_block[1] = block.y; g_last_block[2] = block.z;
    return hipSuccess;
}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t st) { t_grid = grid; t_block = block; t_shmem = shmem; t_stream = st; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* st) { *grid = t_grid; *block = t_block; *shmem = t_shmem; *st = t_stream; return hipSuccess; }
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void**) {}
}
