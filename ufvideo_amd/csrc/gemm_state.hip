// Per-device state of the split-K / stream-K GEMM forms -- the ONLY process-wide mutable state of the library besides the thread-local
// error string, and it is never shared between devices or between launches that can be in flight together:
//   * one record per HIP device (indexed by hipGetDevice), created under a mutex on the first split-K launch on that device or by
//     ufv_gemm_prepare() (call that before stream capture: creation allocates);
//   * a 1 Mi-int (4 MiB) ring of turn flags.  Every launch takes its OWN slice [off, off + tiles) and its own ticket base, so two split-K
//     GEMMs in flight on different streams never see each other's flags; a slice comes round again after >= 1 Mi / tiles launches
//     (thousands), and since ticket bases only grow a stale value can never satisfy a `>=` wait;
//   * a pinned host word the kernels set when a bounded turn wait expires (a part's predecessor was never scheduled: more blocks than
//     the device could keep resident beside some other kernel).  The launch then finishes with a wrong tile instead of hanging, and the
//     next SPLIT-K launch on that device returns UFV_EHIP with the reason (ufv_splitk_acquire reads the word; unsplit GEMMs are not gated);
//     ufv_gemm_error_state() reads it at any time.
#include "common.h"
#include "gemm_state.h"
#include <mutex>

namespace {

constexpr int kMaxDev = 64;
constexpr int kRing = 1 << 20;          // ints

struct DevState {
    std::mutex mu;
    bool ready = false;
    int n_cu = 0;
    int* flags = nullptr;               // [kRing], zero-initialised
    int head = 0;                       // next free slot of the ring
    int ticket = 0;                     // last ticket base handed out
    int* err = nullptr;                 // pinned, mapped host word (0 = fine)
    float* sk_ws = nullptr;             // stream-K (opt-in): accumulator slots + flags, one launch at a time per device
    int* sk_flags = nullptr;
    int sk_cap = 0, sk_epoch = 0;
};

DevState g_dev[kMaxDev];

DevState* current(int* dev_out = nullptr) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) {
        ufv_set_error("ufv_gemm: no current HIP device (or its index is >= %d)", kMaxDev);
        return nullptr;
    }
    if (dev_out) *dev_out = dev;
    return &g_dev[dev];
}

// caller holds s->mu
int ensure(DevState* s, int dev) {
    if (s->ready) return UFV_OK;
    hipDeviceProp_t prop;
    s->n_cu = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipMalloc(&s->flags, (size_t)kRing * sizeof(int)) != hipSuccess || hipMemset(s->flags, 0, (size_t)kRing * sizeof(int)) != hipSuccess ||
        hipHostMalloc(&s->err, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        ufv_set_error("ufv_gemm: could not allocate the split-K turn flags of device %d", dev);
        return UFV_EHIP;
    }
    *s->err = 0;
    s->ready = true;
    return UFV_OK;
}

}  // namespace

int ufv_dev_n_cu() {
    int dev = 0;
    DevState* s = current(&dev);
    if (!s) return 256;
    std::lock_guard<std::mutex> g(s->mu);
    if (s->n_cu == 0) {
        hipDeviceProp_t prop;
        s->n_cu = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return s->n_cu;
}

int ufv_splitk_acquire(int tiles, int parts, int** flags, int* base, int** err) {
    int dev = 0;
    DevState* s = current(&dev);
    if (!s) return UFV_EHIP;
    std::lock_guard<std::mutex> g(s->mu);
    const int rc = ensure(s, dev);
    if (rc != UFV_OK) return rc;
    if (*s->err != 0) {
        ufv_set_error("ufv_gemm: an earlier split-K launch on device %d gave up waiting for a tile's turn (code %d): its blocks were not all "
                      "resident (another kernel held CUs); that launch's output is wrong.  Clear with ufv_gemm_clear_error()", dev, *s->err);
        return UFV_EHIP;
    }
    if (tiles <= 0 || tiles > kRing || parts < 1 || parts > 32) {
        ufv_set_error("ufv_gemm: split-K over %d tiles x %d parts is outside the flag ring (%d tiles, 32 parts)", tiles, parts, kRing);
        return UFV_EUNSUPPORTED;
    }
    if (s->ticket > (1 << 30)) {
        // once per ~2^25 launches: start the tickets over.  Everything queued on the device must be done with the old values first.
        if (hipDeviceSynchronize() != hipSuccess || hipMemset(s->flags, 0, (size_t)kRing * sizeof(int)) != hipSuccess) return UFV_EHIP;
        s->ticket = 0; s->head = 0;
    }
    if (s->head + tiles > kRing) s->head = 0;
    *flags = s->flags + s->head;
    s->head += tiles;
    *base = s->ticket;                  // this launch's flags run base + 1 .. base + parts, all above anything a slot has ever held
    s->ticket += parts + 1;
    *err = s->err;
    return UFV_OK;
}

int ufv_streamk_acquire(int grid, float** ws, int** flags, int* epoch, int** err) {
    int dev = 0;
    DevState* s = current(&dev);
    if (!s) return UFV_EHIP;
    std::lock_guard<std::mutex> g(s->mu);
    const int rc = ensure(s, dev);
    if (rc != UFV_OK) return rc;
    if (grid > s->sk_cap) {
        if (s->sk_ws) (void)hipFree(s->sk_ws);
        if (s->sk_flags) (void)hipFree(s->sk_flags);
        s->sk_ws = nullptr; s->sk_flags = nullptr; s->sk_cap = 0;
        if (hipMalloc(&s->sk_ws, (size_t)grid * 262144) != hipSuccess || hipMalloc(&s->sk_flags, (size_t)grid * sizeof(int)) != hipSuccess ||
            hipMemset(s->sk_flags, 0, (size_t)grid * sizeof(int)) != hipSuccess) {
            ufv_set_error("ufv_gemm: could not allocate the stream-K workspace (%d x 256 KiB)", grid);
            return UFV_EHIP;
        }
        s->sk_cap = grid; s->sk_epoch = 0;
    }
    *ws = s->sk_ws; *flags = s->sk_flags; *epoch = ++s->sk_epoch; *err = s->err;
    return UFV_OK;
}

extern "C" int ufv_gemm_prepare(void) {
    int dev = 0;
    DevState* s = current(&dev);
    if (!s) return UFV_EHIP;
    std::lock_guard<std::mutex> g(s->mu);
    return ensure(s, dev);
}

extern "C" int ufv_gemm_error_state(void) {
    DevState* s = current();
    if (!s) return UFV_EHIP;
    std::lock_guard<std::mutex> g(s->mu);
    return s->ready ? *s->err : 0;
}

extern "C" int ufv_gemm_clear_error(void) {
    DevState* s = current();
    if (!s) return UFV_EHIP;
    std::lock_guard<std::mutex> g(s->mu);
    if (s->ready) *s->err = 0;
    return UFV_OK;
}
