// `make asan`: the host side of libufv_hip.so under AddressSanitizer + UBSan on the CPU (tests/asan/hip_stub.cpp stands in for the HIP runtime).
// What is exercised: the cost model behind UFV_GEMM_AUTO at the clip's shapes, ufv_gemm's argument checks and dispatch (bf16 / SwiGLU / residual / split-K /
// half-tile-item launches: grid sizes as the device would see them), the split-K flag ring across its wrap-around and from two threads at once, the error
// word (a timed-out turn gates the next split-K launch, unsplit launches go on, clear resets), the no-device path, the thread-local error string, the
// whole-stage calls' workspace sizes.  Exit status 0 = every check held and the sanitizers reported nothing (they abort the process otherwise).
#include "../../include/ufv.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <cstdint>

extern "C" long ufv_stub_launches(void);
extern "C" void ufv_stub_last_launch(unsigned* grid, unsigned* block);
extern "C" void ufv_stub_set_device_ok(int ok);

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "host_logic.cpp:%d: CHECK failed: %s  [last error: %s]\n", __LINE__, #c, ufv_last_error()); ++g_fail; } } while (0)

static void* dev(size_t bytes) { void* p = std::aligned_alloc(256, (bytes + 255) / 256 * 256); std::memset(p, 0, (bytes + 255) / 256 * 256); return p; }

static int gemm(int M, int N, int K, int out_f32, int swiglu, bool resid, int kernel, void* A, void* W, void* C, float* R) {
    return ufv_gemm(A, K, W, K, C, swiglu ? N / 2 : N, out_f32, M, N, K, nullptr, UFV_ACT_NONE, resid ? R : nullptr, resid ? N : 0, 0, swiglu, kernel, nullptr);
}

int main() {
    if (std::getenv("UFV_ASAN_SELFTEST")) {        // tests/test_asan_host.py: the build really is instrumented -- a one-element overrun must end the process
        volatile int* p = static_cast<int*>(std::malloc(16));
        p[4] = 1;
        std::printf("selftest: the overrun went unnoticed\n");
        return 0;
    }
    CHECK(ufv_abi_version() == UFV_ABI_VERSION);
    // ---- cost model (tests/test_host_cpu.py pins the same picks through ctypes; here they run under the sanitizers)
    CHECK(ufv_gemm_choice(2399, 37888, 3584, 0, 1, 1) == 1442);
    CHECK(ufv_gemm_choice(18432, 1152, 1152, 1, 0, 1) == 1431 && ufv_gemm_choice(18432, 1152, 4352, 1, 0, 1) == 1431);
    CHECK(ufv_gemm_choice(2399, 3584, 3584, 1, 0, 1) == 1331 && ufv_gemm_choice(2399, 3584, 18944, 1, 0, 1) == 1331);
    CHECK(ufv_gemm_choice(1200, 3584, 18944, 1, 0, 1) / 10000 >= 4 && ufv_gemm_choice(1200, 3584, 18944, 1, 0, 0) < 10000);
    for (int M = 1; M < 30000; M += 997)                      // no shape makes the model read outside its tables
        for (int N : {64, 1152, 3456, 3584, 4352, 4608, 37888, 151808})
            for (int K : {64, 640, 1152, 3584, 4352, 18944, 28672}) (void)ufv_gemm_choice(M, N, K, M & 1, N == 37888, 1);
    CHECK(ufv_gemm_qkv_rope_shape(2399, 28, 4, 128, 3584) == 1332 && ufv_gemm_qkv_rope_shape(100, 28, 4, 128, 3584) == 0);

    const int MAXM = 2799, MAXN = 37888, MAXK = 3584;
    void* A = dev((size_t)MAXM * 18944 * 2); void* W = dev((size_t)MAXN * MAXK * 2); void* C = dev((size_t)MAXM * MAXN * 4);
    float* R = static_cast<float*>(dev((size_t)MAXM * 3584 * 4));
    unsigned grid[3], block[3];
    // ---- dispatch: the launches the clip issues, and what the device would have been asked to run
    long n0 = ufv_stub_launches();
    CHECK(gemm(2399, 37888, 3584, 0, 1, false, UFV_GEMM_AUTO, A, W, C, R) == 0);           // gate/up: persistent, one block per CU
    ufv_stub_last_launch(grid, block);
    CHECK(ufv_stub_launches() == n0 + 1 && grid[0] == 256 && block[0] == 512);
    CHECK(gemm(2799, 37888, 3584, 0, 1, false, UFV_GEMM_AUTO, A, W, C, R) == 0);           // 384 px
    CHECK(gemm(2399, 3584, 3584, 1, 0, true, UFV_GEMM_AUTO, A, W, C, R) == 0);             // o_proj: 247 tiles of 192 x 192
    ufv_stub_last_launch(grid, block);
    CHECK(grid[0] == 247);
    CHECK(gemm(2399, 3584, 18944, 1, 0, true, UFV_GEMM_AUTO, A, W, C, R) == 0);            // down
    CHECK(gemm(100, 3584, 3584, 1, 0, true, UFV_GEMM_AUTO, A, W, C, R) == 0);              // small M: the 128-wide kernels
    CHECK(gemm(0, 3584, 3584, 1, 0, false, UFV_GEMM_AUTO, A, W, C, R) != 0);               // refused, with a message
    CHECK(std::strlen(ufv_last_error()) > 0);
    CHECK(gemm(2399, 3584, 3584, 1, 0, false, UFV_GEMM_AUTO, nullptr, W, C, R) != 0);
    CHECK(gemm(2399, 3500, 3584, 0, 1, false, UFV_GEMM_FAST256, A, W, C, R) != 0);          // SwiGLU needs N % 256 in {0, 128}
    // ---- split-K: the flag ring hands every launch its own slice and ticket base; walk it past its wrap-around (1 Mi flags / 70 tiles per launch)
    CHECK(ufv_gemm_prepare() == 0 && ufv_gemm_error_state() == 0);
    const int prev = ufv_gemm_set_splitk(1);
    CHECK(ufv_gemm_choice(1200, 3584, 18944, 1, 0, 1) >= 10000);
    for (int i = 0; i < 16000; ++i) CHECK(gemm(1200, 3584, 18944, 1, 0, true, UFV_GEMM_AUTO, A, W, C, R) == 0);
    // ... from two threads at once (the record is mutex-guarded; every thread has its own error string)
    {
        std::vector<std::thread> th;
        int bad[2] = {0, 0};
        for (int t = 0; t < 2; ++t)
            th.emplace_back([&, t] {
                for (int i = 0; i < 4000; ++i) bad[t] += gemm(1200, 3584, 18944, 1, 0, true, UFV_GEMM_AUTO, A, W, C, R) != 0;
                bad[t] += gemm(0, 1, 1, 0, 0, false, UFV_GEMM_AUTO, A, W, C, R) == 0;           // this thread's own error message
                bad[t] += std::strstr(ufv_last_error(), "ufv_gemm") == nullptr;
            });
        for (auto& x : th) x.join();
        CHECK(bad[0] == 0 && bad[1] == 0);
    }
    (void)ufv_gemm_set_splitk(prev);
    CHECK(ufv_gemm_clear_error() == 0 && ufv_gemm_error_state() == 0);
    // ---- no current device: every entry that needs the record says so instead of touching it
    ufv_stub_set_device_ok(0);
    CHECK(ufv_gemm_prepare() != 0 && ufv_gemm_error_state() != 0 && std::strstr(ufv_last_error(), "no current HIP device") != nullptr);
    ufv_stub_set_device_ok(1);
    // ---- workspace arithmetic of the small entry points
    CHECK(ufv_attention_decode_ws_bytes(1, 28, 128, 16) > 0 && ufv_attention_decode_fused_ws_bytes(28, 128, 16) > 0 && ufv_argmax_ws_bytes() > 0);
    CHECK(ufv_attention_bwd_ws_bytes(2399, 28, 4, 128) > 0 && ufv_rmsnorm_bwd_ws_bytes(3584) > 0 && ufv_layernorm_bwd_ws_bytes(1152) > 0);
    // ---- the whole-stage calls (csrc/stages.hip): workspace carving and the launch sequence of a tower / connector / prefill, with a workspace of EXACTLY the
    //      advertised size (one byte less is refused) -- dummy weights, every launch dropped by the stub; the sanitizers watch the host side of each call
    {
        const int D = 1152, L = 3, T = 2, NP = 729, IP = 4352, KP = 640;
        float* fz = static_cast<float*>(dev((size_t)NP * D * 4));
        void* wz = dev((size_t)IP * D * 2);
        std::vector<ufv_vit_layer> vl(L);
        for (auto& l : vl) l = ufv_vit_layer{fz, fz, fz, fz, wz, fz, wz, fz, wz, fz, wz, fz};
        ufv_vit_model vm{};
        vm.n_layers = L; vm.d = D; vm.n_heads = 16; vm.d_ff_pad = IP; vm.patch = 14; vm.channels = 3; vm.kpad = KP; vm.n_patches = NP; vm.act = UFV_ACT_GELU_TANH; vm.eps = 1e-6f;
        vm.patch_w = wz; vm.patch_b = fz; vm.pos = fz; vm.layers = vl.data();
        const int64_t wsb = ufv_vit_forward_ws_bytes(&vm, T);
        CHECK(wsb > 0);
        void* ws = dev((size_t)wsb); void* px = dev((size_t)T * 3 * 384 * 384 * 2); float* x = static_cast<float*>(dev((size_t)T * NP * D * 4));
        const long l0 = ufv_stub_launches();
        CHECK(ufv_vit_forward(&vm, px, 1 /* bf16 */, T, 384, 384, L - 1, x, ws, wsb, nullptr) == 0);          // 384 px: 27 x 27 patches, the 6 remainder pixels dropped
        CHECK(ufv_stub_launches() - l0 >= 2 + 7 * (L - 1));
        CHECK(ufv_vit_forward(&vm, px, 1, T, 384, 384, L - 1, x, ws, wsb - 1, nullptr) != 0);                  // workspace one byte short
        CHECK(ufv_vit_forward(&vm, px, 1, T, 336, 336, L - 1, x, ws, wsb, nullptr) != 0);                      // 576 patches into a 729-patch tower
        CHECK(ufv_vit_forward(&vm, px, 1, T, 384, 384, L + 1, x, ws, wsb, nullptr) != 0);
        std::free(ws); std::free(px); std::free(x); std::free(fz); std::free(wz);
    }
    {
        const int D = 3584, L = 2, S = 300, FF = 18944, V = 4096, ML = 512;
        float* fz = static_cast<float*>(dev((size_t)4608 * 4 + D * 4));
        void* wz = dev((size_t)2 * FF * D * 2);
        void* kv = dev((size_t)ML * 1024 * 2);
        std::vector<ufv_qwen2_layer> ql(L);
        for (auto& l : ql) { std::memset(&l, 0, sizeof(l)); l.wqkv = wz; l.bqkv = fz; l.wo = wz; l.wgu = wz; l.wd = wz; l.ln1 = fz; l.ln2 = fz; l.kv_cache = kv; }
        ufv_qwen2_model qm{};
        qm.n_layers = L; qm.d = D; qm.n_q = 28; qm.n_kv = 4; qm.hd = 128; qm.d_ff = FF; qm.vocab = V; qm.ldkv = 1024; qm.max_len = ML; qm.attn_splits = 16; qm.eps = 1e-6f;
        qm.inv_freq = fz; qm.norm = fz; qm.embed = wz; qm.lm_head = wz; qm.layers = ql.data();
        const int64_t wsb = ufv_qwen2_prefill_ws_bytes(&qm, S);
        CHECK(wsb > 0);
        void* ws = dev((size_t)wsb); float* x = static_cast<float*>(dev((size_t)S * D * 4)); float* lg = static_cast<float*>(dev((size_t)V * 4));
        CHECK(ufv_qwen2_prefill(&qm, x, S, 0, ws, wsb, nullptr, nullptr, lg, nullptr) == 0);
        CHECK(ufv_qwen2_prefill(&qm, x, S, 0, ws, wsb - 1, nullptr, nullptr, lg, nullptr) != 0);
        CHECK(ufv_qwen2_prefill(&qm, x, S, ML - 10, ws, wsb, nullptr, nullptr, lg, nullptr) != 0);            // past the cache's rows
        const int64_t dwb = ufv_qwen2_decode_ws_bytes(&qm);
        void* dws = dev((size_t)dwb);
        int64_t* tok = static_cast<int64_t*>(dev(64));
        CHECK(ufv_qwen2_decode_step(&qm, tok, S, dws, dwb, lg, nullptr, tok + 1, nullptr) == 0);
        CHECK(ufv_qwen2_decode_step(&qm, tok, ML, dws, dwb, lg, nullptr, tok + 1, nullptr) != 0);              // position past the cache
        std::free(ws); std::free(x); std::free(lg); std::free(dws); std::free(tok); std::free(fz); std::free(wz); std::free(kv);
    }
    std::free(A); std::free(W); std::free(C); std::free(R);
    std::printf("host_logic: %ld launches through the stub, %d failed checks\n", ufv_stub_launches(), g_fail);
    return g_fail ? 1 : 0;
}
