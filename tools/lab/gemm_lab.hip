// Stand-alone lab for the ping-pong GEMM (not part of the product): builds csrc/gemm256.hip into one executable with an s_memtime
// timeline of one steady-state K-tile (UFV_GSTAMP: waves 0 and 4 of every block, K-tile 8 -- or 100 of a long loop -- of the block's first tile), times a
// shape and prints the median step durations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/lab/gemm_lab.hip -o tools/lab/gemm_lab && tools/lab/gemm_lab M N K shape [f32res]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cstring>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <chrono>
__device__ unsigned long long* g_stamps;      // [blocks][2][16]
#define UFV_GSTAMP_DECL unsigned long long gs0 = 0, gs1 = 0, gs2 = 0, gs3 = 0, gs4 = 0, gs5 = 0, gs6 = 0, gs7 = 0, gs8 = 0, gs9 = 0, gs10 = 0, gs11 = 0, gs12 = 0, gs13 = 0; \
    const bool gs_on = g_stamps != nullptr && round == 1; if (gs_on) gs13 = __builtin_amdgcn_s_memtime();
// the sampled K-tile: the 100th of a long K loop (steady state: the first ones of a launch wait for cold HBM pages), else the 9th
#define UFV_GSTAMP(i) do { if (gs_on && tt == (len > 120 ? 100 : 8)) gs##i = __builtin_amdgcn_s_memtime(); } while (0)
#define UFV_GSTAMP_FLUSH do { if (gs_on && lane == 0 && (wave & 3) == 0) { unsigned long long* p_ = g_stamps + ((size_t)blockIdx.x * 2 + (wave >> 2)) * 16; \
    p_[0] = gs0; p_[1] = gs1; p_[2] = gs2; p_[3] = gs3; p_[4] = gs4; p_[5] = gs5; p_[6] = gs6; p_[7] = gs7; p_[8] = gs8; p_[9] = gs9; p_[10] = gs10; p_[11] = gs11; \
    p_[12] = gs12; p_[13] = gs13; p_[14] = __builtin_amdgcn_s_memtime(); p_[15] = len; } } while (0)
void ufv_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
extern "C" const char* ufv_last_error(void) { return ""; }
#include "../../ufvideo_amd/csrc/gemm256.hip"
#include "../../ufvideo_amd/csrc/gemm256_b.hip"
#include "../../ufvideo_amd/csrc/gemm256_s.hip"
int ufv_launch_pp_shape_fp8(const void*, const void*, const Epi&, int, int, int, int, int, bool, int, hipStream_t) { return 1; }
#include "../../ufvideo_amd/csrc/gemm_state.hip"

static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 18432, N = argc > 2 ? atoi(argv[2]) : 3584, K = argc > 3 ? atoi(argv[3]) : 3584;
    const int shape = argc > 4 ? atoi(argv[4]) : 1442, f32res = argc > 5 ? atoi(argv[5]) : 0;
    std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
    srand(1);
    for (auto& v : ha) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    for (auto& v : hw) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.05f);
    uint16_t *a, *w; void* c; float* r;
    hipMalloc(&a, ha.size() * 2); hipMalloc(&w, hw.size() * 2); hipMalloc(&c, (size_t)M * N * 4); hipMalloc(&r, (size_t)M * N * 4);
    hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemset(r, 0, (size_t)M * N * 4);
    Epi e; memset(&e, 0, sizeof(e));
    e.out = c; e.ldc = N; e.act = 0; e.resid = f32res ? r : nullptr; e.ldr = N;
    unsigned long long* st; const size_t nst = 256 * 2 * 16;
    hipMalloc(&st, nst * 8); hipMemset(st, 0, nst * 8);
    unsigned long long* nullp = nullptr;
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &nullp, sizeof(nullp));
    // COLD=n: rotate over n copies of A and W (more than the 256 MB Infinity Cache holds) so that every launch streams its operands from HBM, as inside the clip
    const int cold = getenv("COLD") ? atoi(getenv("COLD")) : 1;
    std::vector<uint16_t*> as(cold, a), ws(cold, w);
    for (int i = 1; i < cold; ++i) {
        hipMalloc(&as[i], ha.size() * 2); hipMalloc(&ws[i], hw.size() * 2);
        hipMemcpy(as[i], a, ha.size() * 2, hipMemcpyDeviceToDevice); hipMemcpy(ws[i], w, hw.size() * 2, hipMemcpyDeviceToDevice);
    }
    int rot = 0;
    auto run = [&]() { rot = (rot + 1) % cold; return ufv_launch_gemm256(as[rot], ws[rot], e, M, N, K, K, K, f32res != 0, false, false, false, shape, nullptr); };
    for (int i = 0; i < 3; ++i) if (run()) return 1;
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int IT = 20;
    float ms;
    const int gap_us = getenv("GAP_US") ? atoi(getenv("GAP_US")) : 0;      // idle time between launches: back to back the chip throttles to ~1.6 GHz, inside the clip it runs ~2.1
    if (gap_us > 0) {
        ms = 0.f;
        for (int i = 0; i < IT; ++i) {
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < gap_us) {}
            hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
            float m1; hipEventElapsedTime(&m1, e0, e1); ms += m1;
        }
    } else {
        hipEventRecord(e0);
        for (int i = 0; i < IT; ++i) run();
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double us = ms * 1000.0 / IT, fl = 2.0 * M * N * (double)K;
    printf("M %d N %d K %d shape %d %s: %.1f us  %.0f TF/s\n", M, N, K, shape, f32res ? "f32+res" : "bf16", us, fl / us * 1e-6);
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &st, sizeof(st));
    run(); hipDeviceSynchronize();
    std::vector<unsigned long long> h(nst);
    hipMemcpy(h.data(), st, nst * 8, hipMemcpyDeviceToHost);
    const int ns = shape >= 1000 ? 9 : 13;
    {   // tick calibration: one timed launch with stamps on, span of the first tile's K loops over all blocks
        hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms1; hipEventElapsedTime(&ms1, e0, e1);
        hipMemcpy(h.data(), st, nst * 8, hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0;
        for (int b = 0; b < 256; ++b) { const unsigned long long* p = h.data() + ((size_t)b * 2) * 16; if (p[13]) { lo = std::min(lo, p[13]); hi = std::max(hi, p[14]); } }
        printf("  stamped launch %.1f us; first-tile K loops span %llu ticks\n", ms1 * 1000.0, hi - lo);
    }
    for (int g = 0; g < 2; ++g) {
        printf("  group %d (wave %d), median over blocks of stamp[i+1]-stamp[i] (s_memtime ticks):", g, g * 4);
        for (int i = 0; i + 1 < ns; ++i) {
            std::vector<double> d;
            for (int b = 0; b < 256; ++b) { const unsigned long long* p = h.data() + ((size_t)b * 2 + g) * 16; if (p[i] && p[i + 1]) d.push_back((double)(p[i + 1] - p[i])); }
            if (d.empty()) { printf(" -"); continue; }
            std::sort(d.begin(), d.end());
            printf(" %.0f", d[d.size() / 2]);
        }
        std::vector<double> d;
        for (int b = 0; b < 256; ++b) { const unsigned long long* p = h.data() + ((size_t)b * 2 + g) * 16; if (p[0] && p[ns - 1]) d.push_back((double)(p[ns - 1] - p[0])); }
        if (!d.empty()) { std::sort(d.begin(), d.end()); printf("  | K-tile %.0f", d[d.size() / 2]); }
        if (shape >= 1000) {      // the load steps in parts: fragment reads issued | DMA issued | counted wait   (phase A: 0 -> 9 -> 10 -> 1; phase B: 4 -> 11 -> 12 -> 5)
            const int seq[2][4] = {{0, 9, 10, 1}, {4, 11, 12, 5}};
            for (int ph = 0; ph < 2; ++ph) {
                printf("  | load step %c reads/dma/wait", ph ? 'B' : 'A');
                for (int j = 0; j < 3; ++j) {
                    std::vector<double> dd;
                    for (int b = 0; b < 256; ++b) { const unsigned long long* p = h.data() + ((size_t)b * 2 + g) * 16; if (p[seq[ph][j]] && p[seq[ph][j + 1]]) dd.push_back((double)(p[seq[ph][j + 1]] - p[seq[ph][j]])); }
                    if (dd.empty()) { printf(" -"); continue; }
                    std::sort(dd.begin(), dd.end());
                    printf(" %.0f", dd[dd.size() / 2]);
                }
            }
        }
        std::vector<double> lp;
        for (int b = 0; b < 256; ++b) { const unsigned long long* p = h.data() + ((size_t)b * 2 + g) * 16; if (p[13] && p[14]) lp.push_back((double)(p[14] - p[13]) / (double)p[15]); }
        if (!lp.empty()) { std::sort(lp.begin(), lp.end()); printf("  | whole K loop / K-tiles %.0f", lp[lp.size() / 2]); }
        printf("\n");
    }
    return 0;
}
