// Lab: per-TILE timeline of the persistent ping-pong GEMM (not part of the product).  For the first 8 tiles of every block, waves 0 and 4 stamp s_memtime at:
//   0 K loop start (prologue landed)   1..4 end of K-tiles 0..3   5 K loop end   6 next tile's prologue issued (bias fetched)   7 epilogue issued (stores queued)
// Prints the medians of the differences, which show where a tile's time outside its K-tiles goes (the seam) and whether the first K-tiles of a tile run
// slower than the steady state (loads queued behind the previous tile's stores).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/lab/gemm_tile_lab.hip -o tools/lab/gemm_tile_lab && tools/lab/gemm_tile_lab M N K shape [f32res] [act]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cstring>
#include <cstdint>
#include <vector>
#include <algorithm>
__device__ unsigned long long* g_ts;      // [blocks][2 groups][8 tiles][16 slots]
#define UFV_TSTAMP_DECL int ts_tile = 0;
#define UFV_TSTAMP_NEXT ++ts_tile;
#define UFV_TSTAMP(slot) do { if (g_ts && lane == 0 && (wave & 3) == 0 && ts_tile < 8) \
    g_ts[((((size_t)blockIdx.x * 2 + (wave >> 2)) * 8 + ts_tile) * 16) + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define UFV_TSTAMP_K(tt) do { if ((tt) < 4) UFV_TSTAMP(1 + (tt)); } while (0)
void ufv_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
extern "C" const char* ufv_last_error(void) { return ""; }
#include "../../ufvideo_amd/csrc/gemm256.hip"
#include "../../ufvideo_amd/csrc/gemm256_b.hip"
#include "../../ufvideo_amd/csrc/gemm256_s.hip"
int ufv_launch_pp_shape_fp8(const void*, const void*, const Epi&, int, int, int, int, int, bool, int, hipStream_t) { return 1; }
#include "../../ufvideo_amd/csrc/gemm_state.hip"

static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 18432, N = argc > 2 ? atoi(argv[2]) : 3456, K = argc > 3 ? atoi(argv[3]) : 1152;
    const int shape = argc > 4 ? atoi(argv[4]) : 1442, f32res = argc > 5 ? atoi(argv[5]) : 0, act = argc > 6 ? atoi(argv[6]) : 0;
    std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
    srand(1);
    for (auto& v : ha) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    for (auto& v : hw) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.05f);
    uint16_t *a, *w; void* c; float* r; float* bias;
    (void)hipMalloc(&a, ha.size() * 2); (void)hipMalloc(&w, hw.size() * 2); (void)hipMalloc(&c, (size_t)M * N * 4); (void)hipMalloc(&r, (size_t)M * N * 4);
    (void)hipMalloc(&bias, (size_t)N * 4); (void)hipMemset(bias, 0, (size_t)N * 4);
    (void)hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemset(r, 0, (size_t)M * N * 4);
    Epi e; memset(&e, 0, sizeof(e));
    e.out = f32res ? (void*)r : c; e.ldc = N; e.act = act; e.resid = f32res ? r : nullptr; e.ldr = N; e.bias = bias;
    const size_t nst = 256 * 2 * 8 * 16;
    unsigned long long* st; (void)hipMalloc(&st, nst * 8); (void)hipMemset(st, 0, nst * 8);
    unsigned long long* nullp = nullptr;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ts), &nullp, sizeof(nullp));
    auto run = [&]() { return ufv_launch_gemm256(a, w, e, M, N, K, K, K, f32res != 0, false, false, false, shape, nullptr); };
    for (int i = 0; i < 3; ++i) if (run()) return 1;
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int IT = 20;
    (void)hipEventRecord(e0);
    for (int i = 0; i < IT; ++i) run();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000.0 / IT, fl = 2.0 * M * N * (double)K;
    printf("M %d N %d K %d shape %d %s act %d: %.1f us  %.0f TF/s\n", M, N, K, shape, f32res ? "f32+res (in place)" : "bf16", act, us, fl / us * 1e-6);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ts), &st, sizeof(st));
    run(); run();
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(nst);
    (void)hipMemcpy(h.data(), st, nst * 8, hipMemcpyDeviceToHost);
    auto med = [](std::vector<double>& d) { if (d.empty()) return -1.0; std::sort(d.begin(), d.end()); return d[d.size() / 2]; };
    const char* names[8] = {"start->kt0", "kt1", "kt2", "kt3", "kt4..end", "loop end->prologue issued", "epilogue issue", "epilogue end->next K loop start"};
    for (int g = 0; g < 2; ++g) {
        printf(" group %d (wave %d): medians over blocks, ticks\n", g, 4 * g);
        for (int t = 0; t < 5; ++t) {
            printf("  tile %d:", t);
            double tile_total = -1;
            for (int i = 0; i < 8; ++i) {
                std::vector<double> d;
                for (int b = 0; b < 256; ++b) {
                    const unsigned long long* p = h.data() + (((size_t)b * 2 + g) * 8 + t) * 16;
                    const unsigned long long* pn = p + 16;
                    unsigned long long x0 = i < 7 ? p[i] : p[7], x1 = i < 7 ? p[i + 1] : (t < 7 ? pn[0] : 0);
                    if (x0 && x1 && x1 > x0) d.push_back((double)(x1 - x0));
                }
                printf(" %s %.0f |", names[i], med(d));
            }
            std::vector<double> d;
            for (int b = 0; b < 256; ++b) { const unsigned long long* p = h.data() + (((size_t)b * 2 + g) * 8 + t) * 16; if (p[0] && p[16]) d.push_back((double)(p[16] - p[0])); }
            { std::vector<double> d1, d2, d3; for (int b = 0; b < 256; ++b) { const unsigned long long* p = h.data() + (((size_t)b * 2 + g) * 8 + t) * 16; if (p[5] && p[8] && p[9] && p[6]) { d1.push_back((double)(p[8] - p[5])); d2.push_back((double)(p[9] - p[8])); d3.push_back((double)(p[6] - p[9])); } }
              printf(" [bias wait %.0f, next_item + set_src %.0f, DMA issue %.0f]", med(d1), med(d2), med(d3)); }
            tile_total = med(d);
            printf(" tile period %.0f\n", tile_total);
        }
    }
    return 0;
}
