// Stand-alone lab for the attention kernels (not part of the product): builds csrc/attn.hip into one executable with a block-level
// s_memtime timeline (UFV_STAMP), times a kernel id on the ViT shape and checks a few rows against a host fp64 softmax.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/lab/attn_lab.hip -o tools/lab/attn_lab && tools/lab/attn_lab [kernel] [B] [H] [S] [hd] [causal]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cmath>
#include <cstring>
#include <cstdint>
#include <vector>
#include <algorithm>
__device__ unsigned long long* g_stamps;      // [blocks][8]
#define UFV_STAMP(i) do { if (threadIdx.x == 0 && g_stamps) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    g_stamps[(size_t)blockIdx.x * 8 + (i)] = t_; if ((i) == 0) { unsigned id_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id_)); \
    unsigned xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_)); g_stamps[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)xcc_ << 32) | id_; } } } while (0)
void ufv_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
extern "C" const char* ufv_last_error(void) { return ""; }
int ufv_dev_n_cu() { static int n = 0; if (!n) { hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0); n = p.multiProcessorCount; } return n; }      // (gemm_state.hip's, for the lab)
#include "../../ufvideo_amd/csrc/attn.hip"

static inline float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
    int kernel = argc > 1 ? atoi(argv[1]) : 0, B = argc > 2 ? atoi(argv[2]) : 32, H = argc > 3 ? atoi(argv[3]) : 16;
    int S = argc > 4 ? atoi(argv[4]) : 576, hd = argc > 5 ? atoi(argv[5]) : 72, causal = argc > 6 ? atoi(argv[6]) : 0;
    int Hkv = argc > 7 ? atoi(argv[7]) : H;
    const int D = H * hd, W = (H + 2 * Hkv) * hd;
    size_t n = (size_t)B * S * W;
    std::vector<uint16_t> h(n);
    srand(1);
    for (size_t i = 0; i < n; ++i) { float u = 0; for (int j = 0; j < 4; ++j) u += rand() / (float)RAND_MAX - 0.5f; h[i] = f2bf(u * 1.7f); }
    uint16_t *qkv, *o;
    hipMalloc(&qkv, n * 2); hipMalloc(&o, (size_t)B * S * D * 2);
    hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice);
    unsigned long long* st; const size_t nst = 1 << 16;
    hipMalloc(&st, 2 * nst * 8 * 8); hipMemset(st, 0, 2 * nst * 8 * 8);
    unsigned long long* nullp = nullptr;
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &nullp, sizeof(nullp));
    auto run = [&]() {
        return ufv_attention(qkv, (int64_t)S * W, W, qkv + H * hd, (int64_t)S * W, W, qkv + (H + Hkv) * hd, (int64_t)S * W, W, o, (int64_t)S * D, D, B, H, Hkv, S, S,
                             hd, 1.0f / sqrtf((float)hd), causal, 0, kernel, nullptr);
    };
    for (int i = 0; i < 5; ++i) if (run()) return 1;
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 50;
    hipEventRecord(e0);
    for (int i = 0; i < N; ++i) run();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double us = ms * 1000.0 / N, fl = 4.0 * B * H * (double)S * S * hd * (causal ? 0.5 : 1.0);
    printf("kernel %d  B %d H %d/%d S %d hd %d causal %d : %.1f us  %.0f TF/s useful  (%.1f%% of 2.5 PF)\n", kernel, B, H, Hkv, S, hd, causal, us, fl / us * 1e-6,
           fl / us * 1e-6 / 25.0);
    // check a few rows vs fp64
    std::vector<uint16_t> ho((size_t)B * S * D);
    hipMemcpy(ho.data(), o, ho.size() * 2, hipMemcpyDeviceToHost);
    double worst = 0, top = 0;
    for (int t = 0; t < 24; ++t) {
        int b = rand() % B, hq = rand() % H, qi = rand() % S, hk = hq / (H / Hkv);
        std::vector<double> p(S); double mx = -1e30;
        int nk = causal ? qi + 1 : S;
        for (int j = 0; j < nk; ++j) { double s = 0; for (int d = 0; d < hd; ++d) s += (double)bf2f(h[((size_t)b * S + qi) * W + hq * hd + d]) * bf2f(h[((size_t)b * S + j) * W + (H + hk) * hd + d]);
            p[j] = s / sqrt((double)hd); mx = std::max(mx, p[j]); }
        double l = 0; for (int j = 0; j < nk; ++j) { p[j] = exp(p[j] - mx); l += p[j]; }
        for (int d = 0; d < hd; ++d) { double acc = 0; for (int j = 0; j < nk; ++j) acc += p[j] * bf2f(h[((size_t)b * S + j) * W + (H + Hkv + hk) * hd + d]);
            acc /= l; double got = bf2f(ho[((size_t)b * S + qi) * D + hq * hd + d]); worst = std::max(worst, fabs(got - acc)); top = std::max(top, fabs(acc));
            if (getenv("LAB_VERBOSE") && d < 3) printf("  b %d h %d q %d d %d: got %.4f ref %.4f\n", b, hq, qi, d, got, acc); }
    }
    printf("check: max|d| / max|ref| = %.2e\n", worst / top);
    if (getenv("LAB_CMP")) {       // element-wise comparison with another kernel id
        const int other = atoi(getenv("LAB_CMP")); const int keep = kernel;
        kernel = other; run(); hipDeviceSynchronize(); kernel = keep;
        std::vector<uint16_t> h2(ho.size());
        hipMemcpy(h2.data(), o, h2.size() * 2, hipMemcpyDeviceToHost);
        int shown = 0; size_t bad = 0, bits = 0;
        for (size_t i = 0; i < ho.size(); ++i) bits += ho[i] != h2[i];
        printf("elements with different BITS from kernel %d: %zu of %zu\n", other, bits, ho.size());
        for (size_t i = 0; i < ho.size(); ++i) { double a_ = bf2f(ho[i]), b_ = bf2f(h2[i]);
            if (fabs(a_ - b_) > 0.02) { ++bad; if (shown++ < 40) { size_t row = i / D; printf("  row(q) %zu head %zu d %zu: this %.4f other %.4f\n", row % S, (i % D) / hd, i % hd, a_, b_); } } }
        printf("elements differing by > 0.02 from kernel %d: %zu of %zu\n", other, bad, ho.size());
        printf("bad rows (b*S+q):");
        for (size_t r = 0; r < ho.size() / D; ++r) { int nb = 0; for (int c = 0; c < D; ++c) nb += fabs(bf2f(ho[r * D + c]) - bf2f(h2[r * D + c])) > 0.02; if (nb) printf(" %zu(%d)", r, nb); }
        printf("\n");
    }
#ifdef UFV_VIT_P2_CLOCK
    if (kernel == 14 && getenv("LAB_P2_CLOCK")) {
        // In-kernel clock of the ViT attention kernel (MI355X_MICROARCH.md, DVFS give-back item 6): >= 2 s of back-to-back launches on the random operands, then
        // the stamps of the LAST launch: per wave d(s_memtime) / d(s_memrealtime) x 100 MHz, median over the waves; the wall time per launch of the same run by
        // HIP events; MFMA-busy cycles per SIMD from the instruction stream (one wave per SIMD: 81 periods x 22 v_mfma_f32_32x32x16_bf16 x 32 cycles).
        const int nw = B * H / 2 * 4;
        unsigned long long* cb; hipMalloc(&cb, (size_t)nw * 32); hipMemset(cb, 0, (size_t)nw * 32);
        hipMemcpyToSymbol(HIP_SYMBOL(g_vit_p2_clock), &cb, sizeof(cb));
        hipEvent_t c0, c1; hipEventCreate(&c0); hipEventCreate(&c1);
        int launches = 0; float tot_ms = 0;
        while (tot_ms < 2500.f) {
            hipEventRecord(c0);
            for (int i = 0; i < 2000; ++i) run();
            hipEventRecord(c1); hipEventSynchronize(c1);
            float m; hipEventElapsedTime(&m, c0, c1); tot_ms += m; launches += 2000;
        }
        hipEventRecord(c0);
        for (int i = 0; i < 2000; ++i) run();
        hipEventRecord(c1); hipEventSynchronize(c1);
        float m; hipEventElapsedTime(&m, c0, c1);
        const double wall_us = m * 1000.0 / 2000;
        std::vector<unsigned long long> hc((size_t)nw * 4);
        hipMemcpy(hc.data(), cb, hc.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> ghz, cyc;
        for (int w = 0; w < nw; ++w) {
            const double dt = (double)(hc[w * 4 + 1] - hc[w * 4]), dr = (double)(hc[w * 4 + 3] - hc[w * 4 + 2]);
            if (dr > 0) { ghz.push_back(dt / dr * 0.1); cyc.push_back(dt); }
        }
        std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
        const double g_med = ghz[ghz.size() / 2], c_med = cyc[cyc.size() / 2];
        const double mfma_busy = 81.0 * 22.0 * 32.0;
        printf("clock: %d launches warm, then 2000 timed: wall %.2f us / launch; in-kernel clock median %.3f GHz (min %.3f max %.3f over %zu waves); pass loop median %.0f cycles "
               "(max %.0f)\n", launches, wall_us, g_med, ghz.front(), ghz.back(), ghz.size(), c_med, cyc.back());
        printf("clock: MFMA busy %.0f cycles per SIMD = %.3f of wall x clock (%.0f cycles), %.3f of the wave's own pass loop\n", mfma_busy, mfma_busy / (wall_us * 1e3 * g_med),
               wall_us * 1e3 * g_med, mfma_busy / c_med);
        printf("{\"kernel\": \"attn_fwd_vit72_p2\", \"wall_us\": %.3f, \"clock_ghz_median\": %.4f, \"clock_ghz_min\": %.4f, \"clock_ghz_max\": %.4f, \"pass_loop_cycles_median\": %.0f, "
               "\"mfma_busy_cycles_per_simd\": %.0f, \"mfma_busy_over_wall_x_clock\": %.4f, \"mfma_busy_over_pass_loop\": %.4f, \"waves\": %zu}\n",
               wall_us, g_med, ghz.front(), ghz.back(), c_med, mfma_busy, mfma_busy / (wall_us * 1e3 * g_med), mfma_busy / c_med, ghz.size());
    }
#endif
    if (kernel == 14 && getenv("LAB_P2_STAMPS")) {       // per-period cycle anatomy of the generated kernel (built with gen_attn_p2.py --stamps)
        const int nw = B * H / 2 * 4;
        unsigned* sb; hipMalloc(&sb, (size_t)nw * 512); hipMemset(sb, 0, (size_t)nw * 512);
        AttnArgs a;
        a.q = (const bf16*)qkv; a.k = (const bf16*)(qkv + H * hd); a.v = (const bf16*)(qkv + (H + Hkv) * hd); a.o = (bf16*)o;
        a.q_bs = a.k_bs = a.v_bs = (int64_t)S * W; a.q_ss = a.k_ss = a.v_ss = W; a.o_bs = (int64_t)S * D; a.o_ss = D;
        a.B = B; a.Hq = H; a.Hkv = Hkv; a.Sq = S; a.Sk = S; a.hd = hd; a.scale = 1.0f / sqrtf((float)hd); a.q_pos0 = 0; a.lse = (float*)sb;
        for (int i = 0; i < 20; ++i) launch_vit72_p2(a, nullptr);
        hipDeviceSynchronize();
        std::vector<unsigned> hsb((size_t)nw * 128);
        hipMemcpy(hsb.data(), sb, hsb.size() * 4, hipMemcpyDeviceToHost);
        const int NS = 85;
        std::vector<double> d(NS, 0.0);
        for (int w = 0; w < nw; ++w) for (int i = 1; i < NS; ++i) d[i] += (double)(unsigned)(hsb[(size_t)w * 128 + i] - hsb[(size_t)w * 128 + i - 1]);
        printf("stamps (mean cycles over %d waves): prologue %.0f | ", nw, d[1] / nw);
        double tot = 0; for (int i = 1; i < NS; ++i) tot += d[i] / nw;
        for (int p_ = 0; p_ < 3; ++p_) { printf("\n pass %d:", p_); for (int j = 0; j < 9; ++j) { printf("  t%d", j); for (int u = 0; u < 3; ++u) printf(" %4.0f", d[2 + p_ * 27 + j * 3 + u + 1 > NS - 1 ? NS - 1 : 2 + p_ * 27 + j * 3 + u + 1] / nw); } }
        printf("\n epilogue %.0f ; total %.0f cycles\n", d[NS - 1] / nw, tot);
    }
#ifdef UFV_ATTN_C128_STAMPS
    if (kernel == 15) {       // per-segment cycle sums of the compute waves (built with gen_attn_c128.py --stamps)
        unsigned* sb; hipMalloc(&sb, 256 * 4 * 32); hipMemset(sb, 0, 256 * 4 * 32);
        AttnArgs a;
        a.q = (const bf16*)qkv; a.k = (const bf16*)(qkv + H * hd); a.v = (const bf16*)(qkv + (H + Hkv) * hd); a.o = (bf16*)o;
        a.q_bs = a.k_bs = a.v_bs = (int64_t)S * W; a.q_ss = a.k_ss = a.v_ss = W; a.o_bs = (int64_t)S * D; a.o_ss = D;
        a.B = B; a.Hq = H; a.Hkv = Hkv; a.Sq = S; a.Sk = S; a.hd = hd; a.scale = 1.0f / sqrtf((float)hd); a.q_pos0 = 0; a.lse = (float*)sb;
        for (int i = 0; i < 20; ++i) launch_c128(a, nullptr);
        hipDeviceSynchronize();
        std::vector<unsigned> hsb(256 * 4 * 8);
        hipMemcpy(hsb.data(), sb, hsb.size() * 4, hipMemcpyDeviceToHost);
        const char* nm[8] = {"iter tail (rescale, loop)", "barrier wait", "phase A (QK + finish)", "phase B (PV + start)", "item prologue (Q, tile 0)", "last tile", "epilogue + idle barriers", "-"};
        double sum[8] = {0}; int nw = 0; double tot = 0, mx = 0;
        for (int w = 0; w < 1024; ++w) { double t = 0; for (int k = 0; k < 8; ++k) { sum[k] += hsb[w * 8 + k]; t += hsb[w * 8 + k]; } if (t > 0) { ++nw; tot += t; mx = std::max(mx, t); } }
        printf("c128 stamps: %d compute waves, mean total %.0f cycles, max %.0f\n", nw, tot / nw, mx);
        for (int k = 0; k < 7; ++k) printf("   %-28s %8.0f cycles / wave  (%.1f %%)\n", nm[k], sum[k] / nw, 100.0 * sum[k] / tot);
    }
#endif
    // timeline
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &st, sizeof(st));
    run(); hipDeviceSynchronize();
    std::vector<unsigned long long> hs(2 * nst * 8);
    hipMemcpy(hs.data(), st, 2 * nst * 64, hipMemcpyDeviceToHost);
    { double r[7] = {0, 0, 0, 0, 0, 0, 0}; int n_ = 0; for (size_t b = 0; b < nst; ++b) if (hs[(nst + b) * 8]) { ++n_; for (int k = 0; k < 7; ++k) r[k] += (double)hs[(nst + b) * 8 + k]; }
      if (n_) printf("wave-0 regions per block (ticks): R1 %.0f  R2 %.0f  R3+trigger %.0f  K-wait before R1 %.0f  vmcnt wait %.0f  barrier %.0f  DMA issue %.0f\n", r[0] / n_, r[1] / n_, r[2] / n_, r[3] / n_, r[4] / n_, r[5] / n_, r[6] / n_); }
    unsigned long long t0 = ~0ull, t1 = 0; int nb = 0;
    for (size_t b = 0; b < nst; ++b) if (hs[b * 8 + 0] && hs[b * 8 + 4]) { t0 = std::min(t0, hs[b * 8]); t1 = std::max(t1, hs[b * 8 + 4]); ++nb; }
    if (nb) {
        double sum[5] = {0};
        for (size_t b = 0; b < nst; ++b) if (hs[b * 8 + 0] && hs[b * 8 + 4]) for (int i = 1; i < 5; ++i) sum[i] += (double)(hs[b * 8 + i] - hs[b * 8 + i - 1]);
        printf("timeline: %d blocks, span %llu ticks (%.2f ticks/ns) ; mean per block: Q %.0f  first-tiles %.0f  loop %.0f  store %.0f\n", nb, t1 - t0, (t1 - t0) / (us * 1000.0),
               sum[1] / nb, sum[2] / nb, sum[3] / nb, sum[4] / nb);
        // concurrency per CU: key = (xcc, se/sh/cu bits of HW_ID)
        std::vector<std::pair<unsigned long long, size_t>> order;
        for (size_t b = 0; b < nst; ++b) if (hs[b * 8] && hs[b * 8 + 4]) order.push_back({((hs[b * 8 + 7] >> 32) << 32) | (hs[b * 8 + 7] & 0xFF00), b});
        std::sort(order.begin(), order.end());
        int ncu = 0, maxconc = 0; double busy = 0; size_t i = 0;
        while (i < order.size()) {
            size_t j = i; std::vector<std::pair<unsigned long long, int>> ev;
            while (j < order.size() && order[j].first == order[i].first) { ev.push_back({hs[order[j].second * 8], 1}); ev.push_back({hs[order[j].second * 8 + 4], -1}); ++j; }
            std::sort(ev.begin(), ev.end()); int c = 0; unsigned long long last = 0;
            for (auto& e : ev) { if (c > 0) busy += (double)(e.first - last) * 1.0; c += e.second; maxconc = std::max(maxconc, c); last = e.first; }
            ++ncu; i = j;
        }
        printf("CUs seen %d, max co-resident blocks on a CU %d, mean CU busy fraction %.2f\n", ncu, maxconc, busy / ncu / (double)(t1 - t0));
    }
    return 0;
}
