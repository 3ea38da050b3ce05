// tools/screen_probe.hip — screen_top2_kernel (xq_screen.hip.h) against the kernel it replaces
// (gemm_colmax_persistent_kernel<2,2,DT_BF16,CM_TOP2>, xq_gemm.hip.h): same inputs, partial arrays compared group by group
// (values to 2e-6 relative to the bound's scale, position tags decoded and checked against a CPU dot), then both timed, interleaved.
// Usage: screen_probe [n=8192] [K=256] [rounds=5]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <random>
#include "../cn_chess_ai_amd/csrc/xq_screen.hip.h"
using namespace xq;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static uint16_t f2bf(float v) { uint32_t u; memcpy(&u, &v, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static int screen_row_host(int g, int code) { const int q = code & 15; return (g >> 2) * 128 + ((g >> 1) & 1) * 64 + 4 * (g & 1) + (code >> 4) * 32 + (q & 3) + 8 * (q >> 2); }

template <int KU, int NS, int DBG = 0> static void launch_new(const ScreenArgs& a, hipStream_t s) {
    static bool once = false;
    const size_t lds = screen_lds_bytes(a);
    if (!once) { CK(hipFuncSetAttribute((const void*)screen_top2_kernel<KU, NS, SCR_TOP2, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
    hipLaunchKernelGGL((screen_top2_kernel<KU, NS, SCR_TOP2, DBG>), dim3(a.panels * a.ranges), dim3(512), lds, s, a);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192, K = argc > 2 ? atoi(argv[2]) : 256, rounds = argc > 3 ? atoi(argv[3]) : 5;
    const int NO = 8100, NOP = 8192;
    const int npad = screen_padded_samples(n, K);
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> uw(-0.05f, 0.05f), ua(-1.f, 1.f), ub(-0.02f, 0.02f);
    std::vector<uint16_t> W((size_t)NOP * K, 0), A((size_t)npad * K, 0);
    std::vector<float> bias(NOP, 0.f);
    for (int j = 0; j < NO; ++j) { for (int k = 0; k < K; ++k) W[(size_t)j * K + k] = f2bf(uw(rng)); bias[j] = ub(rng); }
    for (int b = 0; b < n; ++b) for (int k = 0; k < K; ++k) A[(size_t)b * K + k] = f2bf(std::tanh(2.f * ua(rng)));
    std::vector<uint16_t> Af((size_t)npad * K, 0);          // the same activations in B-fragment order
    for (int b = 0; b < npad; ++b) for (int k = 0; k < K; ++k) Af[(size_t)scr_afrag_index(b, k, K)] = A[(size_t)b * K + k];
    uint16_t *dW, *dA, *dAf; float *dB, *P1o, *P2o, *P1n, *P2n;
    CK(hipMalloc(&dAf, Af.size() * 2)); CK(hipMemcpy(dAf, Af.data(), Af.size() * 2, hipMemcpyHostToDevice));
    const int tiles_m = NOP / 128, G = 4 * tiles_m;
    CK(hipMalloc(&dW, W.size() * 2)); CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, NOP * 4));
    CK(hipMemcpy(dW, W.data(), W.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, bias.data(), NOP * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&P1o, (size_t)G * npad * 4)); CK(hipMalloc(&P2o, (size_t)G * npad * 4));
    CK(hipMalloc(&P1n, (size_t)G * npad * 4)); CK(hipMalloc(&P2n, (size_t)G * npad * 4));
    CK(hipMemset(P1n, 0xFF, (size_t)G * npad * 4)); CK(hipMemset(P2n, 0xFF, (size_t)G * npad * 4));
    // old kernel
    GemmArgs g; memset(&g, 0, sizeof g);
    g.M = NO; g.N = n; g.K = K / 2; g.lda = g.ldb = K / 2;
    g.A = reinterpret_cast<const float*>(dW); g.B = reinterpret_cast<const float*>(dA); g.bias = dB;
    g.partial = P1o; g.partial2 = P2o; g.a_vec = g.b_vec = 1; g.k_chunk = g.K; g.bias_padded = 1;
    const int total = tiles_m * ((n + 127) / 128), grid_old = std::min(total, 512);
    if (total >= 4 * grid_old) { g.prio_split = grid_old / 2; g.prio_tiles = (total / 2) / tiles_m * tiles_m; }
    const size_t bias_lds = (size_t)tiles_m * 128 * 4;
    auto launch_old = [&](hipStream_t s) {
        hipLaunchKernelGGL((gemm_colmax_persistent_kernel<2, 2, DT_BF16, CM_TOP2>), dim3(grid_old), dim3(256), bias_lds, s, g, tiles_m, total);
    };
    ScreenArgs a; memset(&a, 0, sizeof a);
    a.W = dW; a.A = dAf; a.a_frag = 1; a.bias = dB; a.P1 = P1n; a.P2 = P2n; a.ldp = npad;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    screen_geometry(NO, n, K, prop.multiProcessorCount, a);
    printf("n %d K %d: panels %d ranges %d cpr %d nchunks %d grid %d lds %zu\n", n, K, a.panels, a.ranges, a.cpr, a.nchunks, a.panels * a.ranges, screen_lds_bytes(a));
    auto launch_n = [&](hipStream_t s) { if (K == 256) launch_new<1, 2>(a, s); else launch_new<2, 1>(a, s); };
    launch_old(0); launch_n(0);
    CK(hipDeviceSynchronize());
    std::vector<float> o1((size_t)G * n), o2((size_t)G * n), n1((size_t)G * npad), n2((size_t)G * npad);
    CK(hipMemcpy(o1.data(), P1o, o1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o2.data(), P2o, o2.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(n1.data(), P1n, n1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(n2.data(), P2n, n2.size() * 4, hipMemcpyDeviceToHost));
    // compare: groups that hold real rows only (the old kernel writes all 4 * tiles_m groups, the new one 2 * nchunks)
    const int Gn = 2 * a.nchunks;
    long long bad = 0, tagdiff = 0, cnt = 0; double maxd1 = 0, maxd2 = 0;
    for (int gi = 0; gi < Gn; ++gi)
        for (int b = 0; b < n; ++b) {
            const float vo = o1[(size_t)gi * n + b], vn = n1[(size_t)gi * npad + b];
            const float wo = o2[(size_t)gi * n + b], wn = n2[(size_t)gi * npad + b];
            uint32_t uo, un; memcpy(&uo, &vo, 4); memcpy(&un, &vn, 4);
            const double d1 = std::fabs((double)vo - vn), d2 = std::fabs((double)wo - wn);
            if (vo > -1e30f) { maxd1 = std::max(maxd1, d1); maxd2 = std::max(maxd2, d2); }
            if ((uo & 31) != (un & 31)) ++tagdiff;
            if (!(d1 <= 4e-6 + 1e-5 * std::fabs(vo)) || !(d2 <= 4e-6 + 1e-5 * std::fabs(wo))) { if (bad < 5) printf("  diff g %d b %d: old %g %g new %g %g\n", gi, b, vo, wo, vn, wn); ++bad; }
            ++cnt;
        }
    printf("compared %lld (group, sample) pairs: max |d top1| %.3g, max |d top2| %.3g, out of tolerance %lld, tags that differ %lld\n", cnt, maxd1, maxd2, bad, tagdiff);
    // CPU check of the new kernel's tags on a sample of pairs: the tagged row's exact bf16 dot + bias must equal the value (to fp32 rounding)
    long long tagbad = 0;
    for (int t = 0; t < 4000; ++t) {
        const int gi = (int)(rng() % Gn), b = (int)(rng() % n);
        const float vn = n1[(size_t)gi * npad + b];
        uint32_t un; memcpy(&un, &vn, 4);
        const int row = screen_row_host(gi, (int)(un & 31));
        if (row >= NO) { if (vn > -1e30f) ++tagbad; continue; }
        double z = bias[row];
        for (int k = 0; k < K; ++k) z += (double)bf2f(W[(size_t)row * K + k]) * bf2f(A[(size_t)b * K + k]);
        // and no row of the group may exceed it by more than rounding
        double best = -1e300;
        for (int code = 0; code < 32; ++code) {
            const int r2 = screen_row_host(gi, code);
            if (r2 >= NO) continue;
            double z2 = bias[r2];
            for (int k = 0; k < K; ++k) z2 += (double)bf2f(W[(size_t)r2 * K + k]) * bf2f(A[(size_t)b * K + k]);
            best = std::max(best, z2);
        }
        if (std::fabs(z - vn) > 2e-5 + 1e-5 * std::fabs(z) || best > z + 4e-6) { if (tagbad < 5) printf("  tag check g %d b %d row %d: value %g exact %g group max %g\n", gi, b, row, vn, z, best); ++tagbad; }
    }
    printf("CPU tag / value check on 4000 pairs: %lld bad\n", tagbad);
    // timing, interleaved
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop = 2.0 * NO * (double)n * K;
    for (int r = 0; r < rounds; ++r) {
        float mo = 0, mn = 0;
        hipEventRecord(e0, 0); for (int i = 0; i < 20; ++i) launch_old(0); hipEventRecord(e1, 0); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&mo, e0, e1);
        hipEventRecord(e0, 0); for (int i = 0; i < 20; ++i) launch_n(0); hipEventRecord(e1, 0); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&mn, e0, e1);
        printf("round %d: old %.2f us (%.0f TFLOP/s)   new %.2f us (%.0f TFLOP/s = %.3f of 2.5 PF)\n", r, mo * 50, flop / (mo / 20 * 1e-3) / 1e12,
               mn * 50, flop / (mn / 20 * 1e-3) / 1e12, flop / (mn / 20 * 1e-3) / 2.5e15);
    }
    if (K == 256 && n == 8192) {          // where the time goes: ablations (results of these launches are garbage by design)
        auto timeit = [&](const char* name, auto fn) {
            float ms = 0; fn(); fn();
            hipEventRecord(e0, 0); for (int i = 0; i < 20; ++i) fn(); hipEventRecord(e1, 0); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
            printf("  %-52s %.2f us\n", name, ms * 50);
        };
        timeit("full kernel (fragment-order activations)", [&] { launch_new<1, 2, 0>(a, 0); });
        { ScreenArgs x = a; CK(hipMalloc(&x.R, (size_t)a.ranges * npad * 4)); CK(hipMalloc(&x.na, (size_t)npad * 4));
          timeit("+ per-range maxima and activation norms (what the library runs)", [&] { launch_new<1, 2, 0>(x, 0); }); }
        { ScreenArgs x = a; x.A = dA; x.a_frag = 0; timeit("row-major activations", [&] { launch_new<1, 2, 0>(x, 0); }); }
        { ScreenArgs x = a; x.xcd_rows = 1; timeit("full kernel, XCD = row ranges", [&] { launch_new<1, 2, 0>(x, 0); }); }
        timeit("TIMING ONLY: two v_mfma_f32_16x16x32_bf16 per 32x32x16 (same pipe cycles, meaningless values)", [&] { launch_new<1, 2, 16>(a, 0); });
        timeit("full kernel again", [&] { launch_new<1, 2, 0>(a, 0); });
        timeit("TIMING ONLY: 16x16x32 again", [&] { launch_new<1, 2, 16>(a, 0); });
        timeit("no fold", [&] { launch_new<1, 2, 1>(a, 0); });
        timeit("no activation loads", [&] { launch_new<1, 2, 2>(a, 0); });
        timeit("no fold, no activation loads", [&] { launch_new<1, 2, 3>(a, 0); });
        timeit("no LDS-DMA in the loop (stale weights)", [&] { launch_new<1, 2, 4>(a, 0); });
        timeit("no fold, no activation loads, no DMA", [&] { launch_new<1, 2, 7>(a, 0); });
        {   // in-kernel clocks: s_memtime (shader cycles) against s_memrealtime (100 MHz), prologue and loop, median over the blocks
            unsigned long long* dbg; CK(hipMalloc(&dbg, 256 * 8 * 8)); CK(hipMemset(dbg, 0, 256 * 8 * 8));
            for (int mode = 1; mode >= 0; --mode) {
            ScreenArgs c = a; c.dbg = dbg; c.xcd_rows = mode;
            for (int i = 0; i < 30; ++i) launch_new<1, 2, 8>(c, 0);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> hd(256 * 8);
            CK(hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> pc, pr, lc, lr, st;
            for (int b = 0; b < 256; ++b) { pc.push_back((double)hd[b * 8]); pr.push_back(hd[b * 8 + 1] * 10.0); lc.push_back((double)hd[b * 8 + 2]); lr.push_back(hd[b * 8 + 3] * 10.0); st.push_back(hd[b * 8 + 5] * 10.0); }
            auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            auto mx = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.back(); };
            auto mn = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.front(); };
            printf("  xcd_rows %d stamps (median over 256 blocks): prologue %.0f cycles / %.0f ns (%.2f GHz), loop %.0f cycles / %.0f ns (%.2f GHz) = %.0f cycles per unit; loop ns min %.0f max %.0f; block start spread %.0f ns\n",
                   mode, med(pc), med(pr), med(pc) / med(pr), med(lc), med(lr), med(lc) / med(lr), med(lc) / 8, mn(lr), mx(lr), mx(st) - mn(st));
            }
        }
        ScreenArgs b = a;                    // half the blocks, twice the chunks each (128 CUs): per-unit cost = difference / 8 units
        b.cpr = 16; b.ranges = 8;
        timeit("128 blocks x 16 chunks (half the chip)", [&] { launch_new<1, 2, 0>(b, 0); });
        b.cpr = 4; b.ranges = 16;            // 256 blocks x 4 chunks: half the rows
        timeit("256 blocks x 4 chunks (half the rows)", [&] { launch_new<1, 2, 0>(b, 0); });
    }
    return (bad || tagbad) ? 1 : 0;
}
