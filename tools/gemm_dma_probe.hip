// tools/gemm_dma_probe.hip — gemm_bf16_kernel (xq_gemm_dma.hip.h): the three products of the bf16 Q-net against a CPU reference
// (exact: bf16 inputs, double accumulation) at a small shape, then timing at the configs[4] shapes beside the tile kernel it replaces.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../cn_chess_ai_amd/csrc/xq_gemm_dma.hip.h"
using namespace xq;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static uint16_t f2bf(float v) { uint32_t u; memcpy(&u, &v, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

template <int AL, int BL, int EPI, int DT = DT_BF16, int TI = 2, int TJ = 2> static void launch(const Bf16GemmArgs& g, int gz) {
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void*)gemm_dma_kernel<DT, AL, BL, EPI, TI, TJ>, hipFuncAttributeMaxDynamicSharedMemorySize, bg_lds_bytes(TI, TJ))); once = true; }
    hipLaunchKernelGGL((gemm_dma_kernel<DT, AL, BL, EPI, TI, TJ>), dim3(g.M / (128 * TI), g.N / (64 * TJ), gz), dim3(512), bg_lds_bytes(TI, TJ), 0, g);
}
template <typename T> static T* dev(const std::vector<T>& v) { T* p; CK(hipMalloc(&p, v.size() * sizeof(T))); CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return p; }

int main() {
    std::mt19937 rng(3);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    int bad = 0;
    {   // ---- correctness, M = 512, N = 256, K = 192 (3 k-tiles) ----
        const int M = 512, N = 256, K = 192;
        std::vector<uint16_t> Akc((size_t)M * K), Amc((size_t)K * M), Bkc((size_t)N * K), Bmc((size_t)K * N), Hb((size_t)M * N);
        std::vector<float> bias(N);
        std::vector<double> ref((size_t)M * N, 0.0);
        for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) { const uint16_t v = f2bf(u(rng)); Akc[(size_t)m * K + k] = v; Amc[(size_t)k * M + m] = v; }
        for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) { const uint16_t v = f2bf(0.1f * u(rng)); Bkc[(size_t)n * K + k] = v; Bmc[(size_t)k * N + n] = v; }
        for (auto& x : Hb) x = f2bf(u(rng));
        for (auto& x : bias) x = 0.2f * u(rng);
        for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)bf2f(Akc[(size_t)m * K + k]) * bf2f(Bkc[(size_t)n * K + k]); ref[(size_t)m * N + n] = s; }
        uint16_t *dAkc = dev(Akc), *dAmc = dev(Amc), *dBkc = dev(Bkc), *dBmc = dev(Bmc), *dHb = dev(Hb);
        float* dbias = dev(bias);
        float* dC; uint16_t* dCb; CK(hipMalloc(&dC, (size_t)M * N * 4 * 3)); CK(hipMalloc(&dCb, (size_t)M * N * 2));
        std::vector<float> C((size_t)M * N * 3); std::vector<uint16_t> Cb((size_t)M * N);
        auto check = [&](const char* name, auto expect_f32, bool has_bf, bool frag, int slabs) {
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(Cb.data(), dCb, Cb.size() * 2, hipMemcpyDeviceToHost));
            double maxe = 0, maxb = 0; long long nb = 0;
            for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
                double got = 0; for (int zz = 0; zz < slabs; ++zz) got += C[(size_t)zz * M * N + (size_t)m * N + n];
                const double want = expect_f32(m, n);
                const double e = std::fabs(got - want); maxe = std::max(maxe, e);
                if (e > 2e-3 * (1 + std::fabs(want))) { if (nb < 5) printf("   %s f32 (%d,%d): got %g want %g\n", name, m, n, got, want); ++nb; }
                if (has_bf) {
                    const float gb = bf2f(Cb[frag ? (size_t)scr_afrag_index(m, n, N) : (size_t)m * N + n]);
                    const double eb = std::fabs(gb - want); maxb = std::max(maxb, eb);
                    if (eb > 1e-2 * (1 + std::fabs(want))) { if (nb < 5) printf("   %s bf16 (%d,%d): got %g want %g\n", name, m, n, gb, want); ++nb; }
                }
            }
            printf("%-44s max |err| f32 %.3g  bf16 %.3g  bad %lld\n", name, maxe, maxb, nb);
            bad += nb != 0;
        };
        for (int fr = 0; fr < 2; ++fr) {
            Bf16GemmArgs g; memset(&g, 0, sizeof g);
            g.M = M; g.N = N; g.K = K; g.A = dAkc; g.lda = K; g.B = dBkc; g.ldb = K; g.k_chunk = K; g.bias = dbias; g.C = dC; g.ldc = N; g.Cb = dCb; g.ldcb = N; g.cb_frag_mask = fr;
            CK(hipMemset(dC, 0, (size_t)M * N * 12));
            launch<L_KCONTIG, L_KCONTIG, BG_TANH>(g, 1);
            check(fr ? "forward (k-contig x k-contig, tanh, frag out)" : "forward (k-contig x k-contig, tanh)", [&](int m, int n) { return std::tanh(ref[(size_t)m * N + n] + bias[n]); }, true, fr, 1);
        }
        {
            Bf16GemmArgs g; memset(&g, 0, sizeof g);
            g.M = M; g.N = N; g.K = K; g.A = dAkc; g.lda = K; g.B = dBmc; g.ldb = N; g.k_chunk = K; g.C = dC; g.ldc = N; g.Cb = dCb; g.ldcb = N; g.Hb = dHb; g.ldh = N;
            CK(hipMemset(dC, 0, (size_t)M * N * 12));
            launch<L_KCONTIG, L_MCONTIG, BG_DELTA>(g, 1);
            check("delta (k-contig x row-contig view, 1 - a^2)", [&](int m, int n) { const double a = bf2f(Hb[(size_t)m * N + n]); return ref[(size_t)m * N + n] * (1 - a * a); }, true, false, 1);
        }
        {
            Bf16GemmArgs g; memset(&g, 0, sizeof g);
            g.M = M; g.N = N; g.K = K; g.A = dAmc; g.lda = M; g.B = dBmc; g.ldb = N; g.k_chunk = 64; g.slab_stride = (long long)M * N; g.C = dC; g.ldc = N;
            CK(hipMemset(dC, 0, (size_t)M * N * 12));
            launch<L_MCONTIG, L_MCONTIG, BG_STORE>(g, 3);
            check("weight gradient (row-contig x row-contig, 3 slabs)", [&](int m, int n) { return ref[(size_t)m * N + n]; }, false, false, 3);
        }
    }
    {   // ---- timing at the configs[4] shapes ----
        const int n = 16384, H = 512;
        std::vector<uint16_t> act((size_t)n * H), W((size_t)H * H);
        for (auto& x : act) x = f2bf(std::tanh(2 * u(rng)));
        for (auto& x : W) x = f2bf(0.05f * u(rng));
        uint16_t *dA[3], *dW[3], *dO[3]; float* dbias; float* dC;
        for (int k = 0; k < 3; ++k) { dA[k] = dev(act); dW[k] = dev(W); CK(hipMalloc(&dO[k], (size_t)n * H * 2)); }
        std::vector<float> b(H, 0.01f); dbias = dev(b);
        CK(hipMalloc(&dC, (size_t)std::max((size_t)n * H, (size_t)32 * H * H) * 4));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto timeit = [&](const char* name, double flop, auto fn) {
            float ms = 0; fn(); fn();
            hipEventRecord(e0, 0); for (int i = 0; i < 20; ++i) fn(); hipEventRecord(e1, 0); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
            printf("  %-58s %7.2f us  %6.0f TFLOP/s\n", name, ms * 50, flop / (ms / 20 * 1e-3) / 1e12);
        };
        Bf16GemmArgs g; memset(&g, 0, sizeof g);
        g.M = n; g.N = H; g.K = H; g.A = dA[0]; g.lda = H; g.B = dW[0]; g.ldb = H; g.k_chunk = H; g.bias = dbias; g.Cb = dO[0]; g.ldcb = H;
        timeit("forward 16384 x 512 x 512, one chain", 2.0 * n * H * H, [&] { launch<L_KCONTIG, L_KCONTIG, BG_TANH>(g, 1); });
        g.groups = 3; for (int k = 0; k < 2; ++k) { g.Ax[k] = dA[k + 1]; g.Bx[k] = dW[k + 1]; g.biasx[k] = dbias; g.Cbx[k] = dO[k + 1]; }
        timeit("forward, three chains grouped", 6.0 * n * H * H, [&] { launch<L_KCONTIG, L_KCONTIG, BG_TANH>(g, 3); });
        GemmArgs o; memset(&o, 0, sizeof o);
        o.M = n; o.N = H; o.K = H / 2; o.lda = o.ldb = H / 2; o.A = (const float*)dA[0]; o.B = (const float*)dW[0]; o.bias = dbias; o.Cb = dO[0]; o.ldcb = H; o.ldc = H;
        o.a_vec = o.b_vec = 1; o.k_chunk = o.K; o.grouped = 3;
        for (int k = 0; k < 2; ++k) { o.Ax[k] = (const float*)dA[k + 1]; o.Bx[k] = (const float*)dW[k + 1]; o.biasx[k] = dbias; o.Cbx[k] = dO[k + 1]; }
        timeit("  (tile kernel of xq_gemm.hip.h, three chains grouped)", 6.0 * n * H * H,
               [&] { hipLaunchKernelGGL((gemm_f32_kernel<L_KCONTIG, L_KCONTIG, EPI_BIAS_TANH, 2, 2, DT_BF16>), dim3(n / 128, H / 128, 3), dim3(256), 0, 0, o); });
        memset(&g, 0, sizeof g);
        g.M = n; g.N = H; g.K = H; g.A = dA[0]; g.lda = H; g.B = dW[0]; g.ldb = H; g.k_chunk = H; g.C = dC; g.ldc = H; g.Cb = dO[0]; g.ldcb = H; g.Hb = dA[1]; g.ldh = H;
        timeit("delta 16384 x 512 x 512 (fp32 + bf16 out)", 2.0 * n * H * H, [&] { launch<L_KCONTIG, L_MCONTIG, BG_DELTA>(g, 1); });
        memset(&g, 0, sizeof g);
        g.M = H; g.N = H; g.K = n; g.A = dA[0]; g.lda = H; g.B = dA[1]; g.ldb = H; g.k_chunk = 512; g.slab_stride = (long long)H * H; g.C = dC; g.ldc = H;
        timeit("weight gradient 512 x 512 x 16384, 32 slabs", 2.0 * n * H * H, [&] { launch<L_MCONTIG, L_MCONTIG, BG_STORE>(g, 32); });
    }
    {   // ---- fp32 forward on the same loop: bit-identical to the tile kernel of xq_gemm.hip.h (same k association), timing ----
        for (int H : {512, 256}) {
            const int n = 8192;
            std::vector<float> a((size_t)n * H), w((size_t)H * H), b(H);
            for (auto& x : a) x = u(rng); for (auto& x : w) x = 0.05f * u(rng); for (auto& x : b) x = 0.1f * u(rng);
            float *dA = dev(a), *dW = dev(w), *dB = dev(b), *dC0, *dC1; uint16_t* dCb;
            CK(hipMalloc(&dC0, a.size() * 4)); CK(hipMalloc(&dC1, a.size() * 4)); CK(hipMalloc(&dCb, a.size() * 2));
            GemmArgs o; memset(&o, 0, sizeof o);
            o.M = n; o.N = H; o.K = H; o.A = dA; o.lda = H; o.B = dW; o.ldb = H; o.C = dC0; o.ldc = H; o.bias = dB; o.a_vec = o.b_vec = 1; o.k_chunk = H;
            auto old_fn = [&] { hipLaunchKernelGGL((gemm_f32_kernel<L_KCONTIG, L_KCONTIG, EPI_BIAS_TANH, 1, 1>), dim3(n / 64, H / 64, 1), dim3(256), 0, 0, o); };
            Bf16GemmArgs g; memset(&g, 0, sizeof g);
            g.M = n; g.N = H; g.K = H; g.A = dA; g.lda = H; g.B = dW; g.ldb = H; g.k_chunk = H; g.bias = dB; g.C = dC1; g.ldc = H; g.Cb = dCb; g.ldcb = H; g.cb_frag_mask = 1;
            auto new_fn = [&] { if (H == 512) launch<L_KCONTIG, L_KCONTIG, BG_TANH, DT_F32, 1, 2>(g, 1); else launch<L_KCONTIG, L_KCONTIG, BG_TANH, DT_F32, 1, 1>(g, 1); };
            old_fn(); new_fn(); CK(hipDeviceSynchronize());
            std::vector<float> c0(a.size()), c1(a.size()); std::vector<uint16_t> cb(a.size());
            CK(hipMemcpy(c0.data(), dC0, c0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), dC1, c1.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(cb.data(), dCb, cb.size() * 2, hipMemcpyDeviceToHost));
            long long diff = 0, cbbad = 0; double maxe = 0;
            for (size_t i = 0; i < c0.size(); ++i) { if (memcmp(&c0[i], &c1[i], 4)) ++diff; }
            for (int m = 0; m < n; m += 37) for (int nn = 0; nn < H; ++nn) {
                double s2 = b[nn]; for (int k = 0; k < H; ++k) s2 += (double)a[(size_t)m * H + k] * w[(size_t)nn * H + k];
                maxe = std::max(maxe, std::fabs(std::tanh(s2) - c1[(size_t)m * H + nn]));
                if (cb[(size_t)scr_afrag_index(m, nn, H)] != f2bf(c1[(size_t)m * H + nn])) ++cbbad;
            }
            printf("fp32 forward 8192 x %d x %d: elements that differ from the tile kernel %lld, max |err| vs fp64 %.3g, bf16 fragment-order copy mismatches %lld\n", H, H, diff, maxe, cbbad);
            bad += (diff != 0) + (maxe > 2e-6) + (cbbad != 0);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto timeit = [&](const char* name, double flop, auto fn) {
                float ms = 0; fn(); fn();
                hipEventRecord(e0, 0); for (int i = 0; i < 20; ++i) fn(); hipEventRecord(e1, 0); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
                printf("  %-58s %7.2f us  %6.1f TFLOP/s = %.3f of 157.3\n", name, ms * 50, flop / (ms / 20 * 1e-3) / 1e12, flop / (ms / 20 * 1e-3) / 157.3e12);
            };
            timeit("tile kernel, 64 x 64 tiles", 2.0 * n * H * H, old_fn);
            timeit("LDS-DMA loop", 2.0 * n * H * H, new_fn);
            g.Cb = nullptr;
            timeit("LDS-DMA loop, no bf16 copy", 2.0 * n * H * H, new_fn);
            g.groups = 2; g.Ax[0] = dA; g.Bx[0] = dW; g.biasx[0] = dB; g.Cx[0] = dC0;
            timeit("LDS-DMA loop, two chains grouped", 4.0 * n * H * H, [&] { if (H == 512) launch<L_KCONTIG, L_KCONTIG, BG_TANH, DT_F32, 1, 2>(g, 2); else launch<L_KCONTIG, L_KCONTIG, BG_TANH, DT_F32, 1, 1>(g, 2); });
            if (H == 512) timeit("LDS-DMA loop, two chains grouped, 256 x 128 tiles", 4.0 * n * H * H, [&] { launch<L_KCONTIG, L_KCONTIG, BG_TANH, DT_F32, 2, 2>(g, 2); });
        }
    }
    return bad;
}
