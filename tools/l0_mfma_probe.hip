// tools/l0_mfma_probe.hip — the layer-0 gradient on the bf16 matrix pipe (cn_chess_ai_amd/csrc/xq_l0grad.hip.h) against a CPU fp64
// reference, its time and its ablations: l0_sel_kernel + delta_split_kernel + l0_grad_mfma_kernel at the headline shape
// (8192 samples x 256) or at argv's.  usage: l0_mfma_probe [n] [H] [chunk]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../cn_chess_ai_amd/csrc/xq_l0grad.hip.h"
using namespace xq;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192, H = argc > 2 ? atoi(argv[2]) : 256;
    const int chunk = argc > 3 ? atoi(argv[3]) : 1024, nch = (n + chunk - 1) / chunk, kpad = nch * chunk;
    if (chunk % 64 || H % 32) { printf("chunk %% 64 and H %% 32 must be 0\n"); return 3; }
    std::mt19937 rng(7);
    std::vector<uint32_t> boards((size_t)n * 12, 0);
    std::vector<uint8_t> sq((size_t)n * 90, 0);
    for (int b = 0; b < n; ++b) {
        const int pieces = 8 + rng() % 25;                       // 8..32 pieces on random squares
        for (int p = 0; p < pieces; ++p) { const int s = rng() % 90; sq[(size_t)b * 90 + s] = 1 + rng() % 14; }
        for (int s = 0; s < 90; ++s) boards[(size_t)b * 12 + (s >> 3)] |= (uint32_t)sq[(size_t)b * 90 + s] << (4 * (s & 7));
    }
    std::vector<float> d0((size_t)n * H);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    for (size_t i = 0; i < d0.size(); ++i) { const float e = std::ldexp(1.f, (int)(rng() % 30) - 20); d0[i] = u(rng) * e * 1000.f; }   // 1e-3 .. 5e5
    uint32_t* dB; float* dD; uint16_t* dP; uint16_t* dS; float* dOut;
    const long long plane_stride = (long long)H * kpad;
    CK(hipMalloc(&dB, boards.size() * 4)); CK(hipMalloc(&dD, d0.size() * 4));
    CK(hipMalloc(&dP, l0m_plane_elems(H, kpad) * 2)); CK(hipMemset(dP, 0, l0m_plane_elems(H, kpad) * 2));
    CK(hipMalloc(&dS, l0sel_elems(kpad) * 2)); CK(hipMemset(dS, 0, l0sel_elems(kpad) * 2));
    CK(hipMalloc(&dOut, (size_t)nch * 1260 * H * 4));
    CK(hipMemcpy(dB, boards.data(), boards.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dD, d0.data(), d0.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dOut, 0xFF, (size_t)nch * 1260 * H * 4));
    const size_t lds = l0m_lds_bytes();
    auto grant = [&](auto kern) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); };
    grant(l0_grad_mfma_kernel<0>); grant(l0_grad_mfma_kernel<1>); grant(l0_grad_mfma_kernel<2>); grant(l0_grad_mfma_kernel<3>);
    grant(l0_grad_mfma_kernel<4>); grant(l0_grad_mfma_kernel<6>);
    auto selk = [&] { hipLaunchKernelGGL(l0_sel_kernel, dim3(kpad / 64), dim3(256), 0, 0, dB, n, kpad, dS); };
    auto split = [&] { hipLaunchKernelGGL(delta_split_kernel, dim3(kpad / 64, H / 64), dim3(256), 0, 0, dD, n, H, dP, plane_stride, kpad); };
    auto mfma = [&] { hipLaunchKernelGGL(l0_grad_mfma_kernel<0>, dim3(H / kL0mCols, 4, nch), dim3(256), lds, 0, dS, dP, plane_stride, kpad, H, chunk, dOut); };
    selk(); split(); mfma(); CK(hipDeviceSynchronize());
    std::vector<float> out((size_t)nch * 1260 * H);
    CK(hipMemcpy(out.data(), dOut, out.size() * 4, hipMemcpyDeviceToHost));
    // reference: chunk partial sums in fp64
    double worst_rel = 0; long long checked = 0;
    std::vector<double> ref((size_t)1260 * H), mag((size_t)1260 * H);
    for (int ch = 0; ch < nch; ++ch) {
        std::fill(ref.begin(), ref.end(), 0.0); std::fill(mag.begin(), mag.end(), 0.0);
        for (int b = ch * chunk; b < std::min(n, (ch + 1) * chunk); ++b)
            for (int s = 0; s < 90; ++s) {
                const int c = sq[(size_t)b * 90 + s];
                if (!c) continue;
                double* r = &ref[(size_t)(s * 14 + c - 1) * H]; double* mg = &mag[(size_t)(s * 14 + c - 1) * H];
                const float* dv = &d0[(size_t)b * H];
                for (int k = 0; k < H; ++k) { r[k] += dv[k]; mg[k] += std::fabs(dv[k]); }
            }
        for (size_t i = 0; i < ref.size(); ++i) {
            const double got = out[(size_t)ch * 1260 * H + i];
            const double err = std::fabs(got - ref[i]);
            if (mag[i] > 0) worst_rel = std::max(worst_rel, err / mag[i]);      // relative to the sum of magnitudes (fp32 summation bound)
            else if (got != 0.0) { printf("non-zero where nothing was summed: chunk %d idx %zu = %g\n", ch, i, got); return 1; }
            ++checked;
        }
    }
    printf("n %d H %d chunk %d: %lld outputs checked, worst |err| / sum|terms| = %.3g (fp32 eps 6e-8; sequential fp32 summation of ~350 terms: ~2e-5 worst case)\n",
           n, H, chunk, checked, worst_rel);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    auto time20 = [&](auto fn) { fn(); CK(hipEventRecord(e0, 0)); for (int i = 0; i < 20; ++i) fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 50; };
    for (int rep = 0; rep < 2; ++rep) {
        printf("  l0_sel_kernel        : %7.2f us  (%.1f MB out)\n", time20(selk), l0sel_elems(kpad) * 2 / 1e6);
        printf("  delta_split_kernel   : %7.2f us  (%.1f MB in, %.1f MB out)\n", time20(split), n * (double)H * 4 / 1e6, 3.0 * plane_stride * 2 / 1e6);
        const double us = time20(mfma), fl = 2.0 * 90 * 16 * (double)H * kpad * 3;
        printf("  l0_grad_mfma_kernel  : %7.2f us  grid %d x %d x %d, %.1f GFLOP bf16 = %.0f TFLOP/s\n", us, 4, H / kL0mCols, nch, fl / 1e9, fl / (us * 1e-6) / 1e12);
        printf("  all three            : %7.2f us\n", time20([&] { selk(); split(); mfma(); }));
    }
    auto dbg = [&](auto kern, const char* what) {
        auto fn = [&] { hipLaunchKernelGGL(kern, dim3(H / kL0mCols, 4, nch), dim3(256), lds, 0, dS, dP, plane_stride, kpad, H, chunk, dOut); };
        printf("  ablation %-44s: %7.2f us\n", what, time20(fn));
    };
    dbg(l0_grad_mfma_kernel<1>, "prologue + epilogue only (no stages)");
    dbg(l0_grad_mfma_kernel<2>, "no MFMAs (operands kept alive)");
    dbg(l0_grad_mfma_kernel<3>, "no v_perm (one-hot operand = selector word)");
    dbg(l0_grad_mfma_kernel<4>, "no LDS-DMA in the loop");
    dbg(l0_grad_mfma_kernel<6>, "MFMAs only (no DMA, LDS reads, v_perm)");
    return worst_rel < 1e-5 ? 0 : 2;
}
