// tools/f32_fwd_probe.hip — the fp32 hidden-layer forward product of configs[3] (8192 x 512 x 512, bias + tanh) on the tile kernel of
// xq_gemm.hip.h with different block tiles: which tiling fills 256 CUs best when there are only 256 128x128 tiles?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include <algorithm>
#include <cmath>
#include "../cn_chess_ai_amd/csrc/xq_gemm.hip.h"
using namespace xq;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int TM, int TN, int EPI = EPI_BIAS_TANH> static void run(const char* name, GemmArgs g, hipEvent_t e0, hipEvent_t e1) {
    dim3 grid((g.M + 64 * TM - 1) / (64 * TM), (g.N + 64 * TN - 1) / (64 * TN), g.grouped ? g.grouped : 1);
    auto fn = [&] { hipLaunchKernelGGL((gemm_f32_kernel<L_KCONTIG, L_KCONTIG, EPI, TM, TN>), grid, dim3(256), 0, 0, g); };
    fn(); fn();
    float ms = 0;
    hipEventRecord(e0, 0); for (int i = 0; i < 20; ++i) fn(); hipEventRecord(e1, 0); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * g.M * g.N * g.K * (g.grouped ? g.grouped : 1);
    printf("  %-40s grid %4d x %d x %d: %7.2f us  %6.1f TFLOP/s = %.3f of 157.3\n", name, grid.x, grid.y, grid.z, ms * 50, flop / (ms / 20 * 1e-3) / 1e12, flop / (ms / 20 * 1e-3) / 157.3e12);
}
template <int TM, int TN, int WPE> static void run_persistent(const char* name, GemmArgs g, hipEvent_t e0, hipEvent_t e1) {
    const int tiles_m = g.M / (64 * TM), tiles_n = g.N / (64 * TN), total = tiles_m * tiles_n * (g.grouped ? g.grouped : 1);
    const int grid = std::min(total, 256 * WPE);
    auto fn = [&] { hipLaunchKernelGGL((gemm_fwd_persistent_kernel<TM, TN, WPE>), dim3(grid), dim3(256), 0, 0, g, tiles_m, tiles_n, total); };
    fn(); fn();
    float ms = 0;
    hipEventRecord(e0, 0); for (int i = 0; i < 20; ++i) fn(); hipEventRecord(e1, 0); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * g.M * g.N * g.K * (g.grouped ? g.grouped : 1);
    printf("  %-40s grid %4d (%d tiles)   : %7.2f us  %6.1f TFLOP/s = %.3f of 157.3\n", name, grid, total, ms * 50, flop / (ms / 20 * 1e-3) / 1e12, flop / (ms / 20 * 1e-3) / 157.3e12);
}
// VERDICT r4 Next #2: two waves per SIMD WITHOUT a second tile per CU — a 512-thread block per 128 x 128 tile, waves 0-3 take the first half
// of the k range, waves 4-7 the second (each group with its own operand LDS, the loop of gemm_mainloop unchanged), the two accumulator sets
// are added through LDS and every group finishes HALF of the tile (bias + tanh).  Same k association inside a half; the halves are added last.
template <int EPI>
__global__ __launch_bounds__(512) void gemm_f32_ksplit_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float ks_smem[];
    constexpr int TILE = g_tile_floats(128);
    const int grp = (int)threadIdx.x >> 8, tid = (int)threadIdx.x & 255, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5, wm = wid >> 1, wn = wid & 1;
    float* As = ks_smem + grp * 2 * TILE;
    float* Bs = As + TILE;
    const int m0 = (int)blockIdx.x * 128, n0 = (int)blockIdx.y * 128;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const int kh = g.K / 2;
    gemm_mainloop<L_KCONTIG, L_KCONTIG, 2, 2, true>(g, As, Bs, m0, n0, grp * kh, (grp + 1) * kh, acc);
    __syncthreads();
    // exchange: group 0 finishes row tiles i = 0 of every wave's 64 x 64 quadrant, group 1 row tiles i = 1; each hands the other half over
    float* red = ks_smem;                                     // [2 groups][2 j][16 q][256 threads]
    const int give = grp == 0 ? 1 : 0, keep = 1 - give;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) red[((grp * 2 + j) * 16 + q) * 256 + tid] = acc[give][j][q];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + r;
        const float bias = g.bias[n];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float v = acc[keep][j][q] + red[(((1 - grp) * 2 + j) * 16 + q) * 256 + tid];
            if (EPI == EPI_BIAS_TANH) v = tanh_hidden(v + bias);
            g.C[(long long)(m0 + wm * 64 + keep * 32 + 4 * h + (q & 3) + 8 * (q >> 2)) * g.ldc + n] = v;
        }
    }
}
// the 64 x 64 tile kernel allowed MORE waves per SIMD than the library's amdgpu_waves_per_eu(1, 2): all tiles of an 8192 x 512 product (1024 = 4 per CU)
// resident at once, prologues and epilogues of some under the loops of others
template <int EPI, int WMAX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, WMAX))) void gemm_f32_occ_kernel(const GemmArgs g_in) {
    __shared__ __attribute__((aligned(16))) float As[g_tile_floats(64)];
    __shared__ __attribute__((aligned(16))) float Bs[g_tile_floats(64)];
    gemm_f32_block<L_KCONTIG, L_KCONTIG, EPI, 1, 1, DT_F32>(g_in, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, As, Bs);
}
template <int EPI, int WMAX> static void run_occ(const char* name, GemmArgs g, hipEvent_t e0, hipEvent_t e1) {
    dim3 grid(g.M / 64, g.N / 64, g.grouped ? g.grouped : 1);
    auto fn = [&] { hipLaunchKernelGGL((gemm_f32_occ_kernel<EPI, WMAX>), grid, dim3(256), 0, 0, g); };
    fn(); fn();
    float ms = 0;
    hipEventRecord(e0, 0); for (int i = 0; i < 20; ++i) fn(); hipEventRecord(e1, 0); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * g.M * g.N * g.K * (g.grouped ? g.grouped : 1);
    printf("  %-40s grid %4d x %d x %d: %7.2f us  %6.1f TFLOP/s = %.3f of 157.3\n", name, grid.x, grid.y, grid.z, ms * 50, flop / (ms / 20 * 1e-3) / 1e12, flop / (ms / 20 * 1e-3) / 157.3e12);
}
template <int EPI = EPI_BIAS_TANH> static void run_ksplit(const char* name, GemmArgs g, hipEvent_t e0, hipEvent_t e1) {
    const size_t lds = (size_t)4 * g_tile_floats(128) * sizeof(float);
    CK(hipFuncSetAttribute((const void*)gemm_f32_ksplit_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(g.M / 128, g.N / 128);
    auto fn = [&] { hipLaunchKernelGGL((gemm_f32_ksplit_kernel<EPI>), grid, dim3(512), lds, 0, g); };
    fn(); fn();
    float ms = 0;
    hipEventRecord(e0, 0); for (int i = 0; i < 20; ++i) fn(); hipEventRecord(e1, 0); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * g.M * g.N * g.K;
    printf("  %-40s grid %4d x %d (512 thr, %zu KB LDS): %7.2f us  %6.1f TFLOP/s = %.3f of 157.3\n", name, grid.x, grid.y, lds / 1024, ms * 50, flop / (ms / 20 * 1e-3) / 1e12, flop / (ms / 20 * 1e-3) / 157.3e12);
}
static double max_diff(const float* dX, const float* dY, size_t n) {
    std::vector<float> x(n), y(n);
    CK(hipMemcpy(x.data(), dX, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), dY, n * 4, hipMemcpyDeviceToHost));
    double m = 0; for (size_t i = 0; i < n; ++i) m = std::max(m, (double)fabsf(x[i] - y[i]));
    return m;
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192, H = argc > 2 ? atoi(argv[2]) : 512;
    std::mt19937 rng(1); std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<float> a((size_t)n * H), w((size_t)H * H), b(H, 0.01f);
    for (auto& x : a) x = u(rng); for (auto& x : w) x = 0.05f * u(rng);
    float *dA, *dW, *dB, *dC;
    CK(hipMalloc(&dA, a.size() * 4)); CK(hipMalloc(&dW, w.size() * 4)); CK(hipMalloc(&dB, b.size() * 4)); CK(hipMalloc(&dC, a.size() * 4 * 2));
    CK(hipMemcpy(dA, a.data(), a.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, w.data(), w.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    GemmArgs g; memset(&g, 0, sizeof g);
    g.M = n; g.N = H; g.K = H; g.A = dA; g.lda = H; g.B = dW; g.ldb = H; g.C = dC; g.ldc = H; g.bias = dB; g.a_vec = g.b_vec = 1; g.k_chunk = H;
    printf("n %d H %d\n", n, H);
    run<1, 1>("64 x 64 tiles", g, e0, e1);
    run<2, 1>("128 x 64 tiles", g, e0, e1);
    run<1, 2>("64 x 128 tiles", g, e0, e1);
    run<2, 2>("128 x 128 tiles", g, e0, e1);
    {   // the in-block k split against the tile kernel: same product (the two halves of k are added last: not bitwise the same sum)
        float* dC3; CK(hipMalloc(&dC3, a.size() * 4));
        GemmArgs k = g; k.C = dC3;
        run_ksplit<EPI_BIAS_TANH>("128 x 128 tile, k split over 8 waves", k, e0, e1);
        run<1, 1>("64 x 64 tiles (again)", g, e0, e1);
        run_ksplit<EPI_BIAS_TANH>("128 x 128 tile, k split over 8 waves", k, e0, e1);
        printf("  k-split against the 64 x 64 tile kernel: max |diff| %.3g\n", max_diff(dC, dC3, a.size()));
        CK(hipFree(dC3));
        run_occ<EPI_BIAS_TANH, 2>("64 x 64 tiles, <= 2 waves per SIMD (library)", g, e0, e1);
        run_occ<EPI_BIAS_TANH, 3>("64 x 64 tiles, <= 3 waves per SIMD", g, e0, e1);
        run_occ<EPI_BIAS_TANH, 4>("64 x 64 tiles, <= 4 waves per SIMD", g, e0, e1);
        run_occ<EPI_BIAS_TANH, 2>("64 x 64 tiles, <= 2 waves per SIMD (again)", g, e0, e1);
        run_occ<EPI_BIAS_TANH, 4>("64 x 64 tiles, <= 4 waves per SIMD (again)", g, e0, e1);
    }
    run<1, 1, EPI_STORE>("64 x 64 tiles, plain store (no tanh)", g, e0, e1);
    run<2, 1, EPI_STORE>("128 x 64 tiles, plain store (no tanh)", g, e0, e1);
    run<2, 2, EPI_STORE>("128 x 128 tiles, plain store (no tanh)", g, e0, e1);
    g.grouped = 2; g.Ax[0] = dA; g.Bx[0] = dW; g.Cx[0] = dC + a.size(); g.biasx[0] = dB;
    run<1, 1>("two chains grouped, 64 x 64", g, e0, e1);
    run<2, 1>("two chains grouped, 128 x 64", g, e0, e1);
    run<2, 2>("two chains grouped, 128 x 128", g, e0, e1);
    printf("persistent walk (XQ_FAST_TANH=%d):\n", (int)XQ_FAST_TANH);
    run_persistent<1, 1, 2>("grouped persistent 64 x 64, 2 blocks/CU", g, e0, e1);
    run_persistent<2, 1, 2>("grouped persistent 128 x 64, 2 blocks/CU", g, e0, e1);
    run_persistent<1, 2, 2>("grouped persistent 64 x 128, 2 blocks/CU", g, e0, e1);
    run_persistent<2, 2, 1>("grouped persistent 128 x 128, 1 block/CU", g, e0, e1);
    run_persistent<2, 2, 2>("grouped persistent 128 x 128, 2 blocks/CU", g, e0, e1);
    {   // same bits as the tile kernel
        float* dC2; CK(hipMalloc(&dC2, a.size() * 4 * 2));
        GemmArgs g2 = g; g2.C = dC2; g2.Cx[0] = dC2 + a.size();
        CK(hipMemset(dC2, 0, a.size() * 8));
        hipLaunchKernelGGL((gemm_f32_kernel<L_KCONTIG, L_KCONTIG, EPI_BIAS_TANH, 1, 1>), dim3(g.M / 64, g.N / 64, 2), dim3(256), 0, 0, g);
        const int tm = g.M / 128, tn = g.N / 64;
        hipLaunchKernelGGL((gemm_fwd_persistent_kernel<2, 1, 2>), dim3(std::min(tm * tn * 2, 512)), dim3(256), 0, 0, g2, tm, tn, tm * tn * 2);
        CK(hipDeviceSynchronize());
        printf("  persistent 128 x 64 against the 64 x 64 tile kernel: max |diff| %.3g (both chains)\n", max_diff(dC, dC2, a.size() * 2));
    }
    g.grouped = 0;
    run_persistent<1, 1, 2>("one chain persistent 64 x 64", g, e0, e1);
    run_persistent<2, 1, 2>("one chain persistent 128 x 64", g, e0, e1);
    return 0;
}