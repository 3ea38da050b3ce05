// tools/gemm_bench.hip — micro-benchmark of xq_gemm.cuh on the dominant product (8100 x B x 256, column-max epilogue).
// Diagnostic only (not part of the library).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_bench.hip -o gemm_bench
#include "../cn_chess_ai_amd/csrc/xq_gemm.cuh"
#include <cstdio>
#include <vector>
using namespace xq;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int EPI, int TM, int TN>
float run(GemmArgs g, int iters) {
    dim3 grid((g.M + 64 * TM - 1) / (64 * TM), (g.N + 64 * TN - 1) / (64 * TN), 1);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_f32_kernel<L_KCONTIG, L_KCONTIG, EPI, TM, TN>), grid, dim3(256), 0, 0, g);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_f32_kernel<L_KCONTIG, L_KCONTIG, EPI, TM, TN>), grid, dim3(256), 0, 0, g);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main(int argc, char** argv) {
    const int M = 8100, N = argc > 1 ? atoi(argv[1]) : 8192, K = argc > 2 ? atoi(argv[2]) : 256;
    float *A, *B, *C, *bias, *partial;
    CK(hipMalloc(&A, (size_t)(M + 128) * K * 4)); CK(hipMalloc(&B, (size_t)(N + 128) * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMalloc(&bias, (size_t)M * 4)); CK(hipMalloc(&partial, (size_t)N * 2 * ((M + 63) / 64) * 4));
    std::vector<float> h((size_t)std::max(M, N) * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    CK(hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, (size_t)M * 4));
    GemmArgs g; memset(&g, 0, sizeof g);
    g.M = M; g.N = N; g.K = K; g.A = A; g.lda = K; g.B = B; g.ldb = K; g.C = C; g.ldc = N; g.bias = bias; g.partial = partial;
    g.k_chunk = K; g.a_vec = 1; g.b_vec = 1;
    const double fl = 2.0 * M * N * K;
    {
        float t = run<EPI_COLMAX, 2, 2>(g, 20);
        printf("colmax 128x128          : %8.1f us  %6.1f TF/s\n", t * 1e3, fl / t / 1e9);
    }
    for (int pct : {0, 50, 55, 60, 65, 70, 75, 80}) {        // share of the tiles given to the prioritised half (0 = no priority)
        const int grid = 512;
        const int tiles_m = (M + 127) / 128, total = tiles_m * ((N + 127) / 128);
        g.prio_split = pct ? grid / 2 : 0;
        g.prio_tiles = (int)((long long)total * pct / 100) / tiles_m * tiles_m;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float best = 1e9, sum = 0;
        for (int i = 0; i < 23; ++i) {
            hipEventRecord(a, 0);
            hipLaunchKernelGGL((gemm_colmax_persistent_kernel<2, 2>), dim3(grid), dim3(256), 0, 0, g, tiles_m, total);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            if (i >= 3) { sum += ms; best = std::min(best, ms); }
        }
        printf("persistent prio share %2d%%: avg %8.1f us (%6.1f TF/s)  best %8.1f us\n", pct, sum / 20 * 1e3, fl / (sum / 20) / 1e9, best * 1e3);
    }
    g.prio_split = 256; g.prio_tiles = (int)(4096LL * 60 / 100) / 64 * 64;
    // check: persistent == plain
    {
        std::vector<float> p0((size_t)N * 2 * ((M + 127) / 128)), p1(p0.size());
        hipLaunchKernelGGL((gemm_f32_kernel<L_KCONTIG, L_KCONTIG, EPI_COLMAX, 2, 2>), dim3((M + 127) / 128, (N + 127) / 128, 1), dim3(256), 0, 0, g);
        CK(hipMemcpy(p0.data(), partial, p0.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(partial, 0, p0.size() * 4));
        hipLaunchKernelGGL((gemm_colmax_persistent_kernel<2, 2>), dim3(512), dim3(256), 0, 0, g, (M + 127) / 128, ((M + 127) / 128) * ((N + 127) / 128));   // with priority split
        CK(hipMemcpy(p1.data(), partial, p1.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < p0.size(); ++i) bad += p0[i] != p1[i];
        printf("persistent vs plain: %zu mismatches of %zu\n", bad, p0.size());
    }
    return 0;
}
