// tools/cumask_probe.hip — does spatial partitioning with hipExtStreamCreateWithCUMask pay?  Runs the dominant persistent
// GEMM on a stream confined to `g` CUs while a chain of small 64-tile GEMMs (proxy for the gradient chain) runs on the rest.
// Diagnostic only.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/cumask_probe.hip -o cumask_probe
#include "../cn_chess_ai_amd/csrc/xq_gemm.hip.h"
#include <cstdio>
#include <vector>
using namespace xq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

static int masked_stream(hipStream_t* s, int first, int count) {
    uint32_t m[8] = {0};
    for (int i = first; i < first + count; ++i) m[i >> 5] |= 1u << (i & 31);
    return hipExtStreamCreateWithCUMask(s, 8, m) == hipSuccess ? 0 : 1;
}

int main() {
    const int M = 8100, N = 8192, K = 256;
    float *A, *B, *bias, *partial, *A2, *B2, *C2;
    CK(hipMalloc(&A, (size_t)(M + 128) * K * 4)); CK(hipMalloc(&B, (size_t)(N + 128) * K * 4));
    CK(hipMalloc(&bias, (size_t)(M + 256) * 4)); CK(hipMalloc(&partial, (size_t)N * 2 * ((M + 63) / 64) * 4));
    CK(hipMalloc(&A2, (size_t)8192 * 256 * 4)); CK(hipMalloc(&B2, (size_t)256 * 256 * 4)); CK(hipMalloc(&C2, (size_t)8192 * 256 * 4));
    CK(hipMemset(A, 0, (size_t)(M + 128) * K * 4)); CK(hipMemset(B, 0, (size_t)(N + 128) * K * 4)); CK(hipMemset(bias, 0, (size_t)(M + 256) * 4));
    CK(hipMemset(A2, 0, (size_t)8192 * 256 * 4)); CK(hipMemset(B2, 0, (size_t)256 * 256 * 4));
    GemmArgs g; memset(&g, 0, sizeof g);
    g.M = M; g.N = N; g.K = K; g.A = A; g.lda = K; g.B = B; g.ldb = K; g.ldc = N; g.bias = bias; g.partial = partial;
    g.k_chunk = K; g.a_vec = 1; g.b_vec = 1; g.bias_padded = 1;
    GemmArgs s; memset(&s, 0, sizeof s);
    s.M = 8192; s.N = 256; s.K = 256; s.A = A2; s.lda = 256; s.B = B2; s.ldb = 256; s.C = C2; s.ldc = 256; s.bias = bias;
    s.k_chunk = 256; s.a_vec = 1; s.b_vec = 1;
    const int tiles_m = (M + 127) / 128, total = tiles_m * ((N + 127) / 128);
    hipEvent_t a, b, c, d; hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&c); hipEventCreate(&d);
    for (int big : {256, 224, 192, 160, 128}) {
        hipStream_t sg, sc;
        if (masked_stream(&sg, 0, big)) { printf("CU mask stream creation failed\n"); return 1; }
        const int rest = big < 256 ? 256 - big : 256;
        if (masked_stream(&sc, big < 256 ? big : 0, rest)) { printf("CU mask stream creation failed\n"); return 1; }
        const int grid = 2 * big;
        auto gemm = [&]() { hipLaunchKernelGGL((gemm_colmax_persistent_kernel<2, 2>), dim3(grid), dim3(256), (size_t)tiles_m * 128 * 4, sg, g, tiles_m, total); };
        auto chain = [&]() { for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((gemm_f32_kernel<L_KCONTIG, L_KCONTIG, EPI_BIAS_TANH, 1, 1>), dim3(128, 4, 1), dim3(256), 0, sc, s); };
        float tg = 0, tc = 0, tboth_g = 0, tboth_c = 0;
        for (int i = 0; i < 3; ++i) { gemm(); chain(); }
        CK(hipDeviceSynchronize());
        hipEventRecord(a, sg); for (int i = 0; i < 5; ++i) gemm(); hipEventRecord(b, sg); hipEventSynchronize(b); hipEventElapsedTime(&tg, a, b);
        hipEventRecord(c, sc); for (int i = 0; i < 5; ++i) chain(); hipEventRecord(d, sc); hipEventSynchronize(d); hipEventElapsedTime(&tc, c, d);
        CK(hipDeviceSynchronize());
        hipEventRecord(a, sg); hipEventRecord(c, sc);
        for (int i = 0; i < 5; ++i) { gemm(); chain(); }
        hipEventRecord(b, sg); hipEventRecord(d, sc); hipEventSynchronize(b); hipEventSynchronize(d);
        hipEventElapsedTime(&tboth_g, a, b); hipEventElapsedTime(&tboth_c, c, d);
        printf("GEMM on %3d CUs, chain on %3d CUs: alone gemm %7.1f us chain(10 small gemms) %7.1f us | together gemm %7.1f us chain %7.1f us\n",
               big, rest, tg / 5 * 1e3, tc / 5 * 1e3, tboth_g / 5 * 1e3, tboth_c / 5 * 1e3);
        hipStreamDestroy(sg); hipStreamDestroy(sc);
    }
    return 0;
}
