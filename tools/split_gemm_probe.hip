// tools/split_gemm_probe.hip — measured lead, NOT product code (round 4): the TD step's grouped hidden product (2 chains of
// 8192 x 256 x 256, bias + tanh) with every fp32 operand split EXACTLY into three bf16 planes (x = hi + mid + lo: each plane the
// truncated upper half of what is left, 8 + 8 + 8 significant bits) and all nine plane products on v_mfma_f32_32x32x16_bf16 — the
// same exact partial products as an fp32 multiply, accumulated in fp32, for 9 x 32 matrix-pipe cycles per 16 k against 8 x 64 of
// v_mfma_f32_32x32x2_f32 (1.78 x; the six largest terms alone: 2.67 x, error 2^-23 of |a||b| per product).
//
// Planes are prepared by kernels outside the timed region (in a product they would come out of the producers' epilogues and the SGD
// kernel).  One 512-thread block per 32 samples: both chains' A planes resident in LDS ([3][32][256] bf16 each), W1's planes streamed
// through a double-buffered stage of 16 k ([3][256][16] bf16, 16-byte chunks XOR-swizzled), wave w owns output columns 32 w .. 32 w + 31.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_build/split_gemm_probe tools/split_gemm_probe.hip && tools/_build/split_gemm_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int H = 256, BS = 32, ARS = H + 8;     // A row stride in bf16 (528 B: 16-byte slots of consecutive rows rotate through a 256-B line)

__global__ void split3_kernel(const float* __restrict__ x, long long n, uint16_t* __restrict__ p0, uint16_t* __restrict__ p1, uint16_t* __restrict__ p2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const float hi = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) & 0xFFFF0000u);
    const float r1 = v - hi;
    const float mid = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, r1) & 0xFFFF0000u);
    const float r2 = r1 - mid;
    p0[i] = (uint16_t)(__builtin_bit_cast(uint32_t, hi) >> 16);
    p1[i] = (uint16_t)(__builtin_bit_cast(uint32_t, mid) >> 16);
    p2[i] = (uint16_t)(__builtin_bit_cast(uint32_t, r2) >> 16);          // at most 8 significant bits are left: exact
}

// TERMS = 9: all plane pairs; 6: (0,0) (0,1) (1,0) (0,2) (2,0) (1,1)
template <int TERMS>
__global__ __launch_bounds__(512) void split_gemm_kernel(const uint16_t* __restrict__ As /*[3][n][256]*/, const uint16_t* __restrict__ An, long long plane_a,
                                                         const uint16_t* __restrict__ Wp /*[3][256][256]*/, const float* __restrict__ b1,
                                                         float* __restrict__ out_s, float* __restrict__ out_n) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* A = lds;                                   // [2 chains][3][32][ARS]
    uint16_t* B = A + 2 * 3 * BS * ARS;                  // [2 stages][3][256][16]
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const long long row0 = (long long)blockIdx.x * BS;
    // A planes: 2 chains x 3 planes x 32 rows x 32 chunks of 16 B = 6144 chunks, 12 per thread
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        const int idx = q * 512 + tid, ch = idx & 31, rw = (idx >> 5) & 31, pl = (idx >> 10) % 3, cn = idx / 3072;
        const uint16_t* src = (cn ? An : As) + (long long)pl * plane_a + (row0 + rw) * H + ch * 8;
        *reinterpret_cast<uint4*>(A + ((cn * 3 + pl) * BS + rw) * ARS + ch * 8) = *reinterpret_cast<const uint4*>(src);
    }
    // B stage: 3 planes x 256 rows x 2 chunks = 1536 chunks, 3 per thread; chunk c of row n lands in slot c ^ ((n >> 3) & 1)
    uint4 breg[3];
#define XQ_BLOAD(kc_) \
    _Pragma("unroll") for (int q = 0; q < 3; ++q) { const int idx = q * 512 + tid, c = idx & 1, nn = (idx >> 1) & 255, pl = idx >> 9; \
        breg[q] = *reinterpret_cast<const uint4*>(Wp + ((long long)pl * H + nn) * H + (kc_) * 16 + c * 8); }
    XQ_BLOAD(0)
#define XQ_BSTORE(st_) \
    _Pragma("unroll") for (int q = 0; q < 3; ++q) { const int idx = q * 512 + tid, c = idx & 1, nn = (idx >> 1) & 255, pl = idx >> 9; \
        *reinterpret_cast<uint4*>(B + (((st_) * 3 + pl) * 256 + nn) * 16 + ((c ^ ((nn >> 3) & 1)) * 8)) = breg[q]; }
    XQ_BSTORE(0)
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
    const int ncol = wid * 32 + r;
    const int bslot = (h ^ ((ncol >> 3) & 1)) * 8;
    for (int kc = 0; kc < 16; ++kc) {
        const int st = kc & 1;
        if (kc + 1 < 16) { XQ_BLOAD(kc + 1) }
        bf16x8 fa[2][3], fb[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            fa[0][pl] = *reinterpret_cast<const bf16x8*>(A + ((0 * 3 + pl) * BS + r) * ARS + kc * 16 + h * 8);
            fa[1][pl] = *reinterpret_cast<const bf16x8*>(A + ((1 * 3 + pl) * BS + r) * ARS + kc * 16 + h * 8);
            fb[pl] = *reinterpret_cast<const bf16x8*>(B + ((st * 3 + pl) * 256 + ncol) * 16 + bslot);
        }
        // smallest terms first, so that the fp32 accumulator meets them before the large ones of this k-step
#define XQ_T(pa, pb) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][pa], fb[pb], acc[0], 0, 0, 0); \
                     acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][pa], fb[pb], acc[1], 0, 0, 0);
        if (TERMS == 9) { XQ_T(2, 2) XQ_T(1, 2) XQ_T(2, 1) }
        XQ_T(1, 1) XQ_T(0, 2) XQ_T(2, 0) XQ_T(0, 1) XQ_T(1, 0) XQ_T(0, 0)
#undef XQ_T
        if (kc + 1 < 16) { XQ_BSTORE(st ^ 1) __syncthreads(); }
    }
    const float bo = b1[ncol];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        out_s[(row0 + row) * H + ncol] = tanhf(acc[0][e] + bo);
        out_n[(row0 + row) * H + ncol] = tanhf(acc[1][e] + bo);
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192;
    std::mt19937 rng(3);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<float> as((size_t)n * H), an((size_t)n * H), W((size_t)H * H), b1(H);
    for (auto& x : as) x = std::tanh(2.f * u(rng)); for (auto& x : an) x = std::tanh(2.f * u(rng));
    for (auto& x : W) x = 0.05f * u(rng); for (auto& x : b1) x = 0.01f * u(rng);
    float *dAs, *dAn, *dW, *db, *dOs, *dOn;
    uint16_t *pAs, *pAn, *pW;
    const size_t act = (size_t)n * H;
    CK(hipMalloc(&dAs, act * 4)); CK(hipMalloc(&dAn, act * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&db, H * 4));
    CK(hipMalloc(&dOs, act * 4)); CK(hipMalloc(&dOn, act * 4));
    CK(hipMalloc(&pAs, act * 2 * 3)); CK(hipMalloc(&pAn, act * 2 * 3)); CK(hipMalloc(&pW, W.size() * 2 * 3));
    CK(hipMemcpy(dAs, as.data(), act * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dAn, an.data(), act * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b1.data(), H * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((act + 255) / 256)), dim3(256), 0, 0, dAs, (long long)act, pAs, pAs + act, pAs + 2 * act);
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((act + 255) / 256)), dim3(256), 0, 0, dAn, (long long)act, pAn, pAn + act, pAn + 2 * act);
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((W.size() + 255) / 256)), dim3(256), 0, 0, dW, (long long)W.size(), pW, pW + W.size(), pW + 2 * W.size());
    CK(hipDeviceSynchronize());
    const size_t lds = (size_t)(2 * 3 * BS * ARS + 2 * 3 * 256 * 16) * 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(split_gemm_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(split_gemm_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto check = [&](const char* name) {
        std::vector<float> os(act), on(act);
        CK(hipMemcpy(os.data(), dOs, act * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(on.data(), dOn, act * 4, hipMemcpyDeviceToHost));
        double worst = 0, worst_z = 0;
        for (int t = 0; t < 96; ++t) {
            const int b = (int)(((long long)t * 2654435761u) % n);
            for (int chain = 0; chain < 2; ++chain) {
                const float* a = (chain ? an.data() : as.data()) + (size_t)b * H;
                for (int j = 0; j < H; ++j) {
                    double y = b1[j]; for (int k = 0; k < H; ++k) y += (double)W[(size_t)j * H + k] * (double)a[k];
                    const double got = chain ? on[(size_t)b * H + j] : os[(size_t)b * H + j];
                    worst = std::max(worst, std::fabs(std::tanh(y) - got));
                    worst_z = std::max(worst_z, std::fabs(y - std::atanh(std::min(0.999999, std::max(-0.999999, got)))));
                }
            }
        }
        printf("  %-34s max |tanh(z) - fp64| over 96 samples x 2 chains: %.2e\n", name, worst);
    };
    auto timeit = [&](const char* name, auto fn) {
        fn(); fn(); float ms = 0;
        hipEventRecord(e0, 0); for (int i = 0; i < 20; ++i) fn(); hipEventRecord(e1, 0); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 2 * n * (double)H * H;
        printf("  %-34s %7.2f us   (%.1f TFLOP/s of fp32-equivalent work; the fp32 pipe's peak is 157.3)\n", name, ms * 50, flop / (ms / 20 * 1e-3) / 1e12);
    };
    printf("n %d x %d x %d, 2 chains, LDS %zu B per block, grid %d x 512 threads\n", n, H, H, lds, n / BS);
    timeit("nine bf16 products per fp32 one", [&] { hipLaunchKernelGGL(split_gemm_kernel<9>, dim3(n / BS), dim3(512), lds, 0, pAs, pAn, (long long)act, pW, db, dOs, dOn); });
    CK(hipDeviceSynchronize()); check("nine terms:");
    timeit("six largest terms", [&] { hipLaunchKernelGGL(split_gemm_kernel<6>, dim3(n / BS), dim3(512), lds, 0, pAs, pAn, (long long)act, pW, db, dOs, dOn); });
    CK(hipDeviceSynchronize()); check("six terms:");
    printf("  (the product's fp32-MFMA kernel for the same two chains: 26.3 us, max error 1.6e-07 — tools/f32_fwd_probe.hip, tools/fused_fwd_probe.hip)\n");
    return 0;
}
