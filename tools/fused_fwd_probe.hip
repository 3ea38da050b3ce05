// tools/fused_fwd_probe.hip — feasibility probe (round 4): layer 0 (gather of W0^T rows, s and the derived s') and the first hidden
// product of the TD step's two forward chains in ONE kernel, the activations of layer 1 never leaving the CU.
//
// Today: l0_forward_kernel (one wave per sample, 8192 waves: 14.7 us, L2-bound) -> kernel boundary -> gemm_fwd_persistent_kernel
// (64 x 128 tiles, 2 chains: 26-28 us, matrix-pipe-bound at 0.5 of the fp32 peak) = 42 us + a boundary on the step's dependency chain.
// Here: one 256-thread block per CU owns 32 samples; phase A = every wave gathers 8 samples one after the other with ALL row loads of
// a sample in flight at once (the list padded with the index of a zero row, so the sum keeps the reference's ascending order and
// bits), a_1(s) / a_1(s') land as two [32][256] fp32 A operands in LDS (a_1(s) also goes to HBM: the backward pass reads it);
// phase B = [32 x 256] x W1^T on v_mfma_f32_32x32x2_f32 for both chains from those images, W1 streamed through a double-buffered
// [256][32] LDS stage; bias + tanh epilogue.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_build/fused_fwd_probe tools/fused_fwd_probe.hip && tools/_build/fused_fwd_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int H = 256, kWords = 12, kZeroRow = 1260, BS = 32, ALD = H + 4, BLD = 36;

__device__ __forceinline__ uint32_t nib(const uint32_t* bw, int s) { return (bw[s >> 3] >> (4 * (s & 7))) & 15u; }

// ---- baseline 1: the gather as the product runs it (one wave per sample, 4 loads at a time), both chains written to HBM
__global__ __launch_bounds__(256) void gather_kernel(const uint32_t* __restrict__ boards, const uint32_t* __restrict__ nboards, const float* __restrict__ W0T,
                                                     const float* __restrict__ b0, float* __restrict__ a1s, float* __restrict__ a1n, int n) {
    __shared__ int rows[4][96];
    __shared__ int dpair[4][16];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.x * 4 + wid;
    if (b >= n) return;
    const uint32_t* bw = boards + (long long)b * kWords; const uint32_t* bw2 = nboards + (long long)b * kWords;
    const int s0 = lane, s1 = 64 + lane;
    const uint32_t n0 = nib(bw, s0), n1 = s1 < 90 ? nib(bw, s1) : 0u, p0 = nib(bw2, s0), p1 = s1 < 90 ? nib(bw2, s1) : 0u;
    const unsigned long long m0 = __ballot(n0 != 0), m1 = __ballot(n1 != 0), below = (1ull << lane) - 1ull;
    const int c0 = __popcll(m0), cnt = c0 + __popcll(m1);
    if (n0) rows[wid][__popcll(m0 & below)] = s0 * 14 + (int)n0 - 1;
    if (n1) rows[wid][c0 + __popcll(m1 & below)] = s1 * 14 + (int)n1 - 1;
    const unsigned long long d0 = __ballot(p0 != n0), d1 = __ballot(p1 != n1);
    const int nd = __popcll(d0) + __popcll(d1);
    if (p0 != n0) { const int k = __popcll(d0 & below); dpair[wid][2 * k] = n0 ? s0 * 14 + (int)n0 - 1 : kZeroRow; dpair[wid][2 * k + 1] = p0 ? s0 * 14 + (int)p0 - 1 : kZeroRow; }
    if (p1 != n1) { const int k = __popcll(d0) + __popcll(d1 & below); dpair[wid][2 * k] = n1 ? s1 * 14 + (int)n1 - 1 : kZeroRow; dpair[wid][2 * k + 1] = p1 ? s1 * 14 + (int)p1 - 1 : kZeroRow; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int col = lane * 4;
    float4 acc = *reinterpret_cast<const float4*>(b0 + col);
    int i = 0;
    for (; i + 4 <= cnt; i += 4) {
        const float4 w0 = *reinterpret_cast<const float4*>(W0T + (long long)rows[wid][i] * H + col), w1 = *reinterpret_cast<const float4*>(W0T + (long long)rows[wid][i + 1] * H + col);
        const float4 w2 = *reinterpret_cast<const float4*>(W0T + (long long)rows[wid][i + 2] * H + col), w3 = *reinterpret_cast<const float4*>(W0T + (long long)rows[wid][i + 3] * H + col);
        acc.x = ((acc.x + w0.x) + w1.x) + w2.x + w3.x; acc.y = ((acc.y + w0.y) + w1.y) + w2.y + w3.y;
        acc.z = ((acc.z + w0.z) + w1.z) + w2.z + w3.z; acc.w = ((acc.w + w0.w) + w1.w) + w2.w + w3.w;
    }
    for (; i < cnt; ++i) { const float4 w = *reinterpret_cast<const float4*>(W0T + (long long)rows[wid][i] * H + col); acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w; }
    *reinterpret_cast<float4*>(a1s + (long long)b * H + col) = make_float4(tanhf(acc.x), tanhf(acc.y), tanhf(acc.z), tanhf(acc.w));
    float4 a2 = acc;
    for (int k = 0; k < nd; ++k) {
        const float4 wo = *reinterpret_cast<const float4*>(W0T + (long long)dpair[wid][2 * k] * H + col), wi = *reinterpret_cast<const float4*>(W0T + (long long)dpair[wid][2 * k + 1] * H + col);
        a2.x -= wo.x; a2.y -= wo.y; a2.z -= wo.z; a2.w -= wo.w; a2.x += wi.x; a2.y += wi.y; a2.z += wi.z; a2.w += wi.w;
    }
    *reinterpret_cast<float4*>(a1n + (long long)b * H + col) = make_float4(tanhf(a2.x), tanhf(a2.y), tanhf(a2.z), tanhf(a2.w));
}

// ---- the fused kernel.  MODE: 0 = all, 1 = phase A only (gather + LDS + a_1(s) store), 2 = phase B only (operands from HBM copies)
// NW = waves per block (4: one per SIMD; 8: two per SIMD, each wave then owns one 32-column tile of the product)
template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void fused_kernel(const uint32_t* __restrict__ boards, const uint32_t* __restrict__ nboards, const float* __restrict__ W0T,
                                                    const float* __restrict__ b0, const float* __restrict__ W1, const float* __restrict__ b1,
                                                    float* __restrict__ a1s, const float* __restrict__ a1n_in, float* __restrict__ out_s, float* __restrict__ out_n, int n) {
    constexpr int NT = NW * 64, SPW = BS / NW, NQ = 2048 / NT, TPW = 8 / NW;      // threads, samples per wave, float4 of a W1 chunk per thread, tiles per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* A0 = smem;                       // [32][ALD]  a_1(s)
    float* A1 = A0 + BS * ALD;              // [32][ALD]  a_1(s')
    float* Bst = A1 + BS * ALD;             // [2][256][BLD]  W1 k-chunks of 32
    __shared__ int rows[NW][32];
    __shared__ int dpair[NW][16];
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int blk = blockIdx.x;
    float4 breg[NQ];
    if (MODE != 2) {
        const int col = lane * 4;
        const float4 bias0 = *reinterpret_cast<const float4*>(b0 + col);
        const unsigned long long below = (1ull << lane) - 1ull;
        const int s0 = lane, s1 = 64 + lane;
        long long b = (long long)blk * BS + wid * SPW;
        uint32_t w0a = boards[b * kWords + (s0 >> 3)], w1a = s1 < 90 ? boards[b * kWords + (s1 >> 3)] : 0u;
        uint32_t w0b = nboards[b * kWords + (s0 >> 3)], w1b = s1 < 90 ? nboards[b * kWords + (s1 >> 3)] : 0u;
        for (int i = 0; i < SPW; ++i) {
            const int sl = wid * SPW + i;
            b = (long long)blk * BS + sl;
            const uint32_t n0 = (w0a >> (4 * (s0 & 7))) & 15u, n1 = s1 < 90 ? (w1a >> (4 * (s1 & 7))) & 15u : 0u;
            const uint32_t p0 = (w0b >> (4 * (s0 & 7))) & 15u, p1 = s1 < 90 ? (w1b >> (4 * (s1 & 7))) & 15u : 0u;
            if (i + 1 < SPW) {                                 // next sample's words under this sample's row loads
                w0a = boards[(b + 1) * kWords + (s0 >> 3)]; w1a = s1 < 90 ? boards[(b + 1) * kWords + (s1 >> 3)] : 0u;
                w0b = nboards[(b + 1) * kWords + (s0 >> 3)]; w1b = s1 < 90 ? nboards[(b + 1) * kWords + (s1 >> 3)] : 0u;
            }
            const unsigned long long m0 = __ballot(n0 != 0), m1 = __ballot(n1 != 0);
            const int c0 = __popcll(m0);
            if (lane < 32) rows[wid][lane] = kZeroRow;
            if (lane < 16) dpair[wid][lane] = kZeroRow;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (n0) rows[wid][__popcll(m0 & below)] = s0 * 14 + (int)n0 - 1;
            if (n1) rows[wid][c0 + __popcll(m1 & below)] = s1 * 14 + (int)n1 - 1;
            const unsigned long long d0 = __ballot(p0 != n0), d1 = __ballot(p1 != n1);
            if (p0 != n0) { const int k = __popcll(d0 & below); if (k < 8) { if (n0) dpair[wid][2 * k] = s0 * 14 + (int)n0 - 1; if (p0) dpair[wid][2 * k + 1] = s0 * 14 + (int)p0 - 1; } }
            if (p1 != n1) { const int k = __popcll(d0) + __popcll(d1 & below); if (k < 8) { if (n1) dpair[wid][2 * k] = s1 * 14 + (int)n1 - 1; if (p1) dpair[wid][2 * k + 1] = s1 * 14 + (int)p1 - 1; } }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float4 v[32], u[16];
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = *reinterpret_cast<const float4*>(W0T + (long long)rows[wid][k] * H + col);
#pragma unroll
            for (int k = 0; k < 16; ++k) u[k] = *reinterpret_cast<const float4*>(W0T + (long long)dpair[wid][k] * H + col);
            float4 acc = bias0;
#pragma unroll
            for (int k = 0; k < 32; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
            float4 a2 = acc;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a2.x -= u[2 * k].x; a2.y -= u[2 * k].y; a2.z -= u[2 * k].z; a2.w -= u[2 * k].w;
                a2.x += u[2 * k + 1].x; a2.y += u[2 * k + 1].y; a2.z += u[2 * k + 1].z; a2.w += u[2 * k + 1].w;
            }
            const float4 t = make_float4(tanhf(acc.x), tanhf(acc.y), tanhf(acc.z), tanhf(acc.w));
            const float4 t2 = make_float4(tanhf(a2.x), tanhf(a2.y), tanhf(a2.z), tanhf(a2.w));
            *reinterpret_cast<float4*>(a1s + b * H + col) = t;
            *reinterpret_cast<float4*>(A0 + sl * ALD + col) = t;
            *reinterpret_cast<float4*>(A1 + sl * ALD + col) = t2;
        }
    } else {
        for (int idx = tid; idx < BS * (H / 4); idx += NT) {
            const int sl = idx / (H / 4), c4 = idx % (H / 4);
            *reinterpret_cast<float4*>(A0 + sl * ALD + c4 * 4) = *reinterpret_cast<const float4*>(a1s + ((long long)blk * BS + sl) * H + c4 * 4);
            *reinterpret_cast<float4*>(A1 + sl * ALD + c4 * 4) = *reinterpret_cast<const float4*>(a1n_in + ((long long)blk * BS + sl) * H + c4 * 4);
        }
    }
    if (MODE == 1) return;
    // ---- phase B
#pragma unroll
    for (int q = 0; q < NQ; ++q) { const int idx = q * NT + tid, nn = idx >> 3, kq = idx & 7; breg[q] = *reinterpret_cast<const float4*>(W1 + (long long)nn * H + kq * 4); }
    f32x16 acc[2][TPW];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][t][e] = 0.f;
#define XQ_STAGE_STORE(st_) \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q) { const int idx = q * NT + tid, nn = idx >> 3, kq = idx & 7; \
        *reinterpret_cast<float4*>(Bst + ((long long)(st_) * 256 + nn) * BLD + kq * 4) = breg[q]; }
    XQ_STAGE_STORE(0)
    __syncthreads();                                   // A images + stage 0 visible
    const int r = lane & 31, half = lane >> 5;
    const int col0 = wid * 32 * TPW;
    for (int c = 0; c < 8; ++c) {
        const int st = c & 1;
        if (c + 1 < 8) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) { const int idx = q * NT + tid, nn = idx >> 3, kq = idx & 7; breg[q] = *reinterpret_cast<const float4*>(W1 + (long long)nn * H + (c + 1) * 32 + kq * 4); }
        }
        const float* Bs = Bst + (long long)st * 256 * BLD;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kk = c * 32 + g * 8 + 4 * half;
            const float4 fa0 = *reinterpret_cast<const float4*>(A0 + r * ALD + kk);
            const float4 fa1 = *reinterpret_cast<const float4*>(A1 + r * ALD + kk);
            float4 fb[TPW];
#pragma unroll
            for (int t = 0; t < TPW; ++t) fb[t] = *reinterpret_cast<const float4*>(Bs + (col0 + t * 32 + r) * BLD + g * 8 + 4 * half);
#define XQ_STEP(c_) \
            _Pragma("unroll") for (int t = 0; t < TPW; ++t) { \
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0.c_, fb[t].c_, acc[0][t], 0, 0, 0); \
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1.c_, fb[t].c_, acc[1][t], 0, 0, 0); }
            XQ_STEP(x) XQ_STEP(y) XQ_STEP(z) XQ_STEP(w)
#undef XQ_STEP
        }
        if (c + 1 < 8) { XQ_STAGE_STORE(st ^ 1) __syncthreads(); }
    }
    // ---- epilogue: bias + tanh; C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int colo = col0 + t * 32 + r;
        const float bo = b1[colo];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            const long long o = ((long long)blk * BS + row) * H + colo;
            out_s[o] = tanhf(acc[0][t][e] + bo);
            out_n[o] = tanhf(acc[1][t][e] + bo);
        }
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<uint32_t> boards((size_t)n * kWords, 0), nboards;
    std::vector<std::vector<int>> sq(n, std::vector<int>(90, 0));
    for (int b = 0; b < n; ++b) {
        int placed = 0;
        for (int s = 0; s < 90 && placed < 32; ++s) if ((rng() % 100) < 29) { sq[b][s] = 1 + rng() % 14; ++placed; }
        for (int s = 0; s < 90; ++s) boards[(size_t)b * kWords + (s >> 3)] |= (uint32_t)sq[b][s] << (4 * (s & 7));
    }
    nboards = boards;
    std::vector<std::vector<int>> sq2 = sq;
    for (int b = 0; b < n; ++b) {                    // a move: one occupied square empties, another square takes its piece
        int from = -1; for (int t = 0; t < 200 && from < 0; ++t) { int s = rng() % 90; if (sq2[b][s]) from = s; }
        if (from < 0) continue;
        int to = rng() % 90; if (to == from) to = (to + 1) % 90;
        sq2[b][to] = sq2[b][from]; sq2[b][from] = 0;
        for (int w = 0; w < kWords; ++w) nboards[(size_t)b * kWords + w] = 0;
        for (int s = 0; s < 90; ++s) nboards[(size_t)b * kWords + (s >> 3)] |= (uint32_t)sq2[b][s] << (4 * (s & 7));
    }
    std::vector<float> W0T((size_t)(kZeroRow + 1) * H, 0.f), b0(H), W1((size_t)H * H), b1(H);
    for (size_t i = 0; i < (size_t)kZeroRow * H; ++i) W0T[i] = 0.05f * u(rng);
    for (auto& x : b0) x = 0.01f * u(rng); for (auto& x : W1) x = 0.05f * u(rng); for (auto& x : b1) x = 0.01f * u(rng);
    uint32_t *dB, *dN; float *dW0, *db0, *dW1, *db1, *dA1s, *dA1n, *dOs, *dOn, *dRs, *dRn;
    CK(hipMalloc(&dB, boards.size() * 4)); CK(hipMalloc(&dN, boards.size() * 4));
    CK(hipMalloc(&dW0, W0T.size() * 4)); CK(hipMalloc(&db0, H * 4)); CK(hipMalloc(&dW1, W1.size() * 4)); CK(hipMalloc(&db1, H * 4));
    const size_t act = (size_t)n * H * 4;
    CK(hipMalloc(&dA1s, act)); CK(hipMalloc(&dA1n, act)); CK(hipMalloc(&dOs, act)); CK(hipMalloc(&dOn, act)); CK(hipMalloc(&dRs, act)); CK(hipMalloc(&dRn, act));
    CK(hipMemcpy(dB, boards.data(), boards.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dN, nboards.data(), boards.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW0, W0T.data(), W0T.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db0, b0.data(), H * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW1, W1.data(), W1.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db1, b1.data(), H * 4, hipMemcpyHostToDevice));
    const size_t lds = (size_t)(2 * BS * ALD + 2 * 256 * BLD) * 4;
#define XQ_ATTR(M, W) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_kernel<M, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    XQ_ATTR(0, 4) XQ_ATTR(1, 4) XQ_ATTR(2, 4) XQ_ATTR(0, 8) XQ_ATTR(1, 8) XQ_ATTR(2, 8)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto fn) {
        fn(); fn(); float ms = 0;
        hipEventRecord(e0, 0); for (int i = 0; i < 20; ++i) fn(); hipEventRecord(e1, 0); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
        printf("  %-58s %7.2f us\n", name, ms * 50);
    };
    printf("n %d, H %d, LDS %zu B per block, grid %d\n", n, H, lds, n / BS);
    timeit("gather as the product runs it (wave per sample, 2 chains)", [&] { hipLaunchKernelGGL(gather_kernel, dim3(n / 4), dim3(256), 0, 0, dB, dN, dW0, db0, dRs, dRn, n); });
#define XQ_RUN(M, W, nm) timeit(nm, [&] { hipLaunchKernelGGL((fused_kernel<M, W>), dim3(n / BS), dim3(W * 64), lds, 0, dB, dN, dW0, db0, dW1, db1, (M == 2 ? dRs : dA1s), dRn, dOs, dOn, n); });
    XQ_RUN(1, 4, "fused, 4 waves: phase A only (gather into LDS + a_1(s))")
    XQ_RUN(2, 4, "fused, 4 waves: phase B only (operands from HBM, 2 chains)")
    XQ_RUN(0, 4, "fused, 4 waves: gather + hidden product, one kernel")
    XQ_RUN(1, 8, "fused, 8 waves: phase A only")
    XQ_RUN(2, 8, "fused, 8 waves: phase B only")
    XQ_RUN(0, 8, "fused, 8 waves: gather + hidden product, one kernel")
    CK(hipDeviceSynchronize());
    // ---- check 64 samples against fp64
    std::vector<float> os((size_t)n * H), on((size_t)n * H), a1((size_t)n * H), gs((size_t)n * H);
    CK(hipMemcpy(os.data(), dOs, act, hipMemcpyDeviceToHost)); CK(hipMemcpy(on.data(), dOn, act, hipMemcpyDeviceToHost));
    CK(hipMemcpy(a1.data(), dA1s, act, hipMemcpyDeviceToHost)); CK(hipMemcpy(gs.data(), dRs, act, hipMemcpyDeviceToHost));
    double worst = 0, worst1 = 0; size_t bitdiff = 0;
    for (size_t i = 0; i < a1.size(); ++i) if (a1[i] != gs[i]) ++bitdiff;
    for (int t = 0; t < 64; ++t) {
        const int b = (int)(((long long)t * 2654435761u) % n);
        for (int chain = 0; chain < 2; ++chain) {
            const auto& S = chain ? sq2[b] : sq[b];
            std::vector<double> z(H);
            for (int c = 0; c < H; ++c) z[c] = b0[c];
            for (int s = 0; s < 90; ++s) if (S[s]) for (int c = 0; c < H; ++c) z[c] += W0T[(size_t)(s * 14 + S[s] - 1) * H + c];
            std::vector<double> a(H); for (int c = 0; c < H; ++c) a[c] = std::tanh(z[c]);
            if (!chain) for (int c = 0; c < H; ++c) worst1 = std::max(worst1, std::fabs(a[c] - (double)a1[(size_t)b * H + c]));
            for (int j = 0; j < H; ++j) {
                double y = b1[j]; for (int k = 0; k < H; ++k) y += (double)W1[(size_t)j * H + k] * a[k];
                const double got = chain ? on[(size_t)b * H + j] : os[(size_t)b * H + j];
                worst = std::max(worst, std::fabs(std::tanh(y) - got));
            }
        }
    }
    printf("max |a_1(s) - fp64| %.2e, a_1(s) differs from the wave-per-sample gather in %zu of %zu elements, max |a_2 - fp64| over 64 samples x 2 chains %.2e\n",
           worst1, bitdiff, a1.size(), worst);
    return (worst < 1e-5 && worst1 < 1e-6) ? 0 : 1;
}
