// tools/gemm_bench.hip — micro-benchmark of xq_gemm.hip.h on the dominant product (8100 x B x 256, column-max epilogue).
// Diagnostic only (not part of the library).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_bench.hip -o gemm_bench
#include "../cn_chess_ai_amd/csrc/xq_gemm.hip.h"
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


// probes: how fast can the MFMA pipe run with (0) registers only, (1) + the LDS fragment reads of tile_mma, (2) + one
// barrier per k-step, (3) + LDS restaging from registers between two barriers, (4) + global loads of the next tile
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void mfma_probe_kernel(float* out, int iters, GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[g_tile_floats(128)];
    __shared__ __attribute__((aligned(16))) float Bs[g_tile_floats(128)];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < g_tile_floats(128); i += 256) { As[i] = 0.001f * (i % 97); Bs[i] = 0.002f * (i % 89); }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    if (MODE == 0) {
        float a = 0.5f + lane, b = 0.25f;
        for (int it = 0; it < iters * 16; ++it) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
        }
    } else if (MODE <= 2) {
        for (int it = 0; it < iters; ++it) {
            tile_mma<L_KCONTIG, L_KCONTIG, 128, 128, 2, 2>(As, Bs, wid >> 1, wid & 1, r, h, acc);
            if (MODE == 2) __syncthreads();
        }
    } else {
        float4 va[4], vb[4];
        const int tm = blockIdx.x % 64, tn = (blockIdx.x / 64) % 64;
        stage_load<L_KCONTIG, 128, true>(g, g.A, g.lda, 1, tm * 128, g.M, 0, g.K, va);
        stage_load<L_KCONTIG, 128, true>(g, g.B, g.ldb, 1, tn * 128, g.N, 0, g.K, vb);
        for (int it = 0; it < iters; ++it) {
            __syncthreads();
            stage_store<L_KCONTIG, 128>(As, va);
            stage_store<L_KCONTIG, 128>(Bs, vb);
            __syncthreads();
            if (MODE >= 4) {
                const int k0 = ((it + 1) & 7) * 32;
                stage_load<L_KCONTIG, 128, true>(g, g.A, g.lda, 1, tm * 128, g.M, k0, g.K, va);
                stage_load<L_KCONTIG, 128, true>(g, g.B, g.ldb, 1, ((tn + (it >> 3)) & 63) * 128, g.N, k0, g.K, vb);
            }
            tile_mma<L_KCONTIG, L_KCONTIG, 128, 128, 2, 2>(As, Bs, wid >> 1, wid & 1, r, h, acc);
            if (MODE == 5 && (it & 7) == 7) {
                epilogue_colmax<2, 2>(g, acc, tm * 128, ((tn + (it >> 3)) & 63) * 128, tm);
                for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
            }
        }
    }
    float s = 0; for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) s += acc[i][j][q];
    out[blockIdx.x * 256 + tid] = s;
}


// persistent-kernel variants: EPIV 0 = the library epilogue, 1 = (almost) no epilogue, 2 = bias preloaded into the
// accumulators + pure register max epilogue (no loads, no masks: rows >= M start at -inf)
template <int EPIV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void pers_variant_kernel(const GemmArgs g, int tiles_m, int total) {
    constexpr int TM = 2, TN = 2, BM = 128, BN = 128;
    __shared__ __attribute__((aligned(16))) float As[g_tile_floats(BM)];
    __shared__ __attribute__((aligned(16))) float Bs[g_tile_floats(BN)];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    int t, tend, stride;
    if (g.prio_split > 0) {
        const bool hi = (int)blockIdx.x >= g.prio_split;
        if (hi) __builtin_amdgcn_s_setprio(1);
        stride = hi ? (int)gridDim.x - g.prio_split : g.prio_split;
        t = hi ? (int)blockIdx.x - g.prio_split : g.prio_tiles + (int)blockIdx.x;
        tend = hi ? g.prio_tiles : total;
    } else { t = (int)blockIdx.x; tend = total; stride = (int)gridDim.x; }
    if (t >= tend) return;
    int tm = t % tiles_m, tn = t / tiles_m;
    float4 va[BM / 32], vb[BN / 32];
    stage_load<L_KCONTIG, BM, true>(g, g.A, g.lda, 1, tm * BM, g.M, 0, g.K, va);
    stage_load<L_KCONTIG, BN, true>(g, g.B, g.ldb, 1, tn * BN, g.N, 0, g.K, vb);
    for (;;) {
        f32x16 acc[TM][TN];
        if (EPIV == 4) {
            const float x = __int_as_float(0x3f000000 | (lane << 3));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[i][j][q] = x;
        } else if (EPIV == 3) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.5f;
        } else if (EPIV == 2) {
            const float NEG = -__builtin_inff();
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = tm * BM + wm * 64 + i * 32 + 4 * h;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 x = *reinterpret_cast<const float4*>(g.bias + mb + 8 * gq);
                    const float b0 = mb + 8 * gq + 0 < g.M ? x.x : NEG, b1 = mb + 8 * gq + 1 < g.M ? x.y : NEG;
                    const float b2 = mb + 8 * gq + 2 < g.M ? x.z : NEG, b3 = mb + 8 * gq + 3 < g.M ? x.w : NEG;
#pragma unroll
                    for (int j = 0; j < TN; ++j) { acc[i][j][4 * gq] = b0; acc[i][j][4 * gq + 1] = b1; acc[i][j][4 * gq + 2] = b2; acc[i][j][4 * gq + 3] = b3; }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        }
        const int tnext = t + stride;
        for (int k0 = 0; k0 < g.K; k0 += GBK) {
            __syncthreads();
            stage_store<L_KCONTIG, BM>(As, va);
            stage_store<L_KCONTIG, BN>(Bs, vb);
            __syncthreads();
            if (k0 + GBK < g.K) {
                stage_load<L_KCONTIG, BM, true>(g, g.A, g.lda, 1, tm * BM, g.M, k0 + GBK, g.K, va);
                stage_load<L_KCONTIG, BN, true>(g, g.B, g.ldb, 1, tn * BN, g.N, k0 + GBK, g.K, vb);
            } else if (tnext < tend) {
                stage_load<L_KCONTIG, BM, true>(g, g.A, g.lda, 1, (tnext % tiles_m) * BM, g.M, 0, g.K, va);
                stage_load<L_KCONTIG, BN, true>(g, g.B, g.ldb, 1, (tnext / tiles_m) * BN, g.N, 0, g.K, vb);
            }
            tile_mma<L_KCONTIG, L_KCONTIG, BM, BN, TM, TN>(As, Bs, wm, wn, r, h, acc);
        }
        if (EPIV == 6) {
            // like 5, but the column maximum goes straight into zmax[n] with an order-independent atomic max
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float c = -__builtin_inff();
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const float4 x = *reinterpret_cast<const float4*>(&Bs[wm * 64 + i * 32 + 4 * h + 8 * gq]);
                        c = fmaxf(c, acc[i][j][4 * gq] + x.x); c = fmaxf(c, acc[i][j][4 * gq + 1] + x.y);
                        c = fmaxf(c, acc[i][j][4 * gq + 2] + x.z); c = fmaxf(c, acc[i][j][4 * gq + 3] + x.w);
                    }
                c = fmaxf(c, __shfl_xor(c, 32, 64));
                const int n = tn * BN + wn * 64 + j * 32 + r;
                if (h == 0 && n < g.N) {
                    if (c >= 0.f) atomicMax(reinterpret_cast<int*>(g.partial + n), __float_as_int(c));
                    else atomicMin(reinterpret_cast<unsigned int*>(g.partial + n), __float_as_uint(c));
                }
            }
        } else if (EPIV == 5) {
            // zero-init + bias from LDS added in the epilogue (Bs reused as a stand-in for a staged bias array)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float c = -__builtin_inff();
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const float4 x = *reinterpret_cast<const float4*>(&Bs[wm * 64 + i * 32 + 4 * h + 8 * gq]);
                        c = fmaxf(c, acc[i][j][4 * gq] + x.x); c = fmaxf(c, acc[i][j][4 * gq + 1] + x.y);
                        c = fmaxf(c, acc[i][j][4 * gq + 2] + x.z); c = fmaxf(c, acc[i][j][4 * gq + 3] + x.w);
                    }
                c = fmaxf(c, __shfl_xor(c, 32, 64));
                const int n = tn * BN + wn * 64 + j * 32 + r;
                if (h == 0 && n < g.N) g.partial[((long long)tm * 2 + wm) * g.N + n] = c;
            }
        } else if (EPIV == 0) epilogue_colmax<TM, TN>(g, acc, tm * BM, tn * BN, tm);
        else if (EPIV == 1) {
            const float v = fmaxf(fmaxf(acc[0][0][0], acc[0][1][0]), fmaxf(acc[1][0][0], acc[1][1][0]));
            if (h == 0) g.partial[((long long)tm * 2 + wm) * g.N + tn * BN + wn * 64 + r] = v;
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float c = acc[0][j][0];
#pragma unroll
                for (int q = 1; q < 16; ++q) c = fmaxf(c, acc[0][j][q]);
#pragma unroll
                for (int q = 0; q < 16; ++q) c = fmaxf(c, acc[1][j][q]);
                c = fmaxf(c, __shfl_xor(c, 32, 64));
                const int n = tn * BN + wn * 64 + j * 32 + r;
                if (h == 0 && n < g.N) g.partial[((long long)tm * 2 + wm) * g.N + n] = c;
            }
        }
        if (tnext >= tend) break;
        t = tnext; tm = t % tiles_m; tn = t / tiles_m;
    }
}

int main(int argc, char** argv) {
    const int M = 8100, N = argc > 1 ? atoi(argv[1]) : 8192, K = argc > 2 ? atoi(argv[2]) : 256;
    float *A, *B, *C, *bias, *partial;
    CK(hipMalloc(&A, (size_t)(M + 128) * K * 4)); CK(hipMalloc(&B, (size_t)(N + 128) * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMalloc(&bias, (size_t)(M + 256) * 4)); CK(hipMalloc(&partial, (size_t)N * 2 * ((M + 63) / 64) * 4));
    std::vector<float> h((size_t)std::max(M, N) * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    CK(hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, (size_t)(M + 256) * 4));
    GemmArgs g; memset(&g, 0, sizeof g);
    g.M = M; g.N = N; g.K = K; g.A = A; g.lda = K; g.B = B; g.ldb = K; g.C = C; g.ldc = N; g.bias = bias; g.partial = partial;
    g.k_chunk = K; g.a_vec = 1; g.b_vec = 1; g.bias_padded = 1;
    const double fl = 2.0 * M * N * K;
    {
        float* o; CK(hipMalloc(&o, 1024 * 256 * 4));
        const int iters = 512;                          // k-steps per block; each = 64 MFMAs per wave
        for (int mode = 0; mode < 6; ++mode) for (int grid : {512}) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL((mfma_probe_kernel<0>), dim3(grid), dim3(256), 0, 0, o, iters, g);
                else if (mode == 1) hipLaunchKernelGGL((mfma_probe_kernel<1>), dim3(grid), dim3(256), 0, 0, o, iters, g);
                else if (mode == 2) hipLaunchKernelGGL((mfma_probe_kernel<2>), dim3(grid), dim3(256), 0, 0, o, iters, g);
                else if (mode == 3) hipLaunchKernelGGL((mfma_probe_kernel<3>), dim3(grid), dim3(256), 0, 0, o, iters, g);
                else if (mode == 4) hipLaunchKernelGGL((mfma_probe_kernel<4>), dim3(grid), dim3(256), 0, 0, o, iters, g);
                else hipLaunchKernelGGL((mfma_probe_kernel<5>), dim3(grid), dim3(256), 0, 0, o, iters, g);
            };
            launch(); hipEventRecord(a, 0); launch(); launch(); hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b); ms /= 2;
            const double f = (double)grid * 4 * iters * 64 * 4096.0;
            printf("probe mode %d (0 regs, 1 +frag reads, 2 +barrier, 3 +LDS restage & 2 barriers, 4 +global loads) grid %d: %7.1f TF/s\n", mode, grid, f / ms / 1e9);
        }

    }
    {
        float t = run<EPI_COLMAX, 2, 2>(g, 20);
        printf("colmax 128x128          : %8.1f us  %6.1f TF/s\n", t * 1e3, fl / t / 1e9);
    }
    for (int pct : {0, 50}) {        // share of the tiles given to the prioritised half (0 = no priority)
        const int grid = 512;
        const int tiles_m = (M + 127) / 128, total = tiles_m * ((N + 127) / 128);
        g.prio_split = pct ? grid / 2 : 0;
        g.prio_tiles = (int)((long long)total * pct / 100) / tiles_m * tiles_m;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float best = 1e9, sum = 0;
        for (int i = 0; i < 23; ++i) {
            hipEventRecord(a, 0);
            hipLaunchKernelGGL((gemm_colmax_persistent_kernel<2, 2>), dim3(grid), dim3(256), (size_t)tiles_m * 128 * 4, 0, g, tiles_m, total);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            if (i >= 3) { sum += ms; best = std::min(best, ms); }
        }
        printf("persistent prio share %2d%%: avg %8.1f us (%6.1f TF/s)  best %8.1f us\n", pct, sum / 20 * 1e3, fl / (sum / 20) / 1e9, best * 1e3);
    }
    for (int v = 5; v < 7; ++v) for (int pct : {0, 0}) {
        const int grid = 512;
        const int tiles_m = (M + 127) / 128, total = tiles_m * ((N + 127) / 128);
        g.prio_split = pct ? grid / 2 : 0;
        g.prio_tiles = (int)((long long)total * pct / 100) / tiles_m * tiles_m;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float sum = 0;
        for (int i = 0; i < 23; ++i) {
            hipEventRecord(a, 0);
            if (v == 0) hipLaunchKernelGGL((pers_variant_kernel<0>), dim3(grid), dim3(256), 0, 0, g, tiles_m, total);
            else if (v == 1) hipLaunchKernelGGL((pers_variant_kernel<1>), dim3(grid), dim3(256), 0, 0, g, tiles_m, total);
            else if (v == 2) hipLaunchKernelGGL((pers_variant_kernel<2>), dim3(grid), dim3(256), 0, 0, g, tiles_m, total);
            else if (v == 3) hipLaunchKernelGGL((pers_variant_kernel<3>), dim3(grid), dim3(256), 0, 0, g, tiles_m, total);
            else if (v == 4) hipLaunchKernelGGL((pers_variant_kernel<4>), dim3(grid), dim3(256), 0, 0, g, tiles_m, total);
            else if (v == 5) hipLaunchKernelGGL((pers_variant_kernel<5>), dim3(grid), dim3(256), 0, 0, g, tiles_m, total);
            else hipLaunchKernelGGL((pers_variant_kernel<6>), dim3(grid), dim3(256), 0, 0, g, tiles_m, total);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            if (i >= 3) sum += ms;
        }
        printf("variant %d (0 lib epilogue, 1 none, 2 bias-in-acc, 3 const-in-acc + max, 4 reg-in-acc + max, 5 zero + LDS bias add + max, 6 same with atomic max into zmax[n]) prio %2d%%: avg %8.1f us (%6.1f TF/s)\n", v, pct, sum / 20 * 1e3, fl / (sum / 20) / 1e9);
    }
    g.prio_split = 256; g.prio_tiles = (int)(4096LL * 60 / 100) / 64 * 64;
    g.prio_split = 256; g.prio_tiles = 2048;
    // check: persistent == plain
    {
        std::vector<float> p0((size_t)N * 2 * ((M + 127) / 128)), p1(p0.size());
        hipLaunchKernelGGL((gemm_f32_kernel<L_KCONTIG, L_KCONTIG, EPI_COLMAX, 2, 2>), dim3((M + 127) / 128, (N + 127) / 128, 1), dim3(256), 0, 0, g);
        CK(hipMemcpy(p0.data(), partial, p0.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(partial, 0, p0.size() * 4));
        hipLaunchKernelGGL((gemm_colmax_persistent_kernel<2, 2>), dim3(512), dim3(256), (size_t)((M + 127) / 128) * 128 * 4, 0, g, (M + 127) / 128, ((M + 127) / 128) * ((N + 127) / 128));   // with priority split
        CK(hipMemcpy(p1.data(), partial, p1.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < p0.size(); ++i) bad += p0[i] != p1[i];
        printf("persistent vs plain: %zu mismatches of %zu\n", bad, p0.size());
    }
    return 0;
}
