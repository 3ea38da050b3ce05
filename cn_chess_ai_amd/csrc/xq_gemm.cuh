// xq_gemm.cuh — fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), LDS-tiled for gfx950.
//
// One template serves every dense contraction of the Q-network (reference dqn.cu kernels forwardKernel :184/:275,
// hiddenLayerDeltaKernel :297, updateWeightsBiasesKernel :310 — there one thread per output neuron with a serial
// loop over inputs and batch 1; here batched, 128x128x32 block tiles, 4 waves of 64 lanes, each wave a 64x64
// sub-tile = 2x2 MFMA tiles with 16 accumulator registers each).
//
//   C[m][n] (+epilogue) = sum_k A(m,k) * B(k,n)
//
// Operand layouts (how the logical operand sits in HBM):
//   KCONTIG : X[row][k], ld = row stride      (activations [batch][features]; weights W[out][in] used as B(k,n)=W[n][k])
//   MCONTIG : X[k][row], ld = k stride        (weights used as B(k,n)=W[k][n]; deltas [batch][out] used as A(m=out,k=batch))
//   ONEHOT  : A only: row = feature f in [0,1260), k = sample; value = 1 if the packed board of sample k has piece
//             code f%14+1 on square f/14 — the one-hot of reference chessai.cpp:268-289, never materialised.
// LDS images: KCONTIG tiles are [128][36] floats read with ds_read_b128 (k = 8c+4h+t for MFMA t of chunk c, lane half
// h — both operands use the same k permutation, so any order is a valid contraction order); MCONTIG tiles are
// [32][132] floats read with ds_read_b32.  Both strides are conflict-free for their read instruction.
// fp32 MFMA runs at the fp32 vector rate (64 FLOP/clk/SIMD), 1/16 of bf16: the kernel is MFMA-bound long before LDS
// or L2 bandwidth matter, so staging goes through registers (global_load_dwordx4 -> ds_write_b128) with the next
// tile's loads in flight under the current tile's 64 MFMAs per wave.
#pragma once

#include "xq_common.h"

namespace xq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { L_KCONTIG = 0, L_MCONTIG = 1, L_ONEHOT = 2 };
enum { EPI_STORE = 0, EPI_BIAS_TANH = 1, EPI_ROWMAX = 2, EPI_DELTA = 3 };

constexpr int GBM = 128, GBN = 128, GBK = 32;
constexpr int G_LDK = GBK + 4;    // 36
constexpr int G_LDM = GBM + 4;    // 132
constexpr int G_TILE_FLOATS = GBM * G_LDK;   // 4608 >= 32*132 = 4224

struct GemmArgs {
    int M, N, K;
    const float* A; long long lda;
    const float* B; long long ldb;
    float* C; long long ldc;
    const float* bias;            // EPI_BIAS_TANH / EPI_ROWMAX: [N]
    const float* H; long long ldh;  // EPI_DELTA: activation a = tanh(z) of the layer the delta belongs to
    float* partial;               // EPI_ROWMAX: [gridDim.y*2][M] partial row maxima of (acc + bias)
    int k_chunk;                  // split-K: k range of blockIdx.z is [z*k_chunk, min(K,(z+1)*k_chunk))
    long long slab_stride;        // split-K: C of split z = C + z*slab_stride
    const uint32_t* boards;       // L_ONEHOT
    const int32_t* slots;         // L_ONEHOT: optional row gather (replay slots)
    int a_vec, b_vec;             // 16-byte vector loads allowed (base and ld aligned)
};

// ---- global -> register staging (4 x float4 per thread per operand) --------------------------------------------
template <int LAYOUT>
__device__ __forceinline__ void stage_load(const GemmArgs& g, const float* __restrict__ X, long long ld, int vec_ok,
                                           int rows0, int R, int k0, int kend, float4 (&v)[4]) {
    const int tid = (int)threadIdx.x;
    if (LAYOUT == L_KCONTIG) {
        const int kq = (tid & 7) * 4;
        const int k = k0 + kq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = rows0 + (tid >> 3) + 32 * j;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < R) {
                const float* p = X + (long long)row * ld + k;
                if (vec_ok && k + 3 < kend) {
                    x = *reinterpret_cast<const float4*>(p);
                } else {
                    if (k + 0 < kend) x.x = p[0];
                    if (k + 1 < kend) x.y = p[1];
                    if (k + 2 < kend) x.z = p[2];
                    if (k + 3 < kend) x.w = p[3];
                }
            }
            v[j] = x;
        }
    } else if (LAYOUT == L_MCONTIG) {
        const int mq = (tid & 31) * 4;
        const int row = rows0 + mq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + (tid >> 5) + 8 * j;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < kend) {
                const float* p = X + (long long)k * ld + row;
                if (vec_ok && row + 3 < R) {
                    x = *reinterpret_cast<const float4*>(p);
                } else {
                    if (row + 0 < R) x.x = p[0];
                    if (row + 1 < R) x.y = p[1];
                    if (row + 2 < R) x.z = p[2];
                    if (row + 3 < R) x.w = p[3];
                }
            }
            v[j] = x;
        }
    } else {  // L_ONEHOT: rows = features, k = samples
        const int mq = (tid & 31) * 4;
        const int f0 = rows0 + mq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + (tid >> 5) + 8 * j;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < kend && f0 < R) {
                const int srow = g.slots ? g.slots[k] : k;
                const uint32_t* bw = g.boards + (long long)srow * kBoardWords;
                float e[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = f0 + q;
                    float val = 0.f;
                    if (f < R) {
                        const int s = f / 14, pc = f - s * 14;
                        const uint32_t nib = (bw[s >> 3] >> (4 * (s & 7))) & 15u;
                        val = (nib == (uint32_t)(pc + 1)) ? 1.f : 0.f;
                    }
                    e[q] = val;
                }
                x = make_float4(e[0], e[1], e[2], e[3]);
            }
            v[j] = x;
        }
    }
}

template <int LAYOUT>
__device__ __forceinline__ void stage_store(float* __restrict__ Xs, const float4 (&v)[4]) {
    const int tid = (int)threadIdx.x;
    if (LAYOUT == L_KCONTIG) {
        const int kq = (tid & 7) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (tid >> 3) + 32 * j;
            *reinterpret_cast<float4*>(&Xs[row * G_LDK + kq]) = v[j];
        }
    } else {
        const int mq = (tid & 31) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kr = (tid >> 5) + 8 * j;
            *reinterpret_cast<float4*>(&Xs[kr * G_LDM + mq]) = v[j];
        }
    }
}

// fragments of chunk c (8 k values) for the wave's two 32-row sub-tiles
template <int LAYOUT>
__device__ __forceinline__ void frag_read(const float* __restrict__ Xs, int wbase, int c, int r, int h, float (&f)[2][4]) {
    if (LAYOUT == L_KCONTIG) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 x = *reinterpret_cast<const float4*>(&Xs[(wbase + i * 32 + r) * G_LDK + c * 8 + 4 * h]);
            f[i][0] = x.x; f[i][1] = x.y; f[i][2] = x.z; f[i][3] = x.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) f[i][t] = Xs[(c * 8 + 4 * h + t) * G_LDM + wbase + i * 32 + r];
    }
}

template <int AL, int BL, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[G_TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float Bs[G_TILE_FLOATS];
    constexpr int ASL = (AL == L_ONEHOT) ? L_MCONTIG : AL;

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = (int)blockIdx.x * GBM, n0 = (int)blockIdx.y * GBN;
    const int kbeg = (int)blockIdx.z * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    float4 va[4], vb[4];
    if (kbeg < kend) {
        stage_load<AL>(g, g.A, g.lda, g.a_vec, m0, g.M, kbeg, kend, va);
        stage_load<BL>(g, g.B, g.ldb, g.b_vec, n0, g.N, kbeg, kend, vb);
    }
    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
        __syncthreads();                         // previous tile fully consumed
        stage_store<ASL>(As, va);
        stage_store<BL>(Bs, vb);
        __syncthreads();
        if (k0 + GBK < kend) {                   // next tile's loads fly under this tile's MFMAs
            stage_load<AL>(g, g.A, g.lda, g.a_vec, m0, g.M, k0 + GBK, kend, va);
            stage_load<BL>(g, g.B, g.ldb, g.b_vec, n0, g.N, k0 + GBK, kend, vb);
        }
#pragma unroll
        for (int c = 0; c < GBK / 8; ++c) {
            float fa[2][4], fb[2][4];
            frag_read<ASL>(As, wm * 64, c, r, h, fa);
            frag_read<BL>(Bs, wn * 64, c, r, h, fb);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[j][t], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue.  32x32 accumulator map: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ---------------
    if (EPI == EPI_ROWMAX) {
        const float NEG = -__builtin_inff();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float v = NEG;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n0 + wn * 64 + j * 32 + r;
                    if (n < g.N) v = fmaxf(v, acc[i][j][q] + g.bias[n]);
                }
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));   // stays inside a 32-lane half
                const int m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (r == 0 && m < g.M) g.partial[((long long)blockIdx.y * 2 + wn) * g.M + m] = v;
            }
        }
        return;
    }
    float* Cz = g.C + (long long)blockIdx.z * g.slab_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + r;
            if (n >= g.N) continue;
            float bias = 0.f;
            if (EPI == EPI_BIAS_TANH) bias = g.bias[n];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (m >= g.M) continue;
                float v = acc[i][j][q];
                if (EPI == EPI_BIAS_TANH) v = tanhf(v + bias);
                if (EPI == EPI_DELTA) {
                    const float a = g.H[(long long)m * g.ldh + n];
                    v = v * (1.f - a * a);
                }
                Cz[(long long)m * g.ldc + n] = v;
            }
        }
}

}  // namespace xq
