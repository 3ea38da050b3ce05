// xq_gemm.hip.h — GEMM on the CDNA4 matrix cores, LDS-tiled for gfx950: fp32 (v_mfma_f32_32x32x2_f32, exact fp32) and,
// for the bf16 Q-net of BASELINE configs[4], bf16 with fp32 accumulation (v_mfma_f32_32x32x16_bf16) through the SAME
// staging code: a k-contiguous bf16 operand [rows][K] is addressed as a float matrix [rows][K/2], so one staged 32-float
// k-tile carries 64 bf16 of K and one 16-byte fragment read is exactly the 8 bf16 (k = 8h..8h+7) one lane feeds the MFMA.
//
// One template serves every dense contraction of the Q-network (reference dqn.cu kernels forwardKernel :184/:275,
// hiddenLayerDeltaKernel :297, updateWeightsBiasesKernel :310 — there one thread per output neuron with a serial
// loop over inputs and batch 1; here batched, (64*TM)x(64*TN)x32 block tiles — 128x128 for the large products,
// 64x64 where a 128-tile grid would leave CUs idle — 4 waves of 64 lanes, each wave TM x TN MFMA tiles of 32x32 with
// 16 accumulator registers each).
//
//   C[m][n] (+epilogue) = sum_k A(m,k) * B(k,n)
//
// Operand layouts (how the logical operand sits in HBM):
//   KCONTIG : X[row][k], ld = row stride      (activations [batch][features]; weights W[out][in] used as B(k,n)=W[n][k])
//   MCONTIG : X[k][row], ld = k stride        (weights used as B(k,n)=W[k][n]; deltas [batch][out] used as A(m=out,k=batch))
// LDS images: KCONTIG tiles are [rows][36] floats read with ds_read_b128 (k = 8c+4h+t for MFMA t of chunk c, lane half
// h — both operands use the same k permutation, so any order is a valid contraction order); MCONTIG tiles are
// [32][rows+4] floats read with ds_read_b32.  Both strides are conflict-free for their read instruction.
// fp32 MFMA runs at the fp32 vector rate (64 FLOP/clk/SIMD), 1/16 of bf16: the kernel is MFMA-bound long before LDS
// or L2 bandwidth matter, so staging goes through registers (global_load_dwordx4 -> ds_write_b128) with the next
// tile's loads in flight under the current tile's 64 MFMAs per wave.
#pragma once

#include "xq_common.h"

namespace xq {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { DT_F32 = 0, DT_BF16 = 1 };     // operand element type (accumulation is fp32 either way)
__device__ __forceinline__ float bf16_round(float v) { return (float)(__bf16)v; }                       // RNE, v_cvt_pk_bf16_f32
__device__ __forceinline__ uint16_t bf16_bits(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }
__device__ __forceinline__ float bf16_to_float(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }
// tanh for values that are rounded to bf16 right away (hidden activations of the bf16 Q-net): 1 - 2 / (1 + e^2x) on the hardware
// exp and reciprocal — absolute error ~1e-7, far below half a bf16 ulp except for |x| < 1e-4 where it cannot matter; tanhf costs
// ~40 instructions per value, which the bf16 MFMA no longer hides (fp32 nets: tanh_hidden below for the hidden products, tanhf
// for layer 0 and the output layer)
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }
// tanh of the fp32 hidden layers (forward epilogues of the hidden products; NeuralNetwork::forward, dqn.cu:184-195, calls libm's tanh
// in fp64).  libm's tanhf is ~40 VALU instructions per value and was 10-15 % of an 8192 x 512 x 512 product (tools/f32_fwd_probe.hip);
// this is 1 - 2 / (e^{2|x|} + 1) on the hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32), with the odd Taylor polynomial below
// |x| = 0.05 where that form cancels: absolute error <= 3e-7, relative <= 6e-6 over the whole range (tests/test_dqn_gpu.py::
// test_hidden_tanh_accuracy measures it through the GEMM) — two orders inside the 1e-4 the Q-values are held to.
// XQ_FAST_TANH=0 restores tanhf (A/B builds).
#ifndef XQ_FAST_TANH
#define XQ_FAST_TANH 1
#endif
__device__ __forceinline__ float tanh_hidden(float x) {
#if XQ_FAST_TANH
    const float ax = fabsf(x);
    const float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);          // e^{2|x|}; overflows to +inf => r = 1
    const float r = 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
    const float x2 = ax * ax;
    const float p = ax * __builtin_fmaf(x2, __builtin_fmaf(x2, 0.13333334f, -0.33333334f), 1.f);
    return __builtin_copysignf(ax < 0.05f ? p : r, x);
#else
    return tanhf(x);
#endif
}

// element index (in bf16 units) of activation (sample, k) in B-fragment order: 32-sample group, k-step of 16, lane half, then the
// 32 samples x 8 consecutive k of one MFMA operand wave-instruction (1 KB).  The screening pass (xq_screen.hip.h) loads its
// register-resident operand with fully coalesced 1-KB loads from this layout; the producers of the operand write through it.
__host__ __device__ __forceinline__ long long scr_afrag_index(int sample, int k, int K) {
    return ((long long)((sample >> 5) * (K >> 4) + (k >> 4)) * 2 + ((k >> 3) & 1)) * 256 + (sample & 31) * 8 + (k & 7);
}

enum { L_KCONTIG = 0, L_MCONTIG = 1 };
// EPI_COLMAX: per column n, max over the rows m of (acc + bias[m]) — the GEMM is launched "transposed" (rows = output
// neurons, columns = samples) so that the reduction runs over accumulator REGISTERS of one lane, not across lanes.
// EPI_HEAD (fp32, 64x64 tiles, whole tiles only): EPI_BIAS_TANH of the last hidden layer with the select head folded in — the block's
// fresh 64 x 64 activation tile goes straight on as the A operand of head_slabs[by][m][0..95] = a[m][64 by .. 64 by + 63] .
// W_head[0..95][same columns]^T, one k-slab of the select head per column tile; q_head_finish_kernel adds the slabs, the bias and the
// tanh.  The activations themselves are stored only when C != nullptr (the select chain has no other reader).
enum { EPI_STORE = 0, EPI_BIAS_TANH = 1, EPI_COLMAX = 2, EPI_DELTA = 3, EPI_HEAD = 4 };

constexpr int GBK = 32;
constexpr int G_LDK = GBK + 4;    // 36: row stride of a k-contiguous LDS tile
// block tile = (64*TM) x (64*TN), 4 waves as 2x2, each wave TM x TN MFMA tiles of 32x32
__host__ __device__ constexpr int g_tile_floats(int rows) { return rows * G_LDK > 32 * (rows + 4) ? rows * G_LDK : 32 * (rows + 4); }

struct GemmArgs {
    int M, N, K;
    const float* A; long long lda;
    const float* B; long long ldb;
    float* C; long long ldc;
    const float* bias;            // EPI_BIAS_TANH / EPI_ROWMAX: [N]
    const float* H; long long ldh;  // EPI_DELTA: activation a = tanh(z) of the layer the delta belongs to
    float* partial;               // EPI_COLMAX: [gridDim.x*2][N] partial column maxima of (acc + bias[m])
    int k_chunk;                  // split-K: k range of blockIdx.z is [z*k_chunk, min(K,(z+1)*k_chunk))
    long long slab_stride;        // split-K: C of split z = C + z*slab_stride
    int a_vec, b_vec;             // 16-byte vector loads allowed (base and ld aligned)
    // grouped launch (`grouped` = 2 or 3 independent products of the same shape in one grid, blockIdx.z = group; no split-K then):
    int grouped;
    const float* Ax[2]; const float* Bx[2]; float* Cx[2]; const float* biasx[2];     // groups 1 and 2
    // DT_BF16: A, B point at bf16 data, K / lda / ldb / k_chunk count PAIRS of bf16 (float units).  EPI_BIAS_TANH then rounds
    // its result to bf16 and writes it to Cb (bf16 bits, row stride ldcb) — and, when C != nullptr, the rounded value to C too.
    uint16_t* Cb; long long ldcb; uint16_t* Cbx[2];
    int cb_frag;                  // fp32 EPI_BIAS_TANH: the bf16 copy Cb is written in MFMA B-fragment order (scr_afrag_index, K = N)
    int* partial_idx;             // EPI_COLMAX: != nullptr => also the row index of each partial maximum (first maximum wins)
    float* partial2;              // persistent kernel, CM_TOP2: second-largest value of every 32-row lane group (see below)
    int bias_padded;              // EPI_COLMAX: bias[] is 16-byte aligned and readable up to the last tile's edge
    int prio_split;               // persistent kernel: blocks >= prio_split run at s_setprio 1 (0 = off) ...
    int prio_tiles;               // ... and own tiles [0, prio_tiles); the other blocks own [prio_tiles, total)
    // EPI_HEAD: first 128 rows of the output layer's weights (row stride head_ldw = N), the slabs [N / 64][M][head_ld]
    const float* head_W; long long head_ldw; float* head_slabs; long long head_slab_stride; int head_ld;
    int libm_tanh;                // EPI_BIAS_TANH: libm's tanhf instead of tanh_hidden — the OUTPUT layer (the audited Q-values) keeps it
    // EPI_DELTA, whole tiles only (M % 64 == 0, N % 64 == 0): besides C, the delta as three bf16 planes hi / mid / lo of 0.5 * delta (every
    // residual exact), TRANSPOSED [plane][column n][sample m], rows `split_ld` samples apart — the operand of the matrix-pipe layer-0
    // gradient (xq_l0grad.hip.h), written here instead of by a launch of its own that re-reads C
    uint16_t* split_planes; long long split_plane_stride; int split_ld;
};

// ---- global -> register staging (4 x float4 per thread per operand) --------------------------------------------
// FAST = the whole tile is interior and 16-byte aligned (decided once per block): unconditional global_load_dwordx4,
// no per-element bounds logic — the checked path compiles to scalar loads with waits and must stay off the hot tiles.
template <int LAYOUT, int ROWS, bool FAST, int J0 = 0, int J1 = ROWS / 32>
__device__ __forceinline__ void stage_load(const GemmArgs& g, const float* __restrict__ X, long long ld, int vec_ok,
                                           int rows0, int R, int k0, int kend, float4 (&v)[ROWS / 32]) {
    constexpr int NV = ROWS / 32;          // float4 per thread
    constexpr int TPR = ROWS / 4;          // threads per k-row of an m-contiguous tile
    constexpr int KPP = 256 / TPR;         // k-rows per pass
    const int tid = (int)threadIdx.x & 255;     // (a 512-thread block = two 4-wave groups working side by side: tools/f32_fwd_probe.hip)
    if (FAST && LAYOUT == L_KCONTIG) {
        const float* p = X + (long long)(rows0 + (tid >> 3)) * ld + k0 + (tid & 7) * 4;
#pragma unroll
        for (int j = J0; j < J1; ++j) {    // element-wise copy: a whole-float4 assignment keeps v[] as a stack object
            const float4 x = *reinterpret_cast<const float4*>(p + (long long)(32 * j) * ld);
            v[j].x = x.x; v[j].y = x.y; v[j].z = x.z; v[j].w = x.w;
        }
        return;
    }
    if (FAST && LAYOUT == L_MCONTIG) {
        const float* p = X + (long long)(k0 + (tid / TPR)) * ld + rows0 + (tid % TPR) * 4;
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            const float4 x = *reinterpret_cast<const float4*>(p + (long long)(KPP * j) * ld);
            v[j].x = x.x; v[j].y = x.y; v[j].z = x.z; v[j].w = x.w;
        }
        return;
    }
    if (LAYOUT == L_KCONTIG) {
        const int kq = (tid & 7) * 4;
        const int k = k0 + kq;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
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
        const int mq = (tid % TPR) * 4;
        const int row = rows0 + mq;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int k = k0 + (tid / TPR) + KPP * j;
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
    }
}

template <int LAYOUT, int ROWS>
__device__ __forceinline__ void stage_store(float* __restrict__ Xs, const float4 (&v)[ROWS / 32]) {
    constexpr int NV = ROWS / 32, TPR = ROWS / 4, KPP = 256 / TPR, LDM = ROWS + 4;
    const int tid = (int)threadIdx.x & 255;     // (a 512-thread block = two 4-wave groups working side by side: tools/f32_fwd_probe.hip)
    if (LAYOUT == L_KCONTIG) {
        const int kq = (tid & 7) * 4;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int row = (tid >> 3) + 32 * j;
            *reinterpret_cast<float4*>(&Xs[row * G_LDK + kq]) = v[j];
        }
    } else {
        const int mq = (tid % TPR) * 4;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int kr = (tid / TPR) + KPP * j;
            *reinterpret_cast<float4*>(&Xs[kr * LDM + mq]) = v[j];
        }
    }
}

// fragments of chunk c (8 k values) for the wave's two 32-row sub-tiles
template <int LAYOUT, int ROWS, int T>
__device__ __forceinline__ void frag_read(const float* __restrict__ Xs, int wbase, int c, int r, int h, float (&f)[T][4]) {
    constexpr int LDM = ROWS + 4;
    if (LAYOUT == L_KCONTIG) {
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const float4 x = *reinterpret_cast<const float4*>(&Xs[(wbase + i * 32 + r) * G_LDK + c * 8 + 4 * h]);
            f[i][0] = x.x; f[i][1] = x.y; f[i][2] = x.z; f[i][3] = x.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) f[i][t] = Xs[(c * 8 + 4 * h + t) * LDM + wbase + i * 32 + r];
    }
}

// MFMAs of one staged k-tile (32 deep = 4 chunks of 8).  Fragments of chunk c+1 are read from LDS while the MFMAs of
// chunk c issue (two named register sets, static indexing).
template <int TM, int TN, int DT>
__device__ __forceinline__ void chunk_mma(const float (&fa)[TM][4], const float (&fb)[TN][4], f32x16 (&acc)[TM][TN]) {
    if (DT == DT_BF16) {           // the 4 floats of a fragment ARE 8 consecutive bf16 of k: one 32x32x16 MFMA per tile pair
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const f32x4 a = {fa[i][0], fa[i][1], fa[i][2], fa[i][3]};
                const f32x4 b = {fb[j][0], fb[j][1], fb[j][2], fb[j][3]};
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                                    acc[i][j], 0, 0, 0);
            }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[j][t], acc[i][j], 0, 0, 0);
    }
}

template <int ASL, int BL, int BM, int BN, int TM, int TN, int DT = DT_F32>
__device__ __forceinline__ void tile_mma(const float* __restrict__ As, const float* __restrict__ Bs, int wm, int wn, int r, int h,
                                         f32x16 (&acc)[TM][TN]) {
    float fa0[TM][4], fb0[TN][4], fa1[TM][4], fb1[TN][4];
    frag_read<ASL, BM, TM>(As, wm * 32 * TM, 0, r, h, fa0);
    frag_read<BL, BN, TN>(Bs, wn * 32 * TN, 0, r, h, fb0);
#pragma unroll
    for (int c = 0; c < GBK / 8; c += 2) {
        frag_read<ASL, BM, TM>(As, wm * 32 * TM, c + 1, r, h, fa1);
        frag_read<BL, BN, TN>(Bs, wn * 32 * TN, c + 1, r, h, fb1);
        __builtin_amdgcn_sched_barrier(0);                        // keep the DS reads ahead of the whole MFMA block
        chunk_mma<TM, TN, DT>(fa0, fb0, acc);
        if (c + 2 < GBK / 8) {
            frag_read<ASL, BM, TM>(As, wm * 32 * TM, c + 2, r, h, fa0);
            frag_read<BL, BN, TN>(Bs, wn * 32 * TN, c + 2, r, h, fb0);
            __builtin_amdgcn_sched_barrier(0);
        }
        chunk_mma<TM, TN, DT>(fa1, fb1, acc);
    }
}

template <int AL, int BL, int TM, int TN, bool FAST, int DT = DT_F32>
__device__ __forceinline__ void gemm_mainloop(const GemmArgs& g, float* __restrict__ As, float* __restrict__ Bs, int m0, int n0,
                                              int kbeg, int kend, f32x16 (&acc)[TM][TN]) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    const int tid = (int)threadIdx.x & 255;     // (a 512-thread block = two 4-wave groups working side by side: tools/f32_fwd_probe.hip)
    const int lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    float4 va[BM / 32], vb[BN / 32];
    if (kbeg < kend) {
        stage_load<AL, BM, FAST>(g, g.A, g.lda, g.a_vec, m0, g.M, kbeg, kend, va);
        stage_load<BL, BN, FAST>(g, g.B, g.ldb, g.b_vec, n0, g.N, kbeg, kend, vb);
    }
    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
        __syncthreads();                         // previous tile fully consumed
        stage_store<AL, BM>(As, va);
        stage_store<BL, BN>(Bs, vb);
        __syncthreads();
        if (k0 + GBK < kend) {                   // next tile's loads fly under this tile's MFMAs
            stage_load<AL, BM, FAST>(g, g.A, g.lda, g.a_vec, m0, g.M, k0 + GBK, kend, va);
            stage_load<BL, BN, FAST>(g, g.B, g.ldb, g.b_vec, n0, g.N, k0 + GBK, kend, vb);
        }
        tile_mma<AL, BL, BM, BN, TM, TN, DT>(As, Bs, wm, wn, r, h, acc);
    }
}

// column-max epilogue: partial[(tile_m*2 + wm)][n] = max over the wave's rows of (acc + bias[m]); runs over the
// accumulator registers of one lane, then one exchange between the two half-waves.  With g.partial_idx the row index of
// the maximum travels along: rows are visited in ascending order per lane with a strict >, the half-wave exchange and the
// later reduction over the partials break ties towards the lower row => the FIRST maximum, like dqn.cpp:48.
template <int TM, int TN>
__device__ __forceinline__ void epilogue_colmax(const GemmArgs& g, const f32x16 (&acc)[TM][TN], int m0, int n0, int tile_m) {
    const int lane = (int)threadIdx.x & 63, wid = (int)threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    const float NEG = -__builtin_inff();
    const bool want_idx = g.partial_idx != nullptr;
    float cm[TN];
    int ci[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { cm[j] = NEG; ci[j] = 0x7fffffff; }
    // The 16 rows a lane holds of one 32x32 tile are four runs of 4 consecutive rows (reg 4g..4g+3 -> row 8g + 4h + 0..3):
    // their biases come as four 16-byte loads per tile, issued together (a branch per row compiles to 32 serialised
    // load + wait pairs; one wide batch keeps the register footprint of the epilogue small).  Rows >= M are masked by
    // select; when `bias_padded` the bias array is readable up to the tile edge (caller pads), else indices are clamped.
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mb = m0 + wm * 32 * TM + i * 32 + 4 * h;
        float bm[16];
        if (g.bias_padded) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 x = *reinterpret_cast<const float4*>(g.bias + mb + 8 * gq);
                bm[4 * gq] = x.x; bm[4 * gq + 1] = x.y; bm[4 * gq + 2] = x.z; bm[4 * gq + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) { const int m = mb + (q & 3) + 8 * (q >> 2); bm[q] = g.bias[m < g.M ? m : g.M - 1]; }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = mb + (q & 3) + 8 * (q >> 2);
            const bool ok = row < g.M;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float v = ok ? acc[i][j][q] + bm[q] : NEG;
                if (want_idx) { if (v > cm[j]) { cm[j] = v; ci[j] = row; } }
                else cm[j] = fmaxf(cm[j], v);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const float ov = __shfl_xor(cm[j], 32, 64);                   // the other 4-row groups live in the other half-wave
        const int oi = __shfl_xor(ci[j], 32, 64);
        float v = fmaxf(cm[j], ov);
        int idx = ci[j];
        if (want_idx && (ov > cm[j] || (ov == cm[j] && oi < ci[j]))) idx = oi;
        const int n = n0 + wn * 32 * TN + j * 32 + r;
        if (h == 0 && n < g.N) {
            g.partial[((long long)tile_m * 2 + wm) * g.N + n] = v;
            if (want_idx) g.partial_idx[((long long)tile_m * 2 + wm) * g.N + n] = idx;
        }
    }
}

// One block of the product: tile (bx, by), k-slab / group bz; As / Bs = g_tile_floats(BM) / g_tile_floats(BN) floats of LDS.  The block
// indices are arguments so that the same body serves gemm_f32_kernel and the fused launches of the TD step's tail (td_tail_kernel,
// xq_dqn.hip), where blocks of several kernels share one grid.
template <int AL, int BL, int EPI, int TM, int TN, int DT = DT_F32>
__device__ __forceinline__ void gemm_f32_block(const GemmArgs& g_in, int bx, int by, int bz, float* __restrict__ As, float* __restrict__ Bs) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    GemmArgs g = g_in;
    if (g_in.grouped && bz >= 1) {
        const int k = bz - 1;
        g.A = g_in.Ax[k]; g.B = g_in.Bx[k]; g.C = g_in.Cx[k]; g.bias = g_in.biasx[k]; g.Cb = g_in.Cbx[k];
    }

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = bx * BM, n0 = by * BN;
    const int kbeg = g.grouped ? 0 : bz * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // interior tiles (all rows/columns in range, k range a multiple of 32, 16-byte aligned operands) take the
    // branch-free loaders; the decision is block-uniform
    const bool interior = (m0 + BM <= g.M) && (n0 + BN <= g.N) && (((kend - kbeg) & (GBK - 1)) == 0) && g.a_vec && g.b_vec;
    if (interior) gemm_mainloop<AL, BL, TM, TN, true, DT>(g, As, Bs, m0, n0, kbeg, kend, acc);
    else gemm_mainloop<AL, BL, TM, TN, false, DT>(g, As, Bs, m0, n0, kbeg, kend, acc);

    // ---- epilogue.  32x32 accumulator map: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ---------------
    if (EPI == EPI_COLMAX) {
        epilogue_colmax<TM, TN>(g, acc, m0, n0, bx);
        return;
    }
    if (EPI == EPI_HEAD) {
        // the two 32-deep k-tiles of the head product = the activation columns of the waves wn = 0 / wn = 1; the head's weight tiles
        // (128 rows: 96 used) are requested before the tanh arithmetic, which hides their latency
        float4 vb0[4], vb1[4];
        stage_load<L_KCONTIG, 128, true>(g, g.head_W, g.head_ldw, 1, 0, 128, n0, n0 + 64, vb0);
        stage_load<L_KCONTIG, 128, true>(g, g.head_W, g.head_ldw, 1, 0, 128, n0 + 32, n0 + 64, vb1);
        const int n = n0 + wn * 32 + r;
        const int mb = m0 + wm * 32 + 4 * h;
        const float bias = g.bias[n];
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            v[q] = tanh_hidden(acc[0][0][q] + bias);
            if (g.C) g.C[(long long)(mb + (q & 3) + 8 * (q >> 2)) * g.ldc + n] = v[q];
        }
        f32x16 hacc[1][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) hacc[0][j][q] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            __syncthreads();                                       // the previous tile (main loop / kt = 0) fully consumed
            if (wn == kt) {
#pragma unroll
                for (int q = 0; q < 16; ++q) As[(wm * 32 + 4 * h + (q & 3) + 8 * (q >> 2)) * G_LDK + r] = v[q];
            }
            stage_store<L_KCONTIG, 128>(Bs, kt == 0 ? vb0 : vb1);
            __syncthreads();
            tile_mma<L_KCONTIG, L_KCONTIG, 64, 128, 1, 2>(As, Bs, wm, wn, r, h, hacc);
        }
        float* slab = g.head_slabs + (long long)by * g.head_slab_stride;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wn * 64 + j * 32 + r;
            if (col < g.head_ld) {
#pragma unroll
                for (int q = 0; q < 16; ++q) slab[(long long)(mb + (q & 3) + 8 * (q >> 2)) * g.head_ld + col] = hacc[0][j][q];
            }
        }
        return;
    }
    float* Cz = g.grouped ? g.C : g.C + (long long)bz * g.slab_stride;
    if (DT == DT_BF16 && EPI == EPI_BIAS_TANH) {
        // bf16 Q-net: a = bf16(tanh(acc + bias)); the bf16 bits feed the next layer's MFMA, the (optional) fp32 copy of the
        // SAME rounded value feeds the fp32 backward pass
        if (g.Cb != nullptr && (m0 + BM <= g.M) && (n0 + BN <= g.N) && (g.ldcb & 7) == 0) {
            // interior tile: the accumulator layout gives every lane ONE column (2-byte global stores, 64 of them per lane);
            // instead each wave transposes its (32 TM) x (32 TN) sub-tile through the now idle operand LDS and stores whole
            // 16-byte pieces of rows
            constexpr int WR = 32 * TM, WC = 32 * TN, S = WC + 8;                // per-wave image [WR][S] bf16, conflict-free stride
            __syncthreads();                                                     // every wave is done with As / Bs
            uint16_t* stage = reinterpret_cast<uint16_t*>(wid < 2 ? As : Bs) + (wid & 1) * WR * S;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * WC + j * 32 + r;
                    const float bias = g.bias[n];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int row = i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                        const __bf16 rb = (__bf16)tanh_fast(acc[i][j][q] + bias);
                        stage[row * S + j * 32 + r] = __builtin_bit_cast(uint16_t, rb);
                        if (Cz) Cz[(long long)(m0 + wm * WR + row) * g.ldc + n] = (float)rb;
                    }
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            constexpr int PPR = WC / 8;                                          // 16-byte pieces per row
#pragma unroll
            for (int c = lane; c < WR * PPR; c += 64) {
                const int row = c / PPR, part = c % PPR;
                const uint4 x = *reinterpret_cast<const uint4*>(stage + row * S + part * 8);
                *reinterpret_cast<uint4*>(g.Cb + (long long)(m0 + wm * WR + row) * g.ldcb + n0 + wn * WC + part * 8) = x;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * 32 * TN + j * 32 + r;
                if (n >= g.N) continue;
                const float bias = g.bias[n];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int m = m0 + wm * 32 * TM + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    if (m >= g.M) continue;
                    float v = g.Cb ? tanh_fast(acc[i][j][q] + bias) : tanhf(acc[i][j][q] + bias);
                    if (g.Cb) {                                  // hidden layer: rounded; the Q head (no Cb) stays fp32
                        const __bf16 rb = (__bf16)v;
                        g.Cb[(long long)m * g.ldcb + n] = __builtin_bit_cast(uint16_t, rb);
                        v = (float)rb;
                    }
                    if (Cz) Cz[(long long)m * g.ldc + n] = v;
                }
            }
        return;
    }
    if ((m0 + BM <= g.M) && (n0 + BN <= g.N)) {
        // tile fully inside the output: no per-element bounds logic, so the epilogue's loads (bias / activation for the
        // delta) are issued as one batch instead of a load + wait per row
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * 32 * TN + j * 32 + r;
                const int mb = m0 + wm * 32 * TM + i * 32 + 4 * h;
                float bias = 0.f;
                if (EPI == EPI_BIAS_TANH) bias = g.bias[n];
                float hv[16];
                if (EPI == EPI_DELTA) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) hv[q] = g.H[(long long)(mb + (q & 3) + 8 * (q >> 2)) * g.ldh + n];
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float v = acc[i][j][q];
                    if (EPI == EPI_BIAS_TANH) v = g.libm_tanh ? tanhf(v + bias) : tanh_hidden(v + bias);
                    if (EPI == EPI_DELTA) v = v * (1.f - hv[q] * hv[q]);
                    Cz[(long long)(mb + (q & 3) + 8 * (q >> 2)) * g.ldc + n] = v;
                    // fp32 net: a bf16 COPY of the exact activation beside it (operand of the screening pass)
                    if (EPI == EPI_BIAS_TANH && g.Cb) {
                        const int row = mb + (q & 3) + 8 * (q >> 2);
                        g.Cb[g.cb_frag ? scr_afrag_index(row, n, g.N) : (long long)row * g.ldcb + n] = bf16_bits(v);
                    }
                    if (EPI == EPI_DELTA) hv[q] = v;
                }
                if (EPI == EPI_DELTA && g.split_planes) {
                    // this lane holds column n of 16 of the wave's 32 samples: m32 + {0..3, 8..11, 16..19, 24..27} + 4 h; its partner
                    // lane ^ 32 holds the other 16.  Word w = samples (2 w, 2 w + 1) of the lane's 16; v_permlane32_swap on the word
                    // pairs (0, 2), (1, 3), (4, 6), (5, 7) leaves the h = 0 lanes with sample octets 0 and 2 of the 32 and the h = 1
                    // lanes with octets 1 and 3, each as four words in memory order: two 16-byte stores per plane and lane.
                    const int m32 = m0 + wm * 32 * TM + i * 32;
                    uint32_t w[3][8];
#pragma unroll
                    for (int q = 0; q < 16; q += 2) {
                        uint32_t pk[3] = {0u, 0u, 0u};
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const float x = 0.5f * hv[q + e];        // x 0.5: the one-hot operand is 2.0 (xq_l0grad.hip.h)
                            const uint16_t hi = bf16_bits(x);
                            const float r1 = x - bf16_to_float(hi);   // exact: the residual of a round-to-nearest to 8 bits
                            const uint16_t mid = bf16_bits(r1);
                            const float r2 = r1 - bf16_to_float(mid); // exact, <= 8 significant bits left
                            const uint16_t lo = bf16_bits(r2);
                            pk[0] |= (uint32_t)hi << (16 * e); pk[1] |= (uint32_t)mid << (16 * e); pk[2] |= (uint32_t)lo << (16 * e);
                        }
                        w[0][q >> 1] = pk[0]; w[1][q >> 1] = pk[1]; w[2][q >> 1] = pk[2];
                    }
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        uint4 o[2];
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            const auto s0 = __builtin_amdgcn_permlane32_swap(w[t][4 * half], w[t][4 * half + 2], false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(w[t][4 * half + 1], w[t][4 * half + 3], false, false);
                            o[half] = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                        }
                        uint16_t* dst = g.split_planes + t * g.split_plane_stride + (long long)n * g.split_ld + m32 + 8 * h;
                        *reinterpret_cast<uint4*>(dst) = o[0];           // octet h of the 32 samples
                        *reinterpret_cast<uint4*>(dst + 16) = o[1];      // octet 2 + h
                    }
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 32 * TN + j * 32 + r;
            if (n >= g.N) continue;
            float bias = 0.f;
            if (EPI == EPI_BIAS_TANH) bias = g.bias[n];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + wm * 32 * TM + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (m >= g.M) continue;
                float v = acc[i][j][q];
                if (EPI == EPI_BIAS_TANH) v = g.libm_tanh ? tanhf(v + bias) : tanh_hidden(v + bias);
                if (EPI == EPI_DELTA) {
                    const float a = g.H[(long long)m * g.ldh + n];
                    v = v * (1.f - a * a);
                }
                Cz[(long long)m * g.ldc + n] = v;
                if (EPI == EPI_BIAS_TANH && g.Cb) g.Cb[g.cb_frag ? scr_afrag_index(m, n, g.N) : (long long)m * g.ldcb + n] = bf16_bits(v);
            }
        }
}

template <int AL, int BL, int EPI, int TM, int TN, int DT = DT_F32>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void gemm_f32_kernel(const GemmArgs g_in) {
    __shared__ __attribute__((aligned(16))) float As[g_tile_floats(64 * TM)];
    __shared__ __attribute__((aligned(16))) float Bs[g_tile_floats(EPI == EPI_HEAD ? 128 : 64 * TN)];     // EPI_HEAD: 128 head rows x 32
    gemm_f32_block<AL, BL, EPI, TM, TN, DT>(g_in, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, As, Bs);
}

// Persistent column-max GEMM for the dominant product (rows = output neurons, columns = samples, both operands
// k-contiguous).  The plain kernel runs its 4096 tiles in lock-step rounds (every resident block loads its first tile,
// computes and drains at the same time), so prologue and epilogue are exposed once per round.  Here 2 blocks per CU walk
// the tile list (t, t+grid, ...: same row tile = same weight panel, resident in L2) and the register prefetch runs
// ACROSS tile boundaries: the first k-tile of the next output tile is in flight under the last MFMAs and the epilogue of
// the current one.  Requires K % 32 == 0, 16-byte aligned operands, and both operands readable up to the next multiple of
// 128 rows (the callers pad their allocations, finite contents); rows beyond M never win the max (their bias is -inf in
// LDS), columns beyond N are not stored.  Dynamic LDS: tiles_m * 128 floats (the bias vector).
// MODE = CM_ARG: also the row index of each partial maximum (g.partial_idx; first maximum wins, see epilogue_colmax) — Double DQN.
// MODE = CM_TOP2 (screening pass of the exact fp32 column maximum): no exchange between the half-waves; every lane
// keeps the LARGEST and the SECOND-LARGEST value of the 32 rows it holds of a column, the largest carrying its position in the low
// five mantissa bits (code = 16 i + q; row = tile_m*128 + wm*64 + 32 i + (q&3) + 8 (q>>2) + 4 h).  partial / partial2 are
// [tiles_m * 4][N], group = (tile_m*2 + wm)*2 + h.  Padding rows carry a bias of -3e38 (finite: tagging -inf would make a NaN).
// DT_BF16: bf16 operands (K counts pairs), same loop; the MFMA share of a k-step drops 16x, so the kernel is then bound by
// its LDS staging, not by the matrix pipe.
enum { CM_MAX = 0, CM_ARG = 1, CM_TOP2 = 2 };
constexpr float kColmaxPadBias = -3.0e38f;      // CM_TOP2: bias of the padding rows

template <int TM, int TN, int DT = DT_F32, int MODE = CM_MAX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_colmax_persistent_kernel(const GemmArgs g,
                                                                                                        int tiles_m, int total) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    __shared__ __attribute__((aligned(16))) float As[g_tile_floats(BM)];
    __shared__ __attribute__((aligned(16))) float Bs[g_tile_floats(BN)];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    // Two blocks share every SIMD.  With equal priority their waves alternate MFMAs, finish each k-step together and
    // then sit in their load/barrier phase together — the matrix pipe idles.  Static priority for the second half of
    // the grid (the blocks that land in the second slot of each CU) makes those run their MFMAs back to back while the
    // others fill exactly their gaps; the prioritised half therefore gets the larger share of the tile list
    // (g.prio_tiles of `total`), each half walking its own range with its own stride.
    int t, tend, stride;
    if (g.prio_split > 0) {
        const bool hi = (int)blockIdx.x >= g.prio_split;
        if (hi) __builtin_amdgcn_s_setprio(1);
        stride = hi ? (int)gridDim.x - g.prio_split : g.prio_split;
        t = hi ? (int)blockIdx.x - g.prio_split : g.prio_tiles + (int)blockIdx.x;
        tend = hi ? g.prio_tiles : total;
    } else {
        t = (int)blockIdx.x; tend = total; stride = (int)gridDim.x;
    }
    if (t >= tend) return;
    int tm = t % tiles_m, tn = t / tiles_m;
    float4 va[BM / 32], vb[BN / 32];
    // The whole bias vector lives in LDS for the lifetime of the block (rows >= M hold -inf so they never win the max):
    // the steady-state loop then has no global loads except the operand prefetch — a bias load inside the loop makes the
    // compiler's vmcnt bookkeeping conservative across the back edge and costs 5 % — and the epilogue needs no masks.
    extern __shared__ __attribute__((aligned(16))) float Bias_s[];          // [tiles_m * BM]
    stage_load<L_KCONTIG, BM, true>(g, g.A, g.lda, 1, tm * BM, g.M, 0, g.K, va);
    stage_load<L_KCONTIG, BN, true>(g, g.B, g.ldb, 1, tn * BN, g.N, 0, g.K, vb);
    if (g.bias_padded) {            // bias 16-byte aligned and M % 4 == 0: unconditional (clamped) float4 loads, 8 in flight per thread
        const int n4 = tiles_m * BM / 4, m4 = g.M / 4;
        const float NEGF = MODE == CM_TOP2 ? kColmaxPadBias : -__builtin_inff();
        for (int c0 = 0; c0 < n4; c0 += 256 * 8) {
            float4 bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) bv[u] = reinterpret_cast<const float4*>(g.bias)[min(c0 + u * 256 + tid, m4 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = c0 + u * 256 + tid;
                if (idx < n4) reinterpret_cast<float4*>(Bias_s)[idx] = idx < m4 ? bv[u] : make_float4(NEGF, NEGF, NEGF, NEGF);
            }
        }
    } else {
        for (int m = tid; m < tiles_m * BM; m += 256) Bias_s[m] = m < g.M ? g.bias[m] : (MODE == CM_TOP2 ? kColmaxPadBias : -__builtin_inff());
    }
    for (;;) {
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        const int tnext = t + stride;
        for (int k0 = 0; k0 < g.K; k0 += GBK) {
            __syncthreads();
            stage_store<L_KCONTIG, BM>(As, va);
            stage_store<L_KCONTIG, BN>(Bs, vb);
            __syncthreads();
            if (k0 + GBK < g.K) {
                stage_load<L_KCONTIG, BM, true>(g, g.A, g.lda, 1, tm * BM, g.M, k0 + GBK, g.K, va);
                stage_load<L_KCONTIG, BN, true>(g, g.B, g.ldb, 1, tn * BN, g.N, k0 + GBK, g.K, vb);
            } else if (tnext < tend) {           // first k-tile of the NEXT output tile
                stage_load<L_KCONTIG, BM, true>(g, g.A, g.lda, 1, (tnext % tiles_m) * BM, g.M, 0, g.K, va);
                stage_load<L_KCONTIG, BN, true>(g, g.B, g.ldb, 1, (tnext / tiles_m) * BN, g.N, 0, g.K, vb);
            }
            tile_mma<L_KCONTIG, L_KCONTIG, BM, BN, TM, TN, DT>(As, Bs, wm, wn, r, h, acc);
        }
        // epilogue: partial[(tile_m*2 + wm)][n] = max over the wave's rows of (acc + bias[m]).  32x32 accumulator map:
        // row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) — four runs of 4 consecutive rows per tile, one ds_read_b128 each; the
        // other 4-row groups live in the other half-wave.  (The first k-step's barrier orders the Bias_s fill before this.)
        const float* bt = Bias_s + tm * BM + wm * 32 * TM + 4 * h;
        if (MODE == CM_TOP2) {
            static_assert(TM == 2, "the 5-bit position code assumes 32 rows per lane");
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float m1 = kColmaxPadBias, m2 = kColmaxPadBias;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const float4 x = *reinterpret_cast<const float4*>(bt + i * 32 + 8 * gq);
                        const float bq[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t bits = __builtin_bit_cast(uint32_t, acc[i][j][4 * gq + e] + bq[e]);
                            const float v = __builtin_bit_cast(float, (bits & ~31u) | (uint32_t)(i * 16 + 4 * gq + e));
                            m2 = __builtin_amdgcn_fmed3f(m1, m2, v);        // second-largest so far
                            m1 = fmaxf(m1, v);
                        }
                    }
                const int n = tn * BN + wn * 32 * TN + j * 32 + r;
                if (n < g.N) {
                    const long long o = (((long long)tm * 2 + wm) * 2 + h) * g.N + n;
                    g.partial[o] = m1;
                    g.partial2[o] = m2;
                }
            }
        } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float c = -__builtin_inff();
            int ci = 0x7fffffff;
            const int row0 = tm * BM + wm * 32 * TM + 4 * h;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 x = *reinterpret_cast<const float4*>(bt + i * 32 + 8 * gq);
                    if (MODE == CM_ARG) {          // rows ascending within the lane + strict > : the lane's FIRST maximum
                        const float v0 = acc[i][j][4 * gq] + x.x, v1 = acc[i][j][4 * gq + 1] + x.y;
                        const float v2 = acc[i][j][4 * gq + 2] + x.z, v3 = acc[i][j][4 * gq + 3] + x.w;
                        const int rb = row0 + i * 32 + 8 * gq;
                        if (v0 > c) { c = v0; ci = rb; }
                        if (v1 > c) { c = v1; ci = rb + 1; }
                        if (v2 > c) { c = v2; ci = rb + 2; }
                        if (v3 > c) { c = v3; ci = rb + 3; }
                    } else {
                        c = fmaxf(c, acc[i][j][4 * gq] + x.x); c = fmaxf(c, acc[i][j][4 * gq + 1] + x.y);
                        c = fmaxf(c, acc[i][j][4 * gq + 2] + x.z); c = fmaxf(c, acc[i][j][4 * gq + 3] + x.w);
                    }
                }
            const float oc = __shfl_xor(c, 32, 64);
            const int n = tn * BN + wn * 32 * TN + j * 32 + r;
            if (MODE == CM_ARG) {
                const int oi = __shfl_xor(ci, 32, 64);
                const bool take = oc > c || (oc == c && oi < ci);
                if (h == 0 && n < g.N) {
                    g.partial[((long long)tm * 2 + wm) * g.N + n] = take ? oc : c;
                    g.partial_idx[((long long)tm * 2 + wm) * g.N + n] = take ? oi : ci;
                }
            } else {
                c = fmaxf(c, oc);
                if (h == 0 && n < g.N) g.partial[((long long)tm * 2 + wm) * g.N + n] = c;
            }
        }
        }
        if (tnext >= tend) break;
        t = tnext; tm = t % tiles_m; tn = t / tiles_m;
    }
}

// Persistent forward product of the hidden layers: C_c = tanh(A_c W_c^T + b_c) for the 1..3 grouped chains of a launch (both operands
// k-contiguous, whole (64 TM) x (64 TN) tiles, K % 32 == 0, 16-byte aligned).  gemm_f32_kernel gives every tile a block of its own:
// with K = 256..512 a block is 8-16 k-steps long, and its prologue (first operand tile: one HBM / L2 round trip with nothing to
// overlap), its epilogue and the launch's block rounds are paid per tile (PMC: matrix pipe 39 % busy at 8192 x 256 x 256 x 2).  Here
// WPE blocks per SIMD walk the tile list — t, t + grid, ...; consecutive tiles share the row panel of A — with the register prefetch
// running ACROSS tile boundaries, as gemm_colmax_persistent_kernel does: the next tile's first k-tile is in flight under the last
// MFMAs and the epilogue of the current one.  Epilogue as EPI_BIAS_TANH of gemm_f32_block (same expression, same bits), incl. the
// optional bf16 copy in row-major or MFMA B-fragment order (operand of the screening pass).
template <int TM, int TN, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void gemm_fwd_persistent_kernel(const GemmArgs g, int tiles_m,
                                                                                                            int tiles_n, int total) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    __shared__ __attribute__((aligned(16))) float As[g_tile_floats(BM)];
    __shared__ __attribute__((aligned(16))) float Bs[g_tile_floats(BN)];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    const int per_group = tiles_m * tiles_n;
    int t = (int)blockIdx.x;
    if (t >= total) return;
    const int stride = (int)gridDim.x;
    struct Tile { const float* A; const float* B; float* C; const float* bias; uint16_t* Cb; int m0, n0; };
    // workgroups go to the 8 XCDs round-robin (the grid is a multiple of 8, so a block's whole walk stays on one XCD): the tiles_n tiles
    // that share a row panel of A are given to ONE XCD — list position L = 8 q + x is tile (row panel 8 (q / tiles_n) + x, column
    // q % tiles_n) — instead of tiles_n different ones, each of which fetched the panel from HBM (PMC: 55 MB per launch against 38
    // compulsory at 8192 x 256 x 256 x 2, the activations read twice; four times at 512 wide)
    const bool xcd_walk = (total % (8 * tiles_n)) == 0 && (stride & 7) == 0;
    auto decode = [&](int tt) {
        if (xcd_walk) { const int x = tt & 7, q = tt >> 3; tt = ((q / tiles_n) * 8 + x) * tiles_n + q % tiles_n; }
        const int grp = tt / per_group, rem = tt - grp * per_group;
        Tile x;
        x.A = grp == 0 ? g.A : g.Ax[grp - 1]; x.B = grp == 0 ? g.B : g.Bx[grp - 1]; x.C = grp == 0 ? g.C : g.Cx[grp - 1];
        x.bias = grp == 0 ? g.bias : g.biasx[grp - 1]; x.Cb = grp == 0 ? g.Cb : g.Cbx[grp - 1];
        x.m0 = (rem / tiles_n) * BM; x.n0 = (rem % tiles_n) * BN;
        return x;
    };
    Tile cur = decode(t);
    float4 va[BM / 32], vb[BN / 32];
    stage_load<L_KCONTIG, BM, true>(g, cur.A, g.lda, 1, cur.m0, g.M, 0, g.K, va);
    stage_load<L_KCONTIG, BN, true>(g, cur.B, g.ldb, 1, cur.n0, g.N, 0, g.K, vb);
    for (;;) {
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        const int tnext = t + stride;
        Tile nxt = cur;
        if (tnext < total) nxt = decode(tnext);
        for (int k0 = 0; k0 < g.K; k0 += GBK) {
            __syncthreads();
            stage_store<L_KCONTIG, BM>(As, va);
            stage_store<L_KCONTIG, BN>(Bs, vb);
            __syncthreads();
            if (k0 + GBK < g.K) {
                stage_load<L_KCONTIG, BM, true>(g, cur.A, g.lda, 1, cur.m0, g.M, k0 + GBK, g.K, va);
                stage_load<L_KCONTIG, BN, true>(g, cur.B, g.ldb, 1, cur.n0, g.N, k0 + GBK, g.K, vb);
            } else if (tnext < total) {          // first k-tile of the NEXT output tile
                stage_load<L_KCONTIG, BM, true>(g, nxt.A, g.lda, 1, nxt.m0, g.M, 0, g.K, va);
                stage_load<L_KCONTIG, BN, true>(g, nxt.B, g.ldb, 1, nxt.n0, g.N, 0, g.K, vb);
            }
            tile_mma<L_KCONTIG, L_KCONTIG, BM, BN, TM, TN, DT_F32>(As, Bs, wm, wn, r, h, acc);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = cur.n0 + wn * 32 * TN + j * 32 + r;
                const int mb = cur.m0 + wm * 32 * TM + i * 32 + 4 * h;
                const float bias = cur.bias[n];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int row = mb + (q & 3) + 8 * (q >> 2);
                    const float v = tanh_hidden(acc[i][j][q] + bias);
                    if (cur.C) cur.C[(long long)row * g.ldc + n] = v;
                    if (cur.Cb) cur.Cb[g.cb_frag ? scr_afrag_index(row, n, g.N) : (long long)row * g.ldcb + n] = bf16_bits(v);
                }
            }
        if (tnext >= total) break;
        t = tnext; cur = nxt;
    }
}

}  // namespace xq
