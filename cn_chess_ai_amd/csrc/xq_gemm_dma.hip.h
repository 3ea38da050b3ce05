// xq_gemm_dma.hip.h — the GEMM loop built on LDS-DMA for gfx950: the dense contractions of the bf16 Q-net (BASELINE configs[4]) on
// v_mfma_f32_32x32x16_bf16, and the fp32 forward products (k-contiguous operands) on v_mfma_f32_32x32x2_f32.
//
// Reference: the same three products as xq_gemm.hip.h serves in fp32 — forwardKernel (dqn.cu:184/:275), hiddenLayerDeltaKernel
// (:297), updateWeightsBiasesKernel (:310) — batched; here with bf16 operands and fp32 accumulation.  The fp32 kernel's skeleton
// (register -> LDS restaging, two barriers per 64-deep k-tile, fragments rebuilt from staged floats) reached 0.26-0.33 of the bf16
// peak and carried 4.8 VALU instructions per MFMA (profiles/r02_g_config5_bf16_gemm_pmc.json).  This kernel has its own loop:
//   * (128 TI) x (64 TJ) block tile, 8 waves (4 x 2), each wave TI x TJ MFMA tiles of 32 x 32 — 256 x 128 for the bf16 products,
//     128 x 128 / 128 x 64 where the fp32 forward would otherwise leave CUs idle; k-tile = 128 bytes of k (64 bf16 / 32 fp32);
//   * both operands go global -> LDS by LDS-DMA (`global_load_lds_dwordx4`, 6 one-KB pieces per wave and k-tile, no VGPR round
//     trip), three k-tiles deep, ONE barrier per k-tile, counted vmcnt (two k-tiles stay in flight across the barrier);
//   * an operand is either k-contiguous in memory (X[row][k]: activations, weights [out][in]) or row-contiguous (X[k][row]: the
//     weight VIEW of the hidden delta, deltas / activations [batch][unit] in the weight-gradient product).  A k-contiguous tile
//     is staged as [rows][64 k] and read with one `ds_read_b128` per fragment; a row-contiguous tile is staged as it lies,
//     [64 k][rows], and read with two `ds_read_b64_tr_b16` per fragment — the LDS transposes, no transposed copy exists anywhere.
//     Both images are XOR-swizzled on the 16-byte chunk index (applied to the per-lane SOURCE address of the DMA, whose LDS side
//     is lane-linear, and again on the read) so that every fragment read is bank-conflict free;
//   * what a read returns IS the MFMA operand: no moves between LDS and the matrix pipe.
//   * fp32 (DT_F32, k-contiguous operands only): the same images and DMA; one ds_read_b128 carries the four k a lane feeds to four
//     v_mfma_f32_32x32x2_f32 (k = 8 s + 4 h + t: the k order of the tile kernel in xq_gemm.hip.h, so results are bit-identical to it).
// Shapes: M, N whole block tiles, K whole k-tiles (per split-K slab), 16-byte aligned operands; callers fall back to the tile
// kernel of xq_gemm.hip.h otherwise.
#pragma once

#include "xq_gemm.hip.h"

#include <type_traits>

namespace xq {

enum { BG_TANH = 0, BG_DELTA = 1, BG_STORE = 2 };

struct Bf16GemmArgs {
    int M, N, K;                      // C[m][n] = sum_k A(m, k) B(n, k)
    const void* A; long long lda;     // L_KCONTIG: A[m * lda + k]; L_MCONTIG: A[k * lda + m]   (elements: bf16, or fp32 for DT_F32)
    const void* B; long long ldb;     // L_KCONTIG: B[n * ldb + k]; L_MCONTIG: B[k * ldb + n]
    int groups;                       // > 1: that many independent products of the same shape, blockIdx.z = group (no split-K then)
    const void* Ax[2]; const void* Bx[2];
    int k_chunk;                      // split-K: blockIdx.z = slab, k range [z * k_chunk, (z + 1) * k_chunk)
    long long slab_stride;            // BG_STORE: C of slab z = C + z * slab_stride
    const float* bias; const float* biasx[2];          // BG_TANH: [N]
    float* C; long long ldc; float* Cx[2];             // fp32 result (BG_TANH: optional copy of the ROUNDED value; BG_DELTA; BG_STORE)
    uint16_t* Cb; long long ldcb; uint16_t* Cbx[2];    // bf16 result (BG_TANH: the activation; BG_DELTA: the delta rounded for the next products)
    int cb_frag_mask;                 // BG_TANH: bit g set => group g's Cb is written in B-fragment order (scr_afrag_index, K = N)
    const uint16_t* Hb; long long ldh;                 // BG_DELTA: activation a of the layer the delta belongs to, bf16 [m][n]
};

constexpr int kBgBM = 256, kBgBN = 128, kBgBK = 64;             // the bf16 products' tile (TI = TJ = 2)
constexpr int kBgStages = 3;
constexpr int bg_stage_bytes(int TI, int TJ) { return (128 * TI + 64 * TJ) * 128; }
constexpr int bg_lds_bytes(int TI, int TJ) { return kBgStages * bg_stage_bytes(TI, TJ); }
constexpr int kBgLdsBytes = bg_lds_bytes(2, 2);                 // 144 KB

template <int N> __device__ __forceinline__ void bg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// the LDS-DMA pieces of one wave for one k-tile: NA of the A tile, NB of the B tile (consecutive LDS kilobytes each), per-lane 32-bit
// source offsets from two wave-uniform bases.  Inline asm: invisible to hipcc's vmcnt bookkeeping (the builtin makes it drain
// vmcnt(0) in front of every LDS read); M0 is compiler-reserved, saved and restored.
template <int NA, int NB>
__device__ __forceinline__ void bg_dma(const void* abase, const void* bbase, const unsigned (&oa)[NA], const unsigned (&ob)[NB],
                                       unsigned lds_a, unsigned lds_b) {
    static_assert((NA == 4 && NB == 2) || (NA == 2 && NB == 2) || (NA == 2 && NB == 1), "piece counts of the tiles in use");
    unsigned keep;
    if constexpr (NA == 4 && NB == 2) {
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %7\n\t"
            "s_mov_b32 m0, %10\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %7\n\t"
            "s_mov_b32 m0, %11\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %7\n\t"
            "s_mov_b32 m0, %12\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %7\n\t"
            "s_mov_b32 m0, %13\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %8\n\t"
            "s_mov_b32 m0, %14\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %8\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(oa[0]), "v"(oa[1]), "v"(oa[2]), "v"(oa[3]), "v"(ob[0]), "v"(ob[1]), "s"(abase), "s"(bbase), "s"(lds_a), "s"(lds_a + 1024u),
              "s"(lds_a + 2048u), "s"(lds_a + 3072u), "s"(lds_b), "s"(lds_b + 1024u)
            : "memory");
    } else if constexpr (NA == 2 && NB == 2) {
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
            "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
            "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %6\n\t"
            "s_mov_b32 m0, %10\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %6\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(oa[0]), "v"(oa[1]), "v"(ob[0]), "v"(ob[1]), "s"(abase), "s"(bbase), "s"(lds_a), "s"(lds_a + 1024u), "s"(lds_b), "s"(lds_b + 1024u)
            : "memory");
    } else {
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
            "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
            "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(oa[0]), "v"(oa[1]), "v"(ob[0]), "s"(abase), "s"(bbase), "s"(lds_a), "s"(lds_a + 1024u), "s"(lds_b)
            : "memory");
    }
}

typedef short bg_s4 __attribute__((ext_vector_type(4)));

// fragment of MFMA tile `i` (32 rows), k-step `s` (16 k) of an operand tile at LDS address `base`:
//   L_KCONTIG image [rows][64 k], 128-byte rows, chunk c of row r stored at c ^ ((r >> 1) & 7): one ds_read_b128;
//   L_MCONTIG image [64 k][ROWS], chunk c of k-row k stored at c ^ ((k & 3) << 2): two ds_read_b64_tr_b16 (4 k x 16 rows each).
// `o0` / `o1` are this lane's byte offsets for tile 0 / 1 of its wave (computed once, see the kernel).
template <int LAYOUT, int ROWS>
__device__ __forceinline__ bf16x8 bg_frag(const unsigned char* base, unsigned o0, unsigned o1, int i, int s) {
    if (LAYOUT == L_KCONTIG) {
        return *reinterpret_cast<const bf16x8*>(base + ((i ? o1 : o0) ^ (unsigned)(s << 5)));
    } else {
        const unsigned char* p = base + (i ? o1 : o0) + s * 16 * (ROWS * 2);
        const bg_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bg_s4*)(p));
        const bg_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bg_s4*)(p + 4 * (ROWS * 2)));
        typedef short s8 __attribute__((ext_vector_type(8)));
        const s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <int DT, int AL, int BL, int EPI, int TI, int TJ>
__global__ __launch_bounds__(512) void gemm_dma_kernel(const Bf16GemmArgs g_in) {
    static_assert(DT == DT_BF16 || (AL == L_KCONTIG && BL == L_KCONTIG && EPI == BG_TANH), "fp32: the forward product only");
    static_assert((AL == L_KCONTIG && BL == L_KCONTIG) || (TI == 2 && TJ == 2), "row-contiguous tiles are laid out for 256 x 128 blocks");
    constexpr int ES = DT == DT_F32 ? 4 : 2;                    // bytes per element
    constexpr int BM = 128 * TI, BN = 64 * TJ;                   // block tile
    constexpr int BKE = 128 / ES;                                // elements of k per k-tile of a k-contiguous operand (= 64 k-rows else)
    constexpr int STAGE = bg_stage_bytes(TI, TJ);
    constexpr int NA = 2 * TI, NB = TJ;                          // 1-KB DMA pieces per wave and k-tile
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];       // [3 stages][A tile | B tile]
    Bf16GemmArgs g = g_in;
    const int z = (int)blockIdx.z;
    if (g_in.groups > 1 && z >= 1) {
        g.A = g_in.Ax[z - 1]; g.B = g_in.Bx[z - 1]; g.bias = g_in.biasx[z - 1]; g.C = g_in.Cx[z - 1]; g.Cb = g_in.Cbx[z - 1];
    }
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;                       // wave tile (wm, wn): rows 32 TI wm.., columns 32 TJ wn..
    const int r5 = lane & 31, h = lane >> 5;
    const int m0 = (int)blockIdx.x * BM, n0 = (int)blockIdx.y * BN;
    const int kbeg = g_in.groups > 1 ? 0 : z * g.k_chunk;
    const int nkt = (g_in.groups > 1 ? g.K : min(g.k_chunk, g.K - kbeg)) / BKE;        // k-tiles of this block
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)lds);

    // ---- DMA source offsets of this lane (loop-invariant; the bases advance per k-tile) ----
    // A: pieces NA wid + j; B: pieces NB wid + j; a piece is 1 KB of the LDS image, lane l at byte 16 l of it
    unsigned oa[NA], ob[NB];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int piece = wid * NA + j;
        if (AL == L_KCONTIG) {                                   // 8 rows x 128 B per piece
            const int row = piece * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
            oa[j] = (unsigned)(((long long)row * g.lda) * ES + c * 16);
        } else {                                                 // [64 k][256]: 2 k-rows x 512 B per piece
            const int k = piece * 2 + (lane >> 5), c = (lane & 31) ^ ((k & 3) << 2);
            oa[j] = (unsigned)(((long long)k * g.lda) * ES + c * 16);
        }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int piece = wid * NB + j;
        if (BL == L_KCONTIG) {
            const int row = piece * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
            ob[j] = (unsigned)(((long long)row * g.ldb) * ES + c * 16);
        } else {                                                 // [64 k][128]: 4 k-rows x 256 B per piece
            const int k = piece * 4 + (lane >> 4), c = (lane & 15) ^ ((k & 3) << 2);
            ob[j] = (unsigned)(((long long)k * g.ldb) * ES + c * 16);
        }
    }
    const unsigned char* abase = reinterpret_cast<const unsigned char*>(g.A) +
                                 (AL == L_KCONTIG ? ((long long)m0 * g.lda + kbeg) * ES : ((long long)kbeg * g.lda + m0) * ES);
    const unsigned char* bbase = reinterpret_cast<const unsigned char*>(g.B) +
                                 (BL == L_KCONTIG ? ((long long)n0 * g.ldb + kbeg) * ES : ((long long)kbeg * g.ldb + n0) * ES);
    const long long astep = AL == L_KCONTIG ? 128 : (long long)64 * g.lda * ES;
    const long long bstep = BL == L_KCONTIG ? 128 : (long long)64 * g.ldb * ES;
    auto issue = [&](int kt) {
        const unsigned st = lds0 + (unsigned)((kt % kBgStages) * STAGE);
        bg_dma<NA, NB>(abase + astep * kt, bbase + bstep * kt, oa, ob, st + (unsigned)(wid * NA * 1024), st + (unsigned)(BM * 128 + wid * NB * 1024));
    };
    // ---- fragment read offsets of this lane inside a stage (tile 0 / tile 1 of the wave) ----
    unsigned fa0, fa1, fb0, fb1;
    {
        const int g4 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        if (AL == L_KCONTIG) {
            const unsigned x = (unsigned)((h ^ ((r5 >> 1) & 7)) << 4);
            fa0 = (unsigned)((wm * 32 * TI + r5) * 128) + x; fa1 = fa0 + 32 * 128;
        } else {
            const int krow = 8 * (g4 >> 1) + q, cbase = wm * 8 + 2 * (g4 & 1) + (p >> 1);
            fa0 = (unsigned)(krow * 512 + ((cbase ^ (q << 2)) << 4) + (p & 1) * 8);
            fa1 = (unsigned)(krow * 512 + (((cbase + 4) ^ (q << 2)) << 4) + (p & 1) * 8);
        }
        if (BL == L_KCONTIG) {
            const unsigned x = (unsigned)((h ^ ((r5 >> 1) & 7)) << 4);
            fb0 = (unsigned)((wn * 32 * TJ + r5) * 128) + x; fb1 = fb0 + 32 * 128;
        } else {
            const int krow = 8 * (g4 >> 1) + q, cbase = wn * 8 + 2 * (g4 & 1) + (p >> 1);
            fb0 = (unsigned)(krow * 256 + ((cbase ^ (q << 2)) << 4) + (p & 1) * 8);
            fb1 = (unsigned)(krow * 256 + (((cbase + 4) ^ (q << 2)) << 4) + (p & 1) * 8);
        }
    }
    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    constexpr int NP = NA + NB;                                  // VM operations of one issue()
    issue(0);
    if (nkt > 1) issue(1);
    if (nkt > 1) bg_wait_vm<NP>(); else bg_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 2 < nkt;
        if (more) issue(kt + 2);
        const unsigned char* sa = lds + (kt % kBgStages) * STAGE;
        const unsigned char* sb = sa + BM * 128;
        unsigned xa0 = fa0, xa1 = fa1, xb0 = fb0, xb1 = fb1;
        asm volatile("" : "+v"(xa0), "+v"(xa1), "+v"(xb0), "+v"(xb1));      // keep the per-step offsets out of loop-invariant registers
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (DT == DT_BF16) {
                bf16x8 af[TI], bfr[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) af[i] = bg_frag<AL, BM>(sa, xa0, xa1, i, s);
#pragma unroll
                for (int j = 0; j < TJ; ++j) bfr[j] = bg_frag<BL, BN>(sb, xb0, xb1, j, s);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            } else {
                // fp32: the 16-byte chunk 2 s + h of a row holds k = 8 s + 4 h + t, t = 0..3 — one v_mfma_f32_32x32x2_f32 per t
                float4 af[TI], bfr[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const float4*>(sa + ((i ? xa1 : xa0) ^ (unsigned)(s << 5)));
#pragma unroll
                for (int j = 0; j < TJ; ++j) bfr[j] = *reinterpret_cast<const float4*>(sb + ((j ? xb1 : xb0) ^ (unsigned)(s << 5)));
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < TI; ++i)
#pragma unroll
                        for (int j = 0; j < TJ; ++j) {
                            const float a = t == 0 ? af[i].x : t == 1 ? af[i].y : t == 2 ? af[i].z : af[i].w;
                            const float b = t == 0 ? bfr[j].x : t == 1 ? bfr[j].y : t == 2 ? bfr[j].z : bfr[j].w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
                        }
            }
        }
        // k-tile kt + 1 has landed once everything older than the pieces just issued is complete
        if (more) bg_wait_vm<NP>(); else bg_wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every fragment read of this stage has returned
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue.  32x32 accumulator map: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) ----
    const int mw = m0 + wm * 32 * TI, nw = n0 + wn * 32 * TJ;
    if (EPI == BG_STORE) {
        float* Cz = g.C + (long long)z * g.slab_stride;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    Cz[(long long)(mw + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * h) * g.ldc + nw + 32 * j + r5] = acc[i][j][q];
        return;
    }
    // BG_TANH / BG_DELTA: the value, its fp32 copy straight from the registers (a half-wave writes 128 contiguous bytes), and the
    // bf16 result through a per-wave LDS image [32 TI rows][32 TJ + 8] so that it leaves as 16-byte row pieces (the ring is idle: the
    // last barrier of the loop lies behind every read of it)
    constexpr int SW = 32 * TJ + 8;
    uint16_t* stage = reinterpret_cast<uint16_t*>(lds) + wid * (32 * TI * SW);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int n = nw + 32 * j + r5;
            float bias = 0.f;
            if (EPI == BG_TANH) bias = g.bias[n];
            uint16_t hv[16];
            if (EPI == BG_DELTA) {
#pragma unroll
                for (int q = 0; q < 16; ++q) hv[q] = g.Hb[(long long)(mw + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * h) * g.ldh + n];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = 32 * i + (q & 3) + 8 * (q >> 2) + 4 * h;
                float v = acc[i][j][q];
                if (EPI == BG_TANH && DT == DT_BF16) {
                    const __bf16 rb = (__bf16)tanh_fast(v + bias);
                    stage[row * SW + 32 * j + r5] = __builtin_bit_cast(uint16_t, rb);
                    if (g.C) g.C[(long long)(mw + row) * g.ldc + n] = (float)rb;
                } else if (EPI == BG_TANH) {                     // fp32 net: the exact activation, and (optionally) a bf16 COPY beside it
                    v = tanh_hidden(v + bias);           // (as the fp32 tile kernel's hidden epilogue: same bits)
                    g.C[(long long)(mw + row) * g.ldc + n] = v;
                    if (g.Cb) stage[row * SW + 32 * j + r5] = bf16_bits(v);
                } else {
                    const float a = bf16_to_float(hv[q]);
                    v = v * (1.f - a * a);
                    g.C[(long long)(mw + row) * g.ldc + n] = v;
                    if (g.Cb) stage[row * SW + 32 * j + r5] = bf16_bits(v);
                }
            }
        }
    if (!g.Cb) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool frag = EPI == BG_TANH && ((g_in.cb_frag_mask >> (g_in.groups > 1 ? z : 0)) & 1);
    constexpr int PPR = 4 * TJ;                                  // 16-byte pieces per row of the wave's sub-tile
#pragma unroll
    for (int c = lane; c < 32 * TI * PPR; c += 64) {
        const int row = c / PPR, part = c % PPR;
        const uint4 x = *reinterpret_cast<const uint4*>(stage + row * SW + part * 8);
        const long long o = frag ? scr_afrag_index(mw + row, nw + part * 8, g.N) : (long long)(mw + row) * g.ldcb + nw + part * 8;
        *reinterpret_cast<uint4*>(g.Cb + o) = x;
    }
}

// the bf16 products' instance (256 x 128 tiles)

}  // namespace xq
