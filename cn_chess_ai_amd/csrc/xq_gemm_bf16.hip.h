// xq_gemm_bf16.hip.h — the dense contractions of the bf16 Q-net (BASELINE configs[4]) on v_mfma_f32_32x32x16_bf16, gfx950 only.
//
// Reference: the same three products as xq_gemm.hip.h serves in fp32 — forwardKernel (dqn.cu:184/:275), hiddenLayerDeltaKernel
// (:297), updateWeightsBiasesKernel (:310) — batched; here with bf16 operands and fp32 accumulation.  The fp32 kernel's skeleton
// (register -> LDS restaging, two barriers per 64-deep k-tile, fragments rebuilt from staged floats) reached 0.26-0.33 of the bf16
// peak and carried 4.8 VALU instructions per MFMA (profiles/r02_g_config5_bf16_gemm_pmc.json).  This kernel has its own loop:
//   * 256 x 128 block tile, 8 waves (4 x 2), each wave 64 x 64 = 2 x 2 MFMA tiles; 64-deep k-tile (4 MFMA k-steps);
//   * both operands go global -> LDS by LDS-DMA (`global_load_lds_dwordx4`, 6 one-KB pieces per wave and k-tile, no VGPR round
//     trip), three k-tiles deep, ONE barrier per k-tile, counted vmcnt (two k-tiles stay in flight across the barrier);
//   * an operand is either k-contiguous in memory (X[row][k]: activations, weights [out][in]) or row-contiguous (X[k][row]: the
//     weight VIEW of the hidden delta, deltas / activations [batch][unit] in the weight-gradient product).  A k-contiguous tile
//     is staged as [rows][64 k] and read with one `ds_read_b128` per fragment; a row-contiguous tile is staged as it lies,
//     [64 k][rows], and read with two `ds_read_b64_tr_b16` per fragment — the LDS transposes, no transposed copy exists anywhere.
//     Both images are XOR-swizzled on the 16-byte chunk index (applied to the per-lane SOURCE address of the DMA, whose LDS side
//     is lane-linear, and again on the read) so that every fragment read is bank-conflict free;
//   * what a read returns IS the MFMA operand: no moves between LDS and the matrix pipe.
// Shapes: M % 256 == 0, N % 128 == 0, K % 64 == 0 (per split-K slab), 16-byte aligned operands; callers fall back to the tile
// kernel of xq_gemm.hip.h otherwise.
#pragma once

#include "xq_gemm.hip.h"

#include <type_traits>

namespace xq {

enum { BG_TANH = 0, BG_DELTA = 1, BG_STORE = 2 };

struct Bf16GemmArgs {
    int M, N, K;                      // C[m][n] = sum_k A(m, k) B(n, k)
    const uint16_t* A; long long lda; // L_KCONTIG: A[m * lda + k]; L_MCONTIG: A[k * lda + m]
    const uint16_t* B; long long ldb; // L_KCONTIG: B[n * ldb + k]; L_MCONTIG: B[k * ldb + n]
    int groups;                       // > 1: that many independent products of the same shape, blockIdx.z = group (no split-K then)
    const uint16_t* Ax[2]; const uint16_t* Bx[2];
    int k_chunk;                      // split-K: blockIdx.z = slab, k range [z * k_chunk, (z + 1) * k_chunk)
    long long slab_stride;            // BG_STORE: C of slab z = C + z * slab_stride
    const float* bias; const float* biasx[2];          // BG_TANH: [N]
    float* C; long long ldc; float* Cx[2];             // fp32 result (BG_TANH: optional copy of the ROUNDED value; BG_DELTA; BG_STORE)
    uint16_t* Cb; long long ldcb; uint16_t* Cbx[2];    // bf16 result (BG_TANH: the activation; BG_DELTA: the delta rounded for the next products)
    int cb_frag_mask;                 // BG_TANH: bit g set => group g's Cb is written in B-fragment order (scr_afrag_index, K = N)
    const uint16_t* Hb; long long ldh;                 // BG_DELTA: activation a of the layer the delta belongs to, bf16 [m][n]
};

constexpr int kBgBM = 256, kBgBN = 128, kBgBK = 64;
constexpr int kBgStageBytes = (kBgBM + kBgBN) * kBgBK * 2;      // 48 KB
constexpr int kBgStages = 3;
constexpr int kBgLdsBytes = kBgStages * kBgStageBytes;          // 144 KB

template <int N> __device__ __forceinline__ void bg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// six LDS-DMA pieces of one wave: four of the A tile, two of the B tile (consecutive LDS kilobytes each), per-lane 32-bit source
// offsets from two wave-uniform bases.  Inline asm: invisible to hipcc's vmcnt bookkeeping (the builtin makes it drain vmcnt(0) in
// front of every LDS read); M0 is compiler-reserved, saved and restored.
__device__ __forceinline__ void bg_dma6(const void* abase, const void* bbase, const unsigned (&oa)[4], const unsigned (&ob)[2],
                                        unsigned lds_a, unsigned lds_b) {
    unsigned keep;
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

template <int AL, int BL, int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_kernel(const Bf16GemmArgs g_in) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];       // [3 stages][A tile 32 KB | B tile 16 KB]
    Bf16GemmArgs g = g_in;
    const int z = (int)blockIdx.z;
    if (g_in.groups > 1 && z >= 1) {
        g.A = g_in.Ax[z - 1]; g.B = g_in.Bx[z - 1]; g.bias = g_in.biasx[z - 1]; g.C = g_in.Cx[z - 1]; g.Cb = g_in.Cbx[z - 1];
    }
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;                       // wave tile (wm, wn): rows 64 wm.., columns 64 wn..
    const int r5 = lane & 31, h = lane >> 5;
    const int m0 = (int)blockIdx.x * kBgBM, n0 = (int)blockIdx.y * kBgBN;
    const int kbeg = g_in.groups > 1 ? 0 : z * g.k_chunk;
    const int nkt = (g_in.groups > 1 ? g.K : min(g.k_chunk, g.K - kbeg)) / kBgBK;      // k-tiles of this block
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)lds);

    // ---- DMA source offsets of this lane (loop-invariant; the bases advance per k-tile) ----
    // A: pieces 4 wid + j (j < 4); B: pieces 2 wid + j (j < 2); a piece is 1 KB of the LDS image, lane l at byte 16 l of it
    unsigned oa[4], ob[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = wid * 4 + j;
        if (AL == L_KCONTIG) {                                   // 8 rows x 128 B per piece
            const int row = piece * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
            oa[j] = (unsigned)(((long long)row * g.lda) * 2 + c * 16);
        } else {                                                 // [64 k][256]: 2 k-rows x 512 B per piece
            const int k = piece * 2 + (lane >> 5), c = (lane & 31) ^ ((k & 3) << 2);
            oa[j] = (unsigned)(((long long)k * g.lda) * 2 + c * 16);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int piece = wid * 2 + j;
        if (BL == L_KCONTIG) {
            const int row = piece * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
            ob[j] = (unsigned)(((long long)row * g.ldb) * 2 + c * 16);
        } else {                                                 // [64 k][128]: 4 k-rows x 256 B per piece
            const int k = piece * 4 + (lane >> 4), c = (lane & 15) ^ ((k & 3) << 2);
            ob[j] = (unsigned)(((long long)k * g.ldb) * 2 + c * 16);
        }
    }
    const unsigned char* abase = reinterpret_cast<const unsigned char*>(g.A) +
                                 (AL == L_KCONTIG ? ((long long)m0 * g.lda + kbeg) * 2 : ((long long)kbeg * g.lda + m0) * 2);
    const unsigned char* bbase = reinterpret_cast<const unsigned char*>(g.B) +
                                 (BL == L_KCONTIG ? ((long long)n0 * g.ldb + kbeg) * 2 : ((long long)kbeg * g.ldb + n0) * 2);
    const long long astep = AL == L_KCONTIG ? kBgBK * 2 : (long long)kBgBK * g.lda * 2;
    const long long bstep = BL == L_KCONTIG ? kBgBK * 2 : (long long)kBgBK * g.ldb * 2;
    auto issue = [&](int kt) {
        const unsigned st = lds0 + (unsigned)((kt % kBgStages) * kBgStageBytes);
        bg_dma6(abase + astep * kt, bbase + bstep * kt, oa, ob, st + (unsigned)(wid * 4096), st + (unsigned)(kBgBM * kBgBK * 2 + wid * 2048));
    };
    // ---- fragment read offsets of this lane inside a stage ----
    unsigned fa0, fa1, fb0, fb1;
    {
        const int g4 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        if (AL == L_KCONTIG) {
            const unsigned x = (unsigned)((h ^ ((r5 >> 1) & 7)) << 4);
            fa0 = (unsigned)((wm * 64 + r5) * 128) + x; fa1 = fa0 + 32 * 128;
        } else {
            const int krow = 8 * (g4 >> 1) + q, cbase = wm * 8 + 2 * (g4 & 1) + (p >> 1);
            fa0 = (unsigned)(krow * 512 + ((cbase ^ (q << 2)) << 4) + (p & 1) * 8);
            fa1 = (unsigned)(krow * 512 + (((cbase + 4) ^ (q << 2)) << 4) + (p & 1) * 8);
        }
        if (BL == L_KCONTIG) {
            const unsigned x = (unsigned)((h ^ ((r5 >> 1) & 7)) << 4);
            fb0 = (unsigned)((wn * 64 + r5) * 128) + x; fb1 = fb0 + 32 * 128;
        } else {
            const int krow = 8 * (g4 >> 1) + q, cbase = wn * 8 + 2 * (g4 & 1) + (p >> 1);
            fb0 = (unsigned)(krow * 256 + ((cbase ^ (q << 2)) << 4) + (p & 1) * 8);
            fb1 = (unsigned)(krow * 256 + (((cbase + 4) ^ (q << 2)) << 4) + (p & 1) * 8);
        }
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    issue(0);
    if (nkt > 1) issue(1);
    if (nkt > 1) bg_wait_vm<6>(); else bg_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 2 < nkt;
        if (more) issue(kt + 2);
        const unsigned char* sa = lds + (kt % kBgStages) * kBgStageBytes;
        const unsigned char* sb = sa + kBgBM * kBgBK * 2;
        unsigned xa0 = fa0, xa1 = fa1, xb0 = fb0, xb1 = fb1;
        asm volatile("" : "+v"(xa0), "+v"(xa1), "+v"(xb0), "+v"(xb1));      // keep the per-step offsets out of loop-invariant registers
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = bg_frag<AL, kBgBM>(sa, xa0, xa1, i, s);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = bg_frag<BL, kBgBN>(sb, xb0, xb1, j, s);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        // k-tile kt + 1 has landed once everything older than the 6 pieces just issued is complete
        if (more) bg_wait_vm<6>(); else bg_wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every fragment read of this stage has returned
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue.  32x32 accumulator map: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) ----
    const int mw = m0 + wm * 64, nw = n0 + wn * 64;
    if (EPI == BG_STORE) {
        float* Cz = g.C + (long long)z * g.slab_stride;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    Cz[(long long)(mw + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * h) * g.ldc + nw + 32 * j + r5] = acc[i][j][q];
        return;
    }
    // BG_TANH / BG_DELTA: the value, its optional fp32 copy straight from the registers (a half-wave writes 128 contiguous bytes),
    // and the bf16 result through a per-wave LDS image [64 rows][64 + 8] so that it leaves as 16-byte row pieces (the ring is idle:
    // the last barrier of the loop lies behind every read of it)
    uint16_t* stage = reinterpret_cast<uint16_t*>(lds) + wid * (64 * 72);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
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
                if (EPI == BG_TANH) {
                    const __bf16 rb = (__bf16)tanh_fast(v + bias);
                    stage[row * 72 + 32 * j + r5] = __builtin_bit_cast(uint16_t, rb);
                    if (g.C) g.C[(long long)(mw + row) * g.ldc + n] = (float)rb;
                } else {
                    const float a = bf16_to_float(hv[q]);
                    v = v * (1.f - a * a);
                    g.C[(long long)(mw + row) * g.ldc + n] = v;
                    if (g.Cb) stage[row * 72 + 32 * j + r5] = bf16_bits(v);
                }
            }
        }
    if (!g.Cb) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool frag = EPI == BG_TANH && ((g_in.cb_frag_mask >> (g_in.groups > 1 ? z : 0)) & 1);
#pragma unroll
    for (int c = lane; c < 64 * 8; c += 64) {                    // 64 rows x 8 pieces of 8 bf16
        const int row = c >> 3, part = c & 7;
        const uint4 x = *reinterpret_cast<const uint4*>(stage + row * 72 + part * 8);
        const long long o = frag ? scr_afrag_index(mw + row, nw + part * 8, g.N) : (long long)(mw + row) * g.ldcb + nw + part * 8;
        *reinterpret_cast<uint4*>(g.Cb + o) = x;
    }
}

inline bool bf16_gemm_ok(const Bf16GemmArgs& g, int splits) {
    const int kc = splits > 1 ? g.k_chunk : g.K;
    return g.M % kBgBM == 0 && g.N % kBgBN == 0 && kc % kBgBK == 0 && g.K % kc == 0 && (g.lda % 8) == 0 && (g.ldb % 8) == 0 &&
           (((uintptr_t)g.A) % 16) == 0 && (((uintptr_t)g.B) % 16) == 0;
}

}  // namespace xq
