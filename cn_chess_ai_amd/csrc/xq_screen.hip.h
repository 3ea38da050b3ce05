// xq_screen.hip.h — the bf16 screening pass of max_a' Q(s',a') as a kernel of its own for gfx950.
//
// Reference: chessai.cpp:126 takes max over the 8100 outputs of DQN::getQValues(nextState) (dqn.cpp:65-68, dqn.cu:184-195).
// What this pass computes: for every sample b and every 32-row lane group g of the output layer, the largest and the
// second-largest of  z~_j(b) = b_j + sum_k bf16(W_jk) bf16(a_bk)  (fp32 accumulation on v_mfma_f32_32x32x16_bf16), the largest
// carrying its position in its low five mantissa bits — the same [G][n] partial arrays, group numbering and position code as
// gemm_colmax_persistent_kernel<2,2,DT_BF16,CM_TOP2> (xq_gemm.hip.h), which it replaces.
//
// Why a kernel of its own (the fp32 kernel's skeleton reached 0.26–0.28 of the bf16 peak: register -> LDS restaging of BOTH
// operands, two barriers per 64-deep k-tile, four `v_mov` per staged fragment, one epilogue per 16 MFMAs):
//   * the product is  [8100 outputs] x [n samples] x [K = 256 or 512]  — K is so short that a wave can keep its samples' WHOLE
//     k-range in registers: 32 NS samples x K bf16 = NS * K / 4 VGPRs (128 for NS * K = 512).  The activation operand is loaded
//     ONCE per block, straight from global memory in MFMA fragment order, and never touches LDS;
//   * only the weight operand streams: 64-row x 256-k "units" (32 KB) go global -> LDS by `global_load_lds_dwordx4` (no VGPR
//     round trip, no ds_write), three units deep, ONE barrier per unit, counted vmcnt so that two units stay in flight across
//     the barrier; the LDS image is the plain row-major unit with the 16-byte chunks of row r XOR-ed by (r & 15) — applied to
//     the per-lane SOURCE address (the LDS side of an LDS-DMA is lane-linear) and again on the read — so every ds_read_b128 of
//     a fragment (16 lanes, 16 different rows, same k) is conflict-free, and what it returns IS the MFMA operand: no moves;
//   * all 8 waves of a block read the same unit (different samples): 32 ds_read_b128 feed 32 NS MFMAs per wave;
//   * the bias enters as the C operand of a tile's first MFMA (one ds_read_b128 per 4 rows), so the fold is tag + top-2 only:
//     two values per step, v_max3 / v_med3 / v_max + two v_and_or = 2.5 VALU per value;
//   * grid = sample panels x row ranges, one 512-thread block per CU, blocks that share a row range on one XCD (same weight
//     stream from that L2).
// Error bound: the bias sits inside the accumulation chain, so the screen's bound uses the augmented vectors —
//   |z~_j - z_j| <= eps * sqrt(||a||^2 + 1) * max_j sqrt(||W_j||^2 + b_j^2)   (qmax_refine_kernel, screen_shadow_block).
#pragma once

#include "xq_gemm.hip.h"

#include <type_traits>

namespace xq {

struct ScreenArgs {
    const uint16_t* W;      // bf16 [rows_padded][K], rows >= NO zero, readable up to ranges * cpr * 64 rows
    const uint16_t* A;      // bf16 [panels * samples-per-block][K] (rows >= n: any finite values)
    const float* bias;      // fp32 [NO]
    float* P1; float* P2;   // [G][ldp], G = 2 * number of 64-row chunks; EVERY column < panels * samples-per-block is written
                            //   (P2: second values for SCR_TOP2, int row indices for SCR_ARG, unused for SCR_MAX)
    int ldp;                //   (ldp >= that: the stores are unconditional, so that the counted vmcnt below never depends on data)
    float* R;               // SCR_TOP2: [ranges][ldp] largest P1 of every sample over the groups of one row range (the refine kernel
                            //   looks at 16 of these per sample instead of 254 partials, and then only at the ranges that matter)
    float* na;              // SCR_TOP2: [ldp] sum of squares of the sample's bf16 activations (the bound's ||a||^2, from the registers)
    int NO, n, K;
    int nchunks;            // 64-row chunks that hold real rows: ceil(NO / 64)
    int cpr;                // chunks per row range
    int panels, ranges;     // grid = panels * ranges
    int a_frag;             // 1: A is stored in B-fragment order (scr_afrag_index) instead of row-major [sample][K]
    int xcd_rows;           // 1: the blocks of one XCD share ROW RANGES (weight stream from that L2); 0: they share SAMPLE PANELS;
                            // 2: an XCD takes half of the row ranges and a quarter of the panels (ranges % 2 == 0, panels % 4 == 0, 8 | grid)
    unsigned long long* dbg;  // DBG & 8 (probe only): [grid][8] stamps
};

constexpr int kScrUnitBytes = 64 * 512;      // 64 rows x 256 bf16
constexpr int kScrBufs = 3;

__device__ __forceinline__ long long ocol_of(int s0, int r5) { return (long long)s0 + r5; }
template <int N> __device__ __forceinline__ void scr_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Four 1-KB LDS-DMA pieces of one wave (consecutive LDS kilobytes from lds_dst, per-lane 32-bit source offsets off[0..3] from the
// wave-uniform base).  Inline asm on purpose: hipcc drains vmcnt(0) in front of the next ds_read when it sees the LDS-DMA builtin
// (it cannot prove that the read and the DMA touch different buffers), which serialises the whole pipeline; an asm statement is
// invisible to its bookkeeping and the kernel counts vmcnt itself.  M0 (the DMA's LDS base) is compiler-reserved: saved/restored.
__device__ __forceinline__ void scr_dma4(const void* base, unsigned o0, unsigned o1, unsigned o2, unsigned o3, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
        "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
        "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
        "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(base), "s"(lds_dst), "s"(lds_dst + 1024u), "s"(lds_dst + 2048u), "s"(lds_dst + 3072u)
        : "memory");
}

// KU = K / 256 (units per chunk), NS = 32-sample column tiles per wave; KU * NS == 2 keeps the register-resident operand at 128 VGPRs
// ---- pieces of the main loop (all indices that select registers are template arguments: static indexing, no scratch) ----
// k-steps [SBEG, SBEG + SCNT) of one unit: per step 2 ds_read_b128 (row tiles) and 2 NS MFMAs; the reads of step s + 1 are issued
// before the MFMAs of step s.  fr = this lane's fragment offset inside a unit (see the kernel), buf = the unit's LDS buffer.
template <int KU, int NS, int KUI, int SBEG, int SCNT, bool M16 = false>
__device__ __forceinline__ void scr_mma(const unsigned char* buf, unsigned fr, const bf16x8 (&bfrag)[NS][16 * KU], f32x16 (&acc)[2][NS]) {
    bf16x8 af[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) af[SBEG & 1][i] = *reinterpret_cast<const bf16x8*>(buf + (fr ^ (unsigned)(SBEG << 5)) + 16384 * i);
#pragma unroll
    for (int s = SBEG; s < SBEG + SCNT; ++s) {
        if (s + 1 < SBEG + SCNT) {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[(s + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(buf + (fr ^ (unsigned)((s + 1) << 5)) + 16384 * i);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                if (M16) {
                    // TIMING PROBE ONLY (tools/screen_probe.hip, DBG & 16; the values are meaningless): the same matrix-pipe cycles, operand
                    // registers and LDS reads as the line below, issued as two v_mfma_f32_16x16x32_bf16 — does the chip hold a higher clock
                    // on that shape inside this loop (MI355X_MICROARCH.md, DVFS give-back item 7)?
                    f32x4 c0 = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]}, c1 = {acc[i][j][4], acc[i][j][5], acc[i][j][6], acc[i][j][7]};
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s & 1][i], bfrag[j][KUI * 16 + s], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s & 1][i], bfrag[j][KUI * 16 + s], c1, 0, 0, 0);
                    acc[i][j][0] = c0[0]; acc[i][j][1] = c0[1]; acc[i][j][2] = c0[2]; acc[i][j][3] = c0[3];
                    acc[i][j][4] = c1[0]; acc[i][j][5] = c1[1]; acc[i][j][6] = c1[2]; acc[i][j][7] = c1[3];
                } else {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][i], bfrag[j][KUI * 16 + s], acc[i][j], 0, 0, 0);
                }
            }
    }
}
// accumulators of a chunk = its rows' biases (the first MFMA of every tile adds onto them): one ds_read_b128 per 4 rows
template <int NS>
__device__ __forceinline__ void scr_bias_init(const float* bt, f32x16 (&acc)[2][NS]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        f32x16 bv;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 x = *reinterpret_cast<const float4*>(bt + 32 * i + 8 * gq);
            bv[4 * gq] = x.x; bv[4 * gq + 1] = x.y; bv[4 * gq + 2] = x.z; bv[4 * gq + 3] = x.w;
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) acc[i][j] = bv;
    }
}
// What the fold keeps per (sample, 32-row lane group): SCR_TOP2 the largest value tagged with its position + the second largest
// (exact screening of an fp32 net); SCR_ARG the exact largest value and its row, first maximum (Double DQN on a bf16 net: P2 holds
// int row indices); SCR_MAX the exact largest value only (bf16 net, plain maximum).
enum { SCR_TOP2 = 0, SCR_ARG = 1, SCR_MAX = 2 };
// fold of a finished chunk over the 32 rows a lane holds of each column (code = 16 i + q; row = rowbase + 32 i + (q & 3) + 8 (q >> 2)).
// SCR_TOP2, two values per step: 2 v_and_or (tags), v_med3, v_max3, and one v_max3 per two steps for the running second value =
// 2.25 VALU per value.  SCR_ARG: rows ascending with a strict > (the lane's first maximum): compare + two selects per value.
template <int NS, int MODE, int DBG>
__device__ __forceinline__ void scr_fold(const f32x16 (&acc)[2][NS], float* __restrict__ P1, float* __restrict__ P2, long long o, int rowbase,
                                         float (&bmax)[NS]) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        float m1 = kColmaxPadBias, m2 = kColmaxPadBias;
        int mi = 0x7fffffff;
        if (DBG & 1) { m1 = acc[0][j][0] + acc[1][j][5]; m2 = acc[0][j][9]; }
        else
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                const float vx = acc[i][j][q], vy = acc[i][j][q + 1];      // (a bit_cast of the vector ELEMENT itself reads element 0)
                if (MODE == SCR_TOP2) {
                    const float x = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, vx) & ~31u) | (unsigned)(16 * i + q));
                    const float y = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, vy) & ~31u) | (unsigned)(16 * i + q + 1));
                    const float t = __builtin_amdgcn_fmed3f(m1, x, y);
                    m1 = fmaxf(fmaxf(m1, x), y);          // v_max3_f32
                    m2 = fmaxf(m2, t);
                } else if (MODE == SCR_ARG) {
                    const int rx = rowbase + 32 * i + (q & 3) + 8 * (q >> 2);
                    if (vx > m1) { m1 = vx; mi = rx; }
                    if (vy > m1) { m1 = vy; mi = rx + 1; }
                } else {
                    m1 = fmaxf(fmaxf(m1, vx), vy);
                }
            }
        P1[o + 32 * j] = m1;
        bmax[j] = fmaxf(bmax[j], m1);                 // running maximum of this lane's groups over the block's chunks
        if (MODE == SCR_TOP2) P2[o + 32 * j] = m2;
        if (MODE == SCR_ARG) reinterpret_cast<int*>(P2)[o + 32 * j] = mi;
    }
}
// unit u + 1 has landed once everything older than this iteration's own traffic is complete: the (up to) 4 LDS-DMA pieces of
// unit u + 2 and the 2 NS partial stores of a finished chunk are the only younger operations of this wave
template <int NSTORES>
__device__ __forceinline__ void scr_wait_landed(bool more, bool stored) {
    if (more) { if (stored) scr_wait_vm<4 + NSTORES>(); else scr_wait_vm<4>(); }
    else scr_wait_vm<0>();
}

// DBG (tools/screen_probe.hip only; 0 in the library): 1 = no fold (stores one accumulator element), 2 = no activation loads,
// 4 = no LDS-DMA in the loop, 8 = s_memtime / s_memrealtime stamps into a.dbg, 16 = timing probe of the 16x16x32 MFMA shape (scr_mma)
template <int KU, int NS, int MODE = SCR_TOP2, int DBG = 0>
__global__ __launch_bounds__(512) void screen_top2_kernel(const ScreenArgs a) {
    constexpr int NSTORES = (MODE == SCR_MAX ? 1 : 2) * NS;     // partial stores of one wave per finished chunk
    static_assert(KU * NS == 2, "register budget: NS * K / 4 = 128 VGPRs of activations");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];      // [kScrBufs][32 KB] units, then cpr * 64 bias floats
    constexpr int K = 256 * KU;
    constexpr int SB = 8 * 32 * NS;                  // samples per block
    constexpr int NBUF = kScrBufs;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);                  // provably wave-uniform (scalar registers)
    const int r5 = lane & 31, h = lane >> 5;
    // block -> (row range, sample panel); blocks b and b + 8 share an XCD (observed round-robin placement; speed only)
    const int nb = (int)gridDim.x;
    int lin = (int)blockIdx.x;
    if ((nb & 7) == 0) lin = ((int)blockIdx.x & 7) * (nb >> 3) + ((int)blockIdx.x >> 3);
    int range = a.xcd_rows ? lin / a.panels : lin % a.ranges;
    int panel = a.xcd_rows ? lin - range * a.panels : lin / a.ranges;
    if (a.xcd_rows == 2) {
        // XCD x = (half of the row ranges, quarter of the sample panels): every XCD streams HALF of W and a QUARTER of the activations from
        // HBM (16.4 + 8.4 MB at 8100 x 256 x 8192) instead of all of W and an eighth of the activations (33 + 4.2 MB)
        const int x = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3, r2 = a.ranges >> 1, p4 = a.panels >> 2;
        range = (x >> 2) * r2 + q % r2;
        panel = (x & 3) * p4 + q / r2;
    }
    const int c_first = range * a.cpr;
    const int nch = min(a.cpr, a.nchunks - c_first);             // chunks of this block (>= 1 by construction of the grid)
    const int U = nch * KU;                                      // units to stream
    unsigned long long st0 = 0, sr0 = 0;
    if (DBG & 8) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }
    float* bias_s = reinterpret_cast<float*>(lds + NBUF * kScrUnitBytes);
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)lds);   // LDS byte address of the ring

    // ---- weight stream: unit u = (chunk u / KU, k-half u % KU) -> buffer u % NBUF, four 1-KB LDS-DMA pieces per wave ----
    // piece p = 4 wid + j holds rows 2p, 2p + 1; the lane at linear LDS position (row, c) fetches 16-byte chunk c ^ (row & 15)
    unsigned srcoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wid * 4 + j) * 2 + (lane >> 5);
        srcoff[j] = (unsigned)(row * K * 2 + (((lane & 31) ^ (row & 15)) << 4));
    }
    auto issue_unit = [&](int u) {
        const int c = c_first + u / KU, ku = u % KU;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(a.W) + ((long long)c * 64 * K + ku * 256) * 2;
        scr_dma4(base, srcoff[0], srcoff[1], srcoff[2], srcoff[3], lds0 + (unsigned)((u % NBUF) * kScrUnitBytes + wid * 4096));
    };
    issue_unit(0);
    if (U > 1) issue_unit(1);
    // ---- bias of this block's rows -> LDS (rows >= NO: a finite very negative value; tagging -inf would make a NaN) ----
    for (int i = tid; i < nch * 64; i += 512) {
        const int row = c_first * 64 + i;
        bias_s[i] = row < a.NO ? a.bias[row] : kColmaxPadBias;
    }
    // ---- this wave's samples, whole k-range, in MFMA B-fragment order: lane (r5, h) holds k = 16 s + 8 h .. + 7 of sample r5.
    //      a_frag: the producer wrote the operand in exactly this order (scr_afrag_index) => every load is 1 KB contiguous ----
    const int s0 = panel * SB + wid * 32 * NS;
    bf16x8 bfrag[NS][K / 16];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const uint16_t* ap = a.a_frag ? a.A + ((long long)((s0 >> 5) + j) * (K / 16) * 2 + h) * 256 + r5 * 8
                                      : a.A + (long long)(s0 + 32 * j + r5) * K + 8 * h;
        const int sstride = a.a_frag ? 512 : 16;
#pragma unroll
        for (int s = 0; s < K / 16; ++s) {
            if (DBG & 2) { const short v = (short)(0x3c00 + lane + s); bfrag[j][s] = __builtin_bit_cast(bf16x8, (short __attribute__((ext_vector_type(8)))){v, v, v, v, v, v, v, v}); }
            else bfrag[j][s] = *reinterpret_cast<const bf16x8*>(ap + (long long)sstride * s);
        }
    }
    scr_wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the bias ds_writes (a raw s_barrier waits for no counter)
    __builtin_amdgcn_s_barrier();
    float bmax[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) bmax[j] = kColmaxPadBias;
    if (MODE == SCR_TOP2 && a.na) {
        // ||bf16(a)||^2 of every sample, from the register-resident operand: the (wave, column tile) pairs of a panel are shared out
        // over its row-range blocks (pair p belongs to the block with range == p mod ranges), each pair computed exactly once
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            if ((wid * NS + j) % a.ranges == range) {
                float ss = 0.f;
#pragma unroll
                for (int s = 0; s < K / 16; ++s) {
                    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bf2 x = {bfrag[j][s][2 * e], bfrag[j][s][2 * e + 1]};
                        ss = __builtin_amdgcn_fdot2_f32_bf16(x, x, ss, false);
                    }
                }
                ss += __shfl_xor(ss, 32, 64);                     // the other half-wave holds the other 8 of every 16 k
                if (h == 0) a.na[ocol_of(s0, r5) + 32 * j] = ss;
            }
        }
    }

    // fragment of k-step s, row tile i: chunk (2 s + h) ^ (r5 & 15) of row 32 i + r5.  2 s + h = 2 s ^ h, so the byte offset inside
    // the unit is  (r5 * 512 | ((h ^ (r5 & 15)) << 4)) ^ (s << 5)  + 16384 i : ONE register, one v_xor per k-step, immediates for i
    const unsigned frag0 = (unsigned)(r5 * 512) | (unsigned)((h ^ (r5 & 15)) << 4);
    unsigned long long st1 = 0, sr1 = 0;
    if (DBG & 8) { st1 = __builtin_amdgcn_s_memtime(); sr1 = __builtin_amdgcn_s_memrealtime(); }
    f32x16 acc[2][NS];
    const long long ocol = (long long)s0 + r5;
    int cl = 0;
    // one iteration = one unit; the k-half arrives as a type so that everything that selects registers stays a constant expression
    auto iteration = [&](auto kuc) {
        constexpr int ku = decltype(kuc)::value;
        const int u = cl * KU + ku;
        const bool more = u + 2 < U;
        if (more && !(DBG & 4)) issue_unit(u + 2);
        bool stored = false;
        unsigned fr = frag0;
        asm volatile("" : "+v"(fr));                             // opaque: keeps the 16 per-step offsets out of 16 loop-invariant registers
        const unsigned char* buf = lds + (u % NBUF) * kScrUnitBytes;
        if (ku == 0) scr_bias_init<NS>(bias_s + cl * 64 + 4 * h, acc);
        scr_mma<KU, NS, ku, 0, 16, (DBG & 16) != 0>(buf, fr, bfrag, acc);
        if (ku == KU - 1) {
            scr_fold<NS, MODE, DBG>(acc, a.P1, a.P2, (long long)(2 * (c_first + cl) + h) * a.ldp + ocol, 64 * (c_first + cl) + 4 * h, bmax);
            stored = true;
        }
        if (DBG & 4) scr_wait_vm<0>(); else scr_wait_landed<NSTORES>(more, stored);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every fragment read of this unit has returned: its buffer may be refilled
        __builtin_amdgcn_s_barrier();
    };
    for (; cl < nch; ++cl) {
        iteration(std::integral_constant<int, 0>{});
        if constexpr (KU > 1) iteration(std::integral_constant<int, KU - 1>{});
    }
    if (MODE == SCR_TOP2 && a.R) {
        // the block's maximum per sample: the two half-waves hold different groups of the same 32 samples
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const float v = fmaxf(bmax[j], __shfl_xor(bmax[j], 32, 64));
            if (h == 0) a.R[(long long)range * a.ldp + ocol + 32 * j] = v;
        }
    }
    if (DBG & 8) {
        const unsigned long long st2 = __builtin_amdgcn_s_memtime(), sr2 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            unsigned long long* o = a.dbg + (size_t)blockIdx.x * 8;
            o[0] = st1 - st0; o[1] = sr1 - sr0; o[2] = st2 - st1; o[3] = sr2 - sr1; o[4] = (unsigned long long)U; o[5] = sr0;
        }
    }
}

// grid geometry for (NO outputs, n samples, K): one block per CU when the problem allows
inline void screen_geometry(int NO, int n, int K, int ncu, ScreenArgs& a) {
    const int SB = K == 256 ? 512 : 256;
    a.NO = NO; a.n = n; a.K = K;
    a.nchunks = (NO + 63) / 64;
    a.panels = (n + SB - 1) / SB;
    int ranges = ncu / a.panels;
    if (ranges < 1) ranges = 1;
    if (ranges > a.nchunks) ranges = a.nchunks;
    a.cpr = (a.nchunks + ranges - 1) / ranges;
    a.ranges = (a.nchunks + a.cpr - 1) / a.cpr;
    a.xcd_rows = ((a.ranges & 1) == 0 && (a.panels & 3) == 0) ? 2 : 0;
    if (const char* e = getenv("XQ_SCREEN_XCD")) { if (e[0] == '0') a.xcd_rows = 0; else if (e[0] == '1') a.xcd_rows = 1; }     // A/B knob
}
inline size_t screen_lds_bytes(const ScreenArgs& a) { return (size_t)kScrBufs * kScrUnitBytes + (size_t)a.cpr * 64 * sizeof(float); }
inline int screen_padded_samples(int n, int K) { const int SB = K == 256 ? 512 : 256; return (n + SB - 1) / SB * SB; }

}  // namespace xq
