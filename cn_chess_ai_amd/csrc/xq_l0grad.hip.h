// xq_l0grad.hip.h — layer-0 weight gradient on the bf16 matrix pipe, exact in fp32:  gW0^T[(sq, piece)][col] = sum over the samples
// that have `piece` on `sq` of delta0[sample][col]  (reference: updateWeightsBiasesKernel on layer 0, dqn.cu:310-319, fed by the
// one-hot of chessai.cpp:268-289 — there one thread per neuron and a serial loop over the 1260 inputs, batch 1).
//
// Why a matrix product after all.  The segmented-sum kernel (l0_grad_block, xq_dqn.hip) reads every 1-KB delta row once per occupied
// square of its sample — ~30 times, 250 MB through L2 per 8192-sample step — and was the longest kernel of the step (38-45 us).  As a
// dense product one-hot^T x delta0 every delta value is read once per GROUP of squares (15 groups of 6: 180 MB, 16-byte loads straight
// into MFMA operand registers), and the 18 GFLOP it costs in bf16 are 7 us of the 2.5 PFLOP/s pipe.  Two things make that exact:
//   * the one-hot operand is 0 / 1 — exact in bf16;
//   * delta0 = hi + mid + lo with three bf16 values per fp32 (8 + 8 + 8 significant bits; each residual is formed exactly in fp32), so
//     every product is exact and the MFMA accumulates in fp32: the result differs from a sequential fp32 sum only by summation order.
// Round 2 tried this shape and measured 87.7 us (profiles/r02_e_onehot_mfma_experiment_*): it expanded the one-hot tile with VALU compares,
// split delta0 inside the product kernel once per 64-row block, and had 184 blocks.  Here:
//   * delta_split_kernel writes the three bf16 planes TRANSPOSED ([plane][column][sample]) once: an MFMA operand fragment (8 consecutive
//     k = samples of one column) is then one 16-byte load per lane, 64 contiguous bytes per column and wave-instruction;
//   * the one-hot fragment of (square, 8 samples) is derived in registers from ONE word: the block transposes its samples' piece codes
//     into words of 8 nibbles per (square, sample octet); XOR with the lane's piece code replicated, two OR-folds and a mask leave one
//     bit per matching sample, and two instructions per register move it to bit 14 of its half-word (= bf16 2.0; delta0 is halved by
//     the split) — ~14 VALU per fragment instead of ~25 compares / selects (round 2) or 4 dependent LDS table reads (first version);
//   * v_mfma_f32_16x16x32_bf16 with M = 16 columns of delta0, N = 16 planes of ONE square (14 used), K = 32 samples: the accumulator
//     lane (plane n, columns 4g..4g+3) stores 16 contiguous bytes of row (sq*14 + n);
//   * a wave owns RC x RS = 2 column tiles x 6 squares (48 accumulator registers); the four waves of a block take four groups of 6
//     squares and the SAME 32 columns, whose delta operand tile (6 KB per k-step) goes global -> registers -> LDS once per block:
//     50 MB through L2 per step instead of 180-250 (the first version, every wave loading its own operands, ran at L2 speed:
//     33 us).  Grid 4 square groups x H/32 column blocks x n/1024 sample chunks = 256 blocks at 8192 x 256 — the chunk count (and
//     with it the 10.3 MB of partial sums the SGD kernel adds in fixed order) stays what it was.
#pragma once

#include "xq_gemm.hip.h"

#ifndef XQ_L0M_SCHED
#define XQ_L0M_SCHED 0
#endif

namespace xq {

constexpr int kL0mRS = 6;              // squares per wave tile
constexpr int kL0mRC = 2;              // 16-column tiles per wave
constexpr int kL0mSqB = 4 * kL0mRS;    // squares per block (4 waves): 24 = three board words; 4 groups cover squares 0..95 (90..95: padding)
constexpr int kL0mCols = 16 * kL0mRC;  // columns per block: 32 — the block's four waves share the SAME delta operand tile through LDS
constexpr int kL0mStageB = 3 * kL0mCols * 64;             // 6144 bytes per k-step stage: six 1-KB LDS-DMA pieces = (plane, 16 columns) x 32 k
constexpr int kL0mStages = 3;          // operand ring: two k-steps stay in flight across the barrier

// (the code words of one k-step beyond the chunk exist and are zero: the pipelined loop reads one k-step ahead without a branch)
__host__ __device__ constexpr size_t l0m_lds_bytes(int chunk) {
    return (size_t)kL0mStages * kL0mStageB + (size_t)kL0mSqB * (chunk / 8 + 4) * 4;
}
// bf16 elements the three planes need: [3][H][kpad] + slack for the operand prefetch that runs (harmlessly) past the last k-step
__host__ __device__ constexpr size_t l0m_plane_elems(int H, int kpad) { return (size_t)3 * H * kpad + 8 * 32 + 64; }

// delta0 [n][H] fp32 -> planes[t][col][k] bf16 (t = hi, mid, lo), k padded with zeros to kpad (a multiple of 64).
// 64 x 64 tiles through LDS: coalesced 16-byte reads along the columns, 32-byte writes along k.
__device__ __forceinline__ void delta_split_block(const float* __restrict__ d0, int n, int H, uint16_t* __restrict__ planes,
                                                  long long plane_stride, int kpad, int tile_k, int tile_c, float* __restrict__ tile /* [64][65] */) {
    const int tid = (int)threadIdx.x;
    const int k0 = tile_k * 64, c0 = tile_c * 64;
    {
        const int r = tid >> 4, c4 = (tid & 15) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + r + 16 * i;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < n) x = *reinterpret_cast<const float4*>(d0 + (long long)k * H + c0 + c4);
            float* p = tile + (r + 16 * i) * 65 + c4;
            p[0] = x.x; p[1] = x.y; p[2] = x.z; p[3] = x.w;
        }
    }
    __syncthreads();
    const int col = tid >> 2, kq = (tid & 3) * 16;
    uint32_t w[3][8];
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        uint32_t pk[3];
        // order inside an octet of samples: word i of the 16-byte fragment = samples (i, i + 4) — the order in which
        // l0_grad_mfma_block derives the one-hot operand from a word of 8 nibbles with two instructions per register
        const int oct = j & 8, i2 = (j & 7) >> 1;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float v = tile[(kq + oct + i2 + 4 * e) * 65 + col];
            const uint16_t hi = bf16_bits(0.5f * v);             // x 0.5: the one-hot operand is 2.0 (one bit per half-word), see below
            const float r1 = 0.5f * v - bf16_to_float(hi);       // exact: the residual of a round-to-nearest to 8 bits
            const uint16_t mid = bf16_bits(r1);
            const float r2 = r1 - bf16_to_float(mid);            // exact, <= 8 significant bits left
            const uint16_t lo = bf16_bits(r2);
            if (e == 0) { pk[0] = hi; pk[1] = mid; pk[2] = lo; }
            else { pk[0] |= (uint32_t)hi << 16; pk[1] |= (uint32_t)mid << 16; pk[2] |= (uint32_t)lo << 16; }
        }
        w[0][j >> 1] = pk[0]; w[1][j >> 1] = pk[1]; w[2][j >> 1] = pk[2];
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        uint4* o = reinterpret_cast<uint4*>(planes + t * plane_stride + (long long)(c0 + col) * kpad + k0 + kq);
        o[0] = make_uint4(w[t][0], w[t][1], w[t][2], w[t][3]);
        o[1] = make_uint4(w[t][4], w[t][5], w[t][6], w[t][7]);
    }
}

__global__ __launch_bounds__(256) void delta_split_kernel(const float* __restrict__ d0, int n, int H, uint16_t* __restrict__ planes,
                                                          long long plane_stride, int kpad) {
    __shared__ float tile[64 * 65];
    delta_split_block(d0, n, H, planes, plane_stride, kpad, (int)blockIdx.x, (int)blockIdx.y, tile);
}

// One block: square group `sqg` (24 squares = board words 3 sqg .. 3 sqg + 2), column block `cb` (32 columns), sample chunk `ch`.
// smem: l0m_lds_bytes(chunk).  partial: [chunks][1260][H].  Wave w owns squares 24 sqg + 6 w .. + 5 and all 32 columns.
template <int DBG = 0>
__device__ __forceinline__ void l0_grad_mfma_block(const uint32_t* __restrict__ gboards, const uint16_t* __restrict__ planes,
                                                   long long plane_stride, int kpad, int n, int H, int chunk, float* __restrict__ partial,
                                                   int sqg, int cb, int ch, uint32_t* __restrict__ smem) {
    const int octs = chunk / 8, ocs = octs + 4;                  // ocs: row stride of `codes` (one spare k-step of zeros)
    unsigned char* stage = reinterpret_cast<unsigned char*>(smem);                     // [3 stages][6 pieces][1 KB] (1-KB aligned)
    uint32_t* codes = smem + kL0mStages * kL0mStageB / 4;        // [24][ocs]: 8 nibbles = the piece codes of 8 consecutive samples
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, g = lane >> 4;
    const int c0 = ch * chunk;
    if (tid < kL0mSqB * 4) codes[(tid >> 2) * ocs + octs + (tid & 3)] = 0u;
    // piece codes, transposed: item = (sample octet, board word) -> the 8 x 8 nibble block of 8 samples x 8 squares, one code word per square
    for (int idx = tid; idx < 3 * octs; idx += 256) {
        const int wd = idx / octs, oq = idx - wd * octs;
        uint32_t bw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {                            // (unconditional, clamped: eight independent loads in flight)
            const int bsmp = c0 + oq * 8 + u;
            bw[u] = gboards[(long long)min(bsmp, n - 1) * kBoardWords + 3 * sqg + wd];
            if (bsmp >= n) bw[u] = 0;
        }
#pragma unroll
        for (int sq8 = 0; sq8 < 8; ++sq8) {
            uint32_t wv = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) wv |= ((bw[u] >> (4 * sq8)) & 15u) << (4 * u);
            codes[(wd * 8 + sq8) * ocs + oq] = wv;
        }
    }
    // ---- delta operand: global -> LDS by LDS-DMA (global_load_lds_dwordx4: no register round trip), three k-steps deep ----------------
    // A piece = 1 KB = (plane t, 16 columns) x 32 k: lane L of the DMA carries the 16 bytes (8 k) of column L >> 2, k-octet
    // (L & 3) ^ f(column), f(col) = (col >> 1) & 3 — the XOR sits on the SOURCE address, the LDS image is written linearly; a fragment
    // read of 8 lanes (one k-octet, 8 columns) then touches 8 different 16-byte bank groups: conflict-free ds_read_b128.
    // Every wave issues two pieces per k-step (6 pieces: waves 2 and 3 repeat their first one — same bytes to the same place), so
    // that one vmcnt constant serves all of them.
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)stage);
    const int pc0 = wid, pc1 = wid + 4 < 6 ? wid + 4 : wid;
    unsigned voff[2];
    {
        const int col = lane >> 2, kq = (lane & 3) ^ ((col >> 1) & 3);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pc = q ? pc1 : pc0, t = pc >> 1, c16 = pc & 1;
            voff[q] = (unsigned)((t * plane_stride + (long long)(cb * kL0mCols + c16 * 16 + col) * kpad + c0 + kq * 8) * 2);
        }
    }
    const unsigned char* pbase = reinterpret_cast<const unsigned char*>(planes);
    auto issue = [&](int ks, unsigned st_off) {
        unsigned keep;
        const unsigned char* base = pbase + (long long)ks * 64;
        const unsigned l0 = lds0 + st_off + (unsigned)pc0 * 1024u, l1 = lds0 + st_off + (unsigned)pc1 * 1024u;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
            "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff[0]), "v"(voff[1]), "s"(base), "s"(l0), "s"(l1)
            : "memory");
    };
    issue(0, 0u);
    issue(1, (unsigned)kL0mStageB);
    f32x4 acc[kL0mRC][kL0mRS];
#pragma unroll
    for (int c = 0; c < kL0mRC; ++c)
#pragma unroll
        for (int j = 0; j < kL0mRS; ++j) acc[c][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");             // this wave's pieces of k-step 0 have landed
    __syncthreads();                                             // ... and everybody's; the code words are complete
    const uint32_t* crow = codes + (wid * kL0mRS) * ocs + g;
    // one-hot fragment of (square j, 8 samples) for this lane's plane (piece code m + 1) from the word of 8 nibbles, in registers: XOR
    // with the code replicated makes the matching nibbles zero; OR-folding each nibble onto its bit 0 and inverting leaves bit 4u set
    // iff sample u matches; register i of the fragment holds samples (i, i + 4), so it is ((match << (14 - 4 i)) & 0x40004000): bit 14
    // of a half-word = the bf16 value 2.0 (delta0 was halved by the split, exactly).  ~14 VALU, no LDS — the first version looked the
    // half-words up in a 16-KB LDS table (4 dependent ds_reads per fragment: 5 us of a 27-us kernel).
    const uint32_t code8 = (uint32_t)(m + 1) * 0x11111111u;
    auto onehot = [&](uint32_t wv) {
        uint32_t x = wv ^ code8;
        x |= x >> 1;
        x |= x >> 2;
        const uint32_t match = ~x & 0x11111111u;
        const f32x4 bw = {__builtin_bit_cast(float, (match << 14) & 0x40004000u), __builtin_bit_cast(float, (match << 10) & 0x40004000u),
                          __builtin_bit_cast(float, (match << 6) & 0x40004000u), __builtin_bit_cast(float, (match << 2) & 0x40004000u)};
        return __builtin_bit_cast(bf16x8, bw);
    };
    const int ksteps = DBG == 1 ? 0 : chunk / 32;
    bf16x8 b[kL0mRS], bn[kL0mRS];
    uint32_t wn[kL0mRS];
#pragma unroll
    for (int j = 0; j < kL0mRS; ++j) b[j] = onehot(crow[j * ocs]);
    // fragment of (column tile c, plane t) inside a stage: piece 2 t + c, row m (64 B), k-octet g at slot g ^ f(m)
    const unsigned foff = (unsigned)(m * 64 + ((g ^ ((m >> 1) & 3)) << 4));
    // One barrier per k-step; no branch in the loop body: the DMA of the last two k-steps runs past the chunk into memory that exists
    // (the next column's samples / the slack behind the planes) and lands in stages nobody reads any more; drained behind the loop.
    unsigned st_rd = 0u, st_wr = 2u * kL0mStageB;
    for (int ks = 0; ks < ksteps; ++ks) {
        if (DBG != 4) issue(ks + 2, st_wr);                      // -> the stage read during k-step ks - 1 (everybody is past that barrier)
#pragma unroll
        for (int j = 0; j < kL0mRS; ++j) wn[j] = crow[j * ocs + (ks + 1) * 4];
        bf16x8 a[kL0mRC][3];
        const unsigned char* sp = stage + st_rd + foff;
#pragma unroll
        for (int c = 0; c < kL0mRC; ++c)
#pragma unroll
            for (int t = 0; t < 3; ++t) a[c][t] = *reinterpret_cast<const bf16x8*>(sp + (2 * t + c) * 1024);
        // issue order: a dependent MFMA (same accumulator) comes 12 MFMAs later, not 2 — the pipe never waits for its own result
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < kL0mRS; ++j) {
#pragma unroll
                for (int c = 0; c < kL0mRC; ++c) {
                    if (DBG != 2) acc[c][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[c][t], b[j], acc[c][j], 0, 0, 0);
                    else { acc[c][j][0] += (float)a[c][t][0] * (float)b[j][0]; }
                }
                if (t == 0) bn[j] = DBG == 3 ? __builtin_bit_cast(bf16x8, f32x4{__builtin_bit_cast(float, wn[j]), 0.f, 0.f, 0.f}) : onehot(wn[j]);
            }
#if XQ_L0M_SCHED
        // one MFMA, then up to three of the one-hot's VALU instructions, 36 times: the VALU work rides in the MFMAs' issue gaps
        // instead of stretching the first 12 of them
#pragma unroll
        for (int i = 0; i < 3 * kL0mRS * kL0mRC; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
#endif
#pragma unroll
        for (int j = 0; j < kL0mRS; ++j) b[j] = bn[j];
        st_rd = st_rd + kL0mStageB == kL0mStages * kL0mStageB ? 0u : st_rd + kL0mStageB;
        st_wr = st_wr + kL0mStageB == kL0mStages * kL0mStageB ? 0u : st_wr + kL0mStageB;
        if (DBG != 4) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");         // k-step ks + 1 has landed (this wave's pieces)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every fragment read of this stage has returned
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // no DMA may still be writing LDS when the block ends
    // D: lane (n = lane & 15, g): plane n of every square, columns 16 c + 4 g .. + 3 of the block's 32
    if (m < 14) {
        float* out = partial + ((long long)ch * kStateSize) * H;
#pragma unroll
        for (int j = 0; j < kL0mRS; ++j) {
            const int sq = sqg * kL0mSqB + wid * kL0mRS + j;
            if (sq >= kSquares) continue;
#pragma unroll
            for (int c = 0; c < kL0mRC; ++c) {
                const f32x4 v = acc[c][j];
                *reinterpret_cast<float4*>(out + (long long)(sq * 14 + m) * H + cb * kL0mCols + 16 * c + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

template <int DBG = 0>
__global__ __launch_bounds__(256) void l0_grad_mfma_kernel(const uint32_t* __restrict__ gboards, const uint16_t* __restrict__ planes,
                                                           long long plane_stride, int kpad, int n, int H, int chunk,
                                                           float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) uint32_t l0m_smem[];
    l0_grad_mfma_block<DBG>(gboards, planes, plane_stride, kpad, n, H, chunk, partial, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, l0m_smem);
}

}  // namespace xq
