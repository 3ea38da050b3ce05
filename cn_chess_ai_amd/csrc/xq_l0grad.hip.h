// xq_l0grad.hip.h — layer-0 weight gradient on the bf16 matrix pipe, exact in fp32 (second form, round 5):
//   gW0^T[(sq, piece)][col] = sum over the samples that have `piece` on `sq` of delta0[sample][col]
// (reference: updateWeightsBiasesKernel on layer 0, dqn.cu:310-319, fed by the one-hot of chessai.cpp:268-289, batch 1).
//
// Why a matrix product.  The segmented-sum kernel (l0_grad_block, xq_l0.hip.h) reads every 1-KB delta row once per occupied square of
// its sample — ~24 times, 200-250 MB through L2 per 8192-sample step — and was the longest kernel of the step (37-45 us).  As the dense
// product one-hot^T x delta0 the 18 GFLOP it costs are ~7 us of the 2.5 PFLOP/s bf16 pipe.  Two things make that exact:
//   * the one-hot operand is 0 / 1 — exact in bf16;
//   * delta0 = hi + mid + lo with three bf16 values per fp32 (8 + 8 + 8 significant bits; each residual is formed exactly in fp32), so
//     every product is exact and the MFMA accumulates in fp32: the result differs from a sequential fp32 sum only by summation order.
// v_mfma_f32_16x16x32_bf16 with M = 16 columns of delta0, N = 16 consecutive ROWS of gW0^T (row = square * 14 + plane: the 1260 rows are
// packed into 79 tiles of 16 — a tile spans two or three squares — instead of 90 squares x 16 planes with two planes idle: 17 % fewer
// MFMAs), K = 32 samples; a wave owns 2 column tiles x 5 row tiles (10 accumulators); the four waves of a block take 20 consecutive row
// tiles and the SAME 32 columns, whose operand tile goes global -> LDS once per block (50 MB through L2 per step).  Grid 4 row groups
// x H/32 column blocks x n/1024 chunks.
// The first form (round 4; 28.7 us alone at 8192 x 256, profiles/r04_g_*) derived the one-hot operand from nibble code words in
// registers; its loop was ISSUE-bound, not matrix-pipe-bound: per 32-sample k-step and wave 36 MFMAs (576 cycles) sat beside ~180
// vector instructions — 78 to derive six one-hot fragments (13 each), ~80 accumulator-register moves the compiler made of rotating
// operand registers — 1360 cycles measured, and every block opened with a 6-us transposition of its boards.  Here
//   * the one-hot operand is ONE v_perm_b32 per register.  A pre-pass (l0_sel_block) writes, per (square, sample), two 16-bit
//     "selector" half-words: (piece, 0) << 8 for pieces 1..7 (encoding 0) and (piece - 7) << 8 for pieces 8..14 (encoding 1), 0 otherwise.
//     v_perm_b32 picks every result byte out of an 8-byte per-lane table by a selector byte: lane n (plane n of a square: encoding
//     n >= 7, index n + 1 or n - 6) holds the table "0x40 at my index, 0 elsewhere", so a word of two selector half-words becomes two bf16
//     values 0x4000 (= 2.0; delta0 is halved by the split, exactly) or 0 in one instruction: 4 per fragment instead of 13;
//   * the selector words travel like the delta tile: global -> LDS by LDS-DMA, [square][k-octet][encoding][8 samples] so that a
//     fragment is one conflict-free ds_read_b128 — the block has no transposition prologue (6 of the old kernel's 28 us);
//   * 64 samples per barrier (two MFMA k-steps), a ring of three stages, every fragment read issued one k-step ahead of its MFMAs.
// Per k-step and wave: 30 MFMAs, 20 v_perm, 11 ds_read_b128.
#pragma once

#include "xq_gemm.hip.h"

namespace xq {

constexpr int kL0mRS = 5;                 // 16-row tiles per wave
constexpr int kL0mRC = 2;                 // 16-column tiles per wave
constexpr int kL0mRowsB = 4 * kL0mRS * 16; // rows of gW0^T per block: 320; 4 row groups cover rows 0..1279 (1260..1279: nothing)
constexpr int kL0mSqB = 24;               // squares whose selector words a block stages: rows 320 g .. 320 g + 319 lie on squares
                                          // floor(320 g / 14) .. + 23
constexpr int kL0mCols = 16 * kL0mRC;      // columns per block: 32
constexpr int kL0mStageK = 64;            // samples per stage
constexpr int kL0mAPieces = 2 * 3 * kL0mRC;            // (k-step, plane, 16-column tile) x 1 KB = 16 columns x 32 k
constexpr int kL0mSPieces = 2 * 3;                    // (k-step, 8 squares) x 1 KB = 8 squares x 4 octets x 2 encodings x 16 B
constexpr int kL0mStageB = (kL0mAPieces + kL0mSPieces) * 1024;
constexpr int kL0mStages = 3;
constexpr int kL0mSelSquares = 96;

// bf16 elements the three planes need: [3][H][kpad] + slack for the stages that are prefetched (harmlessly) past the last chunk
__host__ __device__ constexpr size_t l0m_plane_elems(int H, int kpad) { return (size_t)3 * H * kpad + 8 * 32 + 64; }
__host__ __device__ constexpr size_t l0m_lds_bytes() { return (size_t)kL0mStages * kL0mStageB; }
// u16 elements of the selector array [96][kpad / 8][2][8] + slack for the stage that is prefetched (harmlessly) past the last chunk
__host__ __device__ constexpr size_t l0sel_elems(int kpad) { return (size_t)kL0mSelSquares * kpad * 2 + 1024; }

// selector half-words of 64 consecutive samples (block `blk`): boards [n][12] packed nibbles -> sel[sq][octet][enc][8]
__device__ __forceinline__ void l0_sel_block(const uint32_t* __restrict__ gboards, int n, int kpad, uint16_t* __restrict__ sel, int blk,
                                             uint32_t* __restrict__ bw /* LDS [64][13] */) {
    const int tid = (int)threadIdx.x, b0 = blk * 64;
    for (int i = tid; i < 64 * kBoardWords; i += 256) {
        const int s = i / kBoardWords, w = i - s * kBoardWords;
        bw[s * 13 + w] = b0 + s < n ? gboards[(long long)(b0 + s) * kBoardWords + w] : 0u;
    }
    __syncthreads();
    const int octs = kpad >> 3;
    for (int p = tid; p < kL0mSelSquares * 16; p += 256) {            // piece = (square, octet of the block, encoding): 16 bytes
        const int sq = p >> 4, oct = (p >> 1) & 7, enc = p & 1;
        const int wd = sq >> 3, sh = 4 * (sq & 7);
        uint32_t o[4];
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            uint32_t v = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint32_t c = (bw[(oct * 8 + 2 * i2 + e) * 13 + wd] >> sh) & 15u;
                const uint32_t idx = enc ? (c >= 8u ? c - 7u : 0u) : (c <= 7u ? c : 0u);
                v |= idx << (8 + 16 * e);
            }
            o[i2] = v;
        }
        *reinterpret_cast<uint4*>(sel + (((long long)sq * octs + (b0 >> 3) + oct) * 2 + enc) * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
__global__ __launch_bounds__(256) void l0_sel_kernel(const uint32_t* __restrict__ gboards, int n, int kpad, uint16_t* __restrict__ sel) {
    __shared__ uint32_t bw[64 * 13];
    l0_sel_block(gboards, n, kpad, sel, (int)blockIdx.x, bw);
}

// delta0 [n][H] fp32 -> planes[t][col][k] bf16 (t = hi, mid, lo of 0.5 * delta0), k in natural order, zero-padded to kpad (multiple of 64)
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
        uint32_t pk[3] = {0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float v = tile[(kq + j + e) * 65 + col];
            const uint16_t hi = bf16_bits(0.5f * v);             // x 0.5: the one-hot operand is 2.0 (bit 14 of a half-word)
            const float r1 = 0.5f * v - bf16_to_float(hi);       // exact: the residual of a round-to-nearest to 8 bits
            const uint16_t mid = bf16_bits(r1);
            const float r2 = r1 - bf16_to_float(mid);            // exact, <= 8 significant bits left
            const uint16_t lo = bf16_bits(r2);
            pk[0] |= (uint32_t)hi << (16 * e); pk[1] |= (uint32_t)mid << (16 * e); pk[2] |= (uint32_t)lo << (16 * e);
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

// One block: row group `sqg` (rows 320 sqg .. + 319 of gW0^T), column block `cb` (32 columns), sample chunk `ch` (`chunk` samples, a
// multiple of 64).  smem: l0m_lds_bytes().  partial: [chunks][1260][H].  Wave w owns row tiles 20 sqg + 5 w .. + 4 and all 32 columns.
template <int DBG = 0>
__device__ __forceinline__ void l0_grad_mfma_block(const uint16_t* __restrict__ sel, const uint16_t* __restrict__ planes,
                                                    long long plane_stride, int kpad, int H, int chunk, float* __restrict__ partial,
                                                    int sqg, int cb, int ch, unsigned char* __restrict__ smem) {
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, g = lane >> 4;
    const int c0 = ch * chunk, octs = kpad >> 3;
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
    // ---- LDS-DMA: 18 pieces of 1 KB per stage; wave w issues pieces w, w + 4, w + 8 (delta tiles), w + 12 and 16 + (w & 1) (selectors:
    // waves 2 and 3 repeat the last piece of waves 0 and 1 — same bytes to the same place — so that one vmcnt constant serves all) -----
    // delta piece (kk, t, c16): lane L carries the 16 bytes (8 k) of column L >> 2, k-octet (L & 3) ^ f(column), f(col) = (col >> 1) & 3
    // — the XOR sits on the SOURCE address, the LDS image is linear; a fragment read then touches 8 different 16-byte bank groups.
    // selector piece (kk, q): lane L carries square 8 q + (L >> 3), k-octet (L >> 1) & 3, encoding L & 1: the LDS image of a square is
    // 128 contiguous bytes [octet][encoding][8 samples].
    unsigned voff[5];
    int piece[5];
    {
        const int col = lane >> 2, kq = (lane & 3) ^ ((col >> 1) & 3);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int p = wid + 4 * i, kk = p / 6, r = p - 6 * kk, t = r >> 1, c16 = r & 1;
            piece[i] = p;
            voff[i] = (unsigned)((t * plane_stride + (long long)(cb * kL0mCols + c16 * 16 + col) * kpad + c0 + kk * 32 + kq * 8) * 2);
        }
#pragma unroll
        for (int i = 3; i < 5; ++i) {
            const int sp = i == 3 ? wid : 4 + (wid & 1), kk = sp / 3, q = sp - 3 * kk;
            piece[i] = kL0mAPieces + sp;
            const int sq = (sqg * kL0mRowsB) / 14 + q * 8 + (lane >> 3), gg = (lane >> 1) & 3, enc = lane & 1;
            voff[i] = (unsigned)((((long long)sq * octs + (c0 >> 3) + kk * 4 + gg) * 2 + enc) * 16);
        }
    }
    const unsigned char* pA = reinterpret_cast<const unsigned char*>(planes);
    const unsigned char* pS = reinterpret_cast<const unsigned char*>(sel);
    // one 1-KB piece of stage `st` into the ring buffer at byte offset buf_off (i = which of the wave's five)
    auto issue1 = [&](int st, unsigned buf_off, int i) {
        unsigned keep;
        const unsigned char* base = i < 3 ? pA + (long long)st * (kL0mStageK * 2) : pS + (long long)st * (kL0mStageK / 8 * 32);
        const unsigned l = lds0 + buf_off + (unsigned)piece[i] * 1024u;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff[i]), "s"(base), "s"(l) : "memory");
    };
    auto issue = [&](int st, unsigned buf_off) {
#pragma unroll
        for (int i = 0; i < 5; ++i) issue1(st, buf_off, i);
    };
    issue(0, 0u);
    f32x4 acc[kL0mRC][kL0mRS];
#pragma unroll
    for (int c = 0; c < kL0mRC; ++c)
#pragma unroll
        for (int j = 0; j < kL0mRS; ++j) acc[c][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // per-lane tables of v_perm_b32, one per row tile: lane n of tile j is row r = 320 sqg + 16 (5 w + j) + n of gW0^T = plane r % 14 of
    // square r / 14.  Byte `idx` of the 8-byte table {S0 : S1} = 0x40 (the high byte of bf16 2.0); selector 0 (empty square, a piece of
    // the other encoding, every LOW byte of a half-word) picks byte 0 = 0.  Planes 0..6 = pieces 1..7 (encoding 0, index plane + 1), planes
    // 7..13 = pieces 8..14 (encoding 1, index plane - 6); rows >= 1260 have index 8: an all-zero table.
    // fragment offsets inside a stage: the delta fragment as before; the selector fragment of tile j is the lane's OWN square's 16 bytes
    // [octet g][encoding] (the lanes of a tile read two or three squares).
    uint32_t tlo[kL0mRS], thi[kL0mRS];
    unsigned soff[kL0mRS];
    const unsigned aoff = (unsigned)(m * 64 + ((g ^ ((m >> 1) & 3)) << 4));
    const int sq0 = (sqg * kL0mRowsB) / 14;
#pragma unroll
    for (int j = 0; j < kL0mRS; ++j) {
        const int r = sqg * kL0mRowsB + (wid * kL0mRS + j) * 16 + m;
        const int sq = r / 14, pl = r - sq * 14;
        const int enc = pl >= 7 ? 1 : 0;
        const int idx = r >= kStateSize ? 8 : (enc ? pl - 6 : pl + 1);
        tlo[j] = idx < 4 ? 0x40u << (8 * idx) : 0u;
        thi[j] = (idx >= 4 && idx < 8) ? 0x40u << (8 * (idx - 4)) : 0u;
        const int sl = min(sq - sq0, kL0mSqB - 1);
        soff[j] = (unsigned)((kL0mAPieces + (sl >> 3)) * 1024 + (sl & 7) * 128 + g * 32 + enc * 16);
    }
    // ---- the loop: ONE barrier per 64-sample stage, placed in the middle of the stage ------------------------------------------------
    // Stage st's fragments are read one k-step ahead of their MFMAs: k-step 1's at the top of the stage (under k-step 0's MFMAs),
    // k-step 0's of stage st + 1 right behind the barrier in the middle of stage st (under k-step 1's MFMAs) — no MFMA ever waits for
    // an LDS read issued just before it.  Ring of three stages: the DMA of stage st + 2 is issued at the top of stage st into the
    // buffer stage st - 1 was read from; every wave's last read of that buffer (k-step 1 of stage st - 1, requested at the top of that
    // stage) returned before the barrier in the middle of stage st - 1.
    struct Frags { bf16x8 a[kL0mRC][3]; uint4 sw[kL0mRS]; };
    // the 11 fragment reads of a k-step in the order their consumers come: selector word of tile 0 (the previous k-step's last group
    // turns it into a one-hot), then plane by plane
    constexpr int kReads = 6 + kL0mRS;
    auto read_frag = [&](Frags& f, unsigned boff, int kk, int i) {
        const unsigned char* sp = smem + boff;
        constexpr int kind[kReads] = {0, 1, 1, 0, 1, 1, 0, 1, 1, 0, 0};     // 0: selector word, 1: delta fragment
        constexpr int arg[kReads] = {0, 0, 1, 1, 2, 3, 2, 4, 5, 3, 4};      // tile j, or t * 2 + c
        if (kind[i] == 0) f.sw[arg[i]] = *reinterpret_cast<const uint4*>(sp + kk * 3 * 1024 + soff[arg[i]]);
        else f.a[arg[i] & 1][arg[i] >> 1] = *reinterpret_cast<const bf16x8*>(sp + (kk * 6 + arg[i]) * 1024 + aoff);
    };
    auto read_frags = [&](Frags& f, unsigned boff, int kk) {
#pragma unroll
        for (int i = 0; i < kReads; ++i) read_frag(f, boff, kk, i);
    };
    auto onehot2 = [&](int j, uint32_t x, uint32_t y, uint32_t& ox, uint32_t& oy) {
        if (DBG == 3 || DBG == 6) { ox = x; oy = y; }
        else { ox = __builtin_amdgcn_perm(thi[j], tlo[j], x); oy = __builtin_amdgcn_perm(thi[j], tlo[j], y); }
    };
    // one k-step: 5 groups (row tiles) x 3 MFMA pairs (planes) on the wave's 10 accumulators.  Between the pairs, pinned there by
    // sched_barrier: the one-hot operand of the NEXT group (v_perm_b32 x 2 per pair; the next square of this k-step, or square 0 of
    // `nxt`), and side(pair) — ONE of the next k-step's fragment reads or ONE LDS-DMA piece.  An MFMA holds the vector issue for 8 of its
    // 16 cycles, so a pair has room for ~16 cycles of other instructions; left alone hipcc puts a k-step's 24 v_perm, its 12 reads and
    // the DMA in front of its 36 MFMAs, and with one wave per SIMD the matrix pipe idles through all of it.
    // The MFMAs are asm statements with the accumulators tied in place ("+v"): left to the register allocator they came out of the
    // loop body in rotated registers, 68 v_accvgpr moves per iteration.  hipcc pads no hazards inside asm: every one-hot fragment is
    // written >= 2 MFMAs before the MFMA that reads it, except the very first of the block (s_nop 1 below).
    auto kstep = [&](const Frags& f, const Frags& nxt, uint4& o, auto&& side) {
#pragma unroll
        for (int j = 0; j < kL0mRS; ++j) {
            const bf16x8 b = __builtin_bit_cast(bf16x8, o);
            const uint4 nx = j + 1 < kL0mRS ? f.sw[j + 1] : nxt.sw[0];
            const int jn = j + 1 < kL0mRS ? j + 1 : 0;
            uint4 on;
            if (DBG == 2) {                                      // no MFMAs: their operands stay alive, nothing else changes
                asm volatile("" : "+v"(acc[0][j]), "+v"(acc[1][j]) : "v"(f.a[0][0]), "v"(f.a[1][0]), "v"(f.a[0][1]), "v"(f.a[1][1]), "v"(f.a[0][2]), "v"(f.a[1][2]), "v"(b));
                onehot2(jn, nx.x, nx.y, on.x, on.y); onehot2(jn, nx.z, nx.w, on.z, on.w);
                side(3 * j); side(3 * j + 1); side(3 * j + 2);
            } else {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %4, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %3, %4, %1"
                             : "+v"(acc[0][j]), "+v"(acc[1][j]) : "v"(f.a[0][0]), "v"(f.a[1][0]), "v"(b));
                onehot2(jn, nx.x, nx.y, on.x, on.y);
                side(3 * j);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %4, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %3, %4, %1"
                             : "+v"(acc[0][j]), "+v"(acc[1][j]) : "v"(f.a[0][1]), "v"(f.a[1][1]), "v"(b));
                onehot2(jn, nx.z, nx.w, on.z, on.w);
                side(3 * j + 1);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %4, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %3, %4, %1"
                             : "+v"(acc[0][j]), "+v"(acc[1][j]) : "v"(f.a[0][2]), "v"(f.a[1][2]), "v"(b));
                side(3 * j + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            o = on;
        }
    };
    const int nst = DBG == 1 ? 0 : chunk / kL0mStageK;
    issue(1, (unsigned)kL0mStageB);
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");             // stage 0 has landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                                // ... and everybody's
    asm volatile("" ::: "memory");
    Frags f0, f1;
    read_frags(f0, 0u, 0);
    if (DBG == 6) read_frags(f1, 0u, 1);
    uint4 o;
    onehot2(0, f0.sw[0].x, f0.sw[0].y, o.x, o.y);
    onehot2(0, f0.sw[0].z, f0.sw[0].w, o.z, o.w);
    asm volatile("s_nop 1" : "+v"(o.x), "+v"(o.y), "+v"(o.z), "+v"(o.w));
    unsigned b_cur = 0u, b_nxt = (unsigned)kL0mStageB, b_far = 2u * kL0mStageB;
    for (int st = 0; st < nst; ++st) {
        // (the last two stages request 64 / 128 samples past the chunk — the next chunk, the next column or the slack behind the arrays —
        // into buffers nobody reads any more)
        // (15 slots per k-step: 11 reads, then LDS-DMA pieces — four of stage st + 2's five here, the fifth in the second k-step)
        kstep(f0, f1, o, [&](int p) {
            if (DBG == 6) return;
            if (p < kReads) read_frag(f1, b_cur, 1, p);
            else if (p < kReads + 4 && DBG != 4) issue1(st + 2, b_far, p - kReads);
        });
        // stage st + 1 has landed (this wave's pieces): all but the four youngest operations — stage st + 2's first four — are done
        if (DBG != 4 && DBG != 6) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // my reads of stage st have returned
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        kstep(f1, f0, o, [&](int p) {
            if (DBG == 6) return;
            if (p < kReads) read_frag(f0, b_nxt, 0, p);
            else if (p == kReads && DBG != 4) issue1(st + 2, b_far, 4);
        });
        const unsigned t = b_cur; b_cur = b_nxt; b_nxt = b_far; b_far = t;
    }
    // no DMA may still be writing LDS when the block ends; and the last MFMAs' results need 8+ wait states before anything but an
    // accumulating MFMA reads them (hipcc does not know the statements above were MFMAs)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    // D: lane (n = lane & 15, g): row n of every row tile, columns 16 c + 4 g .. + 3 of the block's 32
    {
        float* out = partial + ((long long)ch * kStateSize) * H;
#pragma unroll
        for (int j = 0; j < kL0mRS; ++j) {
            const int r = sqg * kL0mRowsB + (wid * kL0mRS + j) * 16 + m;
            if (r >= kStateSize) continue;
#pragma unroll
            for (int c = 0; c < kL0mRC; ++c) {
                const f32x4 v = acc[c][j];
                *reinterpret_cast<float4*>(out + (long long)r * H + cb * kL0mCols + 16 * c + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

template <int DBG = 0>
__global__ __launch_bounds__(256) void l0_grad_mfma_kernel(const uint16_t* __restrict__ sel, const uint16_t* __restrict__ planes,
                                                            long long plane_stride, int kpad, int H, int chunk, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char l0m_smem[];
    // grid (H / 32, 4, chunks): the column block is the fastest index, so that an XCD (workgroups round-robin in linear order) keeps to
    // its own columns' planes
    l0_grad_mfma_block<DBG>(sel, planes, plane_stride, kpad, H, chunk, partial, (int)blockIdx.y, (int)blockIdx.x, (int)blockIdx.z, l0m_smem);
}

}  // namespace xq
