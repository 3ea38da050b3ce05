// xq_l0.hip.h — layer 0 of the Q-network as gathers over packed boards: forward (l0_forward_kernel, l0_select_kernel: the one-hot of
// chessai.cpp:268-289 is never built, forwardKernel dqn.cu:184-195 on <= 32 rows), the bf16 shadow of the output layer for the screening pass, and
// the segmented-sum form of the layer-0 weight gradient (l0_grad_block; updateWeightsBiasesKernel dqn.cu:310-319)
// (kernel half of xq_dqn.hip, split out in round 5; included by xq_dqn.hip only, inside namespace xq)
#pragma once

namespace xq {


// where sample b of a minibatch lives: identity, an explicit slot list, or the replay sampler's Philox stream recomputed
// in place (ctr = {b, 0, call, 1}, key = seed, % size — identical to replay_sample_kernel).  A windowed sample draws from the
// `size` ring slots that start at `start` (the overlapped trainer excludes the slots a concurrent collect is writing).
struct SlotSrc {
    const int32_t* slots;
    uint32_t implicit, call, seed_lo, seed_hi, size, start, cap;
};
__device__ __forceinline__ int slot_of(const SlotSrc& s, int b) {
    if (s.implicit) {
        uint32_t v = s.start + philox4x32_10((uint32_t)b, 0u, s.call, 1u, s.seed_lo, s.seed_hi).v[0] % s.size;
        if (v >= s.cap) v -= s.cap;
        return (int)v;
    }
    return s.slots ? s.slots[b] : b;
}
// the forward chains of a TD step (s on the online net, s' on the TD net, and for Double DQN s' on the target net as well)
// share one launch per layer
enum { kMaxChains = 3 };
// bf16 copy of a weight matrix [NO][K] (K % 64 == 0) + the largest row norm (exact screening of max_a' Q(s',a'), see
// qmax_refine_kernel).  A quarter-wave per row, 2 rows per quarter, all of a quarter's loads in flight together (pure latency:
// 8 MB in, 4 MB out); block `blk` of 256 threads takes rows [32 blk, 32 blk + 32); rows >= NO of the padded copy stay zero.
struct ShadowJob {
    const float* W; const float* bias; int NO, K; uint16_t* Wb;
    unsigned* w_dyn; unsigned* b_dyn;         // rows 0..95 (kShadowDynBlocks blocks): this step's parity slots
    unsigned* w_stat; unsigned* b_stat;       // rows >= 96
    int nblocks;                              // blocks to run: all of them, or kShadowDynBlocks when rows >= 96 are still valid
};
enum { kShadowRows = 32, kShadowDynBlocks = 3 };
__device__ __forceinline__ void screen_shadow_block(const ShadowJob& S, int blk, float* nrm /* LDS [8] */) {
    const int ql = (int)(threadIdx.x & 15), quarter = (int)(threadIdx.x >> 4);
    const int row0 = blk * kShadowRows + quarter * 2;
    float mx = 0.f;
    float bm = fmaxf(row0 < S.NO ? fabsf(S.bias[row0]) : 0.f, row0 + 1 < S.NO ? fabsf(S.bias[row0 + 1]) : 0.f);
    if (S.K == 256) {
        float4 x[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                x[u][t] = *reinterpret_cast<const float4*>(S.W + (long long)min(row0 + u, S.NO - 1) * 256 + t * 64 + ql * 4);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float ss = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                ss += x[u][t].x * x[u][t].x + x[u][t].y * x[u][t].y + x[u][t].z * x[u][t].z + x[u][t].w * x[u][t].w;
                if (row0 + u < S.NO) {
                    const uint16_t q0 = bf16_bits(x[u][t].x), q1 = bf16_bits(x[u][t].y), q2 = bf16_bits(x[u][t].z), q3 = bf16_bits(x[u][t].w);
                    *reinterpret_cast<uint2*>(S.Wb + (long long)(row0 + u) * 256 + t * 64 + ql * 4) =
                        make_uint2((uint32_t)q0 | ((uint32_t)q1 << 16), (uint32_t)q2 | ((uint32_t)q3 << 16));
                }
            }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
            mx = fmaxf(mx, row0 + u < S.NO ? ss : 0.f);
        }
    } else {
        for (int u = 0; u < 2; ++u) {
            float ss = 0.f;
            if (row0 + u < S.NO)
                for (int k = ql * 4; k < S.K; k += 64) {
                    const float4 y = *reinterpret_cast<const float4*>(S.W + (long long)(row0 + u) * S.K + k);
                    ss += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
                    const uint16_t q0 = bf16_bits(y.x), q1 = bf16_bits(y.y), q2 = bf16_bits(y.z), q3 = bf16_bits(y.w);
                    *reinterpret_cast<uint2*>(S.Wb + (long long)(row0 + u) * S.K + k) = make_uint2((uint32_t)q0 | ((uint32_t)q1 << 16), (uint32_t)q2 | ((uint32_t)q3 << 16));
                }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
            mx = fmaxf(mx, ss);
        }
    }
#pragma unroll
    for (int off = 32; off >= 16; off >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, off, 64)); bm = fmaxf(bm, __shfl_xor(bm, off, 64)); }
    if ((threadIdx.x & 63) == 0) { nrm[threadIdx.x >> 6] = mx; nrm[4 + (threadIdx.x >> 6)] = bm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = sqrtf(fmaxf(fmaxf(nrm[0], nrm[1]), fmaxf(nrm[2], nrm[3])));
        const float b = fmaxf(fmaxf(nrm[4], nrm[5]), fmaxf(nrm[6], nrm[7]));
        unsigned* wslot = blk < kShadowDynBlocks ? S.w_dyn : S.w_stat;
        unsigned* bslot = blk < kShadowDynBlocks ? S.b_dyn : S.b_stat;
        const unsigned mb = __builtin_bit_cast(unsigned, m), bb = __builtin_bit_cast(unsigned, b);
        if (mb > *reinterpret_cast<volatile unsigned*>(wslot)) atomicMax(wslot, mb);
        if (bb > *reinterpret_cast<volatile unsigned*>(bslot)) atomicMax(bslot, bb);
    }
}
__global__ __launch_bounds__(256) void screen_shadow_kernel(ShadowJob S) {
    __shared__ float nrm[8];
    screen_shadow_block(S, (int)blockIdx.x, nrm);
}

// tanh of layer 0 on every route (the gather kernels here, the dense-input product of chain_dense): tanh_hidden, like the hidden layers
// (same-box A/B of the step: 0.1706 -> 0.1697 ms).  XQ_L0_FAST_TANH=0 restores libm's tanhf on both routes (A/B builds).
#ifndef XQ_L0_FAST_TANH
#define XQ_L0_FAST_TANH 1
#endif
__device__ __forceinline__ float tanh_l0(float x) {
#if XQ_L0_FAST_TANH
    return tanh_hidden(x);
#else
    return tanhf(x);
#endif
}

struct L0Jobs {
    const uint32_t* boards[kMaxChains];
    const float* W0T[kMaxChains];
    const uint16_t* W0T_bf[kMaxChains];      // bf16 Q-net: shadow of W0^T (same [1260][H] order)
    const float* b0[kMaxChains];
    float* out[kMaxChains];                  // fp32 activations (may be nullptr in bf16 mode when nothing reads them)
    uint16_t* out_bf[kMaxChains];            // bf16 Q-net: bf16 bits of the activations
    uint32_t* gathered[kMaxChains];
    int njobs;
    int nrows;                               // grid rows that gather (njobs - derive_next); the shadow row, if any, is row nrows
    int derive_next;                         // 1: job 0's waves also produce job 1 (s' = s after one move, SAME net) from their own layer-0 sums:
                                             // z1(s') = z1(s) - rows of the squares that changed + rows of what stands there now
    int out_bf_frag;                         // fp32 net: the bf16 copy out_bf is written in MFMA B-fragment order (scr_afrag_index)
    ShadowJob shadow;                        // W != nullptr: the blocks of grid row y == njobs convert the screening shadow (no extra launch)
};

// Layer 0 from packed boards: a_1 = tanh(b_0 + sum over occupied squares of W0^T[sq*14 + piece-1][:]).
// One wave per sample; ascending square order = the reference's i-ascending accumulation with the zeros skipped.
// BF16: rows come from the bf16 shadow (half the L2 traffic of this gather), the sum runs in fp32, the result is rounded to bf16.
template <bool BF16>
__global__ __launch_bounds__(256) void l0_forward_kernel(L0Jobs J, SlotSrc src, int n, int H) {
    __shared__ __attribute__((aligned(16))) int rows[4][96];
    if ((int)blockIdx.y == J.nrows) {                   // block-uniform: the screening shadow rides in the same grid
        if ((int)blockIdx.x < J.shadow.nblocks) screen_shadow_block(J.shadow, (int)blockIdx.x, reinterpret_cast<float*>(&rows[0][0]));
        return;
    }
    const int wid = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int b = (int)blockIdx.x * 4 + wid;
    if (b >= n) return;
    const int job = (J.derive_next && blockIdx.y >= 1) ? (int)blockIdx.y + 1 : (int)blockIdx.y;    // job 1 is produced by job 0's waves
    const float* __restrict__ W0T = J.W0T[job];
    const uint16_t* __restrict__ W0B = J.W0T_bf[job];
    const float* __restrict__ b0 = J.b0[job];
    float* __restrict__ out = J.out[job];
    uint16_t* __restrict__ out_bf = J.out_bf[job];
    uint32_t* __restrict__ gathered = J.gathered[job];
    const int srow = slot_of(src, b);
    const uint32_t* bw = J.boards[job] + (long long)srow * kBoardWords;
    if (gathered != nullptr && lane < kBoardWords)      // the minibatch's boards, contiguous, for the layer-0 gradient
        gathered[(long long)b * kBoardWords + lane] = bw[lane];
    const int s0 = lane, s1 = 64 + lane;
    const uint32_t n0 = (bw[s0 >> 3] >> (4 * (s0 & 7))) & 15u;
    const uint32_t n1 = s1 < kSquares ? (bw[s1 >> 3] >> (4 * (s1 & 7))) & 15u : 0u;
    const unsigned long long m0 = __ballot(n0 != 0), m1 = __ballot(n1 != 0);
    const int c0 = __popcll(m0);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (n0) rows[wid][__popcll(m0 & below)] = s0 * 14 + (int)n0 - 1;
    if (n1) rows[wid][c0 + __popcll(m1 & below)] = s1 * 14 + (int)n1 - 1;
    const int cnt = c0 + __popcll(m1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // s' from s: the squares whose piece code differs (a move changes two), in ascending order, as (row to take out, row to put in);
    // boards that are not a move apart (an unused slot, a foreign pair) get the full row list of s' instead
    __shared__ __attribute__((aligned(16))) int dpair[4][8][2];
    __shared__ __attribute__((aligned(16))) int rows2[4][96];
    int nd = -1, cnt2 = 0;                                // nd = -1: nothing derived here; nd > 8: cnt2 rows in rows2
    if (J.derive_next && job == 0) {
        const uint32_t* bw2 = J.boards[1] + (long long)srow * kBoardWords;
        const uint32_t p0 = (bw2[s0 >> 3] >> (4 * (s0 & 7))) & 15u;
        const uint32_t p1 = s1 < kSquares ? (bw2[s1 >> 3] >> (4 * (s1 & 7))) & 15u : 0u;
        const unsigned long long d0 = __ballot(p0 != n0), d1 = __ballot(p1 != n1);
        nd = __popcll(d0) + __popcll(d1);
        if (nd <= 8) {
            if (p0 != n0) { const int k = __popcll(d0 & below); dpair[wid][k][0] = n0 ? s0 * 14 + (int)n0 - 1 : -1; dpair[wid][k][1] = p0 ? s0 * 14 + (int)p0 - 1 : -1; }
            if (p1 != n1) { const int k = __popcll(d0) + __popcll(d1 & below); dpair[wid][k][0] = n1 ? s1 * 14 + (int)n1 - 1 : -1; dpair[wid][k][1] = p1 ? s1 * 14 + (int)p1 - 1 : -1; }
        } else {
            const unsigned long long q0 = __ballot(p0 != 0), q1 = __ballot(p1 != 0);
            const int e0 = __popcll(q0);
            if (p0) rows2[wid][__popcll(q0 & below)] = s0 * 14 + (int)p0 - 1;
            if (p1) rows2[wid][e0 + __popcll(q1 & below)] = s1 * 14 + (int)p1 - 1;
            cnt2 = e0 + __popcll(q1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    auto load4 = [&](int row, int col) -> float4 {
        if (BF16) {
            const uint2 x = *reinterpret_cast<const uint2*>(W0B + (long long)row * H + col);
            return make_float4(__builtin_bit_cast(float, x.x << 16), __builtin_bit_cast(float, x.x & 0xFFFF0000u),
                               __builtin_bit_cast(float, x.y << 16), __builtin_bit_cast(float, x.y & 0xFFFF0000u));
        }
        return *reinterpret_cast<const float4*>(W0T + (long long)row * H + col);
    };
    if (BF16 && (H & 7) == 0 && ((H >= 512 && (H & 511) == 0) || (H >= 64 && 512 % H == 0))) {
        // 16-byte loads (8 bf16 per lane): a 1-KB row needs all 64 lanes; narrower rows are shared out — lane group g takes the
        // rows i = g (mod G) — and the groups' partial sums are combined by a fixed shuffle tree
        const int lpr = H >= 512 ? 64 : H / 8;            // lanes per row
        const int G = 64 / lpr, grp = lane / lpr, lc = lane - grp * lpr;
        for (int col = lc * 8; col < H; col += 512) {
            float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            auto add8 = [&](const uint4& x) {
                a[0] += __builtin_bit_cast(float, x.x << 16); a[1] += __builtin_bit_cast(float, x.x & 0xFFFF0000u);
                a[2] += __builtin_bit_cast(float, x.y << 16); a[3] += __builtin_bit_cast(float, x.y & 0xFFFF0000u);
                a[4] += __builtin_bit_cast(float, x.z << 16); a[5] += __builtin_bit_cast(float, x.z & 0xFFFF0000u);
                a[6] += __builtin_bit_cast(float, x.w << 16); a[7] += __builtin_bit_cast(float, x.w & 0xFFFF0000u);
            };
            int i = grp;
            for (; i + 3 * G < cnt; i += 4 * G) {
                const uint4 x0 = *reinterpret_cast<const uint4*>(W0B + (long long)rows[wid][i] * H + col);
                const uint4 x1 = *reinterpret_cast<const uint4*>(W0B + (long long)rows[wid][i + G] * H + col);
                const uint4 x2 = *reinterpret_cast<const uint4*>(W0B + (long long)rows[wid][i + 2 * G] * H + col);
                const uint4 x3 = *reinterpret_cast<const uint4*>(W0B + (long long)rows[wid][i + 3 * G] * H + col);
                add8(x0); add8(x1); add8(x2); add8(x3);
            }
            for (; i < cnt; i += G) add8(*reinterpret_cast<const uint4*>(W0B + (long long)rows[wid][i] * H + col));
            for (int off = lpr; off < 64; off <<= 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] += __shfl_xor(a[k], off, 64);
            }
            if (grp == 0) {
                const float4 ba = *reinterpret_cast<const float4*>(b0 + col), bb = *reinterpret_cast<const float4*>(b0 + col + 4);
                const float bias[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
                uint16_t qv[8];
                float tv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { qv[k] = bf16_bits(tanh_fast(a[k] + bias[k])); tv[k] = bf16_to_float(qv[k]); }
                *reinterpret_cast<uint4*>(out_bf + (long long)b * H + col) =
                    make_uint4((uint32_t)qv[0] | ((uint32_t)qv[1] << 16), (uint32_t)qv[2] | ((uint32_t)qv[3] << 16),
                               (uint32_t)qv[4] | ((uint32_t)qv[5] << 16), (uint32_t)qv[6] | ((uint32_t)qv[7] << 16));
                if (out) {
                    *reinterpret_cast<float4*>(out + (long long)b * H + col) = make_float4(tv[0], tv[1], tv[2], tv[3]);
                    *reinterpret_cast<float4*>(out + (long long)b * H + col + 4) = make_float4(tv[4], tv[5], tv[6], tv[7]);
                }
                if (nd >= 0) {                          // the s' chain of the same sample, same net (xq_dqn_set_l0_derive), from these sums
                    float a2[8];
                    if (nd <= 8) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) a2[k] = a[k];
                        auto acc8 = [&](const uint4& x, float sgn) {
                            a2[0] += sgn * __builtin_bit_cast(float, x.x << 16); a2[1] += sgn * __builtin_bit_cast(float, x.x & 0xFFFF0000u);
                            a2[2] += sgn * __builtin_bit_cast(float, x.y << 16); a2[3] += sgn * __builtin_bit_cast(float, x.y & 0xFFFF0000u);
                            a2[4] += sgn * __builtin_bit_cast(float, x.z << 16); a2[5] += sgn * __builtin_bit_cast(float, x.z & 0xFFFF0000u);
                            a2[6] += sgn * __builtin_bit_cast(float, x.w << 16); a2[7] += sgn * __builtin_bit_cast(float, x.w & 0xFFFF0000u);
                        };
                        for (int k = 0; k < nd; ++k) {     // wave-uniform
                            const int ro = dpair[wid][k][0], ri = dpair[wid][k][1];
                            if (ro >= 0) acc8(*reinterpret_cast<const uint4*>(W0B + (long long)ro * H + col), -1.f);
                            if (ri >= 0) acc8(*reinterpret_cast<const uint4*>(W0B + (long long)ri * H + col), 1.f);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) a2[k] = 0.f;
                        for (int k = 0; k < cnt2; ++k) {
                            const uint4 x = *reinterpret_cast<const uint4*>(W0B + (long long)rows2[wid][k] * H + col);
                            a2[0] += __builtin_bit_cast(float, x.x << 16); a2[1] += __builtin_bit_cast(float, x.x & 0xFFFF0000u);
                            a2[2] += __builtin_bit_cast(float, x.y << 16); a2[3] += __builtin_bit_cast(float, x.y & 0xFFFF0000u);
                            a2[4] += __builtin_bit_cast(float, x.z << 16); a2[5] += __builtin_bit_cast(float, x.z & 0xFFFF0000u);
                            a2[6] += __builtin_bit_cast(float, x.w << 16); a2[7] += __builtin_bit_cast(float, x.w & 0xFFFF0000u);
                        }
                    }
                    uint16_t q2[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) q2[k] = bf16_bits(tanh_fast(a2[k] + bias[k]));
                    *reinterpret_cast<uint4*>(J.out_bf[1] + (long long)b * H + col) =
                        make_uint4((uint32_t)q2[0] | ((uint32_t)q2[1] << 16), (uint32_t)q2[2] | ((uint32_t)q2[3] << 16),
                                   (uint32_t)q2[4] | ((uint32_t)q2[5] << 16), (uint32_t)q2[6] | ((uint32_t)q2[7] << 16));
                    if (J.out[1]) {
                        *reinterpret_cast<float4*>(J.out[1] + (long long)b * H + col) =
                            make_float4(bf16_to_float(q2[0]), bf16_to_float(q2[1]), bf16_to_float(q2[2]), bf16_to_float(q2[3]));
                        *reinterpret_cast<float4*>(J.out[1] + (long long)b * H + col + 4) =
                            make_float4(bf16_to_float(q2[4]), bf16_to_float(q2[5]), bf16_to_float(q2[6]), bf16_to_float(q2[7]));
                    }
                }
            }
        }
    } else if ((H & 3) == 0) {
        for (int col = lane * 4; col < H; col += 256) {
            float4 acc = *reinterpret_cast<const float4*>(b0 + col);
            int i = 0;
            if (!BF16) {                                // fp32: eight rows per round trip, their codes read 16 bytes at a time
                for (; i + 8 <= cnt; i += 8) {
                    const int4 ca = *reinterpret_cast<const int4*>(&rows[wid][i]), cb = *reinterpret_cast<const int4*>(&rows[wid][i + 4]);
                    const float4 w0 = load4(ca.x, col), w1 = load4(ca.y, col), w2 = load4(ca.z, col), w3 = load4(ca.w, col);
                    const float4 w4 = load4(cb.x, col), w5 = load4(cb.y, col), w6 = load4(cb.z, col), w7 = load4(cb.w, col);
                    acc.x = (((((((acc.x + w0.x) + w1.x) + w2.x) + w3.x) + w4.x) + w5.x) + w6.x) + w7.x;
                    acc.y = (((((((acc.y + w0.y) + w1.y) + w2.y) + w3.y) + w4.y) + w5.y) + w6.y) + w7.y;
                    acc.z = (((((((acc.z + w0.z) + w1.z) + w2.z) + w3.z) + w4.z) + w5.z) + w6.z) + w7.z;
                    acc.w = (((((((acc.w + w0.w) + w1.w) + w2.w) + w3.w) + w4.w) + w5.w) + w6.w) + w7.w;
                }
            }
            for (; i + 4 <= cnt; i += 4) {
                const float4 w0 = load4(rows[wid][i], col), w1 = load4(rows[wid][i + 1], col);
                const float4 w2 = load4(rows[wid][i + 2], col), w3 = load4(rows[wid][i + 3], col);
                acc.x = ((acc.x + w0.x) + w1.x) + w2.x + w3.x;
                acc.y = ((acc.y + w0.y) + w1.y) + w2.y + w3.y;
                acc.z = ((acc.z + w0.z) + w1.z) + w2.z + w3.z;
                acc.w = ((acc.w + w0.w) + w1.w) + w2.w + w3.w;
            }
            for (; i < cnt; ++i) {
                const float4 w = load4(rows[wid][i], col);
                acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w;
            }
            float4 t = make_float4(tanh_l0(acc.x), tanh_l0(acc.y), tanh_l0(acc.z), tanh_l0(acc.w));
            if (BF16) {
                const uint16_t q0 = bf16_bits(t.x), q1 = bf16_bits(t.y), q2 = bf16_bits(t.z), q3 = bf16_bits(t.w);
                *reinterpret_cast<uint2*>(out_bf + (long long)b * H + col) = make_uint2((uint32_t)q0 | ((uint32_t)q1 << 16), (uint32_t)q2 | ((uint32_t)q3 << 16));
                t = make_float4(bf16_to_float(q0), bf16_to_float(q1), bf16_to_float(q2), bf16_to_float(q3));
            } else if (out_bf) {            // fp32 net: a bf16 COPY beside the exact activations (screening operand, one hidden layer)
                const uint16_t q0 = bf16_bits(t.x), q1 = bf16_bits(t.y), q2 = bf16_bits(t.z), q3 = bf16_bits(t.w);
                *reinterpret_cast<uint2*>(out_bf + (J.out_bf_frag ? scr_afrag_index(b, col, H) : (long long)b * H + col)) = make_uint2((uint32_t)q0 | ((uint32_t)q1 << 16), (uint32_t)q2 | ((uint32_t)q3 << 16));
            }
            if (out) *reinterpret_cast<float4*>(out + (long long)b * H + col) = t;
            if (!BF16 && nd >= 0) {                       // the s' chain of the same sample, same net
                float4 a2;
                if (nd <= 8) {
                    a2 = acc;
                    for (int k = 0; k < nd; ++k) {        // wave-uniform
                        const int ro = dpair[wid][k][0], ri = dpair[wid][k][1];
                        if (ro >= 0) { const float4 w = load4(ro, col); a2.x -= w.x; a2.y -= w.y; a2.z -= w.z; a2.w -= w.w; }
                        if (ri >= 0) { const float4 w = load4(ri, col); a2.x += w.x; a2.y += w.y; a2.z += w.z; a2.w += w.w; }
                    }
                } else {
                    a2 = *reinterpret_cast<const float4*>(b0 + col);
                    for (int k = 0; k < cnt2; ++k) { const float4 w = load4(rows2[wid][k], col); a2.x += w.x; a2.y += w.y; a2.z += w.z; a2.w += w.w; }
                }
                const float4 t2 = make_float4(tanh_l0(a2.x), tanh_l0(a2.y), tanh_l0(a2.z), tanh_l0(a2.w));
                if (J.out_bf[1]) {
                    const uint16_t q0 = bf16_bits(t2.x), q1 = bf16_bits(t2.y), q2 = bf16_bits(t2.z), q3 = bf16_bits(t2.w);
                    *reinterpret_cast<uint2*>(J.out_bf[1] + (J.out_bf_frag ? scr_afrag_index(b, col, H) : (long long)b * H + col)) =
                        make_uint2((uint32_t)q0 | ((uint32_t)q1 << 16), (uint32_t)q2 | ((uint32_t)q3 << 16));
                }
                if (J.out[1]) *reinterpret_cast<float4*>(J.out[1] + (long long)b * H + col) = t2;
            }
        }
    } else {
        for (int col = lane; col < H; col += 64) {
            float acc = b0[col];
            for (int i = 0; i < cnt; ++i)
                acc += BF16 ? bf16_to_float(W0B[(long long)rows[wid][i] * H + col]) : W0T[(long long)rows[wid][i] * H + col];
            float t = tanh_l0(acc);
            if (BF16) { const uint16_t q = bf16_bits(t); out_bf[(long long)b * H + col] = q; t = bf16_to_float(q); }
            if (out) out[(long long)b * H + col] = t;
        }
    }
}

// Layer 0 of the SELECT chain with its sums kept between plies (one wave per game, fp32): a_1 = tanh(z_1), z_1 = b_0 + sum of the rows of
// the occupied squares.  derive != 0 and at most 8 squares differ from the board this game showed last time (a move changes two; a
// game that ended shows the start position: dozens): z_1 = kept z_1 - rows of what stood on the changed squares + rows of what
// stands there now, ascending square order — 4 row reads instead of ~25.  Otherwise the full sum in l0_forward_kernel's order.
// Either way z_1 and the board are kept for the next ply.  The kept sums are only valid while W0 / b0 do not change (the host
// drops them on every parameter update), i.e. across the plies of one update (bench --config 4: three of four plies).
__global__ __launch_bounds__(256) void l0_select_kernel(const uint32_t* __restrict__ boards, uint32_t* __restrict__ prev_boards,
                                                        const float* __restrict__ W0T, const float* __restrict__ b0, float* __restrict__ z1,
                                                        float* __restrict__ out, int n, int H, int derive) {
    __shared__ int rows[4][96];
    __shared__ int dpair[4][8][2];
    const int wid = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int b = (int)blockIdx.x * 4 + wid;
    if (b >= n) return;
    const uint32_t* bw = boards + (long long)b * kBoardWords;
    uint32_t* pw = prev_boards + (long long)b * kBoardWords;
    const int s0 = lane, s1 = 64 + lane;
    const uint32_t n0 = (bw[s0 >> 3] >> (4 * (s0 & 7))) & 15u;
    const uint32_t n1 = s1 < kSquares ? (bw[s1 >> 3] >> (4 * (s1 & 7))) & 15u : 0u;
    const unsigned long long below = (1ull << lane) - 1ull;
    int nd = 99, cnt = 0;
    if (derive) {
        const uint32_t p0 = (pw[s0 >> 3] >> (4 * (s0 & 7))) & 15u;
        const uint32_t p1 = s1 < kSquares ? (pw[s1 >> 3] >> (4 * (s1 & 7))) & 15u : 0u;
        const unsigned long long d0 = __ballot(p0 != n0), d1 = __ballot(p1 != n1);
        nd = __popcll(d0) + __popcll(d1);
        if (nd <= 8) {
            if (p0 != n0) { const int k = __popcll(d0 & below); dpair[wid][k][0] = p0 ? s0 * 14 + (int)p0 - 1 : -1; dpair[wid][k][1] = n0 ? s0 * 14 + (int)n0 - 1 : -1; }
            if (p1 != n1) { const int k = __popcll(d0) + __popcll(d1 & below); dpair[wid][k][0] = p1 ? s1 * 14 + (int)p1 - 1 : -1; dpair[wid][k][1] = n1 ? s1 * 14 + (int)n1 - 1 : -1; }
        }
    }
    if (nd > 8) {
        const unsigned long long m0 = __ballot(n0 != 0), m1 = __ballot(n1 != 0);
        const int c0 = __popcll(m0);
        if (n0) rows[wid][__popcll(m0 & below)] = s0 * 14 + (int)n0 - 1;
        if (n1) rows[wid][c0 + __popcll(m1 & below)] = s1 * 14 + (int)n1 - 1;
        cnt = c0 + __popcll(m1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int col = lane * 4; col < H; col += 256) {
        float4 acc;
        if (nd <= 8) {
            acc = *reinterpret_cast<const float4*>(z1 + (long long)b * H + col);
            for (int k = 0; k < nd; ++k) {            // wave-uniform
                const int ro = dpair[wid][k][0], ri = dpair[wid][k][1];
                if (ro >= 0) { const float4 w = *reinterpret_cast<const float4*>(W0T + (long long)ro * H + col); acc.x -= w.x; acc.y -= w.y; acc.z -= w.z; acc.w -= w.w; }
                if (ri >= 0) { const float4 w = *reinterpret_cast<const float4*>(W0T + (long long)ri * H + col); acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w; }
            }
        } else {
            acc = *reinterpret_cast<const float4*>(b0 + col);
            int i = 0;
            for (; i + 4 <= cnt; i += 4) {
                const float4 w0 = *reinterpret_cast<const float4*>(W0T + (long long)rows[wid][i] * H + col);
                const float4 w1 = *reinterpret_cast<const float4*>(W0T + (long long)rows[wid][i + 1] * H + col);
                const float4 w2 = *reinterpret_cast<const float4*>(W0T + (long long)rows[wid][i + 2] * H + col);
                const float4 w3 = *reinterpret_cast<const float4*>(W0T + (long long)rows[wid][i + 3] * H + col);
                acc.x = ((acc.x + w0.x) + w1.x) + w2.x + w3.x;
                acc.y = ((acc.y + w0.y) + w1.y) + w2.y + w3.y;
                acc.z = ((acc.z + w0.z) + w1.z) + w2.z + w3.z;
                acc.w = ((acc.w + w0.w) + w1.w) + w2.w + w3.w;
            }
            for (; i < cnt; ++i) {
                const float4 w = *reinterpret_cast<const float4*>(W0T + (long long)rows[wid][i] * H + col);
                acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w;
            }
        }
        *reinterpret_cast<float4*>(z1 + (long long)b * H + col) = acc;
        *reinterpret_cast<float4*>(out + (long long)b * H + col) = make_float4(tanh_l0(acc.x), tanh_l0(acc.y), tanh_l0(acc.z), tanh_l0(acc.w));
    }
    if (lane < kBoardWords) pw[lane] = bw[lane];
}

// block -> square: the squares of the start position first.  A block's run time grows with the number of samples that have a piece on
// its square (a home square of the back rank: nearly all of them; a square in the middle of the board: a few per cent) and the grid
// runs in two rounds of blocks (59 KB of LDS: two per CU) — the long blocks must be in the first round, or the kernel ends with a
// few of them running alone (device timestamps in the training loop: 36.5 us from first block start to last block end with the
// squares in board order, 28.1 us in this order; per block 1.5 us loads + 1.8 compaction + 6.9 streaming (mean) + 2.5 output)
__constant__ unsigned char kL0SquareOrder[90] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 19, 25, 27, 29, 31, 33, 35, 54, 56, 58, 60, 62, 64, 70, 81, 82, 83, 84, 85, 86, 87, 88, 89, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 21, 22, 23, 24, 26, 28, 30, 32, 34, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 55, 57, 59, 61, 63, 65, 66, 67, 68, 69, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80};

// Layer-0 weight gradient gW0^T[(sq,piece)][:] = sum over the samples that have `piece` on `sq` of delta_0[sample][:]
// (the one-hot input of chessai.cpp:268-289 transposed).  A dense one-hot GEMM would spend 2*1260*H FLOP per sample on
// zeros; here block (sq, chunk) compacts the samples of its chunk that occupy `sq` (ascending sample order, so the sums
// are bitwise reproducible), then streams their delta rows (1 KB each, L2-resident) into 14 LDS accumulator rows.
// partial[chunk][sq*14 + piece-1][H]; the ordered chunk reduction is the usual reduce_slabs_kernel.
// Wide layers: blockIdx.z picks a slab of HS columns (grid z = H / HS) — at H = 512 one block per (square, chunk) could keep only two
// accumulator sets in LDS (two of its four waves streaming, 265 us at 16384 x 512); two 256-column slabs are two blocks of the
// H = 256 shape each (four sets, 2 blocks per CU).  Every slab compacts the chunk for itself (cheap) and streams its own columns.
// (block body: `lin` = linear id of the block inside its column slab, nch = chunks, slab = column slab — gridDim (90, nch, H / HS) in
// l0_grad_kernel; td_tail_kernel hands the same triple to its layer-0 blocks)
__device__ __forceinline__ void l0_grad_block(const uint32_t* __restrict__ gboards, const float* __restrict__ delta0_all,
                                              int n, int Hfull, int HS, int chunk, int nsets, float* __restrict__ partial_all,
                                              int lin, int nch, int slab, float* __restrict__ smem) {
    const int H = HS;                                   // width this block works on; rows of delta0 / partial are Hfull apart
    const float* __restrict__ delta0 = delta0_all + (long long)slab * HS;
    float* __restrict__ partial = partial_all + (long long)slab * HS;
    float* acc = smem;                                  // [nsets][14][H]
    uint16_t* list = reinterpret_cast<uint16_t*>(smem + (long long)nsets * 14 * H);   // [chunk] (b_local | piece << 11)
    // workgroups go to the 8 XCDs round-robin in linear order: chunk = linear id mod nchunks keeps all 90 square-blocks of a chunk
    // (they stream the same 1 MB of delta rows, each up to 32 times) behind one XCD's L2 when there are 8 chunks
    const int s = kL0SquareOrder[lin / nch];
    const int c0 = (lin % nch) * chunk;
    const int c1 = min(n, c0 + chunk);
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // the chunk's piece codes first (up to 8 independent loads per thread in flight), accumulator zeroing under their latency
    constexpr int kMaxIters = 8;                         // chunk <= 2048
    uint32_t nibs = 0;
#pragma unroll
    for (int it = 0; it < kMaxIters; ++it) {
        const int b = c0 + it * 256 + tid;
        // (unconditional, clamped: a predicated load compiles to a branch with its own wait, one memory round trip per load)
        uint32_t nib = (gboards[(long long)min(b, c1 - 1) * kBoardWords + (s >> 3)] >> (4 * (s & 7))) & 15u;
        if (b >= c1) nib = 0;
        nibs |= nib << (4 * it);
    }
    if (((nsets * 14 * H) & 3) == 0) {
        float4* a4 = reinterpret_cast<float4*>(acc);
        for (int i = tid; i < nsets * 14 * H / 4; i += 256) a4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (int i = tid; i < nsets * 14 * H; i += 256) acc[i] = 0.f;
    }
    // phase 1: ordered compaction of the occupied samples with two barriers in all: the per-wave counts of every round are
    // published first, the offsets are then prefix sums over (round, wave) that every thread computes for itself
    __shared__ int wc[kMaxIters][4];
    const int iters = (c1 - c0 + 255) / 256;
#pragma unroll
    for (int it = 0; it < kMaxIters; ++it) {
        const uint32_t nib = (nibs >> (4 * it)) & 15u;
        const unsigned long long m = __ballot(nib != 0);
        if (lane == 0) wc[it][wid] = __popcll(m);
    }
    __syncthreads();
    int off = 0;
#pragma unroll
    for (int it = 0; it < kMaxIters; ++it) {
        if (it < iters) {
            const uint32_t nib = (nibs >> (4 * it)) & 15u;
            const unsigned long long m = __ballot(nib != 0);
            int o = off;
            for (int w = 0; w < wid; ++w) o += wc[it][w];
            if (nib) list[o + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)((it * 256 + tid) | (nib << 11));
        }
        off += wc[it][0] + wc[it][1] + wc[it][2] + wc[it][3];
    }
    __syncthreads();
    const int cnt = off;
    // phase 2: wave w streams list entries w, w+nsets, ... (whole 1-KB rows as float4 per lane, 16 rows in flight per
    // wave) into ITS OWN accumulator set; the sets are added in fixed order afterwards => bitwise reproducible
    if (wid < nsets && (H & 3) == 0) {
        float* my = acc + (long long)wid * 14 * H;
        for (int col = lane * 4; col < H; col += 256) {
            // rows of the same piece arrive in runs (a square mostly holds one or two piece kinds): a run is summed in
            // registers and touches its LDS accumulator once, instead of one read-modify-write round trip per row
            int cur = 0;                                   // piece code of the open run (wave-uniform), 0 = none
            float rx = 0.f, ry = 0.f, rz = 0.f, rw = 0.f;
            auto flush = [&]() {
                if (cur != 0) {
                    float4* a = reinterpret_cast<float4*>(my + (cur - 1) * H + col);
                    float4 t = *a;
                    t.x += rx; t.y += ry; t.z += rz; t.w += rw;
                    *a = t;
                }
            };
            int i = wid;
            for (; i + 15 * nsets < cnt; i += 16 * nsets) {
                int e[16];
                float4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    e[u] = list[i + u * nsets];
                    const float4 x = *reinterpret_cast<const float4*>(delta0 + (long long)(c0 + (e[u] & 2047)) * Hfull + col);
                    v[u].x = x.x; v[u].y = x.y; v[u].z = x.z; v[u].w = x.w;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int p = __builtin_amdgcn_readfirstlane(e[u] >> 11);
                    if (p != cur) { flush(); cur = p; rx = v[u].x; ry = v[u].y; rz = v[u].z; rw = v[u].w; }
                    else { rx += v[u].x; ry += v[u].y; rz += v[u].z; rw += v[u].w; }
                }
            }
            for (; i < cnt; i += nsets) {
                const int e1 = list[i];
                const float4 x = *reinterpret_cast<const float4*>(delta0 + (long long)(c0 + (e1 & 2047)) * Hfull + col);
                const int p = __builtin_amdgcn_readfirstlane(e1 >> 11);
                if (p != cur) { flush(); cur = p; rx = x.x; ry = x.y; rz = x.z; rw = x.w; }
                else { rx += x.x; ry += x.y; rz += x.z; rw += x.w; }
            }
            flush();
        }
    } else if ((H & 3) != 0 && wid == 0) {
        for (int col = lane; col < H; col += 64)
            for (int i = 0; i < cnt; ++i) {
                const int e1 = list[i];
                acc[((e1 >> 11) - 1) * H + col] += delta0[(long long)(c0 + (e1 & 2047)) * Hfull + col];
            }
    }
    __syncthreads();
    float* out = partial + ((long long)(lin % nch) * kStateSize + (long long)s * 14) * Hfull;
    const int used = (H & 3) == 0 ? nsets : 1;
    for (int i = tid; i < 14 * H; i += 256) {
        float t = acc[i];
        for (int w = 1; w < used; ++w) t += acc[(long long)w * 14 * H + i];
        out[(long long)(i / H) * Hfull + (i % H)] = t;
    }
}
__global__ __launch_bounds__(256) void l0_grad_kernel(const uint32_t* __restrict__ gboards, const float* __restrict__ delta0_all,
                                                      int n, int Hfull, int HS, int chunk, int nsets, float* __restrict__ partial_all) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    l0_grad_block(gboards, delta0_all, n, Hfull, HS, chunk, nsets, partial_all, (int)(blockIdx.x + gridDim.x * blockIdx.y), (int)gridDim.y,
                  (int)blockIdx.z, smem);
}


}  // namespace xq
