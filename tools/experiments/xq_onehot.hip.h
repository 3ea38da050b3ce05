// EXPERIMENT (round 2), NOT part of libxqhip: measured and rejected — see DESIGN.md §5 "Tried on the layer-0 gradient kernel" and
// profiles/r02_e_onehot_mfma_experiment_bench_all_kernels.json.  Correct (all TD parity tests green, bit-reproducible), but slower
// than the segmented sums it was meant to replace: layer 0 87.7 us vs 52 us, output layer 37.8 us vs 27 us at 8192 x 256.  The
// dense form does 3 x 2 x 1440 x H x B flop (18 GFLOP) where the segmented sums touch only the ~32 occupied rows per sample, it
// re-reads delta_0 once per 64-row block (23 x 8 MB through L2), and with 184 blocks of 4 waves nothing hides the load latency.
// To build it again: copy next to xq_gemm.hip.h, include from xq_dqn.hip, call onehot_grad_kernel<OH_L0 / OH_OUT> from l0_gradient /
// side_gradients (grid: ((90 + 3) / 4 | 2, nchunks, (H / 32 + 7) / 8), chunk a multiple of OH_SUB).
//
// xq_onehot.hip.h — gradients whose left operand is a ONE-HOT matrix, on the bf16 matrix pipe without losing a bit.
//
// Two gradients of the TD step have the form  G[(g, c)][:] = sum over the samples k with code_g(k) == c of x_k[:] :
//   layer 0      g = square (90), c = piece code 1..14, x_k = delta_0[k][:]            (the one-hot input of chessai.cpp:268-289
//                transposed times delta_0 — updateWeightsBiasesKernel dqn.cu:310-319 for the first layer, batched)
//   output layer g = action.to >> 4 (6), c = action.to & 15, x_k = delta_k * a_last[k][:]  (only the row of the played action
//                has a non-zero output delta in a TD step, chessai.cpp:122-128)
// i.e. G = onehot^T * X with a 0/1 matrix.  0 and 1 are exact in bf16, and an fp32 value splits EXACTLY into three bf16 terms
// (x = hi + mid + lo: 8 + 8 + 8 significant bits, each remainder computed exactly in fp32), so
//     G = onehot^T * hi + onehot^T * mid + onehot^T * lo
// on v_mfma_f32_32x32x16_bf16 (fp32 accumulation, every product exact) is an fp32 sum of the very same terms in a fixed order:
// exact fp32 semantics at 16/3 times the fp32-MFMA rate, and no segmented-sum machinery (compaction lists, LDS accumulators,
// two block rounds) at all.  The one-hot tile never exists in HBM: it is expanded in LDS from the packed codes.
//
// Block = 4 waves: 64 rows (4 groups x 16 codes; codes that do not exist are never matched) x up to 256 columns (each wave two
// 32-column tiles) x one chunk of samples; partial[chunk] slabs are summed in fixed order by the caller (deterministic).
#pragma once

#include "xq_gemm.hip.h"

namespace xq {

enum { OH_L0 = 0, OH_OUT = 1 };
constexpr int OH_SUB = 256;                 // samples expanded into the LDS one-hot image at a time
constexpr int OH_LD = OH_SUB + 8;           // row stride of the image in bf16: 528 B, conflict-free for ds_read_b128

struct OneHotArgs {
    const uint32_t* gboards;    // OH_L0: [n][12] packed boards of the minibatch
    const int32_t* act;         // OH_OUT: [n] action.to, -1 = no gradient
    const float* X; long long ldx;          // [n][H]
    const float* scale;         // OH_OUT: [n] the sample's output delta (x_k = scale[k] * X[k][:]); OH_L0: nullptr
    int n, H, chunk;
    float* partial;             // [nchunks][slab_stride]: OH_L0 rows (sq*14 + code-1), OH_OUT rows 0..95 then 96 bias sums at 96*H
    long long slab_stride;
};

// 8 fp32 values -> three bf16x8 fragments with hi + mid + lo == x exactly
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        const float r1 = x[j] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        hi[j] = h; mid[j] = m; lo[j] = (__bf16)r2;
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void onehot_grad_kernel(OneHotArgs P) {
    __shared__ __attribute__((aligned(16))) uint16_t val[4][OH_SUB];        // code of every sample of the sub-chunk, per group
    __shared__ __attribute__((aligned(16))) uint16_t img[64 * OH_LD];       // one-hot image [row][sample] in bf16
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int g0 = (int)blockIdx.x * 4;                                     // first group of the block
    const int c0 = (int)blockIdx.y * P.chunk, c1 = min(P.n, c0 + P.chunk);
    const int ntiles = min(8, P.H / 32 - (int)blockIdx.z * 8);              // 32-column tiles of this block
    const int nt0 = (int)blockIdx.z * 8;
    // wave w owns tiles w and w + 4 of the block (when they exist)
    const bool have0 = wid < ntiles, have1 = wid + 4 < ntiles;
    const int n0a = (nt0 + wid) * 32, n0b = (nt0 + wid + 4) * 32;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    float bias_acc[2] = {0.f, 0.f};                                         // OH_OUT, wave 0: sum of scale over the matching samples

    for (int s0 = c0; s0 < c1; s0 += OH_SUB) {
        // 1. codes of the sub-chunk's samples for the block's four groups (0xFFFF = matches nothing)
        {
            const int k = s0 + tid;
            uint16_t v[4] = {0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF};
            if (k < c1) {
                if (KIND == OH_L0) {
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int sq = g0 + gq;
                        if (sq < kSquares) v[gq] = (uint16_t)((P.gboards[(long long)k * kBoardWords + (sq >> 3)] >> (4 * (sq & 7))) & 15u);
                    }
                } else {
                    const int a = P.act[k];
                    if (a >= 0 && a < 96) {
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) if ((a >> 4) == g0 + gq) v[gq] = (uint16_t)(a & 15);
                    }
                }
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) val[gq][tid] = v[gq];
        }
        __syncthreads();
        // 2. expand to the bf16 one-hot image: row (group gq, code cv) x sample.  Thread t fills a quarter of row t >> 2, two samples
        //    per dword: d = codes ^ cv (per 16-bit half), half != 0  <=>  bit 15 of ((d + 0x7FFF) | d), 1.0bf16 = 0x3F80
        {
            const int row = tid >> 2, part = tid & 3;
            const uint32_t cv = (uint32_t)((row & 15) + (KIND == OH_L0 ? 1 : 0));
            const uint32_t cv2 = cv | (cv << 16);
            const uint32_t* src = reinterpret_cast<const uint32_t*>(&val[row >> 4][part * 64]);
            uint32_t* dst = reinterpret_cast<uint32_t*>(&img[row * OH_LD + part * 64]);
#pragma unroll
            for (int q = 0; q < 32; q += 4) {
                const uint4 x = *reinterpret_cast<const uint4*>(src + q);
                const uint32_t in[4] = {x.x, x.y, x.z, x.w};
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t dd = in[e] ^ cv2;
                    const uint32_t ne = (((dd & 0x7FFF7FFFu) + 0x7FFF7FFFu) | dd) >> 15 & 0x00010001u;     // 1 per half that differs
                    o[e] = 0x3F803F80u - ne * 0x3F80u;
                }
                *reinterpret_cast<uint4*>(dst + q) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
        __syncthreads();
        // 3. sixteen k-steps of 16 samples: A fragments from the image, B fragments straight from global memory (each wave reads
        //    only its own columns: 32 lanes x 4 B contiguous per sample row), split into three exact bf16 terms
        const int ksteps = (min(c1, s0 + OH_SUB) - s0 + 15) / 16;
        // software pipeline: the loads of step ks + 1 are in flight under the splits and MFMAs of step ks
        float xs[8], xa[8], xb[8], nxs[8], nxa[8], nxb[8];
        auto load_step = [&](int ks, float (&sv)[8], float (&va)[8], float (&vb)[8]) {
            const int kb = s0 + ks * 16 + 8 * hh;                            // this lane's first sample of the step
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long long row = min(kb + j, P.n - 1);                  // rows >= n: matched by nothing, any finite value will do
                sv[j] = (KIND == OH_OUT) ? P.scale[row] : 1.f;
                va[j] = have0 ? P.X[row * P.ldx + n0a + r] : 0.f;
                vb[j] = have1 ? P.X[row * P.ldx + n0b + r] : 0.f;
            }
        };
        load_step(0, xs, xa, xb);
        for (int ks = 0; ks < ksteps; ++ks) {
            if (ks + 1 < ksteps) load_step(ks + 1, nxs, nxa, nxb);
            const uint4 a0 = *reinterpret_cast<const uint4*>(&img[r * OH_LD + ks * 16 + 8 * hh]);
            const uint4 a1 = *reinterpret_cast<const uint4*>(&img[(32 + r) * OH_LD + ks * 16 + 8 * hh]);
            const bf16x8 fa0 = __builtin_bit_cast(bf16x8, a0), fa1 = __builtin_bit_cast(bf16x8, a1);
            if (KIND == OH_OUT && wid == 0) {
                const uint32_t w0[4] = {a0.x, a0.y, a0.z, a0.w}, w1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bias_acc[0] += ((w0[e] & 0xFFFFu) ? xs[2 * e] : 0.f) + ((w0[e] >> 16) ? xs[2 * e + 1] : 0.f);
                    bias_acc[1] += ((w1[e] & 0xFFFFu) ? xs[2 * e] : 0.f) + ((w1[e] >> 16) ? xs[2 * e + 1] : 0.f);
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (!(t == 0 ? have0 : have1)) continue;
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = t == 0 ? xa[j] : xb[j];
                    x[j] = (KIND == OH_OUT) ? v * xs[j] : v;
                }
                bf16x8 hi, mid, lo;
                split3(x, hi, mid, lo);
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, hi, acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, hi, acc[1][t], 0, 0, 0);
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, mid, acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, mid, acc[1][t], 0, 0, 0);
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, lo, acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, lo, acc[1][t], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { xs[j] = nxs[j]; xa[j] = nxa[j]; xb[j] = nxb[j]; }
        }
        __syncthreads();                                                    // the image is rewritten by the next sub-chunk
    }

    // epilogue: 32x32 accumulator map col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); rows of codes that do
    // not exist (layer 0: 15, 16; squares >= 90; output rows >= 96) are dropped
    float* out = P.partial + (long long)blockIdx.y * P.slab_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = i * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh;
            const int grp = g0 + (m >> 4), code = m & 15;
            long long row;
            if (KIND == OH_L0) { if (grp >= kSquares || code >= 14) continue; row = (long long)grp * 14 + code; }
            else { row = (long long)grp * 16 + code; if (row >= 96) continue; }
            if (have0) out[row * P.H + n0a + r] = acc[i][0][q];
            if (have1) out[row * P.H + n0b + r] = acc[i][1][q];
        }
    if (KIND == OH_OUT && wid == 0 && blockIdx.z == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float b = bias_acc[i] + __shfl_xor(bias_acc[i], 32, 64);   // the two k halves of the fragment
            const int row = (g0 + (i * 32 + r) / 16) * 16 + (r & 15);
            if (hh == 0 && row < 96) out[96LL * P.H + row] = b;
        }
    }
}

}  // namespace xq
