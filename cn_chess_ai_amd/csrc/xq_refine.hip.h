// xq_refine.hip.h — pass 2 of the exact screening of max_a' Q(s',a') (chessai.cpp:126): candidate selection under the bf16 error bound and the
// fp32 re-evaluation of the candidates, with the TD target / delta arithmetic riding in the same blocks (qmax_refine_kernel, qmax_refine2_kernel)
// (kernel half of xq_dqn.hip, split out in round 5; included by xq_dqn.hip only, inside namespace xq)
#pragma once

namespace xq {

// ---- exact screening of z_max[b] = max_j (W_out[j] . a[b] + b_out[j])  (xq_dqn_set_qmax_mode(XQ_QMAX_SCREENED)) ------
// The fp32 column-max GEMM computes 8100 outputs per sample to keep one.  Screening computes all of them once on the bf16 matrix
// pipe (16x the fp32 MFMA rate), with a rigorous bound on what bf16 operands can hide, and re-evaluates in fp32 only the few
// outputs that could still be the maximum.  With u = 2^-8 (round-to-nearest bf16, 8-bit significand):
//   z~_j = fl32(sum_k bf16(W_jk) bf16(a_k)) + b_j
//   |z~_j - z_j| <= (2u + u^2) sum_k |W_jk a_k|  [operand rounding]  +  2K 2^-23 sum_k |W_jk a_k|  [fp32 accumulation of the exact
//                   products inside and between the MFMAs, K <= 1024]
//               <= B := kScreenEps ||a||_2 max_j ||W_j||_2           [Cauchy-Schwarz; kScreenEps = 2^-7 * 1.0625 >= 2^-7 + 2^-16 + 2^-12]
//   j* = argmax z_j  =>  z~_j* >= z_j* - B >= z_J - B >= z~_J - 2B with J = argmax z~: every output whose screened value is within
//   2B of the screened maximum is a candidate and j* is among them.  The threshold used is m~ - 2B (1 + 2^-5) - 2^-16 (|m~| + 2B):
//   the 2^-5 absorbs the rounding of the fp32 re-evaluation itself (<= K 2^-24 sum|W a| <= 2^-14/kScreenEps B per value), so the
//   result is the maximum over ALL outputs of the fp32-evaluated value, not only a value close to it; the last term covers the
//   5-bit position tag (<= 2^-18 relative at both ends).
// Pass 1 (gemm_colmax_persistent_kernel<.., DT_BF16, CM_TOP2>) leaves, per sample and per 32-row lane group, the largest screened
// value (tagged with its row) and the second largest.  Pass 2 (qmax_refine_kernel): threshold per sample, then one fp32 dot per
// candidate group whose second value is below the threshold (the usual case), 32 dots for a group with two values above it.
constexpr float kScreenEps = 0.0078125f * 1.0625f;
// the bias travels inside the accumulation chain (C operand of a tile's first MFMA, xq_screen.hip.h) or is added behind it (the
// older kernel): either way it adds at most (K + 1) 2^-24 |b_j| of rounding to the screened value and the same to the fp32
// re-evaluation; 2^-9 max_j |b_j| covers both for K <= 1024, including the 2^-5 share of B the threshold reserves for the latter
constexpr float kScreenBiasEps = 0.001953125f;
enum { kScreenCheckEvery = 32, kScreenHoldSteps = 512 };
constexpr double kScreenMaxPairs = 24.0, kScreenMaxWhole = 1.0;     // candidate groups / whole groups per sample above which the
                                                                    // fp32 re-evaluation costs more than the product it replaces

__device__ __forceinline__ int float_order_key(float f) {             // signed-int order == float order (no NaNs here)
    const int b = __builtin_bit_cast(int, f);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float float_from_key(int k) { return __builtin_bit_cast(float, k ^ ((k >> 31) & 0x7fffffff)); }
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Pass 2.  Block = 32 consecutive samples x 8 group phases (thread = (sample, phase); P1 / P2 are [G][n], group-major, so a
// half-wave reads 128 contiguous bytes per group).  G <= 512.  Every thread keeps its G/8 screened values in registers between the
// maximum and the candidate scan.  Dynamic LDS: G * 32 single-row candidates (32 bits) + G * 32 whole-group candidates (16 bits) —
// one entry per (sample, group) pair at most, so neither list can overflow.  The fp32 dots run a quarter-wave per row: singles two
// per quarter and round (32 per round), a whole group as ONE round of the block (its 32 rows over the 16 quarters).
// wmax_next: zeroed for the next step's shadow pass.  stats: [2] += candidate pairs, [3] += pairs recomputed as whole groups.
enum { kRefineSamples = 32, kRefineMaxPerThread = 64 };

// one fp32 dot per quarter-wave (16 lanes x float4 x K/64 passes), all loads of both operands issued before the first fma
template <int KFIX>
__device__ __forceinline__ float quarter_dot(const float* __restrict__ ap, const float* __restrict__ wp, int K, int ql) {
    float acc = 0.f;
    if (KFIX > 0) {
        constexpr int NT = KFIX > 0 ? KFIX / 64 : 1;
        float4 x[NT], w[NT];
#pragma unroll
        for (int t = 0; t < KFIX / 64; ++t) {
            x[t] = *reinterpret_cast<const float4*>(ap + t * 64 + ql * 4);
            w[t] = *reinterpret_cast<const float4*>(wp + t * 64 + ql * 4);
        }
#pragma unroll
        for (int t = 0; t < KFIX / 64; ++t) {
            acc = fmaf(x[t].x, w[t].x, acc); acc = fmaf(x[t].y, w[t].y, acc);
            acc = fmaf(x[t].z, w[t].z, acc); acc = fmaf(x[t].w, w[t].w, acc);
        }
    } else {
        for (int k = ql * 4; k < K; k += 64) {
            const float4 x = *reinterpret_cast<const float4*>(ap + k);
            const float4 w = *reinterpret_cast<const float4*>(wp + k);
            acc = fmaf(x.x, w.x, acc); acc = fmaf(x.y, w.y, acc); acc = fmaf(x.z, w.z, acc); acc = fmaf(x.w, w.w, acc);
        }
    }
    return acc;
}
__device__ __forceinline__ float quarter_sum(float v) {
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ int screen_row(int g, int code) {       // inverse of the CM_TOP2 position code
    const int q = code & 15;
    return (g >> 2) * 128 + ((g >> 1) & 1) * 64 + 4 * (g & 1) + (code >> 4) * 32 + (q & 3) + 8 * (q >> 2);
}

template <int KFIX, int NPT>                 // NPT = screened values per thread = ceil(G / 8), unrolled (32 for 8100 outputs)
__global__ __launch_bounds__(256) void qmax_refine_kernel(const float* __restrict__ P1, const float* __restrict__ P2, int G, int n, long long ldp,
                                                          const float* __restrict__ a_last, int K, const float* __restrict__ W,
                                                          const float* __restrict__ bias, int NO, unsigned* __restrict__ wm, int parity,
                                                          float* __restrict__ zmax, unsigned long long* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cand[];            // [G * 32]: sample | row << 5
    uint16_t* wlist = reinterpret_cast<uint16_t*>(cand + (size_t)G * kRefineSamples);    // [G * 32]: sample | group << 5
    __shared__ float sv[8][32];
    __shared__ float na[32], thr[32];
    __shared__ int best[32];
    __shared__ int cnt, nexp;
    const int tid = (int)threadIdx.x;
    const int sl = tid & 31, phase = tid >> 5;
    const int ql = tid & 15, quarter = tid >> 4;
    const int b0 = (int)blockIdx.x * kRefineSamples;
    const int b = b0 + sl;
    const bool ok = b < n;
    if (tid == 0) { cnt = 0; nexp = 0; if (blockIdx.x == 0) { wm[parity ^ 1] = 0u; wm[2 + (parity ^ 1)] = 0u; } }   // next step's slots
    unsigned long long st_pairs = 0, st_whole = 0;   // candidate counters: [block][2] running totals, one writer per slot (stream order)
    if (tid == 0) { st_pairs = stats[2 * blockIdx.x]; st_whole = stats[2 * blockIdx.x + 1]; }
    if (tid < 32) best[tid] = (int)0x80000000;
    // this thread's screened values: groups phase, phase + 8, ...
    // (unconditional, clamped loads: a predicate per load compiles to a branch per load)
    float v[NPT], v2[NPT];                          // the second values too, up front: one memory round trip less
    const int bc = min(b, n - 1);
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
        const int g = min(phase + 8 * u, G - 1);
        v[u] = P1[(long long)g * ldp + bc];
        v2[u] = P2[(long long)g * ldp + bc];
    }
#pragma unroll
    for (int u = 0; u < NPT; ++u)
        if (!ok || phase + 8 * u >= G) { v[u] = kColmaxPadBias; v2[u] = kColmaxPadBias; }
    // ||a_b||^2: a quarter-wave per sample, two samples per quarter
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = quarter + 16 * r;
        const float* ap = a_last + (long long)min(b0 + i, n - 1) * K;
        const float ss = quarter_sum(quarter_dot<KFIX>(ap, ap, K, ql));
        if (ql == 0) na[i] = ss;
    }
    float m = kColmaxPadBias;
#pragma unroll
    for (int u = 0; u < NPT; ++u) m = fmaxf(m, v[u]);
    sv[phase][sl] = m;
    __syncthreads();
    if (tid < 32) {
        m = sv[0][sl];
#pragma unroll
        for (int p = 1; p < 8; ++p) m = fmaxf(m, sv[p][sl]);
        const float wmx = fmaxf(__builtin_bit_cast(float, wm[parity]), __builtin_bit_cast(float, wm[4]));
        const float bmx = fmaxf(__builtin_bit_cast(float, wm[2 + parity]), __builtin_bit_cast(float, wm[5]));
        const float B = kScreenEps * sqrtf(na[sl]) * wmx + kScreenBiasEps * bmx;
        float t0 = m - 2.f * B * 1.03125f - 1.52587890625e-05f * (fabsf(m) + 2.f * B);
        if (!(t0 == t0)) t0 = -__builtin_inff();      // a non-finite norm (diverged net): every group is a candidate, like the full product
        thr[sl] = b0 + sl < n ? t0 : __builtin_inff();   // no candidates past n
    }
    __syncthreads();
    {
        const float t = thr[sl];
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            const int g = phase + 8 * u;
            if (v[u] >= t) {                         // padding values are far below every threshold
                if (v2[u] >= t) wlist[atomicAdd(&nexp, 1)] = (uint16_t)(sl | (g << 5));
                else cand[atomicAdd(&cnt, 1)] = (uint32_t)sl | ((uint32_t)screen_row(g, (int)(__builtin_bit_cast(uint32_t, v[u]) & 31u)) << 5);
            }
        }
    }
    __syncthreads();
    // fp32 dots of the candidates (the maximum does not depend on the order they are visited in)
    const int singles = cnt, wholes = nexp;
    for (int e0 = 0; e0 < singles; e0 += 64) {                   // 16 quarters x 4 rows per round, all loads of a round in flight
        float z[4];
        int s2[4], row[4];
        bool live[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = e0 + quarter + 16 * r;
            live[r] = e < singles;
            const uint32_t ent = cand[live[r] ? e : 0];
            s2[r] = (int)(ent & 31u);
            row[r] = min((int)(ent >> 5), NO - 1);
            z[r] = quarter_dot<KFIX>(a_last + (long long)(b0 + s2[r]) * K, W + (long long)row[r] * K, K, ql);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            z[r] = quarter_sum(z[r]) + bias[row[r]];
            if (live[r] && ql == 0 && z[r] == z[r]) atomicMax(&best[s2[r]], float_order_key(z[r]));   // (a NaN output never wins: fmaxf semantics)
        }
    }
    for (int e = 0; e < wholes; ++e) {                           // a whole group: its 32 rows over the 16 quarters, one round
        const int ent = wlist[e];
        const int s2 = ent & 31, g = ent >> 5;
        float zb = kColmaxPadBias;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = screen_row(g, quarter + 16 * r);
            const int rc = min(row, NO - 1);
            const float z = quarter_sum(quarter_dot<KFIX>(a_last + (long long)(b0 + s2) * K, W + (long long)rc * K, K, ql)) + bias[rc];
            if (row < NO) zb = fmaxf(zb, z);
        }
        if (ql == 0) atomicMax(&best[s2], float_order_key(zb));
    }
    __syncthreads();
    if (tid < 32 && ok) zmax[b] = float_from_key(best[sl]);
    if (tid == 0) {                                  // this block's own running totals (512 same-address atomics per launch cost the
        stats[2 * blockIdx.x] = st_pairs + (unsigned long long)(singles + wholes);      // step 5 us)
        stats[2 * blockIdx.x + 1] = st_whole + (unsigned long long)wholes;
    }
}

// Pass 2 behind screen_top2_kernel (xq_screen.hip.h), which also leaves R[range][sample] = the largest P1 of the sample over the groups
// of one row range and na[sample] = ||bf16(a)||^2: the threshold needs 16 values per sample instead of 254, and only the groups of the
// ranges that reach it are looked at (usually one: the kernel reads ~1/10 of the partial arrays, and no activation rows for the norm).
// ||a|| <= ||bf16(a)|| (1 + 2^-7): inside the slack of kScreenEps (2^-7 * 0.0625 - 2^-12 - 2^-16 = 2.3e-4 against 6.1e-5 + the fp32
// rounding of the sum of squares).  Same block shape, candidate lists and fp32 re-evaluation as qmax_refine_kernel.
// TD: the work of td_delta_kernel (its fp32, 256-wide fast path: same loads, same arithmetic, same bits) for the block's 32 samples —
// wave w owns samples 8w .. 8w+7.  Everything that does not depend on the maximum (action, reward, Q(s,a) = the dot of two 1-KB rows) is
// requested at the very top and lands under the refine phases; once the block's maxima exist the targets, the scalar deltas and the top
// hidden deltas follow.  One launch and ~10 us of exposed latency chain fewer on the step's critical stream.
struct TdFused {
    SlotSrc src;
    const int32_t* action_to; const float* reward; const uint8_t* done;
    const float* a_s;              // last hidden activations of s on the online net [n][256]
    const float* w_out; const float* b_out;
    const float* view; long long view_ld; int view_kmax;
    float gamma;
    float* dtop; float* dsc; int32_t* act; float* qsa; float* yv; float* lossv;
};
// 32-bit words of the candidate list in qmax_refine2_kernel's dynamic LDS: G * 32 entries, and never less than the 32 KB its staged pass
// parks a group's rows of W in
__host__ __device__ inline size_t refine_cand_words(int G) { const size_t w = (size_t)G * 32; return w < 8192 ? 8192 : w; }
// bytes of the whole-group list, and of the extra dynamic LDS of the staged pass (activation rows + the rows of two more groups), K = 256
__host__ __device__ inline size_t refine_wlist_bytes(int G) { return ((size_t)G * 32 * sizeof(uint16_t) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t refine_stage_bytes() { return (size_t)3 * 32 * 256 * sizeof(float); }
template <int KFIX, bool TD = false>
__global__ __launch_bounds__(256) void qmax_refine2_kernel(const float* __restrict__ R, int ranges, int gpr /* groups per range */,
                                                           const float* __restrict__ P1, const float* __restrict__ P2, int G, int n, long long ldp,
                                                           const float* __restrict__ na_all, const float* __restrict__ a_last, int K,
                                                           const float* __restrict__ W, const float* __restrict__ bias, int NO,
                                                           unsigned* __restrict__ wm, int parity, float* __restrict__ zmax,
                                                           unsigned long long* __restrict__ stats, const TdFused T, int whole_mode) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cand[];            // [G * 32]: sample | row << 5
    uint16_t* wlist = reinterpret_cast<uint16_t*>(cand + refine_cand_words(G));          // [G * 32]: sample | group << 5
    __shared__ float sv[8][32];
    __shared__ float thr[32];
    __shared__ int best[32];
    __shared__ int cnt, nexp;
    __shared__ unsigned gbits[256];     // per group: the samples of the block that ask for it as a whole group (G <= 256: see stage_ok)
    __shared__ unsigned long long gpop[4];      // per wave: the groups tid that >= 8 samples ask for
    const int tid = (int)threadIdx.x;
    const int sl = tid & 31, phase = tid >> 5;
    const int ql = tid & 15, quarter = tid >> 4;
    const int b0 = (int)blockIdx.x * kRefineSamples;
    const int b = b0 + sl;
    const bool ok = b < n;
    const int bc = min(b, n - 1);
    if (tid == 0) { cnt = 0; nexp = 0; if (blockIdx.x == 0) { wm[parity ^ 1] = 0u; wm[2 + (parity ^ 1)] = 0u; } }   // next step's slots
    unsigned long long st_pairs = 0, st_whole = 0;   // candidate counters: [block][2] running totals, one writer per slot (stream order)
    if (tid == 0) { st_pairs = stats[2 * blockIdx.x]; st_whole = stats[2 * blockIdx.x + 1]; }
    if (tid < 32) best[tid] = (int)0x80000000;
    const bool stage_ok = KFIX > 0 && G <= 256;
    gbits[tid] = 0u;                    // (ordered before the scan below by the barriers in between)
    // TD: lanes 0..7 of each wave hold action / reward / done / output bias of the wave's eight samples; zq[i] = Q(s,a) before the tanh
    const int td_lane = tid & 63, td_w = tid >> 6;
    int td_a = -1; float td_r = 0.f, td_bo = 0.f; bool td_dn = false;
    float zq[8];
    if (TD) {
        static_assert(!TD || KFIX == 256, "fused TD delta: 256-wide last hidden layer");
        if (td_lane < 8) {
            const int bb = min(b0 + td_w * 8 + td_lane, n - 1);
            const int sslot = slot_of(T.src, bb);
            td_a = T.action_to[sslot];
            td_r = T.reward[sslot];
            td_dn = T.done[sslot] != 0;
            td_bo = T.b_out[(td_a >= 0 && td_a < 96) ? td_a : 0];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int bb = min(b0 + td_w * 8 + i, n - 1);
            const int a = __shfl(td_a, i, 64);
            const int ac = (a >= 0 && a < 96) ? a : 0;
            const float4 av = *reinterpret_cast<const float4*>(T.a_s + (long long)bb * 256 + td_lane * 4);
            const float4 wv = *reinterpret_cast<const float4*>(T.w_out + (long long)ac * 256 + td_lane * 4);
            zq[i] = (av.x * wv.x + av.y * wv.y) + (av.z * wv.z + av.w * wv.w);
        }
    }
    float m = kColmaxPadBias;
    for (int r = phase; r < ranges; r += 8) m = fmaxf(m, R[(long long)r * ldp + bc]);
    sv[phase][sl] = m;
    __syncthreads();
    if (tid < 32) {
        m = sv[0][sl];
#pragma unroll
        for (int p = 1; p < 8; ++p) m = fmaxf(m, sv[p][sl]);
        const float wmx = fmaxf(__builtin_bit_cast(float, wm[parity]), __builtin_bit_cast(float, wm[4]));
        const float bmx = fmaxf(__builtin_bit_cast(float, wm[2 + parity]), __builtin_bit_cast(float, wm[5]));
        const float B = kScreenEps * sqrtf(na_all[bc]) * wmx + kScreenBiasEps * bmx;
        float t0 = m - 2.f * B * 1.03125f - 1.52587890625e-05f * (fabsf(m) + 2.f * B);
        if (!(t0 == t0)) t0 = -__builtin_inff();      // a non-finite norm (diverged net): every group is a candidate, like the full product
        thr[sl] = ok ? t0 : __builtin_inff();            // no candidates past n
    }
    __syncthreads();
    {
        const float t = thr[sl];
        for (int r = phase; r < ranges; r += 8) {
            if (R[(long long)r * ldp + bc] < t) continue;        // no group of this range reaches the threshold
            const int g0 = r * gpr, g1 = min(G, g0 + gpr);
            // the groups of a range sixteen at a time, all loads in flight, then the second-largest values of the groups that reach the
            // threshold, again all in flight: two round trips per range instead of one per four groups plus one per hit.  (A net whose
            // per-range maxima all lie within the bound of each other — some fresh nets do, for a whole run — scans every range of every
            // sample: that kernel read 65 us instead of 27 with four loads in flight.)
            for (int g = g0; g < g1; g += 16) {
                float v[16], v2[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = P1[(long long)min(g + u, g1 - 1) * ldp + bc];
#pragma unroll
                for (int u = 0; u < 16; ++u) v2[u] = (g + u < g1 && v[u] >= t) ? P2[(long long)(g + u) * ldp + bc] : 0.f;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (g + u < g1 && v[u] >= t) {
                        if (v2[u] >= t) { wlist[atomicAdd(&nexp, 1)] = (uint16_t)(sl | ((g + u) << 5)); if (stage_ok) atomicOr(&gbits[g + u], 1u << sl); }
                        else cand[atomicAdd(&cnt, 1)] = (uint32_t)sl | ((uint32_t)screen_row(g + u, (int)(__builtin_bit_cast(uint32_t, v[u]) & 31u)) << 5);
                    }
                }
            }
        }
    }
    __syncthreads();
    // fp32 dots of the candidates (the maximum does not depend on the order they are visited in)
    const int singles = cnt, wholes = nexp;
    for (int e0 = 0; e0 < singles; e0 += 64) {                   // 16 quarters x 4 rows per round, all loads of a round in flight
        float z[4];
        int s2[4], row[4];
        bool live[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = e0 + quarter + 16 * r;
            live[r] = e < singles;
            const uint32_t ent = cand[live[r] ? e : 0];
            s2[r] = (int)(ent & 31u);
            row[r] = min((int)(ent >> 5), NO - 1);
            z[r] = quarter_dot<KFIX>(a_last + (long long)(b0 + s2[r]) * K, W + (long long)row[r] * K, K, ql);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            z[r] = quarter_sum(z[r]) + bias[row[r]];
            if (live[r] && ql == 0 && z[r] == z[r]) atomicMax(&best[s2[r]], float_order_key(z[r]));   // (a NaN output never wins: fmaxf semantics)
        }
    }
    // Whole groups that MANY samples of the block ask for (>= 8 of 32; nets whose two largest outputs of a sample sit in ONE group within the
    // bound of each other — the trained rows 0..95 in about a fifth of the runs from time-seeded weights — ask for the same group for nearly
    // every sample): the group's 32 rows of W are staged into LDS once (the candidate list's space: the round of single rows is over) and
    // every sample of that group takes them from there, instead of 32 KB through the texture path per sample (262 MB per launch: the kernel
    // read 55-65 us instead of 27 and the step 0.205 instead of 0.173 ms, tools/facade_rep.sh).  Same dots, same order inside a dot, same bits.
    // whole_mode 2 (the host grants it when the launch has one block per CU and K = 256): the block's dynamic LDS has room for the 32 activation
    // rows of its samples and the rows of W of THREE groups (refine_stage_bytes); everything the popular groups need then arrives in ONE round
    // trip and every dot reads LDS.  The loops above and below cost a dependent round trip per four groups: ~3 us each beside the select
    // chain, 24 us for 26 groups per block; this pass: ~6 us.
    const bool staged_pass = stage_ok && KFIX == 256 && wholes >= 8 && whole_mode == 2;
    if (staged_pass) {
        float* Wt0 = reinterpret_cast<float*>(cand);                                                          // [32][256]
        float* At = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(wlist) + refine_wlist_bytes(G));   // [32][256], then W tiles 1, 2
        {   // which groups: one ballot per wave over "thread tid's group has >= 8 samples"
            const unsigned long long m = __ballot(__popc(gbits[tid]) >= 8);
            if ((tid & 63) == 0) gpop[tid >> 6] = m;
        }
        __syncthreads();                                         // (also: every thread has left the candidate list)
        unsigned long long gm[4] = {gpop[0], gpop[1], gpop[2], gpop[3]};       // block-uniform
        bool a_staged = false;
        for (;;) {
            int gs[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                gs[u] = -1;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv)
                    if (gs[u] < 0 && gm[wv]) { gs[u] = wv * 64 + __builtin_ctzll(gm[wv]); gm[wv] &= gm[wv] - 1; }
            }
            if (gs[0] < 0) break;
            // two round trips (16 x 16 bytes per thread in flight each): the activation rows (first sweep) and the rows of up to three groups,
            // with this thread's biases
            float bj[3][2];
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int r = 0; r < 2; ++r) bj[u][r] = gs[u] >= 0 ? bias[min(screen_row(gs[u], quarter + 16 * r), NO - 1)] : 0.f;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float4 ra[4], rw0[4], rw1[4], rw2[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = (half * 4 + q) * 256 + tid, j = i >> 6, c = i & 63;       // float4 i of a [32][256] tile
                    // (unconditional loads — a group that is not there reads group gs[0]'s rows again — so that the arrays stay in registers)
                    ra[q] = *reinterpret_cast<const float4*>(a_last + (long long)min(b0 + j, n - 1) * K + 4 * c);
                    rw0[q] = *reinterpret_cast<const float4*>(W + (long long)min(screen_row(gs[0], j), NO - 1) * K + 4 * c);
                    rw1[q] = *reinterpret_cast<const float4*>(W + (long long)min(screen_row(gs[1] >= 0 ? gs[1] : gs[0], j), NO - 1) * K + 4 * c);
                    rw2[q] = *reinterpret_cast<const float4*>(W + (long long)min(screen_row(gs[2] >= 0 ? gs[2] : gs[0], j), NO - 1) * K + 4 * c);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = (half * 4 + q) * 256 + tid;
                    if (!a_staged) reinterpret_cast<float4*>(At)[i] = ra[q];
                    if (gs[0] >= 0) reinterpret_cast<float4*>(Wt0)[i] = rw0[q];
                    if (gs[1] >= 0) reinterpret_cast<float4*>(At + 1 * 32 * 256)[i] = rw1[q];
                    if (gs[2] >= 0) reinterpret_cast<float4*>(At + 2 * 32 * 256)[i] = rw2[q];
                }
            }
            a_staged = true;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (gs[u] < 0) continue;
                const float* Wt = u == 0 ? Wt0 : At + u * 32 * 256;
                unsigned left = gbits[gs[u]];                    // block-uniform
                while (left) {
                    const int s2 = __builtin_ctz(left);
                    left &= left - 1;
                    float zb = kColmaxPadBias;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const int j = quarter + 16 * r;
                        const float z = quarter_sum(quarter_dot<KFIX>(At + s2 * 256, Wt + j * 256, K, ql)) + bj[u][r];
                        if (screen_row(gs[u], j) < NO) zb = fmaxf(zb, z);
                    }
                    if (ql == 0) atomicMax(&best[s2], float_order_key(zb));
                }
            }
            __syncthreads();                                     // before the next sweep overwrites the tiles
        }
    }
    // whole groups that the staged pass did not take (few samples ask for them, or the launch has no room for the pass): the 32 rows of a
    // group over the 16 quarters, one group per round trip.  (Four groups per round trip, measured: no faster in the regime that has many of
    // them — that loop is a chain of round trips whatever is in flight — and 86 more registers, which cost the usual step 1.5 us.)
    for (int e = 0; e < wholes; ++e) {
        const int ent = wlist[e];
        const int s2 = ent & 31, g = ent >> 5;
        if (staged_pass && __popc(gbits[g]) >= 8) continue;      // block-uniform
        float zb = kColmaxPadBias;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = screen_row(g, quarter + 16 * r);
            const int rc = min(row, NO - 1);
            const float z = quarter_sum(quarter_dot<KFIX>(a_last + (long long)(b0 + s2) * K, W + (long long)rc * K, K, ql)) + bias[rc];
            if (row < NO) zb = fmaxf(zb, z);
        }
        if (ql == 0) atomicMax(&best[s2], float_order_key(zb));
    }
    __syncthreads();
    if (tid < 32 && ok) zmax[b] = float_from_key(best[sl]);
    if (tid == 0) {                                  // this block's own running totals (512 same-address atomics per launch cost the
        stats[2 * blockIdx.x] = st_pairs + (unsigned long long)(singles + wholes);      // step 5 us)
        stats[2 * blockIdx.x + 1] = st_whole + (unsigned long long)wholes;
    }
    if (TD) {
        // the rows for the top hidden delta (L2-hot: read a moment ago / shared by every sample with the same action), all in flight
        float4 av[8], vv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int bb = min(b0 + td_w * 8 + i, n - 1);
            const int a = __shfl(td_a, i, 64);
            const bool has_view = a >= 0 && a < 96 && a < T.view_kmax;
            av[i] = *reinterpret_cast<const float4*>(T.a_s + (long long)bb * 256 + td_lane * 4);
            vv[i] = has_view ? *reinterpret_cast<const float4*>(T.view + (long long)a * T.view_ld + td_lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int bb = b0 + td_w * 8 + i;
            const int a = __shfl(td_a, i, 64);
            const float r = __shfl(td_r, i, 64), bo = __shfl(td_bo, i, 64);
            const bool dn = __shfl((int)td_dn, i, 64) != 0;
            const bool live = a >= 0 && a < 96;
            float z = zq[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) z += __shfl_xor(z, off, 64);
            const float zm = float_from_key(best[td_w * 8 + i]);
            float q = 0.f, y = 0.f, delta = 0.f;
            if (live) {
                q = tanhf(z + bo);
                y = dn ? r : r + T.gamma * tanhf(zm);
                delta = (q - y) * (1.f - q * q) * 1.f;
            }
            if (bb < n) {                                         // wave-uniform
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live && a < T.view_kmax) {
                    o.x = delta * vv[i].x * (1.f - av[i].x * av[i].x); o.y = delta * vv[i].y * (1.f - av[i].y * av[i].y);
                    o.z = delta * vv[i].z * (1.f - av[i].z * av[i].z); o.w = delta * vv[i].w * (1.f - av[i].w * av[i].w);
                }
                *reinterpret_cast<float4*>(T.dtop + (long long)bb * 256 + td_lane * 4) = o;
                if (td_lane == 0) {
                    T.dsc[bb] = delta;
                    T.act[bb] = live ? a : -1;
                    T.qsa[bb] = q; T.yv[bb] = y;
                    T.lossv[bb] = live ? 0.5f * (q - y) * (q - y) : 0.f;
                }
            }
        }
    }
}


}  // namespace xq
