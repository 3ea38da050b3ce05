// xq_tail.hip.h — the gradient half of a TD step: partial-maximum fold, TD target / output delta (outputLayerDeltaKernel dqn.cu:288-295 on
// the one non-zero column of chessai.cpp:121-133), output-layer and bias gradients as ordered sums, the fused launches (td_tail_kernel), the
// ordered slab reduction, the Q head finish and the SGD step (updateWeightsBiasesKernel dqn.cu:310-319, batched)
// (kernel half of xq_dqn.hip, split out in round 5; included by xq_dqn.hip only, inside namespace xq)
#pragma once

namespace xq {

// zmax[b] = max over the column-max GEMM's partial rows t of partial[t][b] (and, for Double DQN, the row index that came with
// the first maximum).  The partials are [n_partial][n]: a block takes 64 consecutive samples so that every wave-instruction reads
// 256 contiguous bytes of one partial row (td_delta_kernel's one-wave-per-sample walk touched a cache line per value); wave w
// folds rows w, w+4, ... with 8 independent loads in flight, the four waves combine through LDS.
// blockIdx.y = one of kReduceParts contiguous ranges of partial rows (4x the blocks in flight: the kernel is pure latency);
// zmax / zidx are [kReduceParts][n], td_delta_kernel folds the last kReduceParts values of its sample itself.
enum { kReduceParts = 4 };
__global__ __launch_bounds__(256) void colmax_reduce_kernel(const float* __restrict__ partial_all, const int* __restrict__ partial_idx_all,
                                                            int n_partial_all, int n, long long ld, float* __restrict__ zmax_all,
                                                            int* __restrict__ zidx_all) {
    const int per = (n_partial_all + kReduceParts - 1) / kReduceParts;
    const int t0 = (int)blockIdx.y * per;
    const int n_partial = max(0, min(per, n_partial_all - t0));
    const float* partial = partial_all + (long long)t0 * ld;          // rows of the partial arrays are `ld` apart (>= n)
    const int* partial_idx = partial_idx_all ? partial_idx_all + (long long)t0 * ld : nullptr;
    float* zmax = zmax_all + (long long)blockIdx.y * n;
    int* zidx = zidx_all + (long long)blockIdx.y * n;
    __shared__ float sv[4][64];
    __shared__ int si[4][64];
    const int lane = (int)(threadIdx.x & 63), wid = (int)(threadIdx.x >> 6);
    const int b = (int)blockIdx.x * 64 + lane;
    const bool ok = b < n;
    float m = -__builtin_inff();
    int mi = 0x7fffffff;
    const bool arg = partial_idx != nullptr;
    int t = wid;
    for (; t + 28 < n_partial; t += 32) {
        float v[8];
        int vi[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v[u] = ok ? partial[(long long)(t + 4 * u) * ld + b] : -__builtin_inff();
            vi[u] = (ok && arg) ? partial_idx[(long long)(t + 4 * u) * ld + b] : 0x7fffffff;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (arg) { if (v[u] > m || (v[u] == m && vi[u] < mi)) { m = v[u]; mi = vi[u]; } }
            else m = fmaxf(m, v[u]);
        }
    }
    for (; t < n_partial; t += 4) {
        const float v = ok ? partial[(long long)t * ld + b] : -__builtin_inff();
        const int vi = (ok && arg) ? partial_idx[(long long)t * ld + b] : 0x7fffffff;
        if (arg) { if (v > m || (v == m && vi < mi)) { m = v; mi = vi; } }
        else m = fmaxf(m, v);
    }
    sv[wid][lane] = m; si[wid][lane] = mi;
    __syncthreads();
    if (wid == 0 && ok) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float v = sv[w][lane];
            const int vi = si[w][lane];
            if (arg) { if (v > m || (v == m && vi < mi)) { m = v; mi = vi; } }
            else m = fmaxf(m, v);
        }
        zmax[b] = m;
        if (arg) zidx[b] = mi;
    }
}

// What the generalised TD step (BASELINE configs[4], build-defined) adds to td_delta_kernel; all optional.
struct TdExtra {
    const int* partial_idx;        // Double DQN: row index of the maximum of every sample (first maximum), reduced
    const float* wout_t; const uint16_t* wout_t_bf; const float* bout_t;   // target net's output layer (fp32 master / bf16 shadow)
    const float* alast_t; const uint16_t* alast_t_bf;                      // a_last(s') of the target net (fp32 / bf16 bits)
    const uint16_t* wout_bf;       // bf16 Q-net: shadow of the online output layer for Q(s,a)
    const float* is_w; const float* is_wmax;     // prioritized replay: raw importance weights [n] and their batch maximum
    float* prio; unsigned* pmax_live;            // prioritized replay: priority table (by ring slot) and the running maximum (float bits)
    float per_eps, per_alpha;
    int double_dqn, nout;
    uint16_t* dtop_bf;             // XQ_PRECISION_BF16_FULL: the top hidden delta rounded to bf16 beside the fp32 one
};

// TD target, output delta and the TOP hidden delta for one sample per wave (chessai.cpp:122-128 +
// outputLayerDeltaKernel dqn.cu:288-295 + hiddenLayerDeltaKernel dqn.cu:297-308 for the last hidden layer).
// The output delta of a TD step has ONE non-zero entry per sample (column action.to), so the last hidden layer's delta
// is a scaled row of the weight view — no GEMM:  dtop[b][i] = delta_b * View[a_b][i] * (1 - a_last[b][i]^2), where
// View[a][i] = view[a*view_ld + i] is the as-written (reference mode: a < view_kmax = width of the last hidden layer,
// stride = width of the layer below) or the textbook (row a of W_out) operand.  Also emits, per sample, the scalar
// delta and the action (gathered through `slots`) for the segmented output-layer gradient.
// Double DQN: the partials carry (max z_online(s'), its row a*); y = r + gamma * tanh(W_out_target[a*] . a_last_target(s') + b).
// Prioritized replay: delta is scaled by w_b / max w, and (|Q(s,a) - y| + eps)^alpha goes back into the priority table.
__global__ __launch_bounds__(256) void td_delta_kernel(int n, SlotSrc src,
                                                       const int32_t* __restrict__ action_to, const float* __restrict__ reward,
                                                       const uint8_t* __restrict__ done, const float* __restrict__ a_last, int H,
                                                       const float* __restrict__ w_out, const float* __restrict__ b_out,
                                                       const float* __restrict__ partial, int n_partial, float gamma,
                                                       const float* __restrict__ view, long long view_ld, int view_kmax,
                                                       float* __restrict__ dtop, float* __restrict__ dsc, int32_t* __restrict__ act,
                                                       float* __restrict__ qsa, float* __restrict__ yv, float* __restrict__ lossv,
                                                       TdExtra X) {
    const int wid = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int b = (int)blockIdx.x * 4 + wid;
    if (b >= n) return;
    const int s = slot_of(src, b);
    const int a = action_to[s];
    const bool live = a >= 0 && a < 96;
    float delta = 0.f, q = 0.f, y = 0.f;
    const float* ar = a_last + (long long)b * H;
    if (H == 256 && !X.wout_bf && !X.double_dqn && n_partial <= 4) {
        // fp32 net, 256-wide last hidden layer, no arg-max: every load that depends only on (b, s, a) is
        // issued up front as one 16-byte load per lane — the general path below is a chain of five dependent memory round trips
        const int ac = live ? a : 0;
        const float4 av = *reinterpret_cast<const float4*>(ar + lane * 4);
        const float4 wv = *reinterpret_cast<const float4*>(w_out + (long long)ac * 256 + lane * 4);
        const bool has_view = live && a < view_kmax;
        const float4 vv = has_view ? *reinterpret_cast<const float4*>(view + (long long)a * view_ld + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float zm = partial[b];                         // the (up to kReduceParts) maxima left per sample
#pragma unroll
        for (int t = 1; t < 4; ++t) zm = fmaxf(zm, partial[(long long)min(t, n_partial - 1) * n + b]);
        const float bo = b_out[ac], r = reward[s];
        const bool dn = done[s] != 0;
        const float isw = X.is_w ? X.is_w[b] / X.is_wmax[0] : 1.f;
        float z = (av.x * wv.x + av.y * wv.y) + (av.z * wv.z + av.w * wv.w);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) z += __shfl_xor(z, off, 64);
        if (live) {
            q = tanhf(z + bo);
            y = dn ? r : r + gamma * tanhf(zm);
            delta = (q - y) * (1.f - q * q) * isw;
        }
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_view) {
            o.x = delta * vv.x * (1.f - av.x * av.x); o.y = delta * vv.y * (1.f - av.y * av.y);
            o.z = delta * vv.z * (1.f - av.z * av.z); o.w = delta * vv.w * (1.f - av.w * av.w);
        }
        *reinterpret_cast<float4*>(dtop + (long long)b * 256 + lane * 4) = o;
    } else if (H == 512 && !X.wout_bf && !X.double_dqn && n_partial <= 4) {
        // fp32 net, 512-wide last hidden layer (BASELINE configs[3]): the 256-wide path with two 16-byte pieces per lane and row
        // (columns 4 lane + 256 v); per-lane partial = piece 0 + piece 1, then the same shuffle tree
        const int ac = live ? a : 0;
        const bool has_view = live && a < view_kmax;
        float4 av[2], wv[2], vv[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            av[v] = *reinterpret_cast<const float4*>(ar + v * 256 + lane * 4);
            wv[v] = *reinterpret_cast<const float4*>(w_out + (long long)ac * 512 + v * 256 + lane * 4);
            vv[v] = has_view ? *reinterpret_cast<const float4*>(view + (long long)a * view_ld + v * 256 + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float zm = partial[b];
#pragma unroll
        for (int t = 1; t < 4; ++t) zm = fmaxf(zm, partial[(long long)min(t, n_partial - 1) * n + b]);
        const float bo = b_out[ac], r = reward[s];
        const bool dn = done[s] != 0;
        const float isw = X.is_w ? X.is_w[b] / X.is_wmax[0] : 1.f;
        float z = ((av[0].x * wv[0].x + av[0].y * wv[0].y) + (av[0].z * wv[0].z + av[0].w * wv[0].w)) +
                  ((av[1].x * wv[1].x + av[1].y * wv[1].y) + (av[1].z * wv[1].z + av[1].w * wv[1].w));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) z += __shfl_xor(z, off, 64);
        if (live) {
            q = tanhf(z + bo);
            y = dn ? r : r + gamma * tanhf(zm);
            delta = (q - y) * (1.f - q * q) * isw;
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (has_view) {
                o.x = delta * vv[v].x * (1.f - av[v].x * av[v].x); o.y = delta * vv[v].y * (1.f - av[v].y * av[v].y);
                o.z = delta * vv[v].z * (1.f - av[v].z * av[v].z); o.w = delta * vv[v].w * (1.f - av[v].w * av[v].w);
            }
            *reinterpret_cast<float4*>(dtop + (long long)b * 512 + v * 256 + lane * 4) = o;
        }
    } else if (H == 512 && X.wout_bf && n_partial <= 4 && (!X.double_dqn || X.wout_t_bf)) {
        // bf16 net, 512-wide last hidden layer (BASELINE configs[4]): the same idea — every load that depends only on (b, s, a) issued up
        // front, 8 columns per lane as 16-byte loads; Double DQN adds ONE dependent round trip (the target net's row of the arg-max)
        const int ac = live ? a : 0;
        const float* arp = ar + lane * 8;
        const float4 av0 = *reinterpret_cast<const float4*>(arp), av1 = *reinterpret_cast<const float4*>(arp + 4);
        const uint4 wq = *reinterpret_cast<const uint4*>(X.wout_bf + (long long)ac * 512 + lane * 8);
        const bool has_view = live && a < view_kmax;
        const float* vp = view + (long long)(has_view ? a : 0) * view_ld + lane * 8;
        float4 vv0 = *reinterpret_cast<const float4*>(vp), vv1 = *reinterpret_cast<const float4*>(vp + 4);
        float pm[4]; int pi[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            pm[t] = partial[(long long)min(t, n_partial - 1) * n + b];
            pi[t] = X.double_dqn ? X.partial_idx[(long long)min(t, n_partial - 1) * n + b] : 0;
        }
        uint4 atq = make_uint4(0u, 0u, 0u, 0u);
        if (X.double_dqn) atq = *reinterpret_cast<const uint4*>(X.alast_t_bf + (long long)b * 512 + lane * 8);
        const float bo = b_out[ac], r = reward[s];
        const bool dn = done[s] != 0;
        const float isw = X.is_w ? X.is_w[b] / X.is_wmax[0] : 1.f;
        auto lo = [](uint32_t x) { return __builtin_bit_cast(float, x << 16); };
        auto hi = [](uint32_t x) { return __builtin_bit_cast(float, x & 0xFFFF0000u); };
        float z = ((av0.x * lo(wq.x) + av0.y * hi(wq.x)) + (av0.z * lo(wq.y) + av0.w * hi(wq.y))) +
                  ((av1.x * lo(wq.z) + av1.y * hi(wq.z)) + (av1.z * lo(wq.w) + av1.w * hi(wq.w)));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) z += __shfl_xor(z, off, 64);
        float zm = pm[0]; int zi = pi[0];
#pragma unroll
        for (int t = 1; t < 4; ++t) {
            if (X.double_dqn) { if (pm[t] > zm || (pm[t] == zm && pi[t] < zi)) { zm = pm[t]; zi = pi[t]; } }
            else zm = fmaxf(zm, pm[t]);
        }
        if (X.double_dqn) {          // value of the online net's greedy action on the TARGET net
            const int astar = (zi >= 0 && zi < X.nout) ? zi : 0;
            const uint4 tq = *reinterpret_cast<const uint4*>(X.wout_t_bf + (long long)astar * 512 + lane * 8);
            float zt = ((lo(atq.x) * lo(tq.x) + hi(atq.x) * hi(tq.x)) + (lo(atq.y) * lo(tq.y) + hi(atq.y) * hi(tq.y))) +
                       ((lo(atq.z) * lo(tq.z) + hi(atq.z) * hi(tq.z)) + (lo(atq.w) * lo(tq.w) + hi(atq.w) * hi(tq.w)));
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) zt += __shfl_xor(zt, off, 64);
            zm = zt + X.bout_t[astar];
        }
        if (live) {
            q = tanhf(z + bo);
            y = dn ? r : r + gamma * tanhf(zm);
            delta = (q - y) * (1.f - q * q) * isw;
        }
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (has_view) {
            o[0] = delta * vv0.x * (1.f - av0.x * av0.x); o[1] = delta * vv0.y * (1.f - av0.y * av0.y);
            o[2] = delta * vv0.z * (1.f - av0.z * av0.z); o[3] = delta * vv0.w * (1.f - av0.w * av0.w);
            o[4] = delta * vv1.x * (1.f - av1.x * av1.x); o[5] = delta * vv1.y * (1.f - av1.y * av1.y);
            o[6] = delta * vv1.z * (1.f - av1.z * av1.z); o[7] = delta * vv1.w * (1.f - av1.w * av1.w);
        }
        float* dp = dtop + (long long)b * 512 + lane * 8;
        *reinterpret_cast<float4*>(dp) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(dp + 4) = make_float4(o[4], o[5], o[6], o[7]);
        if (X.dtop_bf)
            *reinterpret_cast<uint4*>(X.dtop_bf + (long long)b * 512 + lane * 8) =
                make_uint4((uint32_t)bf16_bits(o[0]) | ((uint32_t)bf16_bits(o[1]) << 16), (uint32_t)bf16_bits(o[2]) | ((uint32_t)bf16_bits(o[3]) << 16),
                           (uint32_t)bf16_bits(o[4]) | ((uint32_t)bf16_bits(o[5]) << 16), (uint32_t)bf16_bits(o[6]) | ((uint32_t)bf16_bits(o[7]) << 16));
    } else {
    if (live) {
        float z = 0.f;
        if (X.wout_bf) { const uint16_t* wr = X.wout_bf + (long long)a * H; for (int i = lane; i < H; i += 64) z += bf16_to_float(wr[i]) * ar[i]; }
        else { const float* wr = w_out + (long long)a * H; for (int i = lane; i < H; i += 64) z += wr[i] * ar[i]; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) z += __shfl_xor(z, off, 64);
        z += b_out[a];
        float zm = partial[b];                         // max_k z_k(s'): the kReduceParts values colmax_reduce_kernel left per sample
        int zi = X.double_dqn ? X.partial_idx[b] : 0;
        for (int t = 1; t < n_partial; ++t) {
            const float v = partial[(long long)t * n + b];
            if (X.double_dqn) {
                const int vi = X.partial_idx[(long long)t * n + b];
                if (v > zm || (v == zm && vi < zi)) { zm = v; zi = vi; }
            } else zm = fmaxf(zm, v);
        }
        if (X.double_dqn) {          // value of the online net's greedy action on the TARGET net
            const int astar = (zi >= 0 && zi < X.nout) ? zi : 0;
            float zt = 0.f;
            for (int i = lane; i < H; i += 64) {
                const float wv = X.wout_t_bf ? bf16_to_float(X.wout_t_bf[(long long)astar * H + i]) : X.wout_t[(long long)astar * H + i];
                const float av = X.alast_t_bf ? bf16_to_float(X.alast_t_bf[(long long)b * H + i]) : X.alast_t[(long long)b * H + i];
                zt += wv * av;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) zt += __shfl_xor(zt, off, 64);
            zm = zt + X.bout_t[astar];
        }
        q = tanhf(z);
        const float r = reward[s];
        y = done[s] ? r : r + gamma * tanhf(zm);       // max_k tanh(z_k) = tanh(max_k z_k)
        delta = (q - y) * (1.f - q * q);               // (a - target) * (1 - tanh(z)^2)
        if (X.is_w) delta *= X.is_w[b] / X.is_wmax[0];
    }
    float* drow = dtop + (long long)b * H;
    if (live && a < view_kmax) {
        const float* vr = view + (long long)a * view_ld;
        for (int i = lane; i < H; i += 64) {
            const float h = ar[i];
            const float v = delta * vr[i] * (1.f - h * h);
            drow[i] = v;
            if (X.dtop_bf) X.dtop_bf[(long long)b * H + i] = bf16_bits(v);
        }
    } else {
        for (int i = lane; i < H; i += 64) { drow[i] = 0.f; if (X.dtop_bf) X.dtop_bf[(long long)b * H + i] = 0; }
    }
    }
    if (lane == 0) {
        dsc[b] = delta;
        act[b] = live ? a : -1;
        qsa[b] = q; yv[b] = y;
        lossv[b] = live ? 0.5f * (q - y) * (q - y) : 0.f;
        if (X.prio && live) {
            const float p = powf(fabsf(q - y) + X.per_eps, X.per_alpha);
            X.prio[s] = p;
            // the running maximum rarely moves once training is under way: test first, so that 16 K waves do not queue on one address
            // (positive floats order like their bit patterns; a maximum is order-independent, hence still deterministic)
            if (__float_as_uint(p) > __hip_atomic_load(X.pmax_live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(X.pmax_live, __float_as_uint(p));
        }
    }
}

// Output-layer gradient of a TD minibatch: gW_out[j][:] = sum over the samples with action.to == j of delta_b * a_last[b][:]
// and gb_out[j] = sum of delta_b (j < 96).  Segmented sums instead of a [96 x B] x [B x H] product: every sample's
// activation row is read exactly once.  Block (group of 4 actions, chunk of samples): ordered compaction of the chunk's
// samples whose action falls in the group, then the rows are streamed into per-wave LDS accumulators (combined in fixed
// order => bitwise reproducible).  partial[chunk][96*H + 96] (weights, then biases).
__device__ __forceinline__ void out_grad_block(const int32_t* __restrict__ act, const float* __restrict__ dsc,
                                               const float* __restrict__ a_last, int n, int H, int chunk,
                                               float* __restrict__ partial, int g /* actions 4g .. 4g+3 */, int chunk_id, float* __restrict__ smem) {
    float* acc = smem;                                  // [4 waves][4 actions][H]
    uint16_t* list = reinterpret_cast<uint16_t*>(smem + 16 * H);   // [chunk] (b_local | class << 11)
    __shared__ int total;
    __shared__ float bsum[4][4];
    const int c0 = chunk_id * chunk, c1 = min(n, c0 + chunk);
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // the chunk's actions first (all loads of a thread in flight together), accumulator zeroing under their latency; then the
    // ordered compaction with two barriers in all: per-wave counts of every round published first, offsets = prefix sums over
    // (round, wave) that every thread computes for itself
    constexpr int kMaxIters = 8;                         // chunk <= 2048
    __shared__ int wc[kMaxIters][4];
    int clsv[kMaxIters];
#pragma unroll
    for (int it = 0; it < kMaxIters; ++it) {
        const int b = c0 + it * 256 + tid;
        const int a = act[min(b, c1 - 1)];                // unconditional, clamped (see l0_grad_kernel)
        clsv[it] = (b < c1 && a >= 4 * g && a < 4 * g + 4) ? a - 4 * g : -1;
    }
    for (int i = tid; i < 16 * H; i += 256) acc[i] = 0.f;
#pragma unroll
    for (int it = 0; it < kMaxIters; ++it) {
        const unsigned long long m = __ballot(clsv[it] >= 0);
        if (lane == 0) wc[it][wid] = __popcll(m);
    }
    __syncthreads();
    {
        const int iters = (c1 - c0 + 255) / 256;
        int off = 0;
#pragma unroll
        for (int it = 0; it < kMaxIters; ++it) {
            if (it < iters) {
                const unsigned long long m = __ballot(clsv[it] >= 0);
                int o = off;
                for (int w = 0; w < wid; ++w) o += wc[it][w];
                if (clsv[it] >= 0) list[o + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)((it * 256 + tid) | (clsv[it] << 11));
            }
            off += wc[it][0] + wc[it][1] + wc[it][2] + wc[it][3];
        }
        if (tid == 0) total = off;
    }
    __syncthreads();
    const int cnt = total;
    float* my = acc + (long long)wid * 4 * H;
    float bs0 = 0.f, bs1 = 0.f, bs2 = 0.f, bs3 = 0.f;
    // wave w takes entries w, w+4, ... (fixed assignment), 8 rows in flight
    int i = wid;
    for (; i + 28 < cnt; i += 32) {
        int bb[8], cl[8];
        float dl[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = list[i + 4 * u];
            bb[u] = c0 + (e & 2047); cl[u] = e >> 11;
            dl[u] = dsc[bb[u]];
        }
        for (int col = lane * 4; col < H; col += 256) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 x = *reinterpret_cast<const float4*>(a_last + (long long)bb[u] * H + col);
                v[u].x = x.x; v[u].y = x.y; v[u].z = x.z; v[u].w = x.w;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float4* a = reinterpret_cast<float4*>(my + cl[u] * H + col);
                float4 t = *a;
                t.x += dl[u] * v[u].x; t.y += dl[u] * v[u].y; t.z += dl[u] * v[u].z; t.w += dl[u] * v[u].w;
                *a = t;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (cl[u] == 0) bs0 += dl[u]; else if (cl[u] == 1) bs1 += dl[u]; else if (cl[u] == 2) bs2 += dl[u]; else bs3 += dl[u];
        }
    }
    for (; i < cnt; i += 4) {
        const int e = list[i];
        const int b = c0 + (e & 2047), cls = e >> 11;
        const float d1 = dsc[b];
        if (cls == 0) bs0 += d1; else if (cls == 1) bs1 += d1; else if (cls == 2) bs2 += d1; else bs3 += d1;
        for (int col = lane * 4; col < H; col += 256) {
            const float4 x = *reinterpret_cast<const float4*>(a_last + (long long)b * H + col);
            float4* a = reinterpret_cast<float4*>(my + cls * H + col);
            float4 t = *a;
            t.x += d1 * x.x; t.y += d1 * x.y; t.z += d1 * x.z; t.w += d1 * x.w;
            *a = t;
        }
    }
    if (lane == 0) { bsum[wid][0] = bs0; bsum[wid][1] = bs1; bsum[wid][2] = bs2; bsum[wid][3] = bs3; }
    __syncthreads();
    float* out = partial + (long long)chunk_id * (96LL * H + 96);
    for (int i = tid; i < 4 * H; i += 256)
        out[(long long)4 * g * H + i] = ((acc[i] + acc[4 * H + i]) + acc[8 * H + i]) + acc[12 * H + i];
    if (tid < 4) out[96LL * H + 4 * g + tid] = ((bsum[0][tid] + bsum[1][tid]) + bsum[2][tid]) + bsum[3][tid];
}
__global__ __launch_bounds__(256) void out_grad_kernel(const int32_t* __restrict__ act, const float* __restrict__ dsc,
                                                       const float* __restrict__ a_last, int n, int H, int chunk,
                                                       float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    out_grad_block(act, dsc, a_last, n, H, chunk, partial, (int)blockIdx.x, (int)blockIdx.y, smem);
}

// dense output delta (general DQN::backpropagate target): d = (q - t) * (1 - q^2)
__global__ void out_delta_dense_kernel(const float* __restrict__ q, const float* __restrict__ t, long long total, float* __restrict__ d) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const float a = q[i];
        d[i] = (a - t[i]) * (1.f - a * a);
    }
}

// Bias gradients = column sums of the delta matrices.  All layers of one TD step go through ONE launch (job table)
// of partial sums over row chunks, then ONE ordered reduction launch (deterministic, no atomics).
struct ColsumJobs {
    const float* X[XQ_MAX_LAYERS + 1];
    long long ld[XQ_MAX_LAYERS + 1];
    int C[XQ_MAX_LAYERS + 1];
    float* dst[XQ_MAX_LAYERS + 1];
    long long poff[XQ_MAX_LAYERS + 1];   // first column of the job in the workspace
    int njobs, n, rows_per, R;
    float* work;                         // [R][wld]: row y = the sums over row chunk y, the jobs side by side — for the TD step in the
    long long wld;                       // order of the hidden biases, so that the SGD kernel can take the R rows as slabs (fused_apply)
};
__device__ __forceinline__ void colsum_partial_block(const ColsumJobs& J, int bx, int by, int job) {
    __shared__ float red[4][64];
    const int C = J.C[job];
    const int tx = (int)(threadIdx.x & 63), ty = (int)(threadIdx.x >> 6);
    const int c = bx * 64 + tx;
    if (bx * 64 >= C) return;
    const float* X = J.X[job];
    const long long ld = J.ld[job];
    const int r0 = by * J.rows_per, r1 = min(J.n, r0 + J.rows_per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
        int r = r0 + ty;
        for (; r + 12 < r1; r += 16) {
            s0 += X[(long long)r * ld + c];
            s1 += X[(long long)(r + 4) * ld + c];
            s2 += X[(long long)(r + 8) * ld + c];
            s3 += X[(long long)(r + 12) * ld + c];
        }
        for (; r < r1; r += 4) s0 += X[(long long)r * ld + c];
    }
    red[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty == 0 && c < C)
        J.work[J.poff[job] + (long long)by * J.wld + c] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}
__global__ __launch_bounds__(256) void colsum_partial_kernel(ColsumJobs J) {
    colsum_partial_block(J, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}
__global__ __launch_bounds__(256) void colsum_final_kernel(ColsumJobs J) {   // 64 columns x 4 partial lanes per block
    __shared__ float red[4][64];
    const int job = (int)blockIdx.y;
    const int C = J.C[job];
    const int tx = (int)(threadIdx.x & 63), ty = (int)(threadIdx.x >> 6);
    const int c = (int)blockIdx.x * 64 + tx;
    if ((int)blockIdx.x * 64 >= C) return;
    // the order of reduce_slabs_kernel / sgd_segments_kernel (four chains z = j, j + 4, ..; ((s0 + s1) + (s2 + s3))): the SGD kernel may
    // sum the rows itself (fused_apply) with the same bits
    float s0 = 0.f;
    if (c < C) {
        const float* p = J.work + J.poff[job] + c;
        const int R4 = J.R & ~3;
        for (int z = ty; z < R4; z += 4) s0 += p[(long long)z * J.wld];
        if (ty == 0) for (int z = R4; z < J.R; ++z) s0 += p[(long long)z * J.wld];     // the leftover rows continue chain 0, as there
    }
    red[ty][tx] = s0;
    __syncthreads();
    if (ty == 0 && c < C) J.dst[job][c] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

// The tail of a TD step as TWO launches on one stream instead of seven on two: every event record on the critical stream costs
// ~6 us of idle time and the join at the end 3-12 us, more than the kernels between them are worth.  One launch
// carries the blocks of several kernels ("horizontal fusion"): the grid is the concatenation of their grids, a block finds its body
// from its linear id (block-uniform branch), dynamic LDS = the largest of the bodies present.  Launch 1 (behind td_delta_kernel,
// once per hidden layer below the top one): the delta GEMM of the next layer down + what only needs the deltas already there — the
// weight-gradient GEMM of the layer above it and, first time round, the output-layer segmented sum.  Launch 2: the layer-0
// segmented sum + the bias column sums.  Long blocks come first in the grid.  Every body is the block function of the stand-alone
// kernel, so the results are bitwise those of the two-stream path (tests/test_dqn_gpu.py).
enum { TAIL_DELTA = 1, TAIL_GRAD = 2, TAIL_OUT = 4, TAIL_COLSUM = 8, TAIL_L0 = 16, TAIL_SEL = 32 };
struct TailArgs {
    // grid order: [l0][grad][delta][out][colsum][sel]; n_* = blocks of each part (0 = absent)
    int n_l0, n_grad, n_delta, n_out, n_colsum, n_sel;
    GemmArgs grad;  int grad_gx, grad_gy;          // 64x64 tiles: grid (gx, gy, splits)
    GemmArgs delta; int delta_gx;                  // grid (gx, gy, 1)
    const int32_t* og_act; const float* og_dsc; const float* og_alast; int og_n, og_H, og_chunk; float* og_partial;   // grid (24, chunks)
    const uint32_t* l0_boards; const float* l0_delta; int l0_n, l0_H, l0_HS, l0_chunk, l0_nsets, l0_nch; float* l0_partial;   // (90, nch, H / HS)
    const uint16_t* l0_planes; long long l0_plane_stride; int l0_kpad, l0_ncb;      // != nullptr: the matrix-pipe form, grid (4, H / 32, nch)
    const uint16_t* l0_sel;                        // ... and its selector half-words (xq_l0grad.hip.h)
    const uint32_t* sel_boards; int sel_n, sel_kpad; uint16_t* sel_out;             // TAIL_SEL: those half-words being made, 64 samples per block
    ColsumJobs cj; int cj_gx, cj_gy;               // grid (gx, R, njobs)
};
// L0MFMA: the layer-0 blocks are the matrix-pipe form (xq_dqn_set_l0_grad_mode(1)) — an instantiation of its own: that body needs 180
// VGPRs against 136 for the rest, and behind a run-time branch in the default kernel it capped every block of the launch at two waves
// per SIMD.  (Measured, same box, 3 x 3 x 300 steps: 180 / 136 / 106 VGPRs — the last forced with amdgpu_waves_per_eu(4) — 0.1887-0.1899 /
// 0.1896-0.1903 / 0.1885-0.1897 ms per step: the launch is not bound by its occupancy.)
template <unsigned KINDS, bool L0MFMA = false>
__global__ __launch_bounds__(256) void td_tail_kernel(const TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tail_smem[];
    int b = (int)blockIdx.x;
    if (KINDS & TAIL_L0) {
        if (b < a.n_l0) {
            if (L0MFMA) {
                // column block = b mod (H / 32): workgroups go to the 8 XCDs round-robin in linear order, so every XCD streams ITS columns'
                // planes (1/8 of the 12.6 MB at H = 256: L2-resident) for all row groups and chunks, instead of every XCD streaming half of
                // all planes (PMC: 125 MB of fabric reads per launch with the row group fastest)
                // ... unless the chunks themselves go round the XCDs (8 | chunks): the CHUNK is then the fastest index, XCD x keeps to the planes
                // AND the selector words of its own chunks (1.6 + 0.4 MB of 12.6 + 3.1), nothing is read by two XCDs — with the column block
                // fastest every XCD read all the selector words (25 MB per launch instead of 3.1)
                int rest = b / a.l0_ncb, cb = b % a.l0_ncb, sqg = rest & 3, ch = rest >> 2;
                if ((a.l0_nch & 7) == 0) { ch = b % a.l0_nch; rest = b / a.l0_nch; cb = rest % a.l0_ncb; sqg = rest / a.l0_ncb; }
                l0_grad_mfma_block<0>(a.l0_sel, a.l0_planes, a.l0_plane_stride, a.l0_kpad, a.l0_H, a.l0_chunk, a.l0_partial, sqg, cb, ch,
                                       reinterpret_cast<unsigned char*>(tail_smem));
                return;
            }
            const int per = kSquares * a.l0_nch;
            l0_grad_block(a.l0_boards, a.l0_delta, a.l0_n, a.l0_H, a.l0_HS, a.l0_chunk, a.l0_nsets, a.l0_partial, b % per, a.l0_nch, b / per, tail_smem);
            return;
        }
        b -= a.n_l0;
    }
    if (KINDS & TAIL_GRAD) {
        if (b < a.n_grad) {
            // the k-slab is the fastest index: workgroups go to the XCDs round-robin in linear order, so an XCD keeps to ITS eighth of the
            // samples (both operands: 2 MB) for all tiles, instead of every XCD pulling every slab of both operands through the fabric
            const int per = a.grad_gx * a.grad_gy, nz = a.n_grad / per, z = b % nz, r = b / nz;
            gemm_f32_block<L_MCONTIG, L_MCONTIG, EPI_STORE, 1, 1>(a.grad, r % a.grad_gx, r / a.grad_gx, z, tail_smem, tail_smem + g_tile_floats(64));
            return;
        }
        b -= a.n_grad;
    }
    if (KINDS & TAIL_DELTA) {
        if (b < a.n_delta) {
            gemm_f32_block<L_KCONTIG, L_MCONTIG, EPI_DELTA, 1, 1>(a.delta, b % a.delta_gx, b / a.delta_gx, 0, tail_smem, tail_smem + g_tile_floats(64));
            return;
        }
        b -= a.n_delta;
    }
    if (KINDS & TAIL_OUT) {
        if (b < a.n_out) {
            out_grad_block(a.og_act, a.og_dsc, a.og_alast, a.og_n, a.og_H, a.og_chunk, a.og_partial, b % 24, b / 24, tail_smem);
            return;
        }
        b -= a.n_out;
    }
    if (KINDS & TAIL_COLSUM) {
        if (b < a.n_colsum) {
            const int per = a.cj_gx * a.cj_gy, r = b % per;
            colsum_partial_block(a.cj, r % a.cj_gx, r / a.cj_gx, b / per);
            return;
        }
        b -= a.n_colsum;
    }
    if (KINDS & TAIL_SEL) {
        if (b < a.n_sel) l0_sel_block(a.sel_boards, a.sel_n, a.sel_kpad, a.sel_out, b, reinterpret_cast<uint32_t*>(tail_smem));
    }
}

// out[i] = sum_z slabs[z*stride + i], z ascending (deterministic)
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int nslabs, long long stride, long long len, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int z = 0;
        for (; z + 3 < nslabs; z += 4) {
            s0 += slabs[(long long)z * stride + i];
            s1 += slabs[(long long)(z + 1) * stride + i];
            s2 += slabs[(long long)(z + 2) * stride + i];
            s3 += slabs[(long long)(z + 3) * stride + i];
        }
        for (; z < nslabs; ++z) s0 += slabs[(long long)z * stride + i];
        out[i] = (s0 + s1) + (s2 + s3);
    }
}

struct SegTable {
    float* dst[16];
    const float* src[16];
    long long len[16];
    int nslabs[16];            // > 0: src holds that many partial-sum slabs `stride` apart; they are summed here, in the order
    long long stride[16];      //      of reduce_slabs_kernel (bit-identical to reducing first), instead of by a kernel of their own
    uint16_t* dst_bf[16];      // bf16 Q-net: shadow of dst, refreshed with the rounded new value (nullptr: none)
    int nseg;
    int reduce_only;           // dst = the slab sum itself (no step): the gradient buffer a reader or an all-reduce needs, in one launch
    int vec4[16];              // set by sgd_apply: every pointer 16-byte aligned, len and stride multiples of 4 — four elements per thread
};

// Q head with the k range split over blocks (q_head) or folded into the last hidden product (EPI_HEAD): q[m][j] = tanh(b_j + the sum of
// the k-slabs of the product), slabs added four at a time as (s0 + s1) + (s2 + s3), the groups of four in ascending order (nslabs even).
// One thread per output.
__global__ __launch_bounds__(256) void q_head_finish_kernel(const float* __restrict__ slabs, long long slab_stride, int nslabs, int n, int n_out, int lds_,
                                                            const float* __restrict__ bias, float* __restrict__ q, int ldq) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * n_out) return;
    const int m = (int)(i / n_out), j = (int)(i % n_out);
    const float* p = slabs + (long long)m * lds_ + j;
    float s = 0.f;
    for (int z = 0; z < nslabs; z += 4) {
        float t = p[z * slab_stride] + p[(z + 1) * slab_stride];
        if (z + 3 < nslabs) t += p[(z + 2) * slab_stride] + p[(z + 3) * slab_stride];
        s = z == 0 ? t : s + t;
    }
    q[(long long)m * ldq + j] = tanhf(bias[j] + s);
}

// bf16 shadow of a weight range (set_params / load_model / set_precision)
__global__ void f32_to_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = bf16_bits(src[i]);
}
// SGD: dst -= alpha * src per segment (updateWeightsBiasesKernel dqn.cu:310-319, batched form)
__device__ __forceinline__ float4 f4_add(const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__global__ void sgd_segments_kernel(SegTable t, float alpha) {
    const int sgm = (int)blockIdx.y;
    if (sgm >= t.nseg) return;
    float* d = t.dst[sgm];
    const float* s = t.src[sgm];
    const int nslabs = t.nslabs[sgm];
    const long long len = t.len[sgm], st = t.stride[sgm];
    uint16_t* db = t.dst_bf[sgm];
    if (t.vec4[sgm]) {
        // four consecutive elements per thread (16-byte loads: a quarter of the instructions for the same bytes); per element the same
        // chains and the same order as the scalar loop below, so the bits do not depend on which loop ran
        const long long n4 = len >> 2, st4 = st >> 2;
        const float4* s4 = reinterpret_cast<const float4*>(s);
        float4* d4 = reinterpret_cast<float4*>(d);
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
            float4 g;
            if (nslabs <= 0) g = s4[i];
            else {
                float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
                int z = 0;
                for (; z + 3 < nslabs; z += 4) {
                    s0 = f4_add(s0, s4[(long long)z * st4 + i]);
                    s1 = f4_add(s1, s4[(long long)(z + 1) * st4 + i]);
                    s2 = f4_add(s2, s4[(long long)(z + 2) * st4 + i]);
                    s3 = f4_add(s3, s4[(long long)(z + 3) * st4 + i]);
                }
                for (; z < nslabs; ++z) s0 = f4_add(s0, s4[(long long)z * st4 + i]);
                g = f4_add(f4_add(s0, s1), f4_add(s2, s3));
                if (t.reduce_only) { d4[i] = g; continue; }
            }
            const float4 w = d4[i];
            const float4 v = make_float4(w.x - alpha * g.x, w.y - alpha * g.y, w.z - alpha * g.z, w.w - alpha * g.w);
            d4[i] = v;
            if (db) reinterpret_cast<uint2*>(db)[i] = make_uint2((uint32_t)bf16_bits(v.x) | ((uint32_t)bf16_bits(v.y) << 16),
                                                                 (uint32_t)bf16_bits(v.z) | ((uint32_t)bf16_bits(v.w) << 16));
        }
        return;
    }
    if (nslabs <= 0) {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x) {
            const float v = d[i] - alpha * s[i];
            d[i] = v;
            if (db) db[i] = bf16_bits(v);
        }
        return;
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int z = 0;
        for (; z + 3 < nslabs; z += 4) {
            s0 += s[(long long)z * st + i];
            s1 += s[(long long)(z + 1) * st + i];
            s2 += s[(long long)(z + 2) * st + i];
            s3 += s[(long long)(z + 3) * st + i];
        }
        for (; z < nslabs; ++z) s0 += s[(long long)z * st + i];
        if (t.reduce_only) { d[i] = (s0 + s1) + (s2 + s3); continue; }
        const float v = d[i] - alpha * ((s0 + s1) + (s2 + s3));
        d[i] = v;
        if (db) db[i] = bf16_bits(v);
    }
}


}  // namespace xq
