// xq_dqn.hip — DQN = online + target NeuralNetwork on the CDNA4 matrix cores (fp32), batched.
//
// Reference: include/dqn.h:42-116, src/dqn.cpp, src/dqn.cu (fp64, batch 1, one thread per output neuron, a
// cudaMalloc/H2D/launch/sync/D2H round trip per layer).  Here the whole TD step of a minibatch stays in HBM:
//   layer 0      : the 1260-wide one-hot input (chessai.cpp:268-289) is never built — rows of W0^T are gathered by
//                  (square, piece) straight from the packed board (<= 32 rows of 1 KB per sample, L2-resident);
//   hidden / Q   : xq_gemm.hip.h MFMA GEMMs with fused bias+tanh;
//   max_a' Q(s') : the 8100-wide output GEMM never writes Q — its epilogue reduces max(z) per row (tanh is monotone);
//   backward     : the TD target equals Q(s) except at action.to < 90 (chessai.cpp:122-128), so the output delta
//                  lives in columns 0..95: delta GEMM with K = 96, weight-gradient GEMM with M = 96;
//   layer-0 grad : one-hot^T x delta as per-(square, piece) segmented sums of delta rows (ordered, in LDS accumulators);
//   reductions over the batch are split-K into slabs + an ordered slab sum: bitwise reproducible, no float atomics.
// Device parameter layout (one flat fp32 buffer per net): [W0^T (L0 x L1)] [W_1 .. W_out, reference layout
// row-major [out][in], concatenated] [b_0 .. b_out].  Keeping layers >= 1 in the reference's flat order lets the
// bug-compatible hidden delta (dqn.cu:406-423 as written: wrong stride, reads across layer boundaries) be expressed
// as the same GEMM with a different base/leading dimension.
#include "xq_internal.h"
#include "xq_gemm.hip.h"
#include "xq_screen.hip.h"
#include "xq_gemm_dma.hip.h"
#include "xq_l0grad.hip.h"

#include <algorithm>
#include <cmath>
#include <random>

namespace xq { struct TailArgs; }
struct xq_dqn {
    int ns = 0, nl = 0;
    int L[XQ_MAX_LAYERS + 1] = {0};
    size_t nw = 0, nb = 0;
    size_t wo[XQ_MAX_LAYERS] = {0}, bo[XQ_MAX_LAYERS] = {0};
    double lr = 1e-3, gamma = 0.99;
    uint64_t seed = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // side stream for work that is independent of the main chain inside one TD step (the s-chain forward runs beside the
    // s'-chain + column-max GEMM; the layer-0 gradient beside the other gradient GEMMs); `cur` = stream launches go to
    hipStream_t side = nullptr;
    hipStream_t cur = nullptr;
    int ncu = 256;                              // compute units of the device (persistent-kernel grid = 2 per CU)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_delta = nullptr;
    hipEvent_t ev_qmax = nullptr;               // recorded behind the column-max GEMM of the last TD step (trainer: collect starts here)
    float* params[2] = {nullptr, nullptr};
    uint64_t params_version = 0;                // ++ whenever an operation that rewrites parameters is queued (xq_trainer: is ev_params still current?)
    // bf16 Q-net (xq_dqn_set_precision): bf16 shadow of the WEIGHTS of both nets, same element order as params[] (biases stay fp32);
    // forward kernels read the shadow, the backward pass and the SGD step work on the fp32 master copy and refresh the shadow
    int precision = XQ_PRECISION_F32;
    uint16_t* params_bf[2] = {nullptr, nullptr};
    uint16_t* acts_bf[XQ_MAX_LAYERS] = {nullptr};      // bf16 bits of acts[] (acts[] then holds the same, rounded, values in fp32)
    uint16_t* tacts_bf[2] = {nullptr, nullptr};
    uint16_t* t2acts_bf[2] = {nullptr, nullptr};
    uint16_t* sel_acts_bf[XQ_MAX_LAYERS] = {nullptr};
    uint16_t* deltas_bf[XQ_MAX_LAYERS] = {nullptr};    // XQ_PRECISION_BF16_FULL: the hidden deltas rounded to bf16 (operands of the backward products)
    int cap_bf = 0, sel_cap_bf = 0, cap_dbf = 0;
    bool bg_ready = false;                             // dynamic-LDS attribute of the gemm_bf16_kernel instances set
    float* t2acts[2] = {nullptr, nullptr};             // third forward chain of a Double-DQN step (s' on the target net)
    int cap_t2 = 0;
    int* partial_idx = nullptr;  int cap_idx = 0;      // row index of each column-max partial (Double DQN)
    // workspaces sized for `cap` samples
    int cap = 0;
    float* acts[XQ_MAX_LAYERS] = {nullptr};     // online hidden activations a_{l+1} = tanh(z_l), l = 0..nl-2
    float* tacts[2] = {nullptr, nullptr};       // ping-pong chain for s' / inference
    float* deltas[XQ_MAX_LAYERS] = {nullptr};   // delta_l of hidden layer l
    float* dsc = nullptr;                       // [cap] the one non-zero output delta of each TD sample
    int32_t* act_mb = nullptr;                  // [cap] action.to of each TD sample (gathered), -1 = empty slot
    float* q90 = nullptr;                       // [cap][96]
    // second workspace for the action-select chain, so that it can run on its own stream beside a TD step
    float* sel_acts[XQ_MAX_LAYERS] = {nullptr};
    float* sel_q90 = nullptr;
    int sel_cap = 0;
    // layer-0 sums of the select chain kept from ply to ply (xq_dqn_set_l0_derive, fp32 net): while the online weights do not change
    // — the plies of one update — the next ply's sums are the kept ones minus the rows of the squares that changed plus the rows of
    // what stands there now (l0_select_kernel); `boards` / `n` = whose boards the kept state belongs to.  One set per stream the chain
    // runs on — [0] the handle's stream (xq_dqn_select_q_dev), [1] any other (the trainer's collect stream) — like qh_slabs: a query on
    // the handle's stream neither races with collects in flight nor changes what their next ply derives from
    struct SelKeep {
        float* z1 = nullptr;  uint32_t* prev_boards = nullptr;  int cap = 0;
        bool valid = false;  const uint32_t* boards = nullptr;  int n = 0;
        int calls = 0;  bool pays = false;                // select calls since the last parameter update; the last period had >= 2
    } sel_keep[2];
    void sel_invalidate() { sel_keep[0].valid = sel_keep[1].valid = false; }
    float* qh_slabs[2] = {nullptr, nullptr};    // k-slabs of the select head (handle stream / any other stream)
    size_t qh_cap[2] = {0, 0};
    bool small_tiles = false;                   // force 64x64 GEMM tiles (<= 80 VGPRs: fits beside the persistent GEMM)
    bool l0_derive = false;                     // xq_dqn_set_l0_derive: layer-0 sums of s' from those of s (online TD rule, fp32 net)
    // exact screening of max_a' Q(s', a') (xq_dqn_set_qmax_mode): bf16 copies of the output-layer weights and of
    // the last hidden activations of s', the two screening partial arrays, the largest row norm of the weights, counters
    int qmax_mode = XQ_QMAX_FULL;
    uint16_t* scr_wb = nullptr;                 // [round_up(nout,128)][hlast] (rows >= nout zero)
    uint16_t* scr_ab = nullptr;  int scr_cap = 0;   // [round_up(cap,512)][hlast]: row-major, or B-fragment order for screen_top2_kernel
    float* scr_p1 = nullptr; float* scr_p2 = nullptr;   // [4*tiles_m][round_up(cap,512)]
    float* scr_R = nullptr; float* scr_na = nullptr;    // [chunks][round_up(cap,512)] per-range maxima, [round_up(cap,512)] ||bf16(a)||^2
    // bits of the largest row norm / largest |bias| of the output layer (non-negative floats order like unsigned): [0..1] rows 0..95 by
    // step parity, [2..3] their biases by step parity, [4] rows >= 96, [5] their biases.  Only rows 0..95 change under the TD rule
    // (xq_dqn_td_grads never touches the others), so the shadow of rows >= 96 and slots [4], [5] are kept from step to step:
    unsigned* scr_wmax = nullptr;
    int scr_static_net = -1;                    // net whose rows >= 96 the shadow holds (-1: none — the next step converts everything)
    bool scr_new_kernel_ready = false;          // dynamic-LDS attribute of screen_top2_kernel set
    unsigned long long* scr_stats = nullptr;    // [refine blocks][2] running totals per block: candidate (sample, group) pairs, pairs recomputed as whole groups
    int scr_stat_blocks = 0;                    // blocks the array (and its pinned copy) has room for
    unsigned long long scr_carry[2] = {0, 0};   // totals of an array that was replaced by a larger one
    unsigned long long scr_host_steps = 0, scr_host_samples = 0;
    // guard: every kScreenCheckEvery screened steps the candidate counters are copied back asynchronously and evaluated kScreenCheckEvery
    // steps later; a net that leaves the screen too many candidates (outputs all within the bf16 bound of each other) gets the full
    // product for the next kScreenHoldSteps steps
    unsigned long long* scr_guard_host = nullptr;      // pinned copy of scr_stats
    hipEvent_t scr_guard_ev = nullptr;
    bool scr_guard_pending = false;
    unsigned long long scr_guard_samples = 0;          // scr_host_samples when the pending copy was queued
    unsigned long long scr_guard_queued_at = 0;        // scr_host_steps when it was queued
    unsigned long long scr_seen[3] = {0, 0, 0};        // samples, pairs, whole groups at the last evaluation
    int scr_hold = 0;                                  // > 0: that many TD steps still run the full product
    unsigned long long scr_fallbacks = 0;
    float* partial = nullptr;                   // row-max partials
    float* zmax = nullptr;  int* zidx = nullptr;    // [kReduceParts][cap] their reduction per sample (colmax_reduce_kernel)
    float* qsa = nullptr;
    float* yv = nullptr;
    float* lossv = nullptr;
    uint32_t* gboards = nullptr;                // [cap][12] boards of the current minibatch, gathered
    int last_n = 0;
    // gradients
    float* grads_td = nullptr;  size_t n_grads_td = 0;
    size_t g_w0 = 0, g_wh[XQ_MAX_LAYERS] = {0}, g_wout = 0, g_bh[XQ_MAX_LAYERS] = {0}, g_bout = 0;
    float* grads_full = nullptr;
    float* slabs = nullptr;  size_t slabs_cap = 0;
    float* slabs_l0 = nullptr;  size_t slabs_l0_cap = 0;     // layer-0 gradient partials
    // layer-0 gradient on the bf16 matrix pipe (xq_l0grad.hip.h): delta_0 as three bf16 planes, transposed [plane][column][sample]
    uint16_t* l0_planes = nullptr;  size_t l0_planes_cap = 0;
    bool l0_mfma = false;                       // xq_dqn_set_l0_grad_mode (opt-in: faster alone, not inside the fused launch — DESIGN.md section 5)
    xq_comm* comm = nullptr;                    // xq_dqn_set_comm: bucketed RCCL all-reduce of the gradient buffer inside td_grads
    bool fused_apply = false;                   // xq_dqn_set_fused_apply: apply_grads may sum the layer-0 partials itself
    int l0_pending = 0;                         // > 0: that many layer-0 slabs wait in slabs_l0, not yet reduced into grads_td
    struct PendingSlab { const float* src = nullptr; int nslabs = 0; long long stride = 0; };
    PendingSlab pend_hidden[XQ_MAX_LAYERS], pend_wout, pend_bout;   // same for the hidden / output-layer gradients (fused_apply)
    PendingSlab pend_bh;                        // ... and the hidden biases (rows of column sums, bias_grads)
    float* bias_work = nullptr;  size_t bias_work_cap = 0;
    // dense API scratch
    float* xdense = nullptr;  size_t xdense_cap = 0;
    float* qfull = nullptr;   size_t qfull_cap = 0;
    float* tfull = nullptr;   size_t tfull_cap = 0;
    uint32_t* hb = nullptr;   size_t hb_cap = 0;     // host-batch staging: boards, next boards
    int32_t* ha = nullptr; float* hr = nullptr; uint8_t* hd = nullptr;
    xq::Profiler prof;
    // fused launches of the TD step's tail (td_tail_kernel): while `tail_open`, the launch helpers of the small fp32 GEMMs, the
    // output-layer / layer-0 segmented sums and the bias column sums append their blocks to `tail` instead of launching
    xq::TailArgs* tail = nullptr;
    bool tail_open = false;
    bool td_tail = true;                        // xq_dqn_set_td_tail
    int exchange_overlap = -1;                  // xq_dqn_set_exchange_overlap: -1 auto (on when the communicator has more than one rank), 0, 1
    bool late_gate = false;                     // this TD step records ev_qmax behind its gradients (see tail_gradients)
    size_t tail_lds = 0;  double tail_flops = 0, tail_bytes = 0;

    // partial-sum slabs may stay unreduced until the SGD kernel only when nothing (an all-reduce) reads the buffer in between
    bool force_defer = false;   // the fused launches of the gradient half leave their partial sums pending whatever follows (tail_gradients)
    bool fused() const { return (fused_apply && comm == nullptr) || force_defer; }
    bool bf16() const { return precision != XQ_PRECISION_F32; }               // bf16 forward passes
    bool bf16_bwd() const { return precision == XQ_PRECISION_BF16_FULL; }     // ... and bf16 operands in the backward products
    uint16_t* wl_bf(int net, int l) const { return params_bf[net] + (l == 0 ? 0 : (size_t)L[0] * L[1] + (wo[l] - wo[1])); }
    float* w0t(int net) const { return params[net]; }
    float* wrest(int net) const { return params[net] + (size_t)L[0] * L[1]; }     // layers 1.. in reference flat order
    float* wl(int net, int l) const { return l == 0 ? w0t(net) : wrest(net) + (wo[l] - wo[1]); }
    float* bl(int net, int l) const { return params[net] + nw + bo[l]; }
    int nout() const { return L[nl]; }
    int hlast() const { return L[nl - 1]; }
};

namespace xq {

Profiler* dqn_profiler(xq_dqn* d) { return &d->prof; }
int dqn_fused_apply(const xq_dqn* d) { return d->fused_apply ? 1 : 0; }
xq_comm* dqn_comm(const xq_dqn* d) { return d->comm; }
hipStream_t dqn_stream(xq_dqn* d) { return d->stream; }
hipEvent_t dqn_qmax_event(xq_dqn* d) { return d->ev_qmax; }
uint64_t dqn_params_version(const xq_dqn* d) { return d->params_version; }

struct ProfScope {
    Profiler& p; hipStream_t s; int h; double flops, bytes;
    // (while a fused tail launch is being assembled nothing is launched, so nothing is bracketed: tail_launch has its own scope)
    // A launch on a stream that is neither the handle's nor its side stream belongs to the select chain of the self-play loop (the
    // trainer's collect stream): same kernels, other shapes, off the critical path — bracketed under "<name>@select".
    ProfScope(xq_dqn* d, const char* name, double fl, double by) : p(d->prof), s(d->cur), flops(fl), bytes(by) {
        h = -1;
        if (d->tail_open || !p.enabled) return;
        if (d->cur != d->stream && d->cur != d->side) {
            char nm[48];
            snprintf(nm, sizeof nm, "%s@select", name);
            h = p.begin(nm, s);
        } else {
            h = p.begin(name, s);
        }
    }
    ~ProfScope() { p.end(h, s, flops, bytes); }
};

// ---------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------

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
    __shared__ int rows[4][96];
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
    __shared__ int dpair[4][8][2];
    __shared__ int rows2[4][96];
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
            float4 t = make_float4(tanhf(acc.x), tanhf(acc.y), tanhf(acc.z), tanhf(acc.w));
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
                const float4 t2 = make_float4(tanhf(a2.x), tanhf(a2.y), tanhf(a2.z), tanhf(a2.w));
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
            float t = tanhf(acc);
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
        *reinterpret_cast<float4*>(out + (long long)b * H + col) = make_float4(tanhf(acc.x), tanhf(acc.y), tanhf(acc.z), tanhf(acc.w));
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
enum { TAIL_DELTA = 1, TAIL_GRAD = 2, TAIL_OUT = 4, TAIL_COLSUM = 8, TAIL_L0 = 16 };
struct TailArgs {
    // grid order: [l0][grad][delta][out][colsum]; n_* = blocks of each part (0 = absent)
    int n_l0, n_grad, n_delta, n_out, n_colsum;
    GemmArgs grad;  int grad_gx, grad_gy;          // 64x64 tiles: grid (gx, gy, splits)
    GemmArgs delta; int delta_gx;                  // grid (gx, gy, 1)
    const int32_t* og_act; const float* og_dsc; const float* og_alast; int og_n, og_H, og_chunk; float* og_partial;   // grid (24, chunks)
    const uint32_t* l0_boards; const float* l0_delta; int l0_n, l0_H, l0_HS, l0_chunk, l0_nsets, l0_nch; float* l0_partial;   // (90, nch, H / HS)
    const uint16_t* l0_planes; long long l0_plane_stride; int l0_kpad, l0_ncb;      // != nullptr: the matrix-pipe form, grid (4, H / 32, nch)
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
                const int rest = b >> 2;
                l0_grad_mfma_block<0>(a.l0_boards, a.l0_planes, a.l0_plane_stride, a.l0_kpad, a.l0_n, a.l0_H, a.l0_chunk, a.l0_partial, b & 3,
                                      rest % a.l0_ncb, rest / a.l0_ncb, reinterpret_cast<uint32_t*>(tail_smem));
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
            const int per = a.grad_gx * a.grad_gy, r = b % per;
            gemm_f32_block<L_MCONTIG, L_MCONTIG, EPI_STORE, 1, 1>(a.grad, r % a.grad_gx, r / a.grad_gx, b / per, tail_smem, tail_smem + g_tile_floats(64));
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
        }
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
};
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
template <int KFIX, bool TD = false>
__global__ __launch_bounds__(256) void qmax_refine2_kernel(const float* __restrict__ R, int ranges, int gpr /* groups per range */,
                                                           const float* __restrict__ P1, const float* __restrict__ P2, int G, int n, long long ldp,
                                                           const float* __restrict__ na_all, const float* __restrict__ a_last, int K,
                                                           const float* __restrict__ W, const float* __restrict__ bias, int NO,
                                                           unsigned* __restrict__ wm, int parity, float* __restrict__ zmax,
                                                           unsigned long long* __restrict__ stats, const TdFused T) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cand[];            // [G * 32]: sample | row << 5
    uint16_t* wlist = reinterpret_cast<uint16_t*>(cand + (size_t)G * kRefineSamples);    // [G * 32]: sample | group << 5
    __shared__ float sv[8][32];
    __shared__ float thr[32];
    __shared__ int best[32];
    __shared__ int cnt, nexp;
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
            for (int g = g0; g < g1; g += 4) {                   // four independent loads in flight
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = P1[(long long)min(g + u, g1 - 1) * ldp + bc];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (g + u < g1 && v[u] >= t) {
                        const float v2 = P2[(long long)(g + u) * ldp + bc];
                        if (v2 >= t) wlist[atomicAdd(&nexp, 1)] = (uint16_t)(sl | ((g + u) << 5));
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
__global__ void sgd_segments_kernel(SegTable t, float alpha) {
    const int sgm = (int)blockIdx.y;
    if (sgm >= t.nseg) return;
    float* d = t.dst[sgm];
    const float* s = t.src[sgm];
    const int nslabs = t.nslabs[sgm];
    const long long len = t.len[sgm], st = t.stride[sgm];
    uint16_t* db = t.dst_bf[sgm];
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

// ---------------------------------------------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------------------------------------------
static inline int vec_ok(const void* p, long long ld) { return (((uintptr_t)p) % 16 == 0) && (ld % 4 == 0); }

// tile choice: 128x128 when that grid already fills the chip twice over, else 64x64 (4x the blocks)
template <int AL, int BL, int EPI, int DT = DT_F32>
static int launch_gemm(xq_dqn* d, GemmArgs g, int splits, const char* name, int* used_splits = nullptr, bool force_small = false,
                       bool force_big = false) {
    if (used_splits) *used_splits = 0;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return fail(XQ_ERR_INVALID_ARGUMENT, "empty GEMM %s (%d x %d x %d)", name, g.M, g.N, g.K);
    g.a_vec = vec_ok(g.A, g.lda);
    g.b_vec = vec_ok(g.B, g.ldb);
    if (splits < 1) splits = 1;
    g.k_chunk = round_up((g.K + splits - 1) / splits, GBK);
    splits = (g.K + g.k_chunk - 1) / g.k_chunk;
    if (splits < 1) splits = 1;
    const int groups = g.grouped > 1 ? g.grouped : 1;
    const long long t128 = (long long)((g.M + 127) / 128) * ((g.N + 127) / 128) * (g.grouped ? groups : splits);
    const bool big = force_big || (!force_small && !d->small_tiles && t128 >= 512);
    const double kk = DT == DT_BF16 ? 2.0 * g.K : (double)g.K;       // bf16: K counts pairs
    if (d->tail_open) {                  // fused tail launch (td_tail_kernel): the blocks join the open grid instead of launching
        TailArgs& T = *d->tail;
        const bool is_delta = AL == L_KCONTIG && BL == L_MCONTIG && EPI == EPI_DELTA, is_grad = AL == L_MCONTIG && BL == L_MCONTIG && EPI == EPI_STORE;
        if (DT != DT_F32 || big || g.grouped || !(is_delta || is_grad) || (is_delta && (T.n_delta || splits != 1)) || (is_grad && T.n_grad))
            return fail(XQ_ERR_RUNTIME, "GEMM %s cannot join the fused tail launch", name);
        const int gx = (g.M + 63) / 64, gy = (g.N + 63) / 64;
        if (is_delta) { T.delta = g; T.delta_gx = gx; T.n_delta = gx * gy; }
        else { T.grad = g; T.grad_gx = gx; T.grad_gy = gy; T.n_grad = gx * gy * splits; }
        d->tail_flops += 2.0 * g.M * g.N * kk;
        d->tail_bytes += 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N);
        d->tail_lds = std::max(d->tail_lds, 2 * (size_t)g_tile_floats(64) * sizeof(float));
        if (used_splits) *used_splits = splits;
        return XQ_OK;
    }
    ProfScope ps(d, name, 2.0 * g.M * g.N * kk, 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N));
    if (g.grouped) {
        if (splits != 1) return fail(XQ_ERR_INVALID_ARGUMENT, "grouped GEMM cannot be split-K");
        if (groups > 3) return fail(XQ_ERR_INVALID_ARGUMENT, "at most 3 grouped products");
        for (int k = 0; k + 1 < groups; ++k) {
            g.b_vec = g.b_vec && vec_ok(g.Bx[k], g.ldb);
            g.a_vec = g.a_vec && vec_ok(g.Ax[k], g.lda);
        }
        ps.flops *= groups; ps.bytes *= groups;
    }
    const int gz = g.grouped ? groups : splits;
    if (big) {
        dim3 grid((g.M + 127) / 128, (g.N + 127) / 128, gz);
        hipLaunchKernelGGL((gemm_f32_kernel<AL, BL, EPI, 2, 2, DT>), grid, dim3(256), 0, d->cur, g);
    } else {
        dim3 grid((g.M + 63) / 64, (g.N + 63) / 64, gz);
        hipLaunchKernelGGL((gemm_f32_kernel<AL, BL, EPI, 1, 1, DT>), grid, dim3(256), 0, d->cur, g);
    }
    XQ_HIP(hipGetLastError());
    if (used_splits) *used_splits = splits;
    return XQ_OK;
}
#define XQ_GEMM(expr) XQ_TRY(expr)

// gemm_dma_kernel (xq_gemm_dma.hip.h): (128 TI) x (64 TJ) tiles, up to 144 KB of dynamic LDS (attribute set once per instance)
template <int DT, int AL, int BL, int EPI, int TI, int TJ>
static int launch_dma_gemm(xq_dqn* d, const Bf16GemmArgs& g, int gz, const char* name) {
    static bool ready = false;
    constexpr int lds = bg_lds_bytes(TI, TJ);
    if (!ready) {
        XQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma_kernel<DT, AL, BL, EPI, TI, TJ>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        ready = true;
    }
    const int groups = g.groups > 1 ? g.groups : 1;
    const double es = DT == DT_F32 ? 4.0 : 2.0;
    ProfScope ps(d, name, 2.0 * g.M * g.N * (double)g.K * groups, es * groups * ((double)g.M * g.K + (double)g.N * g.K + 2.0 * g.M * g.N));
    hipLaunchKernelGGL((gemm_dma_kernel<DT, AL, BL, EPI, TI, TJ>), dim3(g.M / (128 * TI), g.N / (64 * TJ), gz), dim3(512), lds, d->cur, g);
    XQ_HIP(hipGetLastError());
    return XQ_OK;
}
template <int AL, int BL, int EPI>
static int launch_bf16_gemm(xq_dqn* d, const Bf16GemmArgs& g, int gz, const char* name) {
    return launch_dma_gemm<DT_BF16, AL, BL, EPI, 2, 2>(d, g, gz, name);
}

static int ensure_slabs(xq_dqn* d, size_t floats) {
    if (floats <= d->slabs_cap) return XQ_OK;
    if (d->slabs) { XQ_HIP(hipStreamSynchronize(d->stream)); XQ_HIP(hipFree(d->slabs)); }
    XQ_HIP(hipMalloc(&d->slabs, floats * sizeof(float)));
    d->slabs_cap = floats;
    return XQ_OK;
}

static int grow(float** p, size_t* cap, size_t floats, hipStream_t s) {
    if (floats <= *cap) return XQ_OK;
    if (*p) { XQ_HIP(hipStreamSynchronize(s)); XQ_HIP(hipFree(*p)); }
    XQ_HIP(hipMalloc(p, floats * sizeof(float)));
    *cap = floats;
    return XQ_OK;
}

static int ensure_capacity(xq_dqn* d, int n) {
    if (n <= d->cap) return XQ_OK;
    XQ_HIP(hipStreamSynchronize(d->stream));
    const size_t cap = (size_t)n;
    int maxh = 0;
    for (int l = 0; l + 1 < d->nl; ++l) maxh = std::max(maxh, d->L[l + 1]);
    for (int l = 0; l + 1 < d->nl; ++l) {
        if (d->acts[l]) XQ_HIP(hipFree(d->acts[l]));
        if (d->deltas[l]) XQ_HIP(hipFree(d->deltas[l]));
        XQ_HIP(hipMalloc(&d->acts[l], cap * d->L[l + 1] * sizeof(float)));
        XQ_HIP(hipMalloc(&d->deltas[l], cap * d->L[l + 1] * sizeof(float)));
    }
    for (int i = 0; i < 2; ++i) {    // rows padded to a multiple of 128 (whole-tile reads of the persistent GEMM)
        if (d->tacts[i]) XQ_HIP(hipFree(d->tacts[i]));
        const size_t rows = (size_t)round_up(n, 128);
        XQ_HIP(hipMalloc(&d->tacts[i], rows * (size_t)maxh * sizeof(float)));
        XQ_HIP(hipMemsetAsync(d->tacts[i], 0, rows * (size_t)maxh * sizeof(float), d->stream));
    }
    const int ntn = (d->nout() + 63) / 64;             // column-max partials: 2 per 64- or 128-row tile
    float** bufs[] = {&d->q90, &d->partial, &d->qsa, &d->yv, &d->lossv, &d->dsc, reinterpret_cast<float**>(&d->act_mb), &d->zmax,
                      reinterpret_cast<float**>(&d->zidx)};
    // (partials: the sample dimension padded to whole blocks of screen_top2_kernel, which stores every column of its panels)
    const size_t sizes[] = {cap * 96, (size_t)round_up(n, 512) * (size_t)ntn * 2, cap, cap, cap, cap, cap, cap * kReduceParts, cap * kReduceParts};
    for (int i = 0; i < 9; ++i) {
        if (*bufs[i]) XQ_HIP(hipFree(*bufs[i]));
        XQ_HIP(hipMalloc(bufs[i], sizes[i] * sizeof(float)));
    }
    if (d->gboards) XQ_HIP(hipFree(d->gboards));
    XQ_HIP(hipMalloc(&d->gboards, cap * kBoardWords * sizeof(uint32_t)));
    d->cap = n;
    return XQ_OK;
}

// the extra buffers of the build-defined modes, allocated on first use: bf16 bits of every activation buffer (bf16 Q-net), the
// third forward chain and the arg-max partials (Double DQN)
static int ensure_ext_capacity(xq_dqn* d, int n, bool want_double) {
    int maxh = 0;
    for (int l = 0; l + 1 < d->nl; ++l) maxh = std::max(maxh, d->L[l + 1]);
    const size_t rows = (size_t)round_up(n, 512);      // whole-tile reads of the persistent GEMM (128) / whole panels of screen_top2_kernel (512)
    if (d->bf16() && n > d->cap_bf) {
        XQ_HIP(hipDeviceSynchronize());
        for (int l = 0; l + 1 < d->nl; ++l) {
            if (d->acts_bf[l]) XQ_HIP(hipFree(d->acts_bf[l]));
            XQ_HIP(hipMalloc(&d->acts_bf[l], rows * d->L[l + 1] * sizeof(uint16_t)));
        }
        for (int i = 0; i < 2; ++i) {
            if (d->tacts_bf[i]) XQ_HIP(hipFree(d->tacts_bf[i]));
            if (d->t2acts_bf[i]) XQ_HIP(hipFree(d->t2acts_bf[i]));
            XQ_HIP(hipMalloc(&d->tacts_bf[i], rows * (size_t)maxh * sizeof(uint16_t)));
            XQ_HIP(hipMalloc(&d->t2acts_bf[i], rows * (size_t)maxh * sizeof(uint16_t)));
            XQ_HIP(hipMemsetAsync(d->tacts_bf[i], 0, rows * (size_t)maxh * sizeof(uint16_t), d->stream));
            XQ_HIP(hipMemsetAsync(d->t2acts_bf[i], 0, rows * (size_t)maxh * sizeof(uint16_t), d->stream));
        }
        d->cap_bf = n;
    }
    if (d->bf16_bwd() && n > d->cap_dbf) {
        XQ_HIP(hipDeviceSynchronize());
        for (int l = 0; l + 1 < d->nl; ++l) {
            if (d->deltas_bf[l]) XQ_HIP(hipFree(d->deltas_bf[l]));
            XQ_HIP(hipMalloc(&d->deltas_bf[l], rows * d->L[l + 1] * sizeof(uint16_t)));
        }
        d->cap_dbf = n;
    }
    if (want_double && n > d->cap_t2) {
        XQ_HIP(hipDeviceSynchronize());
        for (int i = 0; i < 2; ++i) {
            if (d->t2acts[i]) XQ_HIP(hipFree(d->t2acts[i]));
            XQ_HIP(hipMalloc(&d->t2acts[i], rows * (size_t)maxh * sizeof(float)));
            XQ_HIP(hipMemsetAsync(d->t2acts[i], 0, rows * (size_t)maxh * sizeof(float), d->stream));
        }
        d->cap_t2 = n;
    }
    if (want_double && n > d->cap_idx) {
        XQ_HIP(hipDeviceSynchronize());
        if (d->partial_idx) XQ_HIP(hipFree(d->partial_idx));
        const int ntn = (d->nout() + 63) / 64;
        XQ_HIP(hipMalloc(&d->partial_idx, (size_t)round_up(n, 512) * (size_t)ntn * 2 * sizeof(int)));
        d->cap_idx = n;
    }
    return XQ_OK;
}

static SlotSrc explicit_slots(const int32_t* slots) {
    SlotSrc s; memset(&s, 0, sizeof s); s.slots = slots; return s;
}

// One forward chain of a launch group: a_1 .. a_{nl-1} of `net` for n packed boards; outs[l] receives a_{l+1} in fp32 (may be
// nullptr per chain in bf16 mode when only the next layer reads it), outs_bf[l] its bf16 bits (bf16 Q-net only).
// totals of the per-block candidate counters (synchronises the device)
static int screen_stat_sums(xq_dqn* d, unsigned long long sums[2]) {
    sums[0] = d->scr_carry[0]; sums[1] = d->scr_carry[1];
    if (!d->scr_stats) return XQ_OK;
    std::vector<unsigned long long> h((size_t)2 * d->scr_stat_blocks);
    XQ_HIP(hipDeviceSynchronize());
    XQ_HIP(hipMemcpy(h.data(), d->scr_stats, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < d->scr_stat_blocks; ++i) { sums[0] += h[2 * i]; sums[1] += h[2 * i + 1]; }
    return XQ_OK;
}
static void screen_stat_sums_of(const xq_dqn* d, const unsigned long long* h, unsigned long long sums[2]) {
    sums[0] = d->scr_carry[0]; sums[1] = d->scr_carry[1];
    for (int i = 0; i < d->scr_stat_blocks; ++i) { sums[0] += h[2 * i]; sums[1] += h[2 * i + 1]; }
}
// buffers of the screened maximum (xq_dqn_set_qmax_mode), allocated on first use
static int ensure_screen_capacity(xq_dqn* d, int n) {
    const int NO = d->nout(), Hl = d->hlast();
    const size_t wrows = (size_t)round_up(NO, 128);
    if (!d->scr_wb) {
        XQ_HIP(hipMalloc(&d->scr_wb, wrows * Hl * sizeof(uint16_t)));
        XQ_HIP(hipMemset(d->scr_wb, 0, wrows * Hl * sizeof(uint16_t)));
        XQ_HIP(hipMalloc(&d->scr_wmax, 6 * sizeof(unsigned)));
        XQ_HIP(hipMemset(d->scr_wmax, 0, 6 * sizeof(unsigned)));
        XQ_HIP(hipEventCreateWithFlags(&d->scr_guard_ev, hipEventDisableTiming));
        d->scr_static_net = -1;
    }
    if (n > d->scr_cap) {
        XQ_HIP(hipDeviceSynchronize());
        // sample dimension padded to whole blocks of screen_top2_kernel (512): it stores every column of its panels unconditionally
        const size_t rows = (size_t)round_up(n, 512), G = (size_t)4 * ((NO + 127) / 128);
        if (d->scr_ab) XQ_HIP(hipFree(d->scr_ab));
        if (d->scr_p1) XQ_HIP(hipFree(d->scr_p1));
        if (d->scr_p2) XQ_HIP(hipFree(d->scr_p2));
        XQ_HIP(hipMalloc(&d->scr_ab, rows * Hl * sizeof(uint16_t)));
        XQ_HIP(hipMemset(d->scr_ab, 0, rows * Hl * sizeof(uint16_t)));
        XQ_HIP(hipMalloc(&d->scr_p1, G * rows * sizeof(float)));
        XQ_HIP(hipMalloc(&d->scr_p2, G * rows * sizeof(float)));
        if (d->scr_R) XQ_HIP(hipFree(d->scr_R));
        if (d->scr_na) XQ_HIP(hipFree(d->scr_na));
        XQ_HIP(hipMalloc(&d->scr_R, (size_t)((NO + 63) / 64) * rows * sizeof(float)));       // at most one range per 64-row chunk
        XQ_HIP(hipMalloc(&d->scr_na, rows * sizeof(float)));
        // candidate counters: one pair per refine block (32 samples); the totals of the array it replaces are carried on the host
        if (d->scr_stats) {
            unsigned long long sums[2];
            XQ_TRY(screen_stat_sums(d, sums));
            d->scr_carry[0] = sums[0]; d->scr_carry[1] = sums[1];
            XQ_HIP(hipFree(d->scr_stats)); XQ_HIP(hipHostFree(d->scr_guard_host));
            d->scr_guard_pending = false;
        }
        d->scr_stat_blocks = (int)(rows / kRefineSamples);
        XQ_HIP(hipMalloc(&d->scr_stats, (size_t)2 * d->scr_stat_blocks * sizeof(unsigned long long)));
        XQ_HIP(hipMemset(d->scr_stats, 0, (size_t)2 * d->scr_stat_blocks * sizeof(unsigned long long)));
        XQ_HIP(hipHostMalloc(reinterpret_cast<void**>(&d->scr_guard_host), (size_t)2 * d->scr_stat_blocks * sizeof(unsigned long long), hipHostMallocDefault));
        memset(d->scr_guard_host, 0, (size_t)2 * d->scr_stat_blocks * sizeof(unsigned long long));
        d->scr_cap = n;
    }
    return XQ_OK;
}

struct ChainJob {
    int net;
    const uint32_t* boards;
    float* const* outs;
    uint16_t* const* outs_bf;
    uint32_t* gathered;
    uint16_t* last_bf;          // fp32 net: != nullptr => bf16 copy of the chain's LAST hidden activations (screening operand)
    bool last_bf_frag;          //   ... written in MFMA B-fragment order (scr_afrag_index) for screen_top2_kernel
    float* head_slabs;          // fp32 net, one chain, >= 2 hidden layers: != nullptr => the select head's k-slabs [hlast / 64][n][96] come
                                //   out of the last hidden product (EPI_HEAD); the last activations are stored only if outs[nl-2] != nullptr
    int sel_keep;               // fp32 net, one chain: 1 = layer 0 through l0_select_kernel (sums kept in sel_state->z1 for the next ply),
                                //   2 = ... and derived from the sums kept last time
};
// Up to three chains run in the same launches: one gather grid with blockIdx.y = chain, grouped GEMMs with blockIdx.z = chain.
static int chain_boards(xq_dqn* d, const ChainJob* jobs, int njobs, SlotSrc src, int n, const ShadowJob* shadow = nullptr) {
    if (d->L[0] != kStateSize) return fail(XQ_ERR_INVALID_ARGUMENT, "board input needs layer_sizes[0] == 1260 (got %d)", d->L[0]);
    if (njobs < 1 || njobs > kMaxChains) return fail(XQ_ERR_INVALID_ARGUMENT, "1..3 forward chains per launch group");
    const bool bf = d->bf16();
    if (njobs == 1 && jobs[0].sel_keep && !bf && !shadow && !src.implicit && !src.slots) {
        const int H = d->L[1];
        ProfScope ps(d, "l0_forward_gather", 2.0 * n * 32 * H, (double)n * (96 + (jobs[0].sel_keep == 2 ? 4.0 : 32.0) * H * 4 + H * 12));
        xq_dqn::SelKeep& K = d->sel_keep[d->cur == d->stream ? 0 : 1];
        hipLaunchKernelGGL(l0_select_kernel, dim3((n + 3) / 4), dim3(256), 0, d->cur, jobs[0].boards, K.prev_boards, d->w0t(jobs[0].net),
                           d->bl(jobs[0].net, 0), K.z1, jobs[0].outs[0], n, H, jobs[0].sel_keep == 2 ? 1 : 0);
        XQ_HIP(hipGetLastError());
    } else
    {
        const int H = d->L[1];
        L0Jobs J; memset(&J, 0, sizeof J);
        J.njobs = njobs;
        for (int k = 0; k < njobs; ++k) {
            J.boards[k] = jobs[k].boards; J.W0T[k] = d->w0t(jobs[k].net); J.b0[k] = d->bl(jobs[k].net, 0);
            J.out[k] = jobs[k].outs ? jobs[k].outs[0] : nullptr; J.gathered[k] = jobs[k].gathered;
            if (bf) { J.W0T_bf[k] = d->wl_bf(jobs[k].net, 0); J.out_bf[k] = jobs[k].outs_bf[0]; }
            else if (d->nl == 2) { J.out_bf[k] = jobs[k].last_bf; if (jobs[k].last_bf && jobs[k].last_bf_frag) J.out_bf_frag = 1; }
        }
        // the screening shadow of the output-layer weights rides in the same grid (one more row of blocks) when the grid is wide
        // enough for it; a launch of its own otherwise
        // online TD rule on an fp32 net: the s' chain (job 1, same net, same slots) is derived inside job 0's waves
        // (bf16 net: in the 16-byte gather path only; Double DQN's third chain — the target net — is gathered as before)
        const bool wide_bf = bf && (H & 7) == 0 && ((H >= 512 && (H & 511) == 0) || (H >= 64 && 512 % H == 0));
        const bool derive = d->l0_derive && njobs >= 2 && jobs[0].net == jobs[1].net && jobs[1].gathered == nullptr &&
                            (bf ? wide_bf : (njobs == 2 && (H & 3) == 0));
        J.derive_next = derive ? 1 : 0;
        J.nrows = njobs - (derive ? 1 : 0);
        bool ride = false;
        if (shadow) {
            const int sblocks = shadow->nblocks;
            ride = sblocks <= (n + 3) / 4;
            if (ride) J.shadow = *shadow;
            else {
                hipLaunchKernelGGL(screen_shadow_kernel, dim3(sblocks), dim3(256), 0, d->cur, *shadow);
                XQ_HIP(hipGetLastError());
            }
        }
        ProfScope ps(d, "l0_forward_gather", 2.0 * njobs * n * 32 * H, (double)njobs * n * (48 + 32.0 * H * (bf ? 2 : 4) + H * 4));
        // (grid rows: the job rows that are really gathered, then the shadow row; the kernel tests blockIdx.y == J.nrows for it)
        if (bf) hipLaunchKernelGGL(l0_forward_kernel<true>, dim3((n + 3) / 4, J.nrows), dim3(256), 0, d->cur, J, src, n, H);
        else hipLaunchKernelGGL(l0_forward_kernel<false>, dim3((n + 3) / 4, J.nrows + (ride ? 1 : 0)), dim3(256), 0, d->cur, J, src, n, H);
        XQ_HIP(hipGetLastError());
    }
    for (int l = 1; l + 1 < d->nl; ++l) {
        GemmArgs g; memset(&g, 0, sizeof g);
        g.M = n; g.N = d->L[l + 1];
        g.grouped = njobs > 1 ? njobs : 0;
        if (bf && (n % kBgBM) == 0 && (d->L[l + 1] % kBgBN) == 0 && (d->L[l] % kBgBK) == 0) {
            // the bf16 loop of its own (xq_gemm_bf16.hip.h): whole 256 x 128 tiles only, the chains as groups of one launch
            Bf16GemmArgs b; memset(&b, 0, sizeof b);
            b.M = n; b.N = d->L[l + 1]; b.K = d->L[l]; b.lda = b.ldb = d->L[l]; b.k_chunk = b.K;
            b.ldc = d->L[l + 1]; b.ldcb = d->L[l + 1];
            b.groups = njobs;
            for (int k = 0; k < njobs; ++k) {
                const uint16_t* A = jobs[k].outs_bf[l - 1];
                const uint16_t* B = d->wl_bf(jobs[k].net, l);
                float* C = jobs[k].outs ? jobs[k].outs[l] : nullptr;
                if (k == 0) { b.A = A; b.B = B; b.C = C; b.bias = d->bl(jobs[k].net, l); b.Cb = jobs[k].outs_bf[l]; }
                else { b.Ax[k - 1] = A; b.Bx[k - 1] = B; b.Cx[k - 1] = C; b.biasx[k - 1] = d->bl(jobs[k].net, l); b.Cbx[k - 1] = jobs[k].outs_bf[l]; }
                // the chain's last activations in B-fragment order when only screen_top2_kernel reads them (bf16 net: max / arg-max pass)
                if (l == d->nl - 2 && jobs[k].last_bf_frag) b.cb_frag_mask |= 1 << k;
            }
            XQ_TRY((launch_bf16_gemm<L_KCONTIG, L_KCONTIG, BG_TANH>(d, b, njobs, "gemm_hidden_fwd")));
        } else if (bf) {
            if (d->L[l] & 1) return fail(XQ_ERR_INVALID_ARGUMENT, "bf16 Q-net needs even layer widths (layer %d has %d)", l, d->L[l]);
            g.K = d->L[l] / 2; g.lda = g.ldb = d->L[l] / 2;
            g.ldc = d->L[l + 1]; g.ldcb = d->L[l + 1];
            for (int k = 0; k < njobs; ++k) {
                const float* A = reinterpret_cast<const float*>(jobs[k].outs_bf[l - 1]);
                const float* B = reinterpret_cast<const float*>(d->wl_bf(jobs[k].net, l));
                float* C = jobs[k].outs ? jobs[k].outs[l] : nullptr;
                if (k == 0) { g.A = A; g.B = B; g.C = C; g.bias = d->bl(jobs[k].net, l); g.Cb = jobs[k].outs_bf[l]; }
                else { g.Ax[k - 1] = A; g.Bx[k - 1] = B; g.Cx[k - 1] = C; g.biasx[k - 1] = d->bl(jobs[k].net, l); g.Cbx[k - 1] = jobs[k].outs_bf[l]; }
            }
            XQ_GEMM((launch_gemm<L_KCONTIG, L_KCONTIG, EPI_BIAS_TANH, DT_BF16>(d, g, 1, "gemm_hidden_fwd")));
        } else {
            g.K = d->L[l]; g.lda = g.ldb = d->L[l]; g.ldc = d->L[l + 1]; g.ldcb = d->L[l + 1];
            if (l == d->nl - 2 && njobs == 1 && jobs[0].head_slabs) {
                if ((n & 63) || (g.N & 63) || (g.K & 31) || d->nout() < 128) return fail(XQ_ERR_RUNTIME, "select head cannot ride on this hidden product");
                g.A = jobs[0].outs[l - 1]; g.B = d->wl(jobs[0].net, l); g.C = jobs[0].outs[l]; g.bias = d->bl(jobs[0].net, l);
                g.a_vec = vec_ok(g.A, g.lda); g.b_vec = vec_ok(g.B, g.ldb); g.k_chunk = g.K;
                g.head_W = d->wl(jobs[0].net, d->nl - 1); g.head_ldw = g.N;
                g.head_slabs = jobs[0].head_slabs; g.head_slab_stride = (long long)n * 96; g.head_ld = 96;
                if (!g.a_vec || !g.b_vec || !vec_ok(g.head_W, g.head_ldw)) return fail(XQ_ERR_RUNTIME, "select head: unaligned operand");
                ProfScope ps(d, "gemm_hidden_fwd", 2.0 * n * g.N * (g.K + 96.0), 4.0 * ((double)n * g.K + (double)g.N * g.K + (double)(g.N / 64) * n * 96));
                hipLaunchKernelGGL((gemm_f32_kernel<L_KCONTIG, L_KCONTIG, EPI_HEAD, 1, 1>), dim3(n / 64, g.N / 64, 1), dim3(256), 0, d->cur, g);
                XQ_HIP(hipGetLastError());
                continue;
            }
            bool aligned = true;
            for (int k = 0; k < njobs; ++k) {
                uint16_t* cb = (l == d->nl - 2) ? jobs[k].last_bf : nullptr;
                if (cb && jobs[k].last_bf_frag) g.cb_frag = 1;
                if (k == 0) { g.A = jobs[k].outs[l - 1]; g.B = d->wl(jobs[k].net, l); g.C = jobs[k].outs[l]; g.bias = d->bl(jobs[k].net, l); g.Cb = cb; }
                else { g.Ax[k - 1] = jobs[k].outs[l - 1]; g.Bx[k - 1] = d->wl(jobs[k].net, l); g.Cx[k - 1] = jobs[k].outs[l]; g.biasx[k - 1] = d->bl(jobs[k].net, l); g.Cbx[k - 1] = cb; }
                aligned = aligned && vec_ok(jobs[k].outs[l - 1], g.lda) && vec_ok(d->wl(jobs[k].net, l), g.ldb);
            }
            // whole 128-row x 128-column tiles on the handle's stream: the persistent walk (gemm_fwd_persistent_kernel; same bits as the
            // tile kernel).  8192 x 512 x 512 x 2 chains: 90 -> 76 us (128 x 128 tiles, 2 blocks per CU walk 512 tiles); 8192 x 256 x
            // 256 x 2: 27.5 -> 26.3 us (64 x 128 tiles) — tools/f32_fwd_probe.hip.  The select chain keeps the 64 x 64 tile kernel.
            if (aligned && !d->small_tiles && !d->tail_open && (n % 128) == 0 && (g.N % 128) == 0 && (g.K % GBK) == 0) {
                const int groups = njobs;
                const int t128 = (n / 128) * (g.N / 128) * groups;
                g.k_chunk = g.K; g.a_vec = g.b_vec = 1;
                ProfScope ps(d, "gemm_hidden_fwd", 2.0 * n * (double)g.N * g.K * groups,
                             4.0 * groups * ((double)n * g.K + (double)g.N * g.K + (double)n * g.N));
                if (t128 >= 512) {
                    hipLaunchKernelGGL((gemm_fwd_persistent_kernel<2, 2, 2>), dim3(std::min(t128, 2 * d->ncu)), dim3(256), 0, d->cur, g, n / 128,
                                       g.N / 128, t128);
                } else {
                    const int total = (n / 64) * (g.N / 128) * groups;
                    hipLaunchKernelGGL((gemm_fwd_persistent_kernel<1, 2, 2>), dim3(std::min(total, 2 * d->ncu)), dim3(256), 0, d->cur, g, n / 64,
                                       g.N / 128, total);
                }
                XQ_HIP(hipGetLastError());
                continue;
            }
            XQ_GEMM((launch_gemm<L_KCONTIG, L_KCONTIG, EPI_BIAS_TANH>(d, g, 1, "gemm_hidden_fwd")));
        }
    }
    return XQ_OK;
}

// Q head on the first n_out outputs: q[m][0..n_out) = tanh(W_out[0:n_out] a_last + b_out); a_last_bf != nullptr: bf16 operands
// (weights from the shadow), fp32 result (the output layer is never rounded)
static int q_head(xq_dqn* d, int net, const float* a_last, int n, int n_out, float* q, int ldq, const char* name,
                  const uint16_t* a_last_bf = nullptr) {
    GemmArgs g; memset(&g, 0, sizeof g);
    if (a_last_bf) {
        g.M = n; g.N = n_out; g.K = d->hlast() / 2;
        g.A = reinterpret_cast<const float*>(a_last_bf); g.lda = d->hlast() / 2;
        g.B = reinterpret_cast<const float*>(d->wl_bf(net, d->nl - 1)); g.ldb = d->hlast() / 2;
        g.C = q; g.ldc = ldq;
        g.bias = d->bl(net, d->nl - 1);
        XQ_GEMM((launch_gemm<L_KCONTIG, L_KCONTIG, EPI_BIAS_TANH, DT_BF16>(d, g, 1, name)));
        return XQ_OK;
    }
    g.M = n; g.N = n_out; g.K = d->hlast();
    g.A = a_last; g.lda = d->hlast();
    g.B = d->wl(net, d->nl - 1); g.ldb = d->hlast();
    g.bias = d->bl(net, d->nl - 1);
    g.libm_tanh = 1;                     // the output layer's tanh stays libm's (as in q_head_finish_kernel / the env kernel / the TD kernels)
    // The select head (selectAction reads q[action.to] only, dqn.cpp:47): 8192 x 96 x K is two column tiles of 64 — 256 blocks, one per
    // CU, each walking all K / 32 k-tiles behind one another with nothing to hide the load latency behind (20 us at K = 256, 32 us at
    // K = 512 for 0.4 / 0.8 GFLOP).  Four k-slabs per tile put four blocks on every CU; a one-thread-per-output kernel adds the slabs in a
    // fixed order, the bias and the tanh.  fp32 nets, large batches (the small ones are not latency-bound per CU to begin with).
    // The slabs are 64 columns of the last hidden layer each — what the fused form (EPI_HEAD in chain_boards, dqn_q90_boards) produces
    // per column tile, so both forms give the same bits.
    if (n_out <= 96 && n >= 2048 && (g.K % 128) == 0) {
        const int nc = round_up(n_out, 4);
        const int nslabs = g.K / 64;
        const size_t need = (size_t)nslabs * n * nc;
        float** slab = d->cur == d->stream ? &d->qh_slabs[0] : &d->qh_slabs[1];      // the select chain may run on its own stream beside a TD step
        size_t* cap = d->cur == d->stream ? &d->qh_cap[0] : &d->qh_cap[1];
        if (need > *cap) {
            XQ_HIP(hipDeviceSynchronize());
            if (*slab) XQ_HIP(hipFree(*slab));
            XQ_HIP(hipMalloc(slab, need * sizeof(float)));
            *cap = need;
        }
        g.C = *slab; g.ldc = nc; g.slab_stride = (long long)n * nc;
        XQ_GEMM((launch_gemm<L_KCONTIG, L_KCONTIG, EPI_STORE>(d, g, nslabs, name, nullptr, true)));
        const long long total = (long long)n * n_out;
        hipLaunchKernelGGL(q_head_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, d->cur, *slab, g.slab_stride, nslabs, n, n_out, nc,
                           g.bias, q, ldq);
        XQ_HIP(hipGetLastError());
        return XQ_OK;
    }
    g.C = q; g.ldc = ldq;
    XQ_GEMM((launch_gemm<L_KCONTIG, L_KCONTIG, EPI_BIAS_TANH>(d, g, 1, name)));
    return XQ_OK;
}

static int ensure_select_capacity(xq_dqn* d, int n) {
    if (n <= d->sel_cap) return XQ_OK;
    XQ_HIP(hipDeviceSynchronize());
    for (int l = 0; l + 1 < d->nl; ++l) {
        if (d->sel_acts[l]) XQ_HIP(hipFree(d->sel_acts[l]));
        XQ_HIP(hipMalloc(&d->sel_acts[l], (size_t)n * d->L[l + 1] * sizeof(float)));
    }
    if (d->sel_q90) XQ_HIP(hipFree(d->sel_q90));
    XQ_HIP(hipMalloc(&d->sel_q90, (size_t)n * 96 * sizeof(float)));
    d->sel_cap = n;
    return XQ_OK;
}
static int ensure_select_capacity_bf(xq_dqn* d, int n) {
    if (!d->bf16() || n <= d->sel_cap_bf) return XQ_OK;
    XQ_HIP(hipDeviceSynchronize());
    for (int l = 0; l + 1 < d->nl; ++l) {
        if (d->sel_acts_bf[l]) XQ_HIP(hipFree(d->sel_acts_bf[l]));
        XQ_HIP(hipMalloc(&d->sel_acts_bf[l], (size_t)round_up(n, 128) * d->L[l + 1] * sizeof(uint16_t)));
    }
    d->sel_cap_bf = n;
    return XQ_OK;
}

// Q(s)[0..95] of n packed boards on the online net.  `on` != nullptr queues the chain on that stream with its own
// workspace and 64x64 tiles, so it can run concurrently with a TD step queued on the handle's stream.
int dqn_q90_boards(xq_dqn* d, const uint32_t* boards_dev, int n, float** q90_dev, int* q_stride, hipStream_t on, QSource* qs) {
    if (d->nout() < 96) return fail(XQ_ERR_INVALID_ARGUMENT, "self-play select needs >= 96 outputs");
    if (qs) memset(qs, 0, sizeof *qs);
    float* outs[XQ_MAX_LAYERS];
    uint16_t* outs_bf[XQ_MAX_LAYERS] = {nullptr};
    float* q = nullptr;
    const bool bf = d->bf16();
    if (on) {
        XQ_TRY(ensure_select_capacity(d, n));
        XQ_TRY(ensure_select_capacity_bf(d, n));
        for (int l = 0; l + 1 < d->nl; ++l) { outs[l] = bf ? nullptr : d->sel_acts[l]; outs_bf[l] = d->sel_acts_bf[l]; }
        q = d->sel_q90;
        d->cur = on;
        d->small_tiles = true;
    } else {
        XQ_TRY(ensure_capacity(d, n));
        XQ_TRY(ensure_ext_capacity(d, n, false));
        for (int l = 0; l + 1 < d->nl; ++l) { outs[l] = bf ? nullptr : d->acts[l]; outs_bf[l] = d->acts_bf[l]; }
        q = d->q90;
    }
    ChainJob job{XQ_NET_ONLINE, boards_dev, outs, outs_bf, nullptr};
    // layer-0 sums kept from ply to ply (xq_dqn_set_l0_derive) — when that pays: keeping costs 16 MB of stores per call, deriving saves
    // ~20 row reads per board, so it is used from the second call of an update period on and, once a period had several calls, from
    // its first (one ply per update, the headline: never; bench --config 4, four plies: 0.976 -> 0.964 ms)
    xq_dqn::SelKeep& K = d->sel_keep[on ? 1 : 0];
    K.calls += 1;
    bool kept = false;
    if (d->l0_derive && !bf && (d->L[1] & 3) == 0 && (K.pays || K.calls >= 2)) {
        if (n > K.cap) {
            XQ_HIP(hipDeviceSynchronize());
            if (K.z1) XQ_HIP(hipFree(K.z1));
            if (K.prev_boards) XQ_HIP(hipFree(K.prev_boards));
            K.z1 = nullptr; K.prev_boards = nullptr; K.cap = 0; K.valid = false;
            XQ_HIP(hipMalloc(&K.z1, (size_t)n * d->L[1] * sizeof(float)));
            XQ_HIP(hipMalloc(&K.prev_boards, (size_t)n * kBoardWords * sizeof(uint32_t)));
            K.cap = n;
        }
        job.sel_keep = (K.valid && K.boards == boards_dev && K.n == n) ? 2 : 1;
        K.valid = false;                   // true again once the launch that keeps the sums has been queued
        kept = true;
    }
    // fp32 net with >= 2 hidden layers, whole 64 x 64 tiles, the batch sizes q_head splits into k-slabs: the head rides on the last hidden
    // product (same slabs, same bits) — its operand never goes to HBM and back (2 x 17 MB at 8192 x 512), one launch fewer per ply
    const int Hl = d->hlast();
    const bool ride = !bf && d->nl >= 3 && n >= 2048 && (n & 63) == 0 && (Hl % 128) == 0 && (d->L[d->nl - 2] & 31) == 0 && d->nout() >= 128 &&
                      vec_ok(d->wl(XQ_NET_ONLINE, d->nl - 1), Hl) && vec_ok(d->wl(XQ_NET_ONLINE, d->nl - 2), d->L[d->nl - 2]);
    int rc = XQ_OK;
    if (ride) {
        const int nslabs = Hl / 64;
        const size_t need = (size_t)nslabs * n * 96;
        float** slab = d->cur == d->stream ? &d->qh_slabs[0] : &d->qh_slabs[1];
        size_t* cap = d->cur == d->stream ? &d->qh_cap[0] : &d->qh_cap[1];
        if (need > *cap) {
            XQ_HIP(hipDeviceSynchronize());
            if (*slab) XQ_HIP(hipFree(*slab));
            XQ_HIP(hipMalloc(slab, need * sizeof(float)));
            *cap = need;
        }
        job.head_slabs = *slab;
        float* keep = outs[d->nl - 2];
        outs[d->nl - 2] = nullptr;                       // nobody else reads the select chain's last activations
        rc = chain_boards(d, &job, 1, explicit_slots(nullptr), n);
        outs[d->nl - 2] = keep;
        if (rc == XQ_OK && qs) {                         // the env kernel finishes the values it needs itself
            qs->slabs = *slab; qs->slab_stride = (long long)n * 96; qs->nslabs = nslabs; qs->bias = d->bl(XQ_NET_ONLINE, d->nl - 1);
            q = nullptr;
        } else if (rc == XQ_OK) {
            ProfScope ps(d, "gemm_q90_select", (double)n * 96 * nslabs, 4.0 * n * 96 * (nslabs + 1));
            hipLaunchKernelGGL(q_head_finish_kernel, dim3((unsigned)(((long long)n * 96 + 255) / 256)), dim3(256), 0, d->cur, *slab, (long long)n * 96,
                               nslabs, n, 96, 96, d->bl(XQ_NET_ONLINE, d->nl - 1), q, 96);
            if (hipGetLastError() != hipSuccess) rc = fail(XQ_ERR_RUNTIME, "q_head_finish_kernel launch failed");
        }
    } else {
    rc = chain_boards(d, &job, 1, explicit_slots(nullptr), n);
    if (rc == XQ_OK) rc = q_head(d, XQ_NET_ONLINE, outs[d->nl - 2], n, 96, q, 96, "gemm_q90_select", bf ? outs_bf[d->nl - 2] : nullptr);
    }
    d->cur = d->stream;
    d->small_tiles = false;
    XQ_TRY(rc);
    if (kept) { K.valid = true; K.boards = boards_dev; K.n = n; }
    *q90_dev = q;
    *q_stride = 96;
    return XQ_OK;
}

// conditions under which dqn.cu:406-423 as written stays inside its buffers (last paragraph)
static int check_reference_topology(const xq_dqn* d) {
    for (int l = d->nl - 2; l >= 0; --l) {
        const int inputSize = d->L[l + 1], outputSize = d->L[l];
        if (d->L[l + 2] < inputSize || outputSize < d->L[l + 1] ||
            d->wo[l + 1] + (size_t)(inputSize - 1) * outputSize + (size_t)(d->L[l + 1] - 1) >= d->nw)
            return fail(XQ_ERR_UNDEFINED_UPSTREAM,
                        "bug-compatible backprop reads out of bounds upstream for this topology (layer %d)", l);
    }
    return XQ_OK;
}

// hidden deltas l = nl-2 .. 0 from the output-side delta `dnext` ([n][ld_next], only the first k_nz columns can be
// non-zero).  reference mode: delta_l = (dnext[:, :L[l+1]] x View) * (1-a^2), View[i][idx] = Wflat[wo[l+1] + i*L[l] + idx];
// textbook: View[k][idx] = W_{l+1}[k][idx], k < L[l+2].
static int hidden_deltas(xq_dqn* d, int n, const float* dnext, int ld_next, int k_nz, int mode, int l_start = -1, int l_stop = 0) {
    const float* up = dnext;
    int ld_up = ld_next, nz = k_nz;
    if (l_start < 0) l_start = d->nl - 2;
    for (int l = l_start; l >= l_stop; --l) {
        GemmArgs g; memset(&g, 0, sizeof g);
        g.M = n; g.N = d->L[l + 1];
        const int kfull = (mode == XQ_BACKPROP_REFERENCE) ? d->L[l + 1] : d->L[l + 2];
        g.K = std::min(kfull, nz);
        if (d->bf16_bwd() && up == d->deltas[l + 1] && (n % kBgBM) == 0 && (g.N % kBgBN) == 0 && (g.K % kBgBK) == 0) {
            // XQ_PRECISION_BF16_FULL: delta_l = (bf16(delta_{l+1}) . bf16 weight VIEW) (1 - a_l^2) on the bf16 loop — the view is the
            // same (base, leading dimension) trick as below, applied to the bf16 shadow, which keeps the element order of the master
            Bf16GemmArgs b; memset(&b, 0, sizeof b);
            b.M = n; b.N = g.N; b.K = g.K; b.k_chunk = b.K;
            b.A = d->deltas_bf[l + 1]; b.lda = ld_up;
            b.B = d->wl_bf(XQ_NET_ONLINE, l + 1); b.ldb = (mode == XQ_BACKPROP_REFERENCE) ? d->L[l] : d->L[l + 1];
            b.C = d->deltas[l]; b.ldc = d->L[l + 1];
            b.Cb = d->deltas_bf[l]; b.ldcb = d->L[l + 1];
            b.Hb = d->acts_bf[l]; b.ldh = d->L[l + 1];
            XQ_TRY((launch_bf16_gemm<L_KCONTIG, L_MCONTIG, BG_DELTA>(d, b, 1, "gemm_hidden_delta")));
            up = d->deltas[l]; ld_up = d->L[l + 1]; nz = d->L[l + 1];
            continue;
        }
        g.A = up; g.lda = ld_up;
        g.B = d->wrest(XQ_NET_ONLINE) + (d->wo[l + 1] - d->wo[1]);
        g.ldb = (mode == XQ_BACKPROP_REFERENCE) ? d->L[l] : d->L[l + 1];
        g.C = d->deltas[l]; g.ldc = d->L[l + 1];
        g.H = d->acts[l]; g.ldh = d->L[l + 1];
        XQ_GEMM((launch_gemm<L_KCONTIG, L_MCONTIG, EPI_DELTA>(d, g, 1, "gemm_hidden_delta")));
        up = d->deltas[l]; ld_up = d->L[l + 1]; nz = d->L[l + 1];
    }
    return XQ_OK;
}

// gradient GEMMs always use 64x64 tiles: more tiles => fewer k-splits => less slab traffic in the ordered reduction
static int pick_splits(int M, int N, int K) {
    const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
    int s = (512 + tiles - 1) / tiles;
    s = std::min(s, std::max(1, K / 128));
    return std::max(1, std::min(s, 32));
}

// XQ_PRECISION_BF16_FULL: the hidden weight gradient on the bf16 loop when the shape allows (whole 256 x 128 tiles, whole 64-deep
// k-tiles per slab); the slab count then comes from the 256 x 128 tiling: one block per CU
static bool grad_bf16_ok(const xq_dqn* d, int M, int N, int K) {
    return d->bf16_bwd() && (M % kBgBM) == 0 && (N % kBgBN) == 0 && K >= kBgBK && (K % kBgBK) == 0;
}
static int grad_splits(const xq_dqn* d, int M, int N, int K) {
    if (!grad_bf16_ok(d, M, N, K)) return pick_splits(M, N, K);
    const int tiles = (M / kBgBM) * (N / kBgBN);
    int s = std::max(1, d->ncu / tiles);
    s = std::min(s, 16);                                    // (32 slabs = one block per CU: 0.717 ms per step of bench --config 5; 16: 0.713, and
                                                            //  half the slab bytes for the SGD kernel)
    while (s > 1 && (K % (s * kBgBK)) != 0) --s;           // whole k-tiles per slab
    return s;
}

// dst[M][N] = sum over the batch: A(m,k) B(k,n), split-K slabs + ordered reduction.  a_bf / b_bf: the bf16 copies of the operands
// ([batch][M] / [batch][N]) for the bf16 loop.
template <int AL>
static int grad_gemm(xq_dqn* d, GemmArgs g, float* dst, const char* name, float* slab_base = nullptr,
                     xq_dqn::PendingSlab* defer = nullptr, const uint16_t* a_bf = nullptr, const uint16_t* b_bf = nullptr) {
    const bool big = false;    // 64-tiles: more tiles, fewer k-splits, cheaper ordered reduction
    const bool use_bf = AL == L_MCONTIG && a_bf && b_bf && grad_bf16_ok(d, g.M, g.N, g.K);
    int splits = use_bf ? grad_splits(d, g.M, g.N, g.K) : pick_splits(g.M, g.N, g.K);
    if (use_bf) {
        const long long len = (long long)g.M * g.N;
        float* slabs = slab_base;
        Bf16GemmArgs b; memset(&b, 0, sizeof b);
        b.M = g.M; b.N = g.N; b.K = g.K; b.A = a_bf; b.lda = g.lda; b.B = b_bf; b.ldb = g.ldb; b.k_chunk = g.K / splits;
        if (splits > 1) {
            if (!slabs) { XQ_TRY(ensure_slabs(d, (size_t)splits * (size_t)len)); slabs = d->slabs; }
            b.C = slabs; b.ldc = g.N; b.slab_stride = len;
        } else { b.C = dst; b.ldc = g.N; }
        XQ_TRY((launch_bf16_gemm<L_MCONTIG, L_MCONTIG, BG_STORE>(d, b, splits, name)));
        if (splits > 1 && defer) { defer->src = slabs; defer->nslabs = splits; defer->stride = len; return XQ_OK; }
        if (splits > 1) {
            ProfScope ps(d, "reduce_slabs", (double)splits * len, 4.0 * (splits + 1) * len);
            hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)std::min<long long>((len + 255) / 256, 2048)), dim3(256), 0,
                               d->cur, slabs, splits, len, len, dst);
            XQ_HIP(hipGetLastError());
        }
        return XQ_OK;
    }
    const long long len = (long long)g.M * g.N;
    float* slabs = slab_base;
    if (splits > 1) {
        if (!slabs) { XQ_TRY(ensure_slabs(d, (size_t)splits * (size_t)len)); slabs = d->slabs; }
        g.C = slabs; g.ldc = g.N; g.slab_stride = len;
    } else {
        g.C = dst; g.ldc = g.N; g.slab_stride = 0;
    }
    int used = 0;
    XQ_TRY((launch_gemm<AL, L_MCONTIG, EPI_STORE>(d, g, splits, name, &used, !big, big)));
    if (splits > 1 && defer) {             // summed by the SGD kernel (xq_dqn_set_fused_apply)
        defer->src = slabs; defer->nslabs = used; defer->stride = len;
        return XQ_OK;
    }
    if (splits > 1) {
        ProfScope ps(d, "reduce_slabs", (double)used * len, 4.0 * (used + 1) * len);
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)std::min<long long>((len + 255) / 256, 2048)), dim3(256), 0,
                           d->cur, slabs, used, len, len, dst);
        XQ_HIP(hipGetLastError());
    }
    return XQ_OK;
}

struct BiasJobs {
    ColsumJobs J;
    BiasJobs() { memset(&J, 0, sizeof J); }
    void add(const float* X, long long ld, int C, float* dst) {
        const int k = J.njobs++;
        J.X[k] = X; J.ld[k] = ld; J.C[k] = C; J.dst[k] = dst;
    }
};
// defer != nullptr (fused_apply, TD step: the jobs are the hidden layers in ascending order, their destinations contiguous): the rows
// of partial sums are left to the SGD kernel as slabs — no final reduction launch
static int bias_grads(xq_dqn* d, BiasJobs& bj, int n, xq_dqn::PendingSlab* defer = nullptr) {
    ColsumJobs& J = bj.J;
    if (J.njobs == 0) return XQ_OK;
    J.n = n;
    J.R = std::max(1, std::min(64, n / 64));
    J.rows_per = (n + J.R - 1) / J.R;
    long long off = 0;
    int maxc = 0;
    for (int k = 0; k < J.njobs; ++k) { J.poff[k] = off; off += J.C[k]; maxc = std::max(maxc, J.C[k]); }
    J.wld = off;
    off *= J.R;
    if ((size_t)off > d->bias_work_cap) {
        if (d->bias_work) { XQ_HIP(hipStreamSynchronize(d->stream)); XQ_HIP(hipFree(d->bias_work)); }
        XQ_HIP(hipMalloc(&d->bias_work, (size_t)off * sizeof(float)));
        d->bias_work_cap = (size_t)off;
    }
    J.work = d->bias_work;
    if (defer) { defer->src = J.work; defer->nslabs = J.R; defer->stride = J.wld; }
    double tot = 0;
    for (int k = 0; k < J.njobs; ++k) tot += (double)n * J.C[k];
    if (d->tail_open) {                  // fused tail launch: the partial sums join the open grid (always deferred to the SGD kernel there)
        TailArgs& T = *d->tail;
        T.cj = J; T.cj_gx = (maxc + 63) / 64; T.cj_gy = J.R; T.n_colsum = T.cj_gx * T.cj_gy * J.njobs;
        d->tail_flops += tot; d->tail_bytes += 4.0 * tot;
        return XQ_OK;
    }
    ProfScope ps(d, "bias_grad_colsum", tot, 4.0 * tot);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((maxc + 63) / 64, J.R, J.njobs), dim3(256), 0, d->cur, J);
    XQ_HIP(hipGetLastError());
    if (defer) return XQ_OK;
    hipLaunchKernelGGL(colsum_final_kernel, dim3((maxc + 63) / 64, J.njobs), dim3(256), 0, d->cur, J);
    XQ_HIP(hipGetLastError());
    return XQ_OK;
}

// layer 0: per-(square, piece) segmented sums of delta_0 rows (no dense one-hot product); launches on d->cur
static int l0_gradient(xq_dqn* d, int n, float* dst) {
    const int H = d->L[1];
    // samples per block (list entries hold 11 bits of sample index).  8192 samples: 1024 (2048 measured 56 against 52 us, round 2);
    // 16384 samples: 2048 — half the partial slabs for the SGD kernel to sum (41 -> 20 MB), bench --config 5 0.717 -> 0.704 ms
    const int chunk = n >= 16384 ? 2048 : 1024;
    const int nchunks = (n + chunk - 1) / chunk;
    const long long len = (long long)kStateSize * H;
    float* out = dst;
    if (nchunks > 1) {
        const size_t need = (size_t)nchunks * (size_t)len;
        if (need > d->slabs_l0_cap) {
            if (d->slabs_l0) { XQ_HIP(hipDeviceSynchronize()); XQ_HIP(hipFree(d->slabs_l0)); }
            XQ_HIP(hipMalloc(&d->slabs_l0, need * sizeof(float)));
            d->slabs_l0_cap = need;
        }
        out = d->slabs_l0;
    }
    if (d->l0_mfma && (H % 64) == 0 && n >= 256) {
        // the matrix-pipe form (xq_l0grad.hip.h): delta_0 -> three transposed bf16 planes (one launch, 6 us at 8192 x 256), then
        // one-hot^T x planes on v_mfma_f32_16x16x32_bf16 — exact products, fp32 accumulation
        const int kpad = nchunks * chunk;
        const size_t need = l0m_plane_elems(H, kpad);
        if (need > d->l0_planes_cap) {
            if (d->l0_planes) { XQ_HIP(hipDeviceSynchronize()); XQ_HIP(hipFree(d->l0_planes)); }
            XQ_HIP(hipMalloc(&d->l0_planes, need * sizeof(uint16_t)));
            XQ_HIP(hipMemsetAsync(d->l0_planes, 0, need * sizeof(uint16_t), d->cur));      // (the slack behind the planes is read, never used)
            d->l0_planes_cap = need;
        }
        const long long plane_stride = (long long)H * kpad;
        const bool was_open = d->tail_open;
        d->tail_open = false;                        // the split is a launch of its own, in front of the fused launch that is being assembled
        {
            ProfScope ps(d, "l0_delta_split", 8.0 * n * H, (double)n * H * 4 + 6.0 * H * kpad);
            hipLaunchKernelGGL(delta_split_kernel, dim3(kpad / 64, H / 64), dim3(256), 0, d->cur, d->deltas[0], n, H, d->l0_planes, plane_stride, kpad);
        }
        d->tail_open = was_open;
        XQ_HIP(hipGetLastError());
        const size_t shmem = l0m_lds_bytes(chunk);
        const double fl = 2.0 * 96 * 16 * (double)H * kpad * 3, by = 6.0 * H * kpad * 4 + 48.0 * n * (H / 32) + 4.0 * nchunks * len;
        if (d->tail_open) {
            TailArgs& T = *d->tail;
            T.l0_boards = d->gboards; T.l0_n = n; T.l0_H = H; T.l0_chunk = chunk; T.l0_nch = nchunks; T.l0_partial = out;
            T.l0_planes = d->l0_planes; T.l0_plane_stride = plane_stride; T.l0_kpad = kpad; T.l0_ncb = H / kL0mCols;
            T.n_l0 = 4 * (H / kL0mCols) * nchunks;
            d->tail_flops += fl; d->tail_bytes += by;
            d->tail_lds = std::max(d->tail_lds, shmem);
        } else {
            ProfScope ps(d, "l0_grad_segsum", fl, by);
            hipLaunchKernelGGL(l0_grad_mfma_kernel<0>, dim3(4, H / kL0mCols, nchunks), dim3(256), shmem, d->cur, d->gboards, d->l0_planes, plane_stride,
                               kpad, n, H, chunk, out);
            XQ_HIP(hipGetLastError());
        }
    } else {
        ProfScope ps(d, "l0_grad_segsum", 2.0 * n * 32 * H, (double)n * (32.0 * H * 4 + 48) + 4.0 * nchunks * len);
        const int HS = (H > 256 && H % 256 == 0) ? 256 : H;     // column slab per block (grid z): wide layers keep the H = 256 shape
        int nsets = 4;                               // one accumulator set per wave while they fit in 60 KB of LDS
        while (nsets > 1 && (size_t)nsets * 14 * HS * sizeof(float) > 60 * 1024) nsets >>= 1;
        const size_t shmem = (size_t)nsets * 14 * HS * sizeof(float) + (size_t)chunk * sizeof(uint16_t);
        if (shmem > 64 * 1024) return fail(XQ_ERR_INVALID_ARGUMENT, "first hidden layer too wide for the layer-0 gradient kernel (%d)", H);
        if (d->tail_open) {              // fused tail launch
            TailArgs& T = *d->tail;
            T.l0_boards = d->gboards; T.l0_delta = d->deltas[0]; T.l0_n = n; T.l0_H = H; T.l0_HS = HS; T.l0_chunk = chunk; T.l0_nsets = nsets;
            T.l0_nch = nchunks; T.l0_partial = out; T.n_l0 = kSquares * nchunks * (H / HS);
            d->tail_flops += 2.0 * n * 32 * H; d->tail_bytes += (double)n * (32.0 * H * 4 + 48) + 4.0 * nchunks * len;
            d->tail_lds = std::max(d->tail_lds, shmem);
        } else {
        hipLaunchKernelGGL(l0_grad_kernel, dim3(kSquares, nchunks, H / HS), dim3(256), shmem, d->cur, d->gboards, d->deltas[0], n, H, HS,
                           chunk, nsets, out);
        XQ_HIP(hipGetLastError());
        }
    }
    d->l0_pending = 0;
    if (nchunks > 1 && d->fused()) {
        d->l0_pending = nchunks;                     // summed inside the SGD kernel: one kernel fewer on the critical chain
    } else if (nchunks > 1) {
        ProfScope ps(d, "reduce_slabs", (double)nchunks * len, 4.0 * (nchunks + 1) * len);
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)std::min<long long>((len + 255) / 256, 2048)), dim3(256), 0, d->cur,
                           d->slabs_l0, nchunks, len, len, dst);
        XQ_HIP(hipGetLastError());
    }
    return XQ_OK;
}

static int sgd_apply(xq_dqn* d, const SegTable& t, double alpha) {
    long long mx = 0;
    for (int i = 0; i < t.nseg; ++i) mx = std::max(mx, t.len[i]);
    const unsigned bx = (unsigned)std::max<long long>(1, std::min<long long>((mx + 255) / 256, 1024));
    ProfScope ps(d, t.reduce_only ? "reduce_slabs" : "sgd_apply", 0, 0);
    hipLaunchKernelGGL(sgd_segments_kernel, dim3(bx, t.nseg), dim3(256), 0, d->cur, t, (float)alpha);
    XQ_HIP(hipGetLastError());
    return XQ_OK;
}

static void layout_td_grads(xq_dqn* d) {
    size_t off = 0;
    d->g_w0 = off; off += (size_t)d->L[0] * d->L[1];
    for (int l = 1; l + 1 < d->nl; ++l) { d->g_wh[l] = off; off += (size_t)d->L[l] * d->L[l + 1]; }
    d->g_wout = off; off += (size_t)96 * d->hlast();
    d->g_bout = off; off += 96;                      // directly behind the output rows: one ordered reduction fills both
    for (int l = 0; l + 1 < d->nl; ++l) { d->g_bh[l] = off; off += (size_t)d->L[l + 1]; }
    d->n_grads_td = off;
}

}  // namespace xq

using namespace xq;

// =================================================================================================================
// C ABI — dqn
// =================================================================================================================
extern "C" {

static int dqn_init(xq_dqn* d, const int* layer_sizes, int n_sizes, double learning_rate, double gamma, uint64_t seed,
                    void* hip_stream);
static int refresh_shadow(xq_dqn* d, int net);

int xq_dqn_create(const int* layer_sizes, int n_sizes, double learning_rate, double gamma, uint64_t seed, void* hip_stream,
                  xq_dqn** out) {
    if (!out || !layer_sizes) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_dqn_create: null pointer");
    if (n_sizes < 2) return fail(XQ_ERR_INVALID_ARGUMENT, "NeuralNetwork must have at least two layers (input and output).");
    if (n_sizes < 3 || n_sizes > XQ_MAX_LAYERS + 1)
        return fail(XQ_ERR_INVALID_ARGUMENT, "xq_dqn supports 1..%d hidden layers (got %d sizes)", XQ_MAX_LAYERS - 1, n_sizes);
    for (int i = 0; i < n_sizes; ++i)
        if (layer_sizes[i] <= 0) return fail(XQ_ERR_INVALID_ARGUMENT, "layer size must be positive");
    int c = 0;
    XQ_TRY(xq_device_count(&c));
    if (c == 0) return fail(XQ_ERR_NO_DEVICE, "no HIP device: libxqhip has no CPU fallback");
    xq_dqn* d = new xq_dqn();
    const int rc = dqn_init(d, layer_sizes, n_sizes, learning_rate, gamma, seed, hip_stream);
    if (rc != XQ_OK) { xq_dqn_destroy(d); return rc; }      // a failed allocation must not leak the ones before it
    *out = d;
    return XQ_OK;
}

static int dqn_init(xq_dqn* d, const int* layer_sizes, int n_sizes, double learning_rate, double gamma, uint64_t seed,
                    void* hip_stream) {
    d->ns = n_sizes; d->nl = n_sizes - 1;
    for (int i = 0; i < n_sizes; ++i) d->L[i] = layer_sizes[i];
    for (int l = 0; l < d->nl; ++l) {                       // offsets, dqn.cu:125-140
        d->wo[l] = d->nw; d->bo[l] = d->nb;
        d->nw += (size_t)d->L[l] * d->L[l + 1];
        d->nb += (size_t)d->L[l + 1];
    }
    d->lr = learning_rate; d->gamma = gamma; d->seed = seed;
    if (hip_stream) d->stream = (hipStream_t)hip_stream;
    else { XQ_HIP(hipStreamCreate(&d->stream)); d->own_stream = true; }
    d->cur = d->stream;
    {
        hipDeviceProp_t prop; int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            d->ncu = prop.multiProcessorCount;
    }
    XQ_HIP(hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking));
    XQ_HIP(hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming));
    XQ_HIP(hipEventCreateWithFlags(&d->ev_join, hipEventDisableTiming));
    XQ_HIP(hipEventCreateWithFlags(&d->ev_delta, hipEventDisableTiming));
    XQ_HIP(hipEventCreateWithFlags(&d->ev_qmax, hipEventDisableTiming));
    // + 128 rows of the widest layer: the persistent column-max GEMM reads whole 128-row tiles of W_out (rows beyond the
    // last output are masked in its epilogue, but must be readable)
    int widest = 0;
    for (int i = 0; i < n_sizes; ++i) widest = std::max(widest, layer_sizes[i]);
    const size_t pad = (size_t)128 * widest;
    for (int i = 0; i < 2; ++i) {
        XQ_HIP(hipMalloc(&d->params[i], (d->nw + d->nb + pad) * sizeof(float)));
        XQ_HIP(hipMemsetAsync(d->params[i], 0, (d->nw + d->nb + pad) * sizeof(float), d->stream));
    }
    layout_td_grads(d);
    XQ_HIP(hipMalloc(&d->grads_td, d->n_grads_td * sizeof(float)));
    XQ_HIP(hipMemsetAsync(d->grads_td, 0, d->n_grads_td * sizeof(float), d->stream));
    // initializeHostWeightsAndBiases (dqn.cu:96-123): U(-0.05, 0.05) in [layer][out][in] order, biases 0
    std::vector<double> w(d->nw), b(d->nb, 0.0);
    std::mt19937_64 gen(seed);
    std::uniform_real_distribution<double> dis(-0.05, 0.05);
    for (size_t i = 0; i < d->nw; ++i) w[i] = dis(gen);
    XQ_TRY(xq_dqn_set_params(d, XQ_NET_ONLINE, w.data(), b.data()));
    return xq_dqn_update_target(d);                          // DQN ctor, dqn.cpp:18
}

int xq_dqn_destroy(xq_dqn* d) {
    if (!d) return XQ_OK;
    hipStreamSynchronize(d->stream);
    if (d->own_stream) retire_stream(d->stream);
    for (int i = 0; i < 2; ++i) { hipFree(d->params[i]); hipFree(d->tacts[i]); }
    for (int l = 0; l < XQ_MAX_LAYERS; ++l) { hipFree(d->acts[l]); hipFree(d->deltas[l]); hipFree(d->sel_acts[l]); }
    hipFree(d->gboards); hipFree(d->dsc); hipFree(d->act_mb); hipFree(d->q90); hipFree(d->sel_q90); for (auto& K : d->sel_keep) { hipFree(K.z1); hipFree(K.prev_boards); } hipFree(d->partial); hipFree(d->zmax); hipFree(d->zidx); hipFree(d->qsa); hipFree(d->yv); hipFree(d->lossv);
    hipFree(d->grads_td); hipFree(d->grads_full); hipFree(d->slabs); hipFree(d->bias_work); hipFree(d->xdense); hipFree(d->qfull); hipFree(d->tfull);
    hipFree(d->hb); hipFree(d->ha); hipFree(d->hr); hipFree(d->hd);
    d->prof.collect();
    hipFree(d->slabs_l0); hipFree(d->l0_planes);
    for (int i = 0; i < 2; ++i) { hipFree(d->params_bf[i]); hipFree(d->tacts_bf[i]); hipFree(d->t2acts_bf[i]); hipFree(d->t2acts[i]); }
    for (int l = 0; l < XQ_MAX_LAYERS; ++l) { hipFree(d->acts_bf[l]); hipFree(d->sel_acts_bf[l]); }
    hipFree(d->partial_idx);
    hipFree(d->qh_slabs[0]); hipFree(d->qh_slabs[1]);
    for (int l = 0; l < XQ_MAX_LAYERS; ++l) hipFree(d->deltas_bf[l]);
    if (d->side) { hipStreamSynchronize(d->side); hipStreamDestroy(d->side); }
    delete d->tail;
    if (d->ev_fork) hipEventDestroy(d->ev_fork);
    if (d->ev_join) hipEventDestroy(d->ev_join);
    if (d->ev_delta) hipEventDestroy(d->ev_delta);
    if (d->ev_qmax) hipEventDestroy(d->ev_qmax);
    for (void* q : {(void*)d->scr_wb, (void*)d->scr_ab, (void*)d->scr_p1, (void*)d->scr_p2, (void*)d->scr_wmax, (void*)d->scr_stats, (void*)d->scr_R,
                    (void*)d->scr_na})
        if (q) hipFree(q);
    if (d->scr_guard_host) hipHostFree(d->scr_guard_host);
    if (d->scr_guard_ev) hipEventDestroy(d->scr_guard_ev);
    if (d->own_stream) hipStreamDestroy(d->stream);
    delete d;
    return XQ_OK;
}

int xq_dqn_stream(const xq_dqn* d, void** s) {
    if (!d || !s) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    *s = (void*)d->stream;
    return XQ_OK;
}

int xq_dqn_num_params(const xq_dqn* d, size_t* nw, size_t* nb) {
    if (!d) return fail(XQ_ERR_INVALID_ARGUMENT, "null dqn");
    if (nw) *nw = d->nw;
    if (nb) *nb = d->nb;
    return XQ_OK;
}

int xq_dqn_set_params(xq_dqn* d, int net, const double* w, const double* b) {
    if (!d || !w || !b || net < 0 || net > 1) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_dqn_set_params: bad argument");
    std::vector<float> p(d->nw + d->nb);
    const int L0 = d->L[0], L1 = d->L[1];
    for (int j = 0; j < L1; ++j)
        for (int i = 0; i < L0; ++i) p[(size_t)i * L1 + j] = (float)w[(size_t)j * L0 + i];     // W0 [out][in] -> W0^T
    for (size_t i = d->wo[1]; i < d->nw; ++i) p[i] = (float)w[i];
    for (size_t i = 0; i < d->nb; ++i) p[d->nw + i] = (float)b[i];
    XQ_HIP(hipStreamSynchronize(d->stream));
    XQ_HIP(hipMemcpy(d->params[net], p.data(), p.size() * sizeof(float), hipMemcpyHostToDevice));
    if (d->scr_static_net == net) d->scr_static_net = -1;       // screening shadow: rows >= 96 are no longer what it holds
    if (net == XQ_NET_ONLINE) d->sel_invalidate();              // kept layer-0 sums of the select chain belong to the old weights
    d->params_version += 1;
    XQ_TRY(refresh_shadow(d, net));
    XQ_HIP(hipStreamSynchronize(d->stream));                    // a host-buffer entry point returns with the parameters in place
    return XQ_OK;
}

int xq_dqn_get_params(xq_dqn* d, int net, double* w, double* b) {
    if (!d || net < 0 || net > 1) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_dqn_get_params: bad argument");
    std::vector<float> p(d->nw + d->nb);
    XQ_HIP(hipStreamSynchronize(d->stream));
    XQ_HIP(hipMemcpy(p.data(), d->params[net], p.size() * sizeof(float), hipMemcpyDeviceToHost));
    const int L0 = d->L[0], L1 = d->L[1];
    if (w) {
        for (int j = 0; j < L1; ++j)
            for (int i = 0; i < L0; ++i) w[(size_t)j * L0 + i] = (double)p[(size_t)i * L1 + j];
        for (size_t i = d->wo[1]; i < d->nw; ++i) w[i] = (double)p[i];
    }
    if (b) for (size_t i = 0; i < d->nb; ++i) b[i] = (double)p[d->nw + i];
    return XQ_OK;
}

int xq_dqn_update_target(xq_dqn* d) {
    if (!d) return fail(XQ_ERR_INVALID_ARGUMENT, "null dqn");
    ProfScope ps(d, "target_sync_copy", 0, 8.0 * (d->nw + d->nb));
    d->params_version += 1;
    XQ_HIP(hipMemcpyAsync(d->params[1], d->params[0], (d->nw + d->nb) * sizeof(float), hipMemcpyDeviceToDevice, d->stream));
    if (d->scr_static_net == XQ_NET_TARGET) d->scr_static_net = -1;     // screening shadow: every row of the target net changed
    if (d->bf16())
        XQ_HIP(hipMemcpyAsync(d->params_bf[1], d->params_bf[0], d->nw * sizeof(uint16_t), hipMemcpyDeviceToDevice, d->stream));
    return XQ_OK;
}

static int refresh_shadow(xq_dqn* d, int net) {
    if (!d->bf16()) return XQ_OK;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(1024), dim3(256), 0, d->stream, d->params[net], d->params_bf[net], (long long)d->nw);
    XQ_HIP(hipGetLastError());
    return XQ_OK;
}

int xq_dqn_set_l0_derive(xq_dqn* d, int on) {
    if (d) d->sel_invalidate();
    if (!d) return fail(XQ_ERR_INVALID_ARGUMENT, "null dqn");
    d->l0_derive = on != 0;
    return XQ_OK;
}

int xq_dqn_set_l0_grad_mode(xq_dqn* d, int mode) {
    if (!d || (mode != 0 && mode != 1)) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_dqn_set_l0_grad_mode: 0 (segmented sums) or 1 (matrix pipe)");
    d->l0_mfma = mode == 1;
    return XQ_OK;
}

int xq_dqn_set_td_tail(xq_dqn* d, int on) {
    if (!d) return fail(XQ_ERR_INVALID_ARGUMENT, "null dqn");
    d->td_tail = on != 0;
    return XQ_OK;
}

int xq_dqn_set_exchange_overlap(xq_dqn* d, int mode) {
    if (!d || mode < -1 || mode > 1) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_dqn_set_exchange_overlap: mode -1, 0 or 1");
    d->exchange_overlap = mode;
    return XQ_OK;
}

int xq_dqn_set_qmax_mode(xq_dqn* d, int mode) {
    if (!d || (mode != XQ_QMAX_FULL && mode != XQ_QMAX_SCREENED)) return fail(XQ_ERR_INVALID_ARGUMENT, "bad qmax mode");
    d->qmax_mode = mode;
    d->scr_hold = 0;                       // an explicit request starts with the screen switched on again
    return XQ_OK;
}

int xq_dqn_qmax_stats(xq_dqn* d, uint64_t stats[4]) {
    if (!d || !stats) return fail(XQ_ERR_INVALID_ARGUMENT, "null");
    unsigned long long h[2] = {0, 0};
    XQ_TRY(screen_stat_sums(d, h));
    stats[0] = d->scr_host_steps; stats[1] = d->scr_host_samples; stats[2] = h[0]; stats[3] = h[1];
    return XQ_OK;
}

int xq_dqn_qmax_guard(xq_dqn* d, uint64_t* fallbacks, int* hold_steps_left) {
    if (!d) return fail(XQ_ERR_INVALID_ARGUMENT, "null dqn");
    if (fallbacks) *fallbacks = d->scr_fallbacks;
    if (hold_steps_left) *hold_steps_left = d->scr_hold;
    return XQ_OK;
}

int xq_dqn_set_precision(xq_dqn* d, int precision) {
    if (!d || (precision != XQ_PRECISION_F32 && precision != XQ_PRECISION_BF16 && precision != XQ_PRECISION_BF16_FULL))
        return fail(XQ_ERR_INVALID_ARGUMENT, "bad precision");
    if (d->l0_pending > 0 || d->pend_wout.nslabs > 0 || d->pend_bh.nslabs > 0)
        return fail(XQ_ERR_RUNTIME, "xq_dqn_set_precision: a TD step is waiting for its apply_grads");
    if (precision != XQ_PRECISION_F32) {
        for (int l = 1; l <= d->nl - 1; ++l)
            if (d->L[l] & 1) return fail(XQ_ERR_INVALID_ARGUMENT, "bf16 Q-net needs even hidden widths (layer %d has %d)", l, d->L[l]);
        if (!d->params_bf[0]) {
            int widest = 0;
            for (int i = 0; i < d->ns; ++i) widest = std::max(widest, d->L[i]);
            const size_t pad = (size_t)128 * widest;         // whole-tile reads of the persistent GEMM, like params[]
            for (int i = 0; i < 2; ++i) {
                XQ_HIP(hipMalloc(&d->params_bf[i], (d->nw + pad) * sizeof(uint16_t)));
                XQ_HIP(hipMemsetAsync(d->params_bf[i], 0, (d->nw + pad) * sizeof(uint16_t), d->stream));
            }
        }
    }
    d->precision = precision;
    d->params_version += 1;
    XQ_TRY(refresh_shadow(d, 0));
    return refresh_shadow(d, 1);
}

// dense-state chain: a_1 via GEMM against W0^T; outs as in chain_boards
static int chain_dense(xq_dqn* d, int net, const float* x, int n, float* const* outs) {
    GemmArgs g; memset(&g, 0, sizeof g);
    g.M = n; g.N = d->L[1]; g.K = d->L[0];
    g.A = x; g.lda = d->L[0];
    g.B = d->w0t(net); g.ldb = d->L[1];
    g.C = outs[0]; g.ldc = d->L[1];
    g.bias = d->bl(net, 0);
    XQ_GEMM((launch_gemm<L_KCONTIG, L_MCONTIG, EPI_BIAS_TANH>(d, g, 1, "gemm_l0_dense_fwd")));
    for (int l = 1; l + 1 < d->nl; ++l) {
        GemmArgs h; memset(&h, 0, sizeof h);
        h.M = n; h.N = d->L[l + 1]; h.K = d->L[l];
        h.A = outs[l - 1]; h.lda = d->L[l];
        h.B = d->wl(net, l); h.ldb = d->L[l];
        h.C = outs[l]; h.ldc = d->L[l + 1];
        h.bias = d->bl(net, l);
        XQ_GEMM((launch_gemm<L_KCONTIG, L_KCONTIG, EPI_BIAS_TANH>(d, h, 1, "gemm_hidden_fwd")));
    }
    return XQ_OK;
}

static int upload_dense(xq_dqn* d, const double* src, size_t count, float** dev, size_t* cap) {
    XQ_TRY(grow(dev, cap, count, d->stream));
    std::vector<float> tmp(count);
    for (size_t i = 0; i < count; ++i) tmp[i] = (float)src[i];
    XQ_HIP(hipStreamSynchronize(d->stream));
    XQ_HIP(hipMemcpy(*dev, tmp.data(), count * sizeof(float), hipMemcpyHostToDevice));
    return XQ_OK;
}

int xq_dqn_forward(xq_dqn* d, int net, const double* states, int n, double* q_host) {
    if (!d || !states || !q_host || n <= 0 || net < 0 || net > 1)
        return fail(XQ_ERR_INVALID_ARGUMENT, "Input size does not match network input layer size.");
    XQ_TRY(ensure_capacity(d, n));
    XQ_TRY(upload_dense(d, states, (size_t)n * d->L[0], &d->xdense, &d->xdense_cap));
    XQ_TRY(grow(&d->qfull, &d->qfull_cap, (size_t)n * d->nout(), d->stream));
    float* outs[XQ_MAX_LAYERS];
    for (int l = 0; l + 1 < d->nl; ++l) outs[l] = d->acts[l];
    XQ_TRY(chain_dense(d, net, d->xdense, n, outs));
    XQ_TRY(q_head(d, net, outs[d->nl - 2], n, d->nout(), d->qfull, d->nout(), "gemm_q_full"));
    std::vector<float> q((size_t)n * d->nout());
    XQ_HIP(hipMemcpyAsync(q.data(), d->qfull, q.size() * sizeof(float), hipMemcpyDeviceToHost, d->stream));
    XQ_HIP(hipStreamSynchronize(d->stream));
    for (size_t i = 0; i < q.size(); ++i) q_host[i] = (double)q[i];
    return XQ_OK;
}

int xq_dqn_forward_boards_dev(xq_dqn* d, int net, const uint32_t* boards_dev, int n, int n_out, float* q_dev, int ldq) {
    if (!d || !boards_dev || !q_dev || n <= 0 || net < 0 || net > 1 || n_out <= 0 || n_out > d->nout() || ldq < n_out)
        return fail(XQ_ERR_INVALID_ARGUMENT, "xq_dqn_forward_boards_dev: bad argument");
    XQ_TRY(ensure_capacity(d, n));
    XQ_TRY(ensure_ext_capacity(d, n, false));
    float* outs[XQ_MAX_LAYERS];
    uint16_t* outs_bf[XQ_MAX_LAYERS] = {nullptr};
    for (int l = 0; l + 1 < d->nl; ++l) { outs[l] = d->acts[l]; outs_bf[l] = d->acts_bf[l]; }
    ChainJob job{net, boards_dev, outs, outs_bf, nullptr};
    XQ_TRY(chain_boards(d, &job, 1, explicit_slots(nullptr), n));
    return q_head(d, net, outs[d->nl - 2], n, n_out, q_dev, ldq, n_out <= 96 ? "gemm_q90_select" : "gemm_q_full",
                  d->bf16() ? outs_bf[d->nl - 2] : nullptr);
}

int xq_dqn_select_q_dev(xq_dqn* d, const uint32_t* boards_dev, int n, float* q_dev) {
    if (!d || !boards_dev || !q_dev || n <= 0) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_dqn_select_q_dev: bad argument");
    float* q = nullptr;
    int stride = 0;
    XQ_TRY(dqn_q90_boards(d, boards_dev, n, &q, &stride, nullptr, nullptr));
    XQ_HIP(hipMemcpyAsync(q_dev, q, (size_t)n * 96 * sizeof(float), hipMemcpyDeviceToDevice, d->stream));
    return XQ_OK;
}

int xq_dqn_backpropagate(xq_dqn* d, const double* states, const double* targets, int n, double lr, double grad_scale, int mode) {
    if (!d || !states || !targets || n <= 0) return fail(XQ_ERR_INVALID_ARGUMENT, "Input size does not match network input layer size.");
    if (mode != XQ_BACKPROP_REFERENCE && mode != XQ_BACKPROP_TEXTBOOK) return fail(XQ_ERR_INVALID_ARGUMENT, "bad backprop mode");
    if (mode == XQ_BACKPROP_REFERENCE) XQ_TRY(check_reference_topology(d));
    XQ_TRY(ensure_capacity(d, n));
    const int NO = d->nout();
    XQ_TRY(upload_dense(d, states, (size_t)n * d->L[0], &d->xdense, &d->xdense_cap));
    XQ_TRY(upload_dense(d, targets, (size_t)n * NO, &d->tfull, &d->tfull_cap));
    XQ_TRY(grow(&d->qfull, &d->qfull_cap, (size_t)n * NO, d->stream));
    if (!d->grads_full) XQ_HIP(hipMalloc(&d->grads_full, (d->nw + d->nb) * sizeof(float)));
    float* outs[XQ_MAX_LAYERS];
    for (int l = 0; l + 1 < d->nl; ++l) outs[l] = d->acts[l];
    XQ_TRY(chain_dense(d, XQ_NET_ONLINE, d->xdense, n, outs));
    XQ_TRY(q_head(d, XQ_NET_ONLINE, outs[d->nl - 2], n, NO, d->qfull, NO, "gemm_q_full"));
    const long long total = (long long)n * NO;
    hipLaunchKernelGGL(out_delta_dense_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0,
                       d->cur, d->qfull, d->tfull, total, d->tfull);      // delta overwrites the target buffer
    XQ_HIP(hipGetLastError());
    const float* dout = d->tfull;
    XQ_TRY(hidden_deltas(d, n, dout, NO, NO, mode));
    // gradients, full layout = parameter layout
    float* gw = d->grads_full;
    float* gb = d->grads_full + d->nw;
    BiasJobs bj;
    {   // layer 0: gW0^T[in][out] = X^T delta_0
        GemmArgs g; memset(&g, 0, sizeof g);
        g.M = d->L[0]; g.N = d->L[1]; g.K = n;
        g.A = d->xdense; g.lda = d->L[0];
        g.B = d->deltas[0]; g.ldb = d->L[1];
        XQ_TRY((grad_gemm<L_MCONTIG>(d, g, gw, "gemm_grad_l0_dense")));
        bj.add(d->deltas[0], d->L[1], d->L[1], gb + d->bo[0]);
    }
    for (int l = 1; l < d->nl; ++l) {   // gW_l[out][in] = delta_l^T a_l
        const float* dl = (l == d->nl - 1) ? dout : d->deltas[l];
        const int ldd = (l == d->nl - 1) ? NO : d->L[l + 1];
        GemmArgs g; memset(&g, 0, sizeof g);
        g.M = d->L[l + 1]; g.N = d->L[l]; g.K = n;
        g.A = dl; g.lda = ldd;
        g.B = d->acts[l - 1]; g.ldb = d->L[l];
        XQ_TRY((grad_gemm<L_MCONTIG>(d, g, gw + d->wo[l], "gemm_grad_dense")));
        bj.add(dl, ldd, d->L[l + 1], gb + d->bo[l]);
    }
    XQ_TRY(bias_grads(d, bj, n));
    SegTable t; memset(&t, 0, sizeof t);
    t.nseg = 1; t.dst[0] = d->params[0]; t.src[0] = d->grads_full; t.len[0] = (long long)(d->nw + d->nb);
    d->params_version += 1;
    XQ_TRY(sgd_apply(d, t, lr * grad_scale));
    XQ_TRY(refresh_shadow(d, XQ_NET_ONLINE));          // bf16 Q-net: every later bf16 forward must see the updated weights
    if (d->scr_static_net == XQ_NET_ONLINE) d->scr_static_net = -1;    // dense update: every output row changed (screening shadow)
    d->sel_invalidate();
    XQ_HIP(hipStreamSynchronize(d->stream));
    return XQ_OK;
}

// prioritized replay hooks of one TD step (device pointers owned by the replay ring; nullptr = uniform replay)
struct PerOpts {
    const float* is_w; const float* is_wmax;
    float* prio; unsigned* pmax_live;
    float eps, alpha;
};
static int td_grads_impl(xq_dqn* d, const uint32_t* boards, const uint32_t* next_boards, const int32_t* action_to,
                         const float* reward, const uint8_t* done, SlotSrc slots, int n, int td_net, int mode, const PerOpts* per);

int xq_dqn_td_grads(xq_dqn* d, const uint32_t* boards, const uint32_t* next_boards, const int32_t* action_to,
                    const float* reward, const uint8_t* done, const int32_t* slots, int n, int td_net, int mode) {
    return td_grads_impl(d, boards, next_boards, action_to, reward, done, explicit_slots(slots), n, td_net, mode, nullptr);
}

// gradients that run on the side stream (d->cur == d->side), see td_grads_impl
static int side_gradients(xq_dqn* d, int n, float* const* outs, float* G) {
    const int nl = d->nl, Hl = d->hlast();
    BiasJobs bj;
    const bool fused = d->fused();
    const int chunk = 256;                           // samples per block: small chunks level the load between popular and rare destination squares
                                                     // (1024: 21.7 us and a side stream that ends with the critical one; 256: 16.4 us, step -10 us)
    const int nchunks = (n + chunk - 1) / chunk;
    const long long len_out = 96LL * Hl + 96;
    // fused_apply: every partial-sum slab of this step stays alive until the SGD kernel sums it => one region each
    size_t off_out = 0, off_h[XQ_MAX_LAYERS] = {0}, need = 0;
    if (fused) {
        if (nchunks > 1) need += (size_t)nchunks * (size_t)len_out;
        for (int l = nl - 2; l >= 1; --l) {
            off_h[l] = need;
            const int sp = grad_splits(d, d->L[l + 1], d->L[l], n);
            if (sp > 1) need += (size_t)sp * (size_t)d->L[l + 1] * (size_t)d->L[l];
        }
        XQ_TRY(ensure_slabs(d, need));
        for (int l = 0; l < XQ_MAX_LAYERS; ++l) d->pend_hidden[l] = xq_dqn::PendingSlab();
        d->pend_wout = d->pend_bout = xq_dqn::PendingSlab();
    }
    {   // output layer rows 0..95 + their biases: segmented sums by action
        float* dst = G + d->g_wout;                      // g_bout follows directly
        float* out = dst;
        if (nchunks > 1) {
            if (fused) out = d->slabs + off_out;
            else { XQ_TRY(ensure_slabs(d, (size_t)nchunks * (size_t)len_out)); out = d->slabs; }
        }
        {
            ProfScope ps(d, "out_grad_segsum", 2.0 * n * Hl, (double)n * (Hl * 4 + 8) + 4.0 * nchunks * len_out);
            const size_t shmem = (size_t)16 * Hl * sizeof(float) + (size_t)chunk * sizeof(uint16_t);
            if (shmem > 64 * 1024 || (Hl & 3)) return fail(XQ_ERR_INVALID_ARGUMENT, "last hidden layer width %d unsupported by the output-gradient kernel (multiple of 4, <= 960)", Hl);
            hipLaunchKernelGGL(out_grad_kernel, dim3(24, nchunks), dim3(256), shmem, d->cur, d->act_mb, d->dsc, outs[nl - 2], n, Hl,
                               chunk, out);
            XQ_HIP(hipGetLastError());
        }
        if (nchunks > 1 && fused) {
            d->pend_wout.src = out; d->pend_wout.nslabs = nchunks; d->pend_wout.stride = len_out;
            d->pend_bout.src = out + 96LL * Hl; d->pend_bout.nslabs = nchunks; d->pend_bout.stride = len_out;
        } else if (nchunks > 1) {
            ProfScope ps(d, "reduce_slabs", (double)nchunks * len_out, 4.0 * (nchunks + 1) * len_out);
            hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((len_out + 255) / 256)), dim3(256), 0, d->cur, out, nchunks, len_out,
                               len_out, dst);
            XQ_HIP(hipGetLastError());
        }
    }
    bool waited = false;
    for (int l = nl - 2; l >= 1; --l) {
        if (l < nl - 2 && !waited) { XQ_HIP(hipStreamWaitEvent(d->cur, d->ev_delta, 0)); waited = true; }   // delta_l, l < top, comes from the GEMMs
        GemmArgs g; memset(&g, 0, sizeof g);
        g.M = d->L[l + 1]; g.N = d->L[l]; g.K = n;
        g.A = d->deltas[l]; g.lda = d->L[l + 1];
        g.B = outs[l - 1]; g.ldb = d->L[l];
        XQ_TRY((grad_gemm<L_MCONTIG>(d, g, G + d->g_wh[l], "gemm_grad_hidden", fused ? d->slabs + off_h[l] : nullptr,
                                     fused ? &d->pend_hidden[l] : nullptr, d->bf16_bwd() ? d->deltas_bf[l] : nullptr,
                                     d->bf16_bwd() ? d->acts_bf[l - 1] : nullptr)));
    }
    for (int l = 0; l <= nl - 2; ++l) bj.add(d->deltas[l], d->L[l + 1], d->L[l + 1], G + d->g_bh[l]);     // the order of the bias vector
    if (!waited) XQ_HIP(hipStreamWaitEvent(d->cur, d->ev_delta, 0));
    d->pend_bh = xq_dqn::PendingSlab();
    return bias_grads(d, bj, n, fused ? &d->pend_bh : nullptr);
}

// ---- the same gradients as fused launches on the handle's stream (td_tail_kernel): no side stream, no event --------------------
static void tail_begin(xq_dqn* d) {
    if (!d->tail) d->tail = new TailArgs();
    memset(d->tail, 0, sizeof(TailArgs));
    d->tail_open = true; d->tail_lds = 0; d->tail_flops = d->tail_bytes = 0;
}
static int tail_launch(xq_dqn* d, bool last, const char* name) {
    d->tail_open = false;
    const TailArgs& T = *d->tail;
    const long long total = (long long)T.n_l0 + T.n_grad + T.n_delta + T.n_out + T.n_colsum;
    if (total <= 0) return XQ_OK;
    ProfScope ps(d, name, d->tail_flops, d->tail_bytes);
    auto launch = [&](auto kern) {
        static size_t granted = 48 * 1024;            // per instantiation
        if (d->tail_lds > granted) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            granted = 64 * 1024;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), d->tail_lds, d->cur, T);
    };
    if (last && T.l0_planes != nullptr) launch(td_tail_kernel<TAIL_L0 | TAIL_GRAD | TAIL_OUT | TAIL_COLSUM, true>);
    else if (last) launch(td_tail_kernel<TAIL_L0 | TAIL_GRAD | TAIL_OUT | TAIL_COLSUM>);
    else launch(td_tail_kernel<TAIL_GRAD | TAIL_DELTA | TAIL_OUT>);
    XQ_HIP(hipGetLastError());
    return XQ_OK;
}
// shapes the fused launches take: fp32 products on 64x64 tiles (what launch_gemm picks for them anyway).  Their partial sums stay
// pending: with fused_apply and no communicator the SGD kernel adds them; otherwise ONE launch reduces all of them into the gradient
// buffer behind the second fused launch (reduce_pending), and a communicator all-reduces that buffer behind it.
static bool tail_eligible(xq_dqn* d, int n) {
    if (!d->td_tail || d->bf16()) return false;
    for (int l = 0; l + 2 < d->nl; ++l)
        if (!d->small_tiles && (long long)((n + 127) / 128) * ((d->L[l + 1] + 127) / 128) >= 512) return false;   // launch_gemm would go 128x128
    return true;
}
// every pending partial-sum slab of the step summed into its place in the gradient buffer, one launch (the order of reduce_slabs_kernel)
static int reduce_pending(xq_dqn* d) {
    SegTable t; memset(&t, 0, sizeof t);
    float* G = d->grads_td;
    int k = 0;
    auto add = [&](float* dst, long long len, xq_dqn::PendingSlab& p) {
        if (p.nslabs > 0) { t.dst[k] = dst; t.src[k] = p.src; t.len[k] = len; t.nslabs[k] = p.nslabs; t.stride[k] = p.stride; ++k; }
        p = xq_dqn::PendingSlab();
    };
    if (d->l0_pending > 0) {
        xq_dqn::PendingSlab p; p.src = d->slabs_l0; p.nslabs = d->l0_pending; p.stride = (long long)d->L[0] * d->L[1];
        add(G + d->g_w0, p.stride, p);
        d->l0_pending = 0;
    }
    for (int l = 1; l + 1 < d->nl; ++l) add(G + d->g_wh[l], (long long)d->L[l] * d->L[l + 1], d->pend_hidden[l]);
    add(G + d->g_wout, 96LL * d->hlast(), d->pend_wout);
    add(G + d->g_bout, 96, d->pend_bout);
    add(G + d->g_bh[0], (long long)d->bo[d->nl - 1], d->pend_bh);
    if (k == 0) return XQ_OK;
    t.nseg = k; t.reduce_only = 1;
    return sgd_apply(d, t, 0.0);
}
static int tail_gradients_impl(xq_dqn* d, int n, float* const* outs, float* G, int mode);
static int tail_gradients(xq_dqn* d, int n, float* const* outs, float* G, int mode) {
    const bool leave_pending = d->fused();          // fused_apply, no communicator: the SGD kernel adds the slabs
    d->force_defer = true;
    int rc = tail_gradients_impl(d, n, outs, G, mode);
    d->force_defer = false;
    d->tail_open = false;                            // a failed assembly must not leave the launch helpers appending to a dead grid
    if (d->late_gate) {                              // the select chain starts here, beside the exchange — also behind a failed step: a
        // collect that waits for ev_qmax must find this step's record, not the previous one's
        if (hipEventRecord(d->ev_qmax, d->stream) != hipSuccess && rc == XQ_OK) rc = fail(XQ_ERR_RUNTIME, "hipEventRecord failed");
    }
    if (rc == XQ_OK && !leave_pending) {
        rc = reduce_pending(d);
        // data-parallel step: the whole buffer in one collective on the handle's stream, right behind its last producer
        if (rc == XQ_OK && d->comm) {
            ProfScope ps(d, "rccl_allreduce_grads", 0, 4.0 * d->n_grads_td);
            rc = comm_allreduce_on(d->comm, G, d->n_grads_td, d->stream);
        }
    }
    return rc;
}
static int tail_gradients_impl(xq_dqn* d, int n, float* const* outs, float* G, int mode) {
    const int nl = d->nl, Hl = d->hlast();
    const int chunk = 256;                           // out_grad: samples per block (see side_gradients)
    const int nchunks = (n + chunk - 1) / chunk;
    const long long len_out = 96LL * Hl + 96;
    size_t off_h[XQ_MAX_LAYERS] = {0}, need = 0;
    if (nchunks > 1) need += (size_t)nchunks * (size_t)len_out;
    for (int l = nl - 2; l >= 1; --l) {
        off_h[l] = need;
        const int sp = grad_splits(d, d->L[l + 1], d->L[l], n);
        if (sp > 1) need += (size_t)sp * (size_t)d->L[l + 1] * (size_t)d->L[l];
    }
    XQ_TRY(ensure_slabs(d, need));
    for (int l = 0; l < XQ_MAX_LAYERS; ++l) d->pend_hidden[l] = xq_dqn::PendingSlab();
    d->pend_wout = d->pend_bout = xq_dqn::PendingSlab();
    const size_t og_lds = (size_t)16 * Hl * sizeof(float) + (size_t)chunk * sizeof(uint16_t);
    if (og_lds > 64 * 1024 || (Hl & 3)) return fail(XQ_ERR_INVALID_ARGUMENT, "last hidden layer width %d unsupported by the output-gradient kernel (multiple of 4, <= 960)", Hl);
    auto add_out_grad = [&]() {
        TailArgs& T = *d->tail;
        float* out = nchunks > 1 ? d->slabs : G + d->g_wout;
        T.og_act = d->act_mb; T.og_dsc = d->dsc; T.og_alast = outs[nl - 2]; T.og_n = n; T.og_H = Hl; T.og_chunk = chunk; T.og_partial = out;
        T.n_out = 24 * nchunks;
        d->tail_flops += 2.0 * n * Hl; d->tail_bytes += (double)n * (Hl * 4 + 8) + 4.0 * nchunks * len_out;
        d->tail_lds = std::max(d->tail_lds, og_lds);
        if (nchunks > 1) {
            d->pend_wout.src = out; d->pend_wout.nslabs = nchunks; d->pend_wout.stride = len_out;
            d->pend_bout.src = out + 96LL * Hl; d->pend_bout.nslabs = nchunks; d->pend_bout.stride = len_out;
        }
    };
    // one launch per hidden layer below the top one: delta_l from delta_{l+1}, the weight gradient of layer l+1 (delta_{l+1}, a_l)
    // beside it, the output-layer sums beside the first
    auto add_grad = [&](int ll) -> int {             // weight gradient of hidden layer ll: delta_ll^T a_{ll-1}
        GemmArgs g; memset(&g, 0, sizeof g);
        g.M = d->L[ll + 1]; g.N = d->L[ll]; g.K = n;
        g.A = d->deltas[ll]; g.lda = d->L[ll + 1];
        g.B = outs[ll - 1]; g.ldb = d->L[ll];
        return grad_gemm<L_MCONTIG>(d, g, G + d->g_wh[ll], "gemm_grad_hidden", d->slabs + off_h[ll], &d->pend_hidden[ll]);
    };
    // two hidden layers: the one weight-gradient product runs beside the layer-0 sums (L2-bound) rather than beside the delta product
    // (both MFMA-bound, and the select chain's Q head is on the chip at that time): 0.2065 against 0.2086 ms per step of the headline
    // bench; three hidden layers (bench --config 4): no difference, kept beside the delta products
    const bool defer_grad = nl == 3;
    int waiting = -1;
    for (int l = nl - 3; l >= 0; --l) {
        tail_begin(d);
        XQ_TRY(hidden_deltas(d, n, d->deltas[l + 1], d->L[l + 2], d->L[l + 2], mode, l, l));
        if (defer_grad) { if (waiting >= 0) XQ_TRY(add_grad(waiting)); waiting = l + 1; }
        else XQ_TRY(add_grad(l + 1));
        if (l == nl - 3) add_out_grad();
        XQ_TRY(tail_launch(d, false, "td_tail_deltas"));
    }
    // last launch: the layer-0 sums and the bias column sums of every hidden delta (+ the output-layer sums of a net without a second
    // hidden layer)
    tail_begin(d);
    XQ_TRY(l0_gradient(d, n, G + d->g_w0));
    if (waiting >= 0) XQ_TRY(add_grad(waiting));
    if (nl < 3) add_out_grad();
    BiasJobs bj;
    for (int l = 0; l <= nl - 2; ++l) bj.add(d->deltas[l], d->L[l + 1], d->L[l + 1], G + d->g_bh[l]);
    d->pend_bh = xq_dqn::PendingSlab();
    XQ_TRY(bias_grads(d, bj, n, &d->pend_bh));
    return tail_launch(d, true, "td_tail_l0");
}

static int td_grads_impl(xq_dqn* d, const uint32_t* boards, const uint32_t* next_boards, const int32_t* action_to,
                         const float* reward, const uint8_t* done, SlotSrc slots, int n, int td_net, int mode, const PerOpts* per) {
    if (!d || !boards || !next_boards || !action_to || !reward || !done || n <= 0)
        return fail(XQ_ERR_INVALID_ARGUMENT, "xq_dqn_td_grads: bad argument");
    if (td_net != XQ_TD_ONLINE_NET && td_net != XQ_TD_TARGET_NET && td_net != XQ_TD_DOUBLE) return fail(XQ_ERR_INVALID_ARGUMENT, "bad td_net");
    if (mode != XQ_BACKPROP_REFERENCE && mode != XQ_BACKPROP_TEXTBOOK) return fail(XQ_ERR_INVALID_ARGUMENT, "bad backprop mode");
    if (d->nout() < 96) return fail(XQ_ERR_INVALID_ARGUMENT, "TD path needs >= 96 outputs (action.to indexes outputs 0..89)");
    if (mode == XQ_BACKPROP_REFERENCE) XQ_TRY(check_reference_topology(d));
    const bool dbl = td_net == XQ_TD_DOUBLE, bf = d->bf16();
    // Data-parallel step with more than one rank: the trainer's select chain (it waits for ev_qmax) starts behind the GRADIENTS
    // instead of behind the max pass, so that it runs beside the all-reduce — the exchange is then hidden behind work the step has to
    // do anyway, and the gradient kernels have the chip to themselves.  On one GPU there is nothing to hide behind and the early start
    // is the faster one.
    d->late_gate = d->comm != nullptr && tail_eligible(d, n) &&
                   (d->exchange_overlap == 1 || (d->exchange_overlap < 0 && comm_world(d->comm) > 1));
    XQ_TRY(ensure_capacity(d, n));
    XQ_TRY(ensure_ext_capacity(d, n, dbl));
    const int nl = d->nl, Hl = d->hlast(), NO = d->nout();
    if (bf && (Hl & 1)) return fail(XQ_ERR_INVALID_ARGUMENT, "bf16 Q-net needs even layer widths");
    // 1. the forward chains in the same launches (one gather grid, grouped hidden GEMMs): s on the online net with the
    //    activations kept for the backward pass, s' on the net that selects (online, or target for XQ_TD_TARGET_NET) and,
    //    for Double DQN, s' on the target net that evaluates
    const int sel_net = td_net == XQ_TD_TARGET_NET ? XQ_NET_TARGET : XQ_NET_ONLINE;
    float* outs[XQ_MAX_LAYERS]; uint16_t* outs_bf[XQ_MAX_LAYERS];
    float* touts[XQ_MAX_LAYERS]; uint16_t* touts_bf[XQ_MAX_LAYERS];
    float* t2outs[XQ_MAX_LAYERS]; uint16_t* t2outs_bf[XQ_MAX_LAYERS];
    for (int l = 0; l + 1 < nl; ++l) {
        outs[l] = d->acts[l]; outs_bf[l] = d->acts_bf[l];
        touts[l] = bf ? nullptr : d->tacts[l & 1]; touts_bf[l] = d->tacts_bf[l & 1];
        t2outs[l] = bf ? nullptr : d->t2acts[l & 1]; t2outs_bf[l] = d->t2acts_bf[l & 1];
    }
    // launched transposed (rows = output neurons, columns = samples): the max over the 8100 outputs then runs over
    // accumulator registers inside one lane instead of across the 32 lanes of a row
    const bool big_tiles = (long long)((NO + 127) / 128) * ((n + 127) / 128) >= 512;
    const int n_partial = 2 * (big_tiles ? (NO + 127) / 128 : (NO + 63) / 64);
    const size_t bias_lds_all = (size_t)((NO + 127) / 128) * 128 * sizeof(float);
    const bool want_screen = d->qmax_mode == XQ_QMAX_SCREENED && !bf && !dbl && big_tiles && (Hl % 64) == 0 && Hl <= 1024 &&
                          bias_lds_all <= 40 * 1024 && (NO + 127) / 128 * 4 <= 8 * kRefineMaxPerThread;
    bool screened = want_screen;
    bool td_fused = false;             // td_delta_kernel's work done inside the refine kernel (below)
    if (screened) {
        XQ_TRY(ensure_screen_capacity(d, n));
        // the counters queued at one check boundary are evaluated at the NEXT one (32 screened steps later: the copy finished long
        // ago, the wait returns at once) — the step at which a fallback begins is a function of the step count alone, never of how
        // far the host runs ahead of the device
        if (d->scr_guard_pending && d->scr_host_steps % kScreenCheckEvery == 0 && d->scr_host_steps != d->scr_guard_queued_at) {
            XQ_HIP(hipEventSynchronize(d->scr_guard_ev));
            d->scr_guard_pending = false;
            unsigned long long h[2];
            screen_stat_sums_of(d, d->scr_guard_host, h);
            const double ds = (double)(d->scr_guard_samples - d->scr_seen[0]);
            if (ds > 0 && ((double)(h[0] - d->scr_seen[1]) > kScreenMaxPairs * ds || (double)(h[1] - d->scr_seen[2]) > kScreenMaxWhole * ds)) {
                d->scr_hold = kScreenHoldSteps;
                d->scr_fallbacks += 1;
            }
            d->scr_seen[0] = d->scr_guard_samples; d->scr_seen[1] = h[0]; d->scr_seen[2] = h[1];
        }
        if (d->scr_hold > 0) { --d->scr_hold; screened = false; }
    }
    // bf16 copy + largest row norm / |bias| of the selecting net's output-layer weights.  Only rows 0..95 change under the TD rule:
    // while `scr_static_net` says that rows >= 96 of the shadow (and their maxima, slots [4], [5]) still belong to this net, the
    // shadow pass converts three blocks of 32 rows instead of 254; everything is converted again after set_params / load_model /
    // update_target / a dense backpropagate / a change of the selecting net.  The maxima of rows 0..95 land in the slots of this
    // step's parity, which the refine kernel of the previous screened step zeroed.
    const int parity = (int)(d->scr_host_steps & 1);
    // screen_top2_kernel (xq_screen.hip.h) for the widths it is built for; the persistent tile kernel's CM_TOP2 mode otherwise
    const bool scr_new = screened && (Hl == 256 || Hl == 512);
    ShadowJob shadow; memset(&shadow, 0, sizeof shadow);
    if (screened) {
        const bool full = d->scr_static_net != sel_net;
        if (full) XQ_HIP(hipMemsetAsync(d->scr_wmax + 4, 0, 2 * sizeof(unsigned), d->cur));
        shadow.W = d->wl(sel_net, nl - 1); shadow.bias = d->bl(sel_net, nl - 1); shadow.NO = NO; shadow.K = Hl; shadow.Wb = d->scr_wb;
        shadow.w_dyn = d->scr_wmax + parity; shadow.b_dyn = d->scr_wmax + 2 + parity;
        shadow.w_stat = d->scr_wmax + 4; shadow.b_stat = d->scr_wmax + 5;
        shadow.nblocks = full ? (NO + kShadowRows - 1) / kShadowRows : std::min((int)kShadowDynBlocks, (NO + kShadowRows - 1) / kShadowRows);
        d->scr_static_net = sel_net;
    }
    // bf16 net: the s' chain's last activations feed only the max / arg-max pass; when both that pass and the forward product run on
    // their own loops (whole tiles), the product writes them in fragment order
    const bool bf_frag = bf && nl >= 3 && (Hl == 256 || Hl == 512) && n >= 1024 && (n % kBgBM) == 0 && (d->L[nl - 2] % kBgBK) == 0;
    ChainJob jobs[3] = {{XQ_NET_ONLINE, boards, outs, outs_bf, d->gboards, nullptr, false},
                        {sel_net, next_boards, touts, touts_bf, nullptr, screened ? d->scr_ab : nullptr, scr_new || bf_frag},
                        {XQ_NET_TARGET, next_boards, t2outs, t2outs_bf, nullptr, nullptr, false}};
    XQ_TRY(chain_boards(d, jobs, dbl ? 3 : 2, slots, n, screened ? &shadow : nullptr));
    // The select chain of the trainer starts HERE when max_a' Q(s',a') runs on the bf16 matrix pipe (screening pass of an fp32 net, or
    // the output layer of a bf16 net): its layer-0 gather (L2-bound) then runs beside the screening pass (matrix-pipe-bound) and
    // is gone when the refine kernel — a chain of dependent memory round trips that the gather doubles in length — starts.  Same-box
    // A/B, round 4 (3 x 300 steps per leg): behind the screening pass 0.1946-0.1969 ms, here 0.1888-0.1937; behind the layer-0 gather of
    // this step 0.1886-0.1896 against 0.1914-0.1921; at the very top of the step no difference; --config 4 / 5 -0.3 % / -0.9 %.
    // The full fp32 product keeps the chip to itself: there the chain starts behind it (below).
    const bool gate_early = (screened || bf) && !d->late_gate;
    if (gate_early) XQ_HIP(hipEventRecord(d->ev_qmax, d->stream));
    int zparts = kReduceParts;
    if (screened) {
        const int tiles_m = (NO + 127) / 128, total = tiles_m * ((n + 127) / 128);
        // 32-row lane groups: the tile kernel writes all 4 per 128-row tile, screen_top2_kernel only those of 64-row chunks with real rows
        const int G = scr_new ? 2 * ((NO + 63) / 64) : 4 * tiles_m;
        long long ldp = n;
        int scr_ranges = 0, scr_gpr = 0;
        if (scr_new) {
            ScreenArgs a; memset(&a, 0, sizeof a);
            a.W = d->scr_wb; a.A = d->scr_ab; a.a_frag = 1; a.bias = d->bl(sel_net, nl - 1);
            a.P1 = d->scr_p1; a.P2 = d->scr_p2; a.R = d->scr_R; a.na = d->scr_na;
            screen_geometry(NO, n, Hl, d->ncu, a);
            a.ldp = screen_padded_samples(n, Hl);
            ldp = a.ldp;
            scr_ranges = a.ranges; scr_gpr = 2 * a.cpr;
            const size_t lds = screen_lds_bytes(a);
            if (!d->scr_new_kernel_ready) {
                XQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(screen_top2_kernel<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                XQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(screen_top2_kernel<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                d->scr_new_kernel_ready = true;
            }
            ProfScope ps(d, "gemm_qmax_screen", 2.0 * NO * (double)n * Hl, 2.0 * ((double)NO * Hl + (double)n * Hl) + 8.0 * G * n);
            if (Hl == 256) hipLaunchKernelGGL((screen_top2_kernel<1, 2>), dim3(a.panels * a.ranges), dim3(512), lds, d->cur, a);
            else hipLaunchKernelGGL((screen_top2_kernel<2, 1>), dim3(a.panels * a.ranges), dim3(512), lds, d->cur, a);
            XQ_HIP(hipGetLastError());
        } else {
            GemmArgs g; memset(&g, 0, sizeof g);
            g.M = NO; g.N = n;
            g.K = Hl / 2; g.lda = g.ldb = Hl / 2;
            g.A = reinterpret_cast<const float*>(d->scr_wb);
            g.B = reinterpret_cast<const float*>(d->scr_ab);
            g.bias = d->bl(sel_net, nl - 1);
            g.partial = d->scr_p1; g.partial2 = d->scr_p2;
            g.a_vec = g.b_vec = 1; g.k_chunk = g.K;
            g.bias_padded = ((((uintptr_t)g.bias) % 16 == 0) && (NO % 4) == 0) ? 1 : 0;
            const int grid = std::min(total, 2 * d->ncu);
            if (grid >= 2 && (grid & 1) == 0 && total >= 4 * grid) {
                g.prio_split = grid / 2;
                g.prio_tiles = (total / 2) / tiles_m * tiles_m;
                if (g.prio_tiles <= 0 || g.prio_tiles >= total) { g.prio_split = 0; g.prio_tiles = 0; }
            }
            ProfScope ps(d, "gemm_qmax_screen", 2.0 * g.M * g.N * Hl, 2.0 * ((double)g.M * Hl + (double)g.N * Hl) + 8.0 * G * g.N);
            hipLaunchKernelGGL((gemm_colmax_persistent_kernel<2, 2, DT_BF16, CM_TOP2>), dim3(grid), dim3(256), bias_lds_all, d->cur, g, tiles_m, total);
            XQ_HIP(hipGetLastError());
        }
        // (the select chain of the trainer started in front of the screening pass, above.  Behind the refine kernel instead — which would
        // then have the chip to itself, 17 instead of 27-34 us — the select chain ends after the gradients and the step waits for
        // it: 0.197 -> 0.207 ms)
        {
            ProfScope ps(d, "qmax_refine", 2.0 * n * Hl * 3, 12.0 * G * n + 4.0 * n * Hl);
            const size_t lds = (size_t)G * kRefineSamples * (sizeof(uint32_t) + sizeof(uint16_t));
            const dim3 grid((n + kRefineSamples - 1) / kRefineSamples);
            auto launch = [&](auto kern) {
                hipLaunchKernelGGL(kern, grid, dim3(256), lds, d->cur, d->scr_p1, d->scr_p2, G, n, ldp, touts[nl - 2], Hl, d->wl(sel_net, nl - 1),
                                   d->bl(sel_net, nl - 1), NO, d->scr_wmax, parity, d->zmax, d->scr_stats);
            };
            const bool small = G <= 8 * 32;
            if (scr_new) {
                TdFused T; memset(&T, 0, sizeof T);
                // the TD target / delta kernel rides in the refine blocks (fp32 net, 256-wide last hidden layer, uniform replay)
                td_fused = d->td_tail && Hl == 256 && !per && !d->bf16_bwd();
                if (td_fused) {
                    const int lt = nl - 2;
                    T.src = slots; T.action_to = action_to; T.reward = reward; T.done = done;
                    T.a_s = outs[lt]; T.w_out = d->wl(XQ_NET_ONLINE, nl - 1); T.b_out = d->bl(XQ_NET_ONLINE, nl - 1);
                    T.view = d->wrest(XQ_NET_ONLINE) + (d->wo[lt + 1] - d->wo[1]);
                    T.view_ld = (mode == XQ_BACKPROP_REFERENCE) ? d->L[lt] : d->L[lt + 1];
                    T.view_kmax = (mode == XQ_BACKPROP_REFERENCE) ? d->L[lt + 1] : NO;
                    T.gamma = (float)d->gamma;
                    T.dtop = d->deltas[lt]; T.dsc = d->dsc; T.act = d->act_mb; T.qsa = d->qsa; T.yv = d->yv; T.lossv = d->lossv;
                }
                auto launch2 = [&](auto kern) {
                    hipLaunchKernelGGL(kern, grid, dim3(256), lds, d->cur, d->scr_R, scr_ranges, scr_gpr, d->scr_p1, d->scr_p2, G, n, ldp, d->scr_na,
                                       touts[nl - 2], Hl, d->wl(sel_net, nl - 1), d->bl(sel_net, nl - 1), NO, d->scr_wmax, parity, d->zmax, d->scr_stats, T);
                };
                if (td_fused) launch2(qmax_refine2_kernel<256, true>);
                else if (Hl == 256) launch2(qmax_refine2_kernel<256>); else launch2(qmax_refine2_kernel<512>);
            } else
            if (Hl == 256) { if (small) launch(qmax_refine_kernel<256, 32>); else launch(qmax_refine_kernel<256, 64>); }
            else if (Hl == 512) { if (small) launch(qmax_refine_kernel<512, 32>); else launch(qmax_refine_kernel<512, 64>); }
            else { if (small) launch(qmax_refine_kernel<0, 32>); else launch(qmax_refine_kernel<0, 64>); }
            XQ_HIP(hipGetLastError());
        }
        d->scr_host_steps += 1; d->scr_host_samples += (unsigned long long)n;
        if (d->scr_host_steps % kScreenCheckEvery == 0 && !d->scr_guard_pending) {
            XQ_HIP(hipMemcpyAsync(d->scr_guard_host, d->scr_stats, (size_t)2 * d->scr_stat_blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost, d->cur));
            XQ_HIP(hipEventRecord(d->scr_guard_ev, d->cur));
            d->scr_guard_pending = true;
            d->scr_guard_samples = d->scr_host_samples;
            d->scr_guard_queued_at = d->scr_host_steps;
        }
        zparts = 1;
    } else {
    long long part_ld = n;              // row stride of the partial arrays
    int n_part = n_partial;
    if (bf && (Hl == 256 || Hl == 512) && n >= 1024) {
        // bf16 Q-net: the same kernel as the screening pass (xq_screen.hip.h) in its exact max / arg-max mode — here the bf16 product
        // IS the net's output layer, not a screen: per 32-row lane group the largest value (+ its row, first maximum: Double DQN)
        ScreenArgs a; memset(&a, 0, sizeof a);
        a.W = d->wl_bf(sel_net, nl - 1); a.A = touts_bf[nl - 2]; a.a_frag = bf_frag ? 1 : 0; a.bias = d->bl(sel_net, nl - 1);
        a.P1 = d->partial; a.P2 = reinterpret_cast<float*>(d->partial_idx);
        screen_geometry(NO, n, Hl, d->ncu, a);
        a.ldp = screen_padded_samples(n, Hl);
        part_ld = a.ldp; n_part = 2 * a.nchunks;
        const size_t lds = screen_lds_bytes(a);
        auto launch = [&](auto kern) {
            static bool ready = false;       // per instantiation
            if (!ready) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); ready = true; }
            hipLaunchKernelGGL(kern, dim3(a.panels * a.ranges), dim3(512), lds, d->cur, a);
        };
        ProfScope ps(d, "gemm_qmax_rowmax", 2.0 * NO * (double)n * Hl, 2.0 * ((double)NO * Hl + (double)n * Hl) + 8.0 * n_part * n);
        if (Hl == 256) { if (dbl) launch(screen_top2_kernel<1, 2, SCR_ARG>); else launch(screen_top2_kernel<1, 2, SCR_MAX>); }
        else { if (dbl) launch(screen_top2_kernel<2, 1, SCR_ARG>); else launch(screen_top2_kernel<2, 1, SCR_MAX>); }
        XQ_HIP(hipGetLastError());
    } else {
        GemmArgs g; memset(&g, 0, sizeof g);
        g.M = NO; g.N = n;
        if (bf) {
            g.K = Hl / 2; g.lda = g.ldb = Hl / 2;
            g.A = reinterpret_cast<const float*>(d->wl_bf(sel_net, nl - 1));
            g.B = reinterpret_cast<const float*>(touts_bf[nl - 2]);
        } else {
            g.K = Hl; g.lda = g.ldb = Hl;
            g.A = d->wl(sel_net, nl - 1);
            g.B = touts[nl - 2];
        }
        g.bias = d->bl(sel_net, nl - 1);
        g.partial = d->partial;
        g.partial_idx = dbl ? d->partial_idx : nullptr;
        // the persistent kernel keeps the whole bias vector in LDS beside its operand tiles (2 blocks per CU must fit)
        const size_t bias_lds = (size_t)((NO + 127) / 128) * 128 * sizeof(float);
        if (big_tiles && (g.K % GBK) == 0 && vec_ok(g.A, g.lda) && vec_ok(g.B, g.ldb) && bias_lds <= 40 * 1024) {
            // persistent form: 2 blocks per CU walk the tile list with the prefetch running across tile boundaries
            const int tiles_m = (NO + 127) / 128, total = tiles_m * ((n + 127) / 128);
            const int ncu = d->ncu;
            g.a_vec = g.b_vec = 1; g.k_chunk = g.K;
            g.bias_padded = ((((uintptr_t)g.bias) % 16 == 0) && (NO % 4) == 0) ? 1 : 0;     // wide bias preload
            const int grid = std::min(total, 2 * ncu);
            // static priority for the second half of the grid (see the kernel): each half walks its own half of the tiles
            g.prio_split = 0; g.prio_tiles = 0;
            if (grid >= 2 && (grid & 1) == 0 && total >= 4 * grid) {
                g.prio_split = grid / 2;
                g.prio_tiles = (total / 2) / tiles_m * tiles_m;
                if (g.prio_tiles <= 0 || g.prio_tiles >= total) { g.prio_split = 0; g.prio_tiles = 0; }
            }
            ProfScope ps(d, "gemm_qmax_rowmax", 2.0 * g.M * g.N * Hl, (bf ? 2.0 : 4.0) * ((double)g.M * Hl + (double)g.N * Hl) + 8.0 * tiles_m * g.N);
            if (bf && dbl) hipLaunchKernelGGL((gemm_colmax_persistent_kernel<2, 2, DT_BF16, CM_ARG>), dim3(grid), dim3(256), bias_lds, d->cur, g, tiles_m, total);
            else if (bf) hipLaunchKernelGGL((gemm_colmax_persistent_kernel<2, 2, DT_BF16, CM_MAX>), dim3(grid), dim3(256), bias_lds, d->cur, g, tiles_m, total);
            else if (dbl) hipLaunchKernelGGL((gemm_colmax_persistent_kernel<2, 2, DT_F32, CM_ARG>), dim3(grid), dim3(256), bias_lds, d->cur, g, tiles_m, total);
            else hipLaunchKernelGGL((gemm_colmax_persistent_kernel<2, 2, DT_F32, CM_MAX>), dim3(grid), dim3(256), bias_lds, d->cur, g, tiles_m, total);
            XQ_HIP(hipGetLastError());
        } else if (bf) {
            XQ_GEMM((launch_gemm<L_KCONTIG, L_KCONTIG, EPI_COLMAX, DT_BF16>(d, g, 1, "gemm_qmax_rowmax")));
        } else {
            XQ_GEMM((launch_gemm<L_KCONTIG, L_KCONTIG, EPI_COLMAX>(d, g, 1, "gemm_qmax_rowmax")));
        }
    }
    if (!d->late_gate && !gate_early) XQ_HIP(hipEventRecord(d->ev_qmax, d->stream));
    {   // the partial maxima of every sample folded into kReduceParts values (+ row indices): coalesced, block-cooperative
        ProfScope ps(d, "colmax_reduce", (double)n * n_part, (dbl ? 8.0 : 4.0) * n * (n_part + kReduceParts));
        hipLaunchKernelGGL(colmax_reduce_kernel, dim3((n + 63) / 64, kReduceParts), dim3(256), 0, d->cur, d->partial,
                           dbl ? d->partial_idx : nullptr, n_part, n, part_ld, d->zmax, d->zidx);
        XQ_HIP(hipGetLastError());
    }
    }   // !screened
    // 3. Q(s, a), target, the scalar output delta and the delta of the last hidden layer (one launch, no GEMM)
    if (!td_fused) {
        const int lt = nl - 2;                               // last hidden layer
        const float* view = d->wrest(XQ_NET_ONLINE) + (d->wo[lt + 1] - d->wo[1]);
        const long long view_ld = (mode == XQ_BACKPROP_REFERENCE) ? d->L[lt] : d->L[lt + 1];
        const int view_kmax = (mode == XQ_BACKPROP_REFERENCE) ? d->L[lt + 1] : NO;
        ProfScope ps(d, "td_target_delta", 4.0 * n * Hl, (double)n * (Hl * 16 + n_partial * 4));
        TdExtra X; memset(&X, 0, sizeof X);
        X.nout = NO;
        if (dbl) {
            X.double_dqn = 1; X.partial_idx = d->zidx;
            X.bout_t = d->bl(XQ_NET_TARGET, nl - 1);
            if (bf) { X.wout_t_bf = d->wl_bf(XQ_NET_TARGET, nl - 1); X.alast_t_bf = t2outs_bf[nl - 2]; }
            else { X.wout_t = d->wl(XQ_NET_TARGET, nl - 1); X.alast_t = t2outs[nl - 2]; }
        }
        if (bf) X.wout_bf = d->wl_bf(XQ_NET_ONLINE, nl - 1);
        if (d->bf16_bwd()) X.dtop_bf = d->deltas_bf[lt];
        if (per) {
            X.is_w = per->is_w; X.is_wmax = per->is_wmax; X.prio = per->prio; X.pmax_live = per->pmax_live;
            X.per_eps = per->eps; X.per_alpha = per->alpha;
        }
        hipLaunchKernelGGL(td_delta_kernel, dim3((n + 3) / 4), dim3(256), 0, d->cur, n, slots, action_to, reward, done,
                           outs[nl - 2], Hl, d->wl(XQ_NET_ONLINE, nl - 1), d->bl(XQ_NET_ONLINE, nl - 1), d->zmax, zparts,
                           (float)d->gamma, view, view_ld, view_kmax, d->deltas[lt], d->dsc, d->act_mb, d->qsa, d->yv, d->lossv, X);
        XQ_HIP(hipGetLastError());
    }
    d->last_n = n;
    // From here two chains run side by side.  Critical (handle stream): the remaining hidden deltas (GEMMs) and the
    // layer-0 segmented sum that consumes delta_0.  Side stream: everything that only needs what td_delta_kernel wrote —
    // the output-layer gradient, the top hidden layer's gradient GEMM — and, once the deltas exist, the lower gradient
    // GEMMs and the bias column sums.
    float* G = d->grads_td;
    if (tail_eligible(d, n)) return tail_gradients(d, n, outs, G, mode);
    XQ_HIP(hipEventRecord(d->ev_fork, d->stream));
    XQ_HIP(hipStreamWaitEvent(d->side, d->ev_fork, 0));
    if (nl >= 3) XQ_TRY(hidden_deltas(d, n, d->deltas[nl - 2], d->L[nl - 1], d->L[nl - 1], mode, nl - 3));
    XQ_HIP(hipEventRecord(d->ev_delta, d->stream));
    const size_t n0 = (size_t)d->L[0] * d->L[1];
    if (d->comm) {
        // data-parallel step: the gradient buffer is all-reduced in two buckets, each ON THE STREAM OF ITS PRODUCER right behind
        // it — no communicator stream, no event of its own.  RCCL runs the collectives of one communicator in issue order, so
        // the side bucket (hidden + output-layer weights, all biases: ready first) is issued first and the layer-0 bucket
        // (the last thing computed, the only exposed one) second; every rank issues in this order.
        d->cur = d->side;
        int rc = side_gradients(d, n, outs, G);
        d->cur = d->stream;
        if (rc != XQ_OK) return rc;
        XQ_TRY(comm_allreduce_on(d->comm, G + n0, d->n_grads_td - n0, d->side));
        XQ_HIP(hipEventRecord(d->ev_join, d->side));
        XQ_TRY(l0_gradient(d, n, G + d->g_w0));
        XQ_TRY(comm_allreduce_on(d->comm, G, n0, d->stream));
        XQ_HIP(hipStreamWaitEvent(d->stream, d->ev_join, 0));
        return XQ_OK;
    }
    XQ_TRY(l0_gradient(d, n, G + d->g_w0));
    d->cur = d->side;
    const int rc = side_gradients(d, n, outs, G);
    d->cur = d->stream;
    if (rc != XQ_OK) return rc;
    XQ_HIP(hipEventRecord(d->ev_join, d->side));
    XQ_HIP(hipStreamWaitEvent(d->stream, d->ev_join, 0));
    return XQ_OK;
}

int xq_dqn_set_comm(xq_dqn* d, xq_comm* comm) {
    if (!d) return fail(XQ_ERR_INVALID_ARGUMENT, "null dqn");
    if (d->l0_pending > 0 || d->pend_wout.nslabs > 0 || d->pend_bh.nslabs > 0)
        return fail(XQ_ERR_RUNTIME, "xq_dqn_set_comm: a TD step is waiting for its apply_grads");
    d->comm = comm;
    return XQ_OK;
}

int xq_allreduce_grads(xq_dqn* d, xq_comm* comm) {
    if (!d || !comm) return fail(XQ_ERR_INVALID_ARGUMENT, "null handle");
    if (d->l0_pending > 0 || d->pend_wout.nslabs > 0 || d->pend_bh.nslabs > 0)
        return fail(XQ_ERR_RUNTIME, "xq_allreduce_grads: gradient slabs are still unreduced (xq_dqn_set_fused_apply is on)");
    return comm_allreduce_on(comm, d->grads_td, d->n_grads_td, d->stream);      // in order on the handle's stream
}

int xq_dqn_apply_grads(xq_dqn* d, double lr, double grad_scale) {
    if (!d) return fail(XQ_ERR_INVALID_ARGUMENT, "null dqn");
    SegTable t; memset(&t, 0, sizeof t);
    const float* G = d->grads_td;
    int k = 0;
    auto take = [&](xq_dqn::PendingSlab& p) {
        if (p.nslabs > 0) { t.src[k] = p.src; t.nslabs[k] = p.nslabs; t.stride[k] = p.stride; }
        p = xq_dqn::PendingSlab();
    };
    const bool bf = d->bf16();
    t.dst[k] = d->w0t(0); t.src[k] = G + d->g_w0; t.len[k] = (long long)d->L[0] * d->L[1];
    if (bf) t.dst_bf[k] = d->wl_bf(0, 0);
    if (d->l0_pending > 0) { t.src[k] = d->slabs_l0; t.nslabs[k] = d->l0_pending; t.stride[k] = t.len[k]; d->l0_pending = 0; }
    ++k;
    for (int l = 1; l + 1 < d->nl; ++l) {
        t.dst[k] = d->wl(0, l); t.src[k] = G + d->g_wh[l]; t.len[k] = (long long)d->L[l] * d->L[l + 1];
        if (bf) t.dst_bf[k] = d->wl_bf(0, l);
        take(d->pend_hidden[l]);
        ++k;
    }
    t.dst[k] = d->wl(0, d->nl - 1); t.src[k] = G + d->g_wout; t.len[k] = 96LL * d->hlast();
    if (bf) t.dst_bf[k] = d->wl_bf(0, d->nl - 1);
    take(d->pend_wout); ++k;
    // hidden biases are contiguous in both layouts
    t.dst[k] = d->bl(0, 0); t.src[k] = G + d->g_bh[0]; t.len[k] = (long long)(d->bo[d->nl - 1]); take(d->pend_bh); ++k;
    t.dst[k] = d->bl(0, d->nl - 1); t.src[k] = G + d->g_bout; t.len[k] = 96; take(d->pend_bout); ++k;
    t.nseg = k;
    d->sel_invalidate();                              // W0 / b0 change: the select chain's kept layer-0 sums are stale
    for (auto& K : d->sel_keep) { K.pays = K.calls >= 2; K.calls = 0; }
    d->params_version += 1;
    return sgd_apply(d, t, lr * grad_scale);
}

int xq_dqn_set_fused_apply(xq_dqn* d, int on) {
    if (!d) return fail(XQ_ERR_INVALID_ARGUMENT, "null dqn");
    if (d->l0_pending > 0 || d->pend_wout.nslabs > 0 || d->pend_bh.nslabs > 0)
        return fail(XQ_ERR_RUNTIME, "xq_dqn_set_fused_apply: a TD step is waiting for its apply_grads");
    d->fused_apply = on != 0;
    return XQ_OK;
}

int xq_dqn_grad_buffer(xq_dqn* d, float** grads_dev, size_t* n_floats) {
    if (!d) return fail(XQ_ERR_INVALID_ARGUMENT, "null dqn");
    if (grads_dev) *grads_dev = d->grads_td;
    if (n_floats) *n_floats = d->n_grads_td;
    return XQ_OK;
}

int xq_dqn_td_grads_replay(xq_dqn* d, xq_replay* r, int batch, int td_net, int mode) {
    if (!d || !r) return fail(XQ_ERR_INVALID_ARGUMENT, "null handle");
    SlotSrc src = explicit_slots(nullptr);
    if (batch > 0) {
        if (r->last_batch != batch) return fail(XQ_ERR_INVALID_ARGUMENT, "call xq_replay_sample(batch) first");
        if (r->implicit) {                  // the trainer's virtual sample: slots recomputed inside the consumer kernels
            src.implicit = 1; src.call = r->implicit_call; src.size = (uint32_t)r->implicit_size;
            src.start = (uint32_t)r->implicit_start; src.cap = (uint32_t)r->dev.capacity;
            src.seed_lo = (uint32_t)r->seed; src.seed_hi = (uint32_t)(r->seed >> 32);
        } else {
            if (!r->slots_dev) return fail(XQ_ERR_INVALID_ARGUMENT, "call xq_replay_sample(batch) first");
            src.slots = r->slots_dev;
        }
    } else {
        batch = r->size;       // identity over the filled part of the ring (on-policy use)
    }
    if (batch <= 0) return fail(XQ_ERR_RUNTIME, "replay is empty");
    PerOpts per; memset(&per, 0, sizeof per);
    const bool prioritized = r->per.enabled && r->per.last_prioritized && !r->implicit && src.slots != nullptr;
    if (prioritized) {        // importance weights in, TD-error priorities out (xq_replay_sample_prioritized drew the slots)
        per.is_w = r->per.is_w; per.is_wmax = reinterpret_cast<const float*>(r->per.scalars + 2);
        per.prio = r->dev.prio; per.pmax_live = r->per.scalars + 0;
        per.eps = r->per.eps; per.alpha = r->per.alpha;
    }
    // ring, slot list and priorities may have been written on other streams (env steps, the draw, a rebuild): this step starts behind
    // them, and whatever touches them next — the next draw overwrites the list, the next env step the slots — behind this step
    XQ_TRY(replay_consumer_begin(r, d->stream, src.slots != nullptr, prioritized));
    return td_grads_impl(d, r->dev.boards, r->dev.next_boards, r->dev.action_to, r->dev.reward, r->dev.done, src, batch, td_net,
                         mode, prioritized ? &per : nullptr);
}

int xq_dqn_td_update_host(xq_dqn* d, int n, const uint8_t* boards90, const uint8_t* next_boards90, const int32_t* action_to,
                          const float* reward, const uint8_t* done, int td_net, int mode, double lr, double grad_scale,
                          float* q_sa_out, float* y_out) {
    if (!d || n <= 0 || !boards90 || !next_boards90 || !action_to || !reward || !done)
        return fail(XQ_ERR_INVALID_ARGUMENT, "xq_dqn_td_update_host: bad argument");
    for (size_t i = 0; i < (size_t)n * 90; ++i)       // code 15 would index one-hot plane 14 of 14 in the layer-0 kernels
        if (boards90[i] > 14 || next_boards90[i] > 14) return fail(XQ_ERR_INVALID_ARGUMENT, "piece code > 14");
    if ((size_t)n > d->hb_cap) {
        XQ_HIP(hipStreamSynchronize(d->stream));
        hipFree(d->hb); hipFree(d->ha); hipFree(d->hr); hipFree(d->hd);
        XQ_HIP(hipMalloc(&d->hb, (size_t)n * 2 * kBoardWords * sizeof(uint32_t)));
        XQ_HIP(hipMalloc(&d->ha, (size_t)n * sizeof(int32_t)));
        XQ_HIP(hipMalloc(&d->hr, (size_t)n * sizeof(float)));
        XQ_HIP(hipMalloc(&d->hd, (size_t)n));
        d->hb_cap = (size_t)n;
    }
    std::vector<uint32_t> w((size_t)n * 2 * kBoardWords);
    for (int i = 0; i < n; ++i) {
        pack_board(boards90 + (size_t)i * 90, &w[(size_t)i * kBoardWords]);
        pack_board(next_boards90 + (size_t)i * 90, &w[((size_t)n + i) * kBoardWords]);
    }
    XQ_HIP(hipStreamSynchronize(d->stream));
    XQ_HIP(hipMemcpy(d->hb, w.data(), w.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    XQ_HIP(hipMemcpy(d->ha, action_to, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
    XQ_HIP(hipMemcpy(d->hr, reward, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    XQ_HIP(hipMemcpy(d->hd, done, (size_t)n, hipMemcpyHostToDevice));
    XQ_TRY(xq_dqn_td_grads(d, d->hb, d->hb + (size_t)n * kBoardWords, d->ha, d->hr, d->hd, nullptr, n, td_net, mode));
    XQ_TRY(xq_dqn_apply_grads(d, lr, grad_scale));
    XQ_HIP(hipStreamSynchronize(d->stream));
    if (q_sa_out) XQ_HIP(hipMemcpy(q_sa_out, d->qsa, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    if (y_out) XQ_HIP(hipMemcpy(y_out, d->yv, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return XQ_OK;
}

int xq_dqn_last_loss(xq_dqn* d, double* loss) {
    if (!d || !loss) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    std::vector<float> v((size_t)std::max(d->last_n, 0));
    XQ_HIP(hipStreamSynchronize(d->stream));
    if (!v.empty()) XQ_HIP(hipMemcpy(v.data(), d->lossv, v.size() * sizeof(float), hipMemcpyDeviceToHost));
    double s = 0;
    for (float x : v) s += x;
    *loss = s;
    return XQ_OK;
}

int xq_dqn_last_td_values(xq_dqn* d, int n, float* q_sa_host, float* y_host) {
    if (!d || n < 0 || n > d->last_n) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_dqn_last_td_values: n exceeds the last TD step's %d samples", d ? d->last_n : 0);
    XQ_HIP(hipStreamSynchronize(d->stream));
    if (q_sa_host && n) XQ_HIP(hipMemcpy(q_sa_host, d->qsa, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    if (y_host && n) XQ_HIP(hipMemcpy(y_host, d->yv, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return XQ_OK;
}

int xq_dqn_kernel_stats(xq_dqn* d, int enable, xq_kernel_stat* stats, int max_stats, int* n_stats) {
    if (!d) return fail(XQ_ERR_INVALID_ARGUMENT, "null dqn");
    XQ_HIP(hipStreamSynchronize(d->stream));
    d->prof.collect();
    int n = 0;
    if (stats && n_stats) {
        for (size_t i = 0; i < d->prof.cats.size() && n < max_stats; ++i, ++n) {
            const Profiler::Cat& c = d->prof.cats[i];
            memset(&stats[n], 0, sizeof stats[n]);
            strncpy(stats[n].name, c.name, sizeof stats[n].name - 1);
            stats[n].ms = c.ms; stats[n].launches = c.launches; stats[n].flops = c.flops; stats[n].bytes = c.bytes;
        }
    }
    if (n_stats) *n_stats = n;
    if (enable >= 0) {   // 0 off, 1 on, 2 on + clear, 3 on + clear, only the kernels bench.py prices (roofline leg), 4 = 3 sampled 1-in-4
        if ((enable != 0) != d->prof.enabled || enable >= 2) d->prof.reset();
        d->prof.enabled = enable != 0;
        d->prof.roofline_only = enable == 3 || enable == 4;
        d->prof.sample_period = enable == 4 ? 4 : 1;
        for (auto& o : d->prof.only) o.phase = 0;
    }
    return XQ_OK;
}

int xq_dqn_kernel_filter(xq_dqn* d, const char* names_csv) {
    if (!d) return fail(XQ_ERR_INVALID_ARGUMENT, "null dqn");
    d->prof.set_only(names_csv && *names_csv ? names_csv : "gemm_qmax_rowmax,gemm_qmax_screen,env_selfplay_step");
    return XQ_OK;
}

int xq_dqn_kernel_timeline(xq_dqn* d, xq_kernel_span* spans, int max_spans, int* n_spans) {
    if (!d || !n_spans) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    int n = 0;
    for (size_t i = 0; i < d->prof.spans.size() && spans && n < max_spans; ++i, ++n) {
        const Profiler::Span& sp = d->prof.spans[i];
        memset(&spans[n], 0, sizeof spans[n]);
        memcpy(spans[n].name, sp.name, sizeof spans[n].name);
        spans[n].start_ms = sp.start_ms; spans[n].end_ms = sp.end_ms;
    }
    *n_spans = n;
    return XQ_OK;
}

// DQN::saveModel / loadModel, dqn.cpp:76-154
static void put_be(FILE* f, uint64_t v, int bytes) { for (int i = bytes - 1; i >= 0; --i) fputc((int)((v >> (8 * i)) & 0xFF), f); }
static bool get_be(FILE* f, uint64_t* v, int bytes) {
    uint64_t x = 0;
    for (int i = 0; i < bytes; ++i) { int c = fgetc(f); if (c == EOF) return false; x = (x << 8) | (uint64_t)(c & 0xFF); }
    *v = x;
    return true;
}

int xq_dqn_save_model(xq_dqn* d, const char* path) {
    if (!d || !path) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    std::vector<double> w(d->nw), b(d->nb);
    XQ_TRY(xq_dqn_get_params(d, XQ_NET_ONLINE, w.data(), b.data()));
    FILE* f = fopen(path, "wb");
    if (!f) return fail(XQ_ERR_IO, "Unable to open file for saving model.");
    bool ok = fwrite(w.data(), sizeof(double), w.size(), f) == w.size();       // raw (little-endian host) fp64
    ok = ok && fwrite(b.data(), sizeof(double), b.size(), f) == b.size();
    put_be(f, (uint64_t)d->ns, 8);                                             // QDataStream: big-endian quint64
    for (int i = 0; i < d->ns; ++i) put_be(f, (uint64_t)(uint32_t)d->L[i], 4); // big-endian qint32
    ok = ok && !ferror(f);
    fclose(f);
    return ok ? XQ_OK : fail(XQ_ERR_IO, "Error writing weights to model file.");
}

int xq_dqn_load_model(xq_dqn* d, const char* path) {
    if (!d || !path) return fail(XQ_ERR_INVALID_ARGUMENT, "null pointer");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(XQ_ERR_IO, "Unable to open file for loading model.");
    std::vector<double> w(d->nw), b(d->nb);
    bool ok = fread(w.data(), sizeof(double), w.size(), f) == w.size();
    ok = ok && fread(b.data(), sizeof(double), b.size(), f) == b.size();
    if (!ok) { fclose(f); return fail(XQ_ERR_IO, "Error reading weights from model file."); }
    uint64_t cnt = 0;
    ok = get_be(f, &cnt, 8);
    std::vector<int> sizes;
    for (uint64_t i = 0; ok && i < cnt && i < 64; ++i) { uint64_t v; ok = get_be(f, &v, 4); sizes.push_back((int)(uint32_t)v); }
    fclose(f);
    bool same = ok && cnt == (uint64_t)d->ns;
    for (int i = 0; same && i < d->ns; ++i) same = sizes[i] == d->L[i];
    if (!same) return fail(XQ_ERR_IO, "Layer sizes in the model file do not match the current network architecture.");
    return xq_dqn_set_params(d, XQ_NET_ONLINE, w.data(), b.data());           // qNetwork only, like upstream (:153)
}

}  // extern "C"
