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
// This file = the handle (struct xq_dqn) + host orchestration + the C ABI; the kernels live in xq_l0.hip.h (layer-0 gathers and
// segmented sums), xq_tail.hip.h (TD delta, gradient sums, fused launches, SGD), xq_refine.hip.h (screening pass 2), xq_l0grad.hip.h
// (layer-0 gradient on the matrix pipe), xq_gemm*.hip.h and xq_screen.hip.h (matrix-pipe products).
#include "xq_internal.h"
#include <hip/hip_ext.h>
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
    hipEvent_t fwd_stop_ev = nullptr;           // set: the last grouped forward product of chain_boards completes this event itself (taken = reset)
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
    int refine_stage = -1;                             // xq_dqn_set_refine_stage: -1 by the counters (scr_stage_whole), 0 never, 1 whenever it fits
    bool scr_stage_whole = false;                      // the last counter window showed whole groups for many samples: the refine kernel is
                                                       // launched with the LDS of its staged pass (145 KB: otherwise it would keep the select
                                                       // chain's blocks off its CUs for nothing — +3 us per step, same-box A/B)
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
    uint16_t* l0_sel = nullptr;     size_t l0_sel_cap = 0;       // selector half-words of the minibatch's boards (xq_l0grad.hip.h)
    bool l0_split_done = false;                 // this step's layer-0 delta product wrote the planes in its epilogue
    bool l0_sel_done = false;                   // this step's selector words rode in an earlier fused launch
    bool l0_mfma = true;                        // xq_dqn_set_l0_grad_mode (default since round 5: xq_l0grad.hip.h)
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
    // xq_dqn_calibrate_exchange: mean duration of an all-reduce of the gradient buffer on this communicator, the threshold it was held
    // against and the start of the select chain chosen from the two (exchange_overlap == -1 only)
    bool exch_calibrated = false, exch_late = false;
    double exch_allreduce_us = 0.0, exch_threshold_us = 0.0;
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
    // attach: the scope is ONE kernel launch that takes start() / stop() as its own events (hipExtLaunchKernelGGL): kernel-exact time
    ProfScope(xq_dqn* d, const char* name, double fl, double by, bool attach = false) : p(d->prof), s(d->cur), flops(fl), bytes(by) {
        h = -1;
        if (d->tail_open || !p.enabled) return;
        if (d->cur != d->stream && d->cur != d->side) {
            char nm[48];
            snprintf(nm, sizeof nm, "%s@select", name);
            h = p.begin(nm, s, attach);
        } else {
            h = p.begin(name, s, attach);
        }
    }
    hipEvent_t start() const { return h >= 0 ? p.recs[h].a : nullptr; }
    hipEvent_t stop() const { return h >= 0 ? p.recs[h].b : nullptr; }
    ~ProfScope() { p.end(h, s, flops, bytes); }
};

}  // namespace xq

// kernels (each header opens namespace xq itself)
#include "xq_l0.hip.h"
#include "xq_tail.hip.h"
#include "xq_refine.hip.h"

namespace xq {

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
        if (const char* e = getenv("XQ_DEBUG_PTRS")) if (e[0] == '1')
            fprintf(stderr, "[xq] screening buffers: P1 %p P2 %p R %p ab %p na %p params %p acts %p\n", (void*)d->scr_p1, (void*)d->scr_p2, (void*)d->scr_R,
                    (void*)d->scr_ab, (void*)d->scr_na, (void*)d->params[0], (void*)d->tacts[0]);
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
        ProfScope ps(d, "l0_forward_gather", 2.0 * n * 32 * H, (double)n * (96 + (jobs[0].sel_keep == 2 ? 4.0 : 32.0) * H * 4 + H * 12), true);
        xq_dqn::SelKeep& K = d->sel_keep[d->cur == d->stream ? 0 : 1];
        hipExtLaunchKernelGGL(l0_select_kernel, dim3((n + 3) / 4), dim3(256), 0, d->cur, ps.start(), ps.stop(), 0, jobs[0].boards, K.prev_boards, d->w0t(jobs[0].net),
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
        ProfScope ps(d, "l0_forward_gather", 2.0 * njobs * n * 32 * H, (double)njobs * n * (48 + 32.0 * H * (bf ? 2 : 4) + H * 4), true);
        // (grid rows: the job rows that are really gathered, then the shadow row; the kernel tests blockIdx.y == J.nrows for it)
        if (bf) hipExtLaunchKernelGGL(l0_forward_kernel<true>, dim3((n + 3) / 4, J.nrows), dim3(256), 0, d->cur, ps.start(), ps.stop(), 0, J, src, n, H);
        else hipExtLaunchKernelGGL(l0_forward_kernel<false>, dim3((n + 3) / 4, J.nrows + (ride ? 1 : 0)), dim3(256), 0, d->cur, ps.start(), ps.stop(), 0, J, src, n, H);
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
                ProfScope ps(d, "gemm_hidden_fwd", 2.0 * n * g.N * (g.K + 96.0), 4.0 * ((double)n * g.K + (double)g.N * g.K + (double)(g.N / 64) * n * 96), true);
                hipExtLaunchKernelGGL((gemm_f32_kernel<L_KCONTIG, L_KCONTIG, EPI_HEAD, 1, 1>), dim3(n / 64, g.N / 64, 1), dim3(256), 0, d->cur, ps.start(), ps.stop(), 0, g);
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
                             4.0 * groups * ((double)n * g.K + (double)g.N * g.K + (double)n * g.N), true);
                // the event the trainer's select chain waits for rides on this kernel's own completion signal when it is the last of
                // the chains (no marker packet on the stream: ~1.5 us less than a record, tools/sync_probe.hip variant 5)
                // (a launch that is being timed carries the profiler's events instead: the fork is then recorded by the caller)
                hipEvent_t stop = (l == d->nl - 2 && d->cur == d->stream && !ps.stop()) ? d->fwd_stop_ev : nullptr;
                if (stop) d->fwd_stop_ev = nullptr;
                hipEvent_t start = ps.start();
                if (ps.stop()) stop = ps.stop();
                if (t128 >= 512) {
                    hipExtLaunchKernelGGL((gemm_fwd_persistent_kernel<2, 2, 2>), dim3(std::min(t128, 2 * d->ncu)), dim3(256), 0, d->cur, start, stop, 0,
                                          g, n / 128, g.N / 128, t128);
                } else {
                    const int total = (n / 64) * (g.N / 128) * groups;
                    hipExtLaunchKernelGGL((gemm_fwd_persistent_kernel<1, 2, 2>), dim3(std::min(total, 2 * d->ncu)), dim3(256), 0, d->cur, start, stop, 0,
                                          g, n / 64, g.N / 128, total);
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
// matrix-pipe form of the layer-0 gradient (xq_dqn_set_l0_grad_mode(1), xq_l0grad.hip.h): taken at these shapes
static bool l0_mfma_shape(const xq_dqn* d, int n) { return d->l0_mfma && (d->L[1] % 64) == 0 && n >= 256; }
static int l0_chunk_of(int n) { return n >= 16384 ? 2048 : 1024; }
static int ensure_l0_mfma(xq_dqn* d, int n) {
    const int H = d->L[1], chunk = l0_chunk_of(n), kpad = (n + chunk - 1) / chunk * chunk;
    const size_t need = l0m_plane_elems(H, kpad);
    if (need > d->l0_planes_cap) {
        if (d->l0_planes) { XQ_HIP(hipDeviceSynchronize()); XQ_HIP(hipFree(d->l0_planes)); }
        XQ_HIP(hipMalloc(&d->l0_planes, need * sizeof(uint16_t)));
        XQ_HIP(hipMemsetAsync(d->l0_planes, 0, need * sizeof(uint16_t), d->cur));      // (the slack behind the planes is read, never used)
        d->l0_planes_cap = need;
    }
    const size_t need_sel = l0sel_elems(kpad);
    if (need_sel > d->l0_sel_cap) {
        if (d->l0_sel) { XQ_HIP(hipDeviceSynchronize()); XQ_HIP(hipFree(d->l0_sel)); }
        XQ_HIP(hipMalloc(&d->l0_sel, need_sel * sizeof(uint16_t)));
        XQ_HIP(hipMemsetAsync(d->l0_sel, 0, need_sel * sizeof(uint16_t), d->cur));
        d->l0_sel_cap = need_sel;
    }
    return XQ_OK;
}

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
        if (l == 0 && l0_mfma_shape(d, n) && (n % l0_chunk_of(n)) == 0) {
            // delta_0's planes for the matrix-pipe layer-0 gradient straight from this product's epilogue (64 x 64 tiles; whole tiles
            // and whole chunks only: the planes' zero padding is then empty) — no split launch, no re-read of delta_0
            const long long t128 = (long long)((g.M + 127) / 128) * ((g.N + 127) / 128);
            if (d->small_tiles || t128 < 512) {
                XQ_TRY(ensure_l0_mfma(d, n));
                g.split_planes = d->l0_planes; g.split_ld = n; g.split_plane_stride = (long long)d->L[1] * n;
                d->l0_split_done = true;
            }
        }
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
    const int chunk = l0_chunk_of(n);
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
    if (l0_mfma_shape(d, n)) {
        // the matrix-pipe form (xq_l0grad.hip.h): boards -> selector half-words, delta_0 -> three transposed bf16 planes, then
        // one-hot^T x planes on v_mfma_f32_16x16x32_bf16 — exact products, fp32 accumulation
        const int kpad = nchunks * chunk;
        XQ_TRY(ensure_l0_mfma(d, n));
        const size_t need_sel = l0sel_elems(kpad);
        const long long plane_stride = (long long)H * kpad;
        const bool was_open = d->tail_open;
        d->tail_open = false;                        // launches of their own, in front of the fused launch that is being assembled
        if (!d->l0_sel_done) {                       // (fused TD step of a net with >= 2 hidden layers: rode in the launch before this one)
            ProfScope ps(d, "l0_sel_words", 0.0, (double)n * 48 + 2.0 * need_sel);
            hipLaunchKernelGGL(l0_sel_kernel, dim3(kpad / 64), dim3(256), 0, d->cur, d->gboards, n, kpad, d->l0_sel);
        }
        if (!d->l0_split_done) {                     // (whole tiles: the delta product's epilogue wrote the planes)
            ProfScope ps(d, "l0_delta_split", 8.0 * n * H, (double)n * H * 4 + 6.0 * H * kpad);
            hipLaunchKernelGGL(delta_split_kernel, dim3(kpad / 64, H / 64), dim3(256), 0, d->cur, d->deltas[0], n, H, d->l0_planes, plane_stride, kpad);
        }
        d->l0_sel_done = d->l0_split_done = false;
        d->tail_open = was_open;
        XQ_HIP(hipGetLastError());
        const size_t shmem = l0m_lds_bytes();
        const double fl = 2.0 * 80 * 16 * (double)H * kpad * 3, by = 6.0 * H * kpad + 2.0 * need_sel + 4.0 * nchunks * len;
        if (d->tail_open) {
            TailArgs& T = *d->tail;
            T.l0_n = n; T.l0_H = H; T.l0_chunk = chunk; T.l0_nch = nchunks; T.l0_partial = out;
            T.l0_sel = d->l0_sel; T.l0_planes = d->l0_planes; T.l0_plane_stride = plane_stride; T.l0_kpad = kpad; T.l0_ncb = H / kL0mCols;
            T.n_l0 = 4 * (H / kL0mCols) * nchunks;
            d->tail_flops += fl; d->tail_bytes += by;
            d->tail_lds = std::max(d->tail_lds, shmem);
        } else {
            ProfScope ps(d, "l0_grad_segsum", fl, by);
            static bool granted = false;
            if (!granted) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(l0_grad_mfma_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
                granted = true;
            }
            hipLaunchKernelGGL(l0_grad_mfma_kernel<0>, dim3(H / kL0mCols, 4, nchunks), dim3(256), shmem, d->cur, d->l0_sel, d->l0_planes, plane_stride,
                               kpad, H, chunk, out);
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

static int sgd_apply(xq_dqn* d, SegTable t, double alpha) {
    static const bool vec_ok4 = [] { const char* e = getenv("XQ_SGD_SCALAR"); return !(e && e[0] == '1'); }();
    long long mx = 0;
    for (int i = 0; i < t.nseg; ++i) {
        const uintptr_t bits = (uintptr_t)t.dst[i] | (uintptr_t)t.src[i] | ((uintptr_t)t.dst_bf[i] << 1);   // the bf16 shadow: 8-byte pieces
        t.vec4[i] = vec_ok4 && (bits & 15) == 0 && (t.len[i] & 3) == 0 && (t.nslabs[i] <= 0 || (t.stride[i] & 3) == 0);
        mx = std::max(mx, t.vec4[i] ? t.len[i] / 4 : t.len[i]);
    }
    const unsigned bx = (unsigned)std::max<long long>(1, std::min<long long>((mx + 255) / 256, 1024));
    ProfScope ps(d, t.reduce_only ? "reduce_slabs" : "sgd_apply", 0, 0, true);
    hipExtLaunchKernelGGL(sgd_segments_kernel, dim3(bx, t.nseg), dim3(256), 0, d->cur, ps.start(), ps.stop(), 0, t, (float)alpha);
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
    XQ_HIP(hipEventCreateWithFlags(&d->ev_fork, stream_event_flags()));
    XQ_HIP(hipEventCreateWithFlags(&d->ev_join, stream_event_flags()));
    XQ_HIP(hipEventCreateWithFlags(&d->ev_delta, stream_event_flags()));
    XQ_HIP(hipEventCreateWithFlags(&d->ev_qmax, stream_event_flags()));
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
    retire_stream(d->stream);                                // (synchronised above: nothing queued on it is left to wait for — also when the stream is the caller's, which may destroy it next)
    for (int i = 0; i < 2; ++i) { hipFree(d->params[i]); hipFree(d->tacts[i]); }
    for (int l = 0; l < XQ_MAX_LAYERS; ++l) { hipFree(d->acts[l]); hipFree(d->deltas[l]); hipFree(d->sel_acts[l]); }
    hipFree(d->gboards); hipFree(d->dsc); hipFree(d->act_mb); hipFree(d->q90); hipFree(d->sel_q90); for (auto& K : d->sel_keep) { hipFree(K.z1); hipFree(K.prev_boards); } hipFree(d->partial); hipFree(d->zmax); hipFree(d->zidx); hipFree(d->qsa); hipFree(d->yv); hipFree(d->lossv);
    hipFree(d->grads_td); hipFree(d->grads_full); hipFree(d->slabs); hipFree(d->bias_work); hipFree(d->xdense); hipFree(d->qfull); hipFree(d->tfull);
    hipFree(d->hb); hipFree(d->ha); hipFree(d->hr); hipFree(d->hd);
    d->prof.collect();
    hipFree(d->slabs_l0); hipFree(d->l0_planes); hipFree(d->l0_sel);
    for (int i = 0; i < 2; ++i) { hipFree(d->params_bf[i]); hipFree(d->tacts_bf[i]); hipFree(d->t2acts_bf[i]); hipFree(d->t2acts[i]); }
    for (int l = 0; l < XQ_MAX_LAYERS; ++l) { hipFree(d->acts_bf[l]); hipFree(d->sel_acts_bf[l]); }
    hipFree(d->partial_idx);
    hipFree(d->qh_slabs[0]); hipFree(d->qh_slabs[1]);
    for (int l = 0; l < XQ_MAX_LAYERS; ++l) hipFree(d->deltas_bf[l]);
    if (d->side) { hipStreamSynchronize(d->side); retire_stream(d->side); hipStreamDestroy(d->side); }
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

int xq_dqn_set_refine_stage(xq_dqn* d, int mode) {
    if (!d || mode < -1 || mode > 1) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_dqn_set_refine_stage: -1 (by the screen's counters), 0 (never) or 1 (whenever the launch has room)");
    d->refine_stage = mode;
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
    g.libm_tanh = !XQ_L0_FAST_TANH;      // layer 0 takes the SAME tanh on both routes (tanh_l0 in the packed-board gather kernels): a dense
                                         // one-hot and the board it encodes then differ by summation order only
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
    const long long total = (long long)T.n_l0 + T.n_grad + T.n_delta + T.n_out + T.n_colsum + T.n_sel;
    if (total <= 0) return XQ_OK;
    ProfScope ps(d, name, d->tail_flops, d->tail_bytes, true);      // one launch: timed by its own start / stop events
    auto launch = [&](auto kern) {
        static size_t granted = 48 * 1024;            // per instantiation
        if (d->tail_lds > granted) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            granted = 64 * 1024;
        }
        hipExtLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), d->tail_lds, d->cur, ps.start(), ps.stop(), 0, T);
    };
    if (last && T.l0_planes != nullptr) launch(td_tail_kernel<TAIL_L0 | TAIL_GRAD | TAIL_OUT | TAIL_COLSUM, true>);
    else if (last) launch(td_tail_kernel<TAIL_L0 | TAIL_GRAD | TAIL_OUT | TAIL_COLSUM>);
    else if (T.n_sel) launch(td_tail_kernel<TAIL_GRAD | TAIL_DELTA | TAIL_OUT | TAIL_SEL>);
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
    static const bool grad_early = getenv("XQ_TAIL_GRAD_EARLY") != nullptr;     // A/B switch (tools/ab_env.sh)
    const bool defer_grad = nl == 3 && !grad_early;
    int waiting = -1;
    for (int l = nl - 3; l >= 0; --l) {
        tail_begin(d);
        XQ_TRY(hidden_deltas(d, n, d->deltas[l + 1], d->L[l + 2], d->L[l + 2], mode, l, l));
        if (defer_grad) { if (waiting >= 0) XQ_TRY(add_grad(waiting)); waiting = l + 1; }
        else XQ_TRY(add_grad(l + 1));
        if (l == nl - 3) add_out_grad();
        if (l == 0 && l0_mfma_shape(d, n)) {
            // the selector half-words of the minibatch's boards (operand of the matrix-pipe layer-0 gradient in the NEXT launch): they
            // depend on the boards alone, so their blocks ride at the end of this grid, off the dependency chain
            XQ_TRY(ensure_l0_mfma(d, n));
            TailArgs& T = *d->tail;
            const int chunk0 = l0_chunk_of(n), kpad = (n + chunk0 - 1) / chunk0 * chunk0;
            T.sel_boards = d->gboards; T.sel_n = n; T.sel_kpad = kpad; T.sel_out = d->l0_sel; T.n_sel = kpad / 64;
            d->tail_bytes += (double)n * 48 + 2.0 * l0sel_elems(kpad);
            d->tail_lds = std::max(d->tail_lds, (size_t)64 * 13 * sizeof(uint32_t));
            d->l0_sel_done = true;
        }
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
    d->l0_split_done = d->l0_sel_done = false;       // (set by this step's launches, consumed by l0_gradient)
    // Data-parallel step with more than one rank: the trainer's select chain (it waits for ev_qmax) starts behind the GRADIENTS
    // instead of behind the max pass, so that it runs beside the all-reduce — the exchange is then hidden behind work the step has to
    // do anyway, and the gradient kernels have the chip to themselves.  On one GPU there is nothing to hide behind and the early start
    // is the faster one.
    d->late_gate = d->comm != nullptr && tail_eligible(d, n) &&
                   (d->exchange_overlap == 1 ||
                    (d->exchange_overlap < 0 && (d->exch_calibrated ? d->exch_late : comm_world(d->comm) > 1)));
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
            if (ds > 0) {                                // whole groups per sample over the window, with hysteresis
                const double share = (double)(h[1] - d->scr_seen[2]) / ds;
                if (share > 0.25) d->scr_stage_whole = true; else if (share < 0.10) d->scr_stage_whole = false;
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
    const bool gate_early = (screened || bf) && !d->late_gate;
    // (same-box A/B, 3 x 3 x 300 steps: 0.1740 -> 0.1728 ms; XQ_FORK_STOP_EVENT=0 records a marker instead)
    static const bool stop_event_fork = [] { const char* e = getenv("XQ_FORK_STOP_EVENT"); return !(e && e[0] == '0'); }();
    d->fwd_stop_ev = (gate_early && stop_event_fork) ? d->ev_qmax : nullptr;
    const int chain_rc = chain_boards(d, jobs, dbl ? 3 : 2, slots, n, screened ? &shadow : nullptr);
    const bool fork_recorded = gate_early && stop_event_fork && d->fwd_stop_ev == nullptr;
    d->fwd_stop_ev = nullptr;
    XQ_TRY(chain_rc);
    // The select chain of the trainer starts HERE when max_a' Q(s',a') runs on the bf16 matrix pipe (screening pass of an fp32 net, or
    // the output layer of a bf16 net): its layer-0 gather (L2-bound) then runs beside the screening pass (matrix-pipe-bound) and
    // is gone when the refine kernel — a chain of dependent memory round trips that the gather doubles in length — starts.  Same-box
    // A/B, round 4 (3 x 300 steps per leg): behind the screening pass 0.1946-0.1969 ms, here 0.1888-0.1937; behind the layer-0 gather of
    // this step 0.1886-0.1896 against 0.1914-0.1921; at the very top of the step no difference; --config 4 / 5 -0.3 % / -0.9 %.
    // The full fp32 product keeps the chip to itself: there the chain starts behind it (below).
    if (gate_early && !fork_recorded) XQ_HIP(hipEventRecord(d->ev_qmax, d->stream));
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
            // (bracketed by its own start / stop events: the live figure is the kernel's duration as rocprofv3 reports it)
            ProfScope ps(d, "gemm_qmax_screen", 2.0 * NO * (double)n * Hl, 2.0 * ((double)NO * Hl + (double)n * Hl) + 8.0 * G * n, true);
            if (Hl == 256) hipExtLaunchKernelGGL((screen_top2_kernel<1, 2>), dim3(a.panels * a.ranges), dim3(512), lds, d->cur, ps.start(), ps.stop(), 0, a);
            else hipExtLaunchKernelGGL((screen_top2_kernel<2, 1>), dim3(a.panels * a.ranges), dim3(512), lds, d->cur, ps.start(), ps.stop(), 0, a);
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
            ProfScope ps(d, "qmax_refine", 2.0 * n * Hl * 3, 12.0 * G * n + 4.0 * n * Hl, true);      // one launch: its own start / stop events
            const dim3 grid((n + kRefineSamples - 1) / kRefineSamples);
            // one block per CU and K = 256: room in LDS for the staged pass of qmax_refine2_kernel (whole groups that many samples ask for)
            const bool stage = scr_new && Hl == 256 && (int)grid.x <= d->ncu && (d->refine_stage > 0 || (d->refine_stage < 0 && d->scr_stage_whole));
            const size_t lds = refine_cand_words((int)G) * sizeof(uint32_t) + refine_wlist_bytes((int)G) + (stage ? refine_stage_bytes() : 0);
            auto launch = [&](auto kern) {
                hipExtLaunchKernelGGL(kern, grid, dim3(256), lds, d->cur, ps.start(), ps.stop(), 0, d->scr_p1, d->scr_p2, G, n, ldp, touts[nl - 2], Hl,
                                      d->wl(sel_net, nl - 1), d->bl(sel_net, nl - 1), NO, d->scr_wmax, parity, d->zmax, d->scr_stats);
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
                // whole groups of qmax_refine2_kernel: 2 = the popular ones through LDS when the launch has room for it, 1 = all of them four per
                // round trip from global memory (XQ_REFINE_WHOLE=1: A/B knob, same bits)
                static const bool whole_staged = [] { const char* e = getenv("XQ_REFINE_WHOLE"); return !(e && e[0] == '1'); }();
                const int whole_mode = (whole_staged && stage) ? 2 : 1;
                auto launch2 = [&](auto kern) {
                    static bool granted = false;          // per instantiation: the staged pass asks for ~145 KB of dynamic LDS
                    if (!granted) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); granted = true; }
                    hipExtLaunchKernelGGL(kern, grid, dim3(256), lds, d->cur, ps.start(), ps.stop(), 0, d->scr_R, scr_ranges, scr_gpr, d->scr_p1, d->scr_p2, G, n,
                                          ldp, d->scr_na, touts[nl - 2], Hl, d->wl(sel_net, nl - 1), d->bl(sel_net, nl - 1), NO, d->scr_wmax, parity, d->zmax,
                                          d->scr_stats, T, whole_mode);
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
        hipEvent_t ev0 = nullptr, ev1 = nullptr;     // (the launch's own start / stop events when it is being timed)
        auto launch = [&](auto kern) {
            static bool ready = false;       // per instantiation
            if (!ready) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); ready = true; }
            hipExtLaunchKernelGGL(kern, dim3(a.panels * a.ranges), dim3(512), lds, d->cur, ev0, ev1, 0, a);
        };
        ProfScope ps(d, "gemm_qmax_rowmax", 2.0 * NO * (double)n * Hl, 2.0 * ((double)NO * Hl + (double)n * Hl) + 8.0 * n_part * n, true);
        ev0 = ps.start(); ev1 = ps.stop();
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

// The late start of the select chain (beside the all-reduce instead of beside the gradient kernels) costs a step ~41 us when the
// collective is free and wins by the part of the collective above that (0.2430 against 0.2021 ms with a one-rank
// communicator, where the all-reduce launches nothing).
static const double kExchangeThresholdUs = 41.0;

int xq_dqn_calibrate_exchange(xq_dqn* d, double threshold_us, double* allreduce_us, int* late) {
    if (!d) return fail(XQ_ERR_INVALID_ARGUMENT, "null dqn");
    if (!d->comm) return fail(XQ_ERR_INVALID_ARGUMENT, "xq_dqn_calibrate_exchange: no communicator attached");
    if (d->n_grads_td == 0) layout_td_grads(d);
    // COLLECTIVE: every rank times the same 4 + 20 all-reduces of a scratch buffer of the gradient buffer's size on the handle's stream
    float* scratch = nullptr;
    XQ_HIP(hipMalloc(&scratch, d->n_grads_td * sizeof(float)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = XQ_OK;
    float ms = 0.f;
    const int reps = 20;
    do {
        if (hipMemsetAsync(scratch, 0, d->n_grads_td * sizeof(float), d->stream) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
            hipEventCreate(&e1) != hipSuccess) { rc = fail(XQ_ERR_RUNTIME, "xq_dqn_calibrate_exchange: HIP setup failed"); break; }
        for (int i = 0; i < 4 && rc == XQ_OK; ++i) rc = comm_allreduce_on(d->comm, scratch, d->n_grads_td, d->stream);
        if (rc != XQ_OK) break;
        if (hipEventRecord(e0, d->stream) != hipSuccess) { rc = fail(XQ_ERR_RUNTIME, "hipEventRecord failed"); break; }
        for (int i = 0; i < reps && rc == XQ_OK; ++i) rc = comm_allreduce_on(d->comm, scratch, d->n_grads_td, d->stream);
        if (rc != XQ_OK) break;
        if (hipEventRecord(e1, d->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
            rc = fail(XQ_ERR_RUNTIME, "xq_dqn_calibrate_exchange: timing failed");
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipStreamSynchronize(d->stream);
    (void)hipFree(scratch);
    if (rc != XQ_OK) return rc;
    // the ranks must agree (a rank that starts its select chain late while its neighbours start early only wastes what the rule
    // saves): the decision is taken on the MEAN over the ranks, in integer nanoseconds so that every rank computes the same number
    uint64_t ns = (uint64_t)((double)ms * 1e6 / reps + 0.5);
    if (ns == 0) ns = 1;                             // (a one-rank all-reduce launches nothing: two back-to-back event records)
    XQ_TRY(xq_comm_sum_u64(d->comm, &ns));
    d->exch_allreduce_us = (double)ns / 1e3 / (double)comm_world(d->comm);
    d->exch_threshold_us = threshold_us >= 0.0 ? threshold_us : kExchangeThresholdUs;
    d->exch_late = d->exch_allreduce_us > d->exch_threshold_us;
    d->exch_calibrated = true;
    if (allreduce_us) *allreduce_us = d->exch_allreduce_us;
    if (late) *late = d->exch_late ? 1 : 0;
    return XQ_OK;
}

int xq_dqn_exchange_calibration(const xq_dqn* d, int* calibrated, double* allreduce_us, double* threshold_us, int* late) {
    if (!d) return fail(XQ_ERR_INVALID_ARGUMENT, "null dqn");
    if (calibrated) *calibrated = d->exch_calibrated ? 1 : 0;
    if (allreduce_us) *allreduce_us = d->exch_allreduce_us;
    if (threshold_us) *threshold_us = d->exch_threshold_us;
    if (late) *late = d->exch_late ? 1 : 0;
    return XQ_OK;
}

int xq_dqn_set_comm(xq_dqn* d, xq_comm* comm) {
    if (!d) return fail(XQ_ERR_INVALID_ARGUMENT, "null dqn");
    if (d->l0_pending > 0 || d->pend_wout.nslabs > 0 || d->pend_bh.nslabs > 0)
        return fail(XQ_ERR_RUNTIME, "xq_dqn_set_comm: a TD step is waiting for its apply_grads");
    d->comm = comm;
    d->exch_calibrated = false;
    // more than one rank: measure the exchange now (collective: every rank attaches its communicator at the same point of the program)
    // and let the measurement, not the rank count, say where the select chain starts (the static guess cost 20 % at
    // the one point it had been measured)
    if (comm && comm_world(comm) > 1 && d->exchange_overlap < 0) XQ_TRY(xq_dqn_calibrate_exchange(d, -1.0, nullptr, nullptr));
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
            stats[n].ms = c.ms; stats[n].launches = c.launches; stats[n].flops = c.flops; stats[n].bytes = c.bytes; stats[n].exact_launches = c.exact;
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
