/* TEST INFRASTRUCTURE — CPU oracle, never shipped, never on the product path.
 *
 * BUILD-DEFINED extensions of the training path for BASELINE configs[4] ("Double-DQN + prioritized replay, bf16 MFMA
 * Q-net").  The reference has NONE of this (`grep -i "double\|priorit\|bf16" /root/reference/src` is empty): there is
 * nothing upstream to restate, so this file IS the definition the HIP path is tested against — "parity unpinned" by
 * construction.  It generalises the one TD step the reference does have:
 *   ChessAI::train   chessai.cpp:122-131   y = done ? r : r + gamma * max_k Q_online(s')[k];  backprop of 0.5|Q(s)-target|^2
 *   DQN::train       dqn.cpp:157-172       the same with the target network
 * and keeps the reference's conventions: tanh on every layer incl. the output (dqn.cu:184-195), the maximum over ALL
 * outputs, first strict maximum wins (dqn.cpp:48), bug-compatible or textbook hidden delta (mode).
 *
 *   Double DQN (van Hasselt, Guez, Silver, AAAI 2016):  a* = argmax_k z_online(s')[k]  (first maximum, pre-activation z:
 *       tanh is monotone),  y = done ? r : r + gamma * tanh(z_target(s')[a*]).
 *   Prioritized replay, proportional variant (Schaul, Quan, Antonoglou, Silver, ICLR 2016):
 *       p_i = (|delta_i| + eps)^alpha,  P(i) = p_i / sum p,  stratified draw (segment k of B gets one sample),
 *       importance weight w_i = (N * P(i))^-beta / max_batch w,  gradient of sample i scaled by w_i,
 *       new transitions enter with the largest priority assigned so far.  The sums live in a radix-32 tree whose every
 *       node is the sequential fp32 sum of its 32 children — fixed association, so the device reproduces it bit for bit.
 *   bf16 Q-net: forward in bf16 — weights rounded to bf16 (RNE, from the fp32 master copy), hidden activations rounded to
 *       bf16 after tanh, products accumulated in >= fp32, biases and the output pre-activation z kept in fp32; backward in
 *       fp32 on the master weights with the rounded activations (1 - a^2 from the stored a).
 *   bf16 = 2 (XQ_PRECISION_BF16_FULL): the same forward, and bf16 operands in the dense products of the backward pass too
 *       (mixed precision as in Micikevicius et al., ICLR 2018, without loss scaling: the deltas of this loss are O(1e-3..1)):
 *       the hidden deltas BELOW the top one multiply the bf16-rounded weights with the upstream delta rounded to bf16, the hidden
 *       weight gradients (layers 1 .. nl-2) multiply the layer's delta rounded to bf16 with the (already bf16) activations;
 *       products are exact, sums >= fp32.  Unrounded: the top hidden delta (one scaled row of the master weights), every
 *       bias gradient, the layer-0 and the output-layer gradients, the master weights and the SGD step.
 */
#include "xq_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

float xqo_bf16_round(float x) {                 /* round-to-nearest-even to 8 significant bits (v_cvt_pk_bf16_f32) */
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return x;          /* inf / nan unchanged */
    u += 0x7FFFu + ((u >> 16) & 1u);
    u &= 0xFFFF0000u;
    memcpy(&x, &u, 4);
    return x;
}

static void ext_offsets(const int* L, int nl, size_t* wo, size_t* bo) {
    size_t tw = 0, tb = 0;
    for (int l = 0; l < nl; ++l) { wo[l] = tw; bo[l] = tb; tw += (size_t)L[l] * L[l + 1]; tb += (size_t)L[l + 1]; }
}

/* forward, bias first (the training kernel's order, dqn.cu:275-286).  bf16 = 1: weights through fp32 then bf16, hidden
 * activations through fp32 then bf16.  acts[l] (l = 0..nl-2) = hidden activations, z_out = output pre-activations. */
static void ext_forward(const int* L, int nl, const double* w, const double* b, const double* in, int bf16,
                        double** acts, double* z_out) {
    size_t wo[XQO_MAX_LAYERS], bo[XQO_MAX_LAYERS];
    ext_offsets(L, nl, wo, bo);
    const double* cur = in;
    for (int l = 0; l < nl; ++l) {
        const int I = L[l], O = L[l + 1];
        const double* W = w + wo[l];
        double* out = l == nl - 1 ? z_out : acts[l];
        for (int j = 0; j < O; ++j) {
            double sum = bf16 ? (double)(float)b[bo[l] + j] : b[bo[l] + j];
            for (int i = 0; i < I; ++i) {
                if (cur[i] == 0.0) continue;
                const double wij = bf16 ? (double)xqo_bf16_round((float)W[(size_t)j * I + i]) : W[(size_t)j * I + i];
                sum += cur[i] * wij;
            }
            if (l == nl - 1) out[j] = sum;
            else out[j] = bf16 ? (double)xqo_bf16_round((float)tanh(sum)) : tanh(sum);
        }
        cur = out;
    }
}

int xqo_ext_forward(const int* L, int ns, const double* w, const double* b, const double* in, int bf16,
                    double* hidden_acts, double* z_out) {
    const int nl = ns - 1;
    if (nl < 2 || nl > XQO_MAX_LAYERS) return -1;
    double* acts[XQO_MAX_LAYERS];
    size_t tot = 0;
    for (int l = 0; l + 1 < nl; ++l) tot += (size_t)L[l + 1];
    double* pool = hidden_acts ? hidden_acts : (double*)malloc(sizeof(double) * tot);
    double* p = pool;
    for (int l = 0; l + 1 < nl; ++l) { acts[l] = p; p += L[l + 1]; }
    double* z = z_out ? z_out : (double*)malloc(sizeof(double) * (size_t)L[nl]);
    ext_forward(L, nl, w, b, in, bf16, acts, z);
    if (!z_out) free(z);
    if (!hidden_acts) free(pool);
    return 0;
}

/* One transition of the generalised TD step: accumulates  gw += weight * dL/dw,  gb += weight * dL/db  at the given
 * (pre-update) parameters and reports Q(s,a), y and the TD error Q(s,a) - y.
 *   td_rule 0: y from max_k tanh(z_online(s'))           (chessai.cpp:126-127)
 *           1: y from max_k tanh(z_target(s'))           (dqn.cpp:166-167)
 *           2: Double DQN — argmax on the online net, value from the target net
 *   mode 0 / 1: bug-compatible / textbook hidden delta (see xqo_nn_backprop); the output delta of a TD step has the single
 *               non-zero entry (Q(s,a) - y)(1 - Q(s,a)^2) at action_to (target == Q(s) elsewhere).
 * Returns -1 where mode 0 is undefined upstream for the topology. */
int xqo_ext_td_accum(const int* L, int ns, const double* w, const double* b, const double* wt, const double* bt,
                     const double* state, const double* next_state, int action_to, double reward, int done, double gamma,
                     int td_rule, int mode, int bf16, double weight, double* gw, double* gb,
                     double* q_sa, double* y_out, int* a_star) {
    const int nl = ns - 1;
    if (nl < 2 || nl > XQO_MAX_LAYERS || action_to < 0 || action_to >= L[nl]) return -1;
    size_t wo[XQO_MAX_LAYERS], bo[XQO_MAX_LAYERS];
    ext_offsets(L, nl, wo, bo);
    size_t nw = 0, hid = 0;
    for (int l = 0; l < nl; ++l) nw += (size_t)L[l] * L[l + 1];
    for (int l = 0; l + 1 < nl; ++l) hid += (size_t)L[l + 1];
    const int NO = L[nl];
    double* pool = (double*)calloc(2 * hid + 3 * (size_t)NO, sizeof(double));
    double* a[XQO_MAX_LAYERS + 1];
    double* d[XQO_MAX_LAYERS];
    double* p = pool;
    for (int l = 0; l + 1 < nl; ++l) { a[l + 1] = p; p += L[l + 1]; }
    for (int l = 0; l + 1 < nl; ++l) { d[l] = p; p += L[l + 1]; }
    double* z = p; p += NO;
    double* zn = p; p += NO;
    double* zt = p;
    a[0] = (double*)state;
    ext_forward(L, nl, w, b, state, bf16, a + 1, z);
    const double q = tanh(z[action_to]);
    double y = reward;
    int astar = -1;
    if (!done) {
        if (td_rule == 0 || td_rule == 2) xqo_ext_forward(L, ns, w, b, next_state, bf16, NULL, zn);
        if (td_rule == 1 || td_rule == 2) xqo_ext_forward(L, ns, wt, bt, next_state, bf16, NULL, zt);
        const double* sel = td_rule == 1 ? zt : zn;
        astar = 0;
        for (int k = 1; k < NO; ++k) if (sel[k] > sel[astar]) astar = k;     /* first strict maximum (dqn.cpp:48) */
        const double* val = td_rule == 0 ? zn : zt;
        y = reward + gamma * tanh(val[astar]);
    }
    const double delta = (q - y) * (1.0 - q * q) * weight;           /* the one non-zero output delta, x importance weight */
    int rc = 0;
    /* hidden deltas from the single non-zero output delta; master weights */
    for (int l = nl - 2; l >= 0 && rc == 0; --l) {
        const int up_n = (l == nl - 2) ? NO : L[l + 2];
        for (int idx = 0; idx < L[l + 1]; ++idx) {
            double sum = 0.0;
            if (mode == 0) {
                const int inputSize = L[l + 1], outputSize = L[l];
                if (L[l + 2] < inputSize || outputSize < L[l + 1] ||
                    wo[l + 1] + (size_t)(inputSize - 1) * outputSize + (size_t)(L[l + 1] - 1) >= nw) { rc = -1; break; }
                if (l == nl - 2) { if (action_to < inputSize) sum = w[wo[l + 1] + (size_t)action_to * outputSize + idx] * delta; }
                else for (int i = 0; i < inputSize; ++i) {
                    const double wv = w[wo[l + 1] + (size_t)i * outputSize + idx], dv = d[l + 1][i];
                    sum += bf16 == 2 ? (double)xqo_bf16_round((float)wv) * (double)xqo_bf16_round((float)dv) : wv * dv;
                }
            } else {
                if (l == nl - 2) sum = w[wo[l + 1] + (size_t)action_to * L[l + 1] + idx] * delta;
                else for (int k = 0; k < up_n; ++k) {
                    const double wv = w[wo[l + 1] + (size_t)k * L[l + 1] + idx], dv = d[l + 1][k];
                    sum += bf16 == 2 ? (double)xqo_bf16_round((float)wv) * (double)xqo_bf16_round((float)dv) : wv * dv;
                }
            }
            d[l][idx] = sum * (1.0 - a[l + 1][idx] * a[l + 1][idx]);
        }
    }
    if (rc == 0) {
        const int lo = nl - 1;
        gb[bo[lo] + action_to] += delta;
        for (int i = 0; i < L[lo]; ++i) gw[wo[lo] + (size_t)action_to * L[lo] + i] += delta * a[lo][i];
        for (int l = 0; l + 1 < nl; ++l)
            for (int j = 0; j < L[l + 1]; ++j) {
                const double dj = d[l][j];
                gb[bo[l] + j] += dj;
                const double dw = (bf16 == 2 && l >= 1) ? (double)xqo_bf16_round((float)dj) : dj;     /* layer 0 sums fp32 delta rows */
                if (dw != 0.0)
                    for (int i = 0; i < L[l]; ++i)
                        if (a[l][i] != 0.0) gw[wo[l] + (size_t)j * L[l] + i] += dw * a[l][i];
            }
    }
    if (q_sa) *q_sa = q;
    if (y_out) *y_out = y;
    if (a_star) *a_star = astar;
    free(pool);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Prioritized replay: radix-32 sum tree in fp32.  level 0 = leaves (capacity, zero-padded to a multiple of 32);
 * level j+1 node i = ((c[32i] + c[32i+1]) + c[32i+2]) + ... (sequential, index order); the last level has one node.
 * ---------------------------------------------------------------------------------------------- */
int xqo_per_levels(int capacity, int* n /* [8] nodes per level */, int* padded /* [8] allocated per level */) {
    int cnt = capacity, lv = 0;
    for (;;) {
        if (lv >= 8) return -1;
        n[lv] = cnt;
        padded[lv] = (cnt + 31) / 32 * 32;
        ++lv;
        if (cnt <= 1) break;
        cnt = padded[lv - 1] / 32;
    }
    return lv;
}

size_t xqo_per_tree_floats(int capacity) {
    int n[8], p[8];
    const int nlv = xqo_per_levels(capacity, n, p);
    size_t t = 0;
    for (int lv = 0; lv < nlv; ++lv) t += (size_t)p[lv];
    return t;
}

/* tree = concatenation of the padded levels (level 0 first); leaves copied from prio[capacity] */
void xqo_per_build(const float* prio, int capacity, float* tree) {
    int n[8], p[8];
    const int nlv = xqo_per_levels(capacity, n, p);
    memset(tree, 0, sizeof(float) * xqo_per_tree_floats(capacity));
    memcpy(tree, prio, sizeof(float) * (size_t)capacity);
    float* cur = tree;
    for (int lv = 1; lv < nlv; ++lv) {
        float* nxt = cur + p[lv - 1];
        for (int i = 0; i < n[lv]; ++i) {
            float s = 0.f;
            for (int c = 0; c < 32; ++c) s += cur[(size_t)i * 32 + c];      /* sequential: ((c0 + c1) + c2) + ... */
            nxt[i] = s;
        }
        cur = nxt;
    }
}

float xqo_per_total(const float* tree, int capacity) {
    int n[8], p[8];
    const int nlv = xqo_per_levels(capacity, n, p);
    size_t root = 0;
    for (int lv = 0; lv + 1 < nlv; ++lv) root += (size_t)p[lv];
    return tree[root];
}

/* descent for one mass u in [0, total): at every node walk the 32 children in index order; child c is taken when
 * u < v[c] (after subtracting the children before it); falling off the end through rounding takes the last child with
 * v > 0 and continues below it with u = 0.  Returns the leaf index. */
int xqo_per_descend(const float* tree, int capacity, float u) {
    int n[8], p[8];
    const int nlv = xqo_per_levels(capacity, n, p);
    size_t off[8];
    off[0] = 0;
    for (int lv = 1; lv < nlv; ++lv) off[lv] = off[lv - 1] + (size_t)p[lv - 1];
    int node = 0;
    for (int lv = nlv - 2; lv >= 0; --lv) {
        const float* v = tree + off[lv] + (size_t)node * 32;
        int c = 0, last = -1;
        for (; c < 32; ++c) {
            if (v[c] > 0.f) last = c;
            if (u < v[c]) break;
            u -= v[c];
        }
        if (c == 32) { c = last < 0 ? 0 : last; u = 0.f; }
        node = node * 32 + c;
    }
    return node;
}

/* stratified minibatch: u_k = (k + r_k) * (total / B), r_k = (philox(k, 0, call, 2; seed).v[0] >> 8) * 2^-24;
 * slots[k] = leaf, w_raw[k] = (n_eligible * p_leaf / total)^-beta (fp32); returns max_k w_raw (the normaliser) */
float xqo_per_sample(const float* tree, int capacity, int batch, uint64_t seed, uint32_t call, int n_eligible, float beta,
                     int32_t* slots, float* w_raw) {
    const float total = xqo_per_total(tree, capacity);
    const float seg = total / (float)batch;
    float wmax = 0.f;
    for (int k = 0; k < batch; ++k) {
        const uint32_t ctr[4] = {(uint32_t)k, 0u, call, 2u};
        const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        uint32_t o[4];
        xqo_philox4x32(ctr, key, o);
        const float r = (float)(o[0] >> 8) * (1.0f / 16777216.0f);
        const float u = ((float)k + r) * seg;
        const int leaf = xqo_per_descend(tree, capacity, u);
        const float prob = tree[leaf] / total;
        const float wr = powf((float)n_eligible * prob, -beta);
        slots[k] = leaf;
        w_raw[k] = wr;
        if (wr > wmax) wmax = wr;
    }
    return wmax;
}

float xqo_per_priority(float td_error, float eps, float alpha) { return powf(fabsf(td_error) + eps, alpha); }
