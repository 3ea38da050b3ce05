// TEST INFRASTRUCTURE — not product code.
//
// Driver around the REAL reference neural-network runtime, executed on the MI355X.
//
// /root/reference/src/dqn.cu and /root/reference/include/dqn.h are CUDA sources; this image has no nvcc and no
// CUDA runtime, but it ships AMD's own source translator (/opt/rocm/bin/hipify-perl).  oracle/Makefile, target
// `refnn`, pipes the two files — read where they lie — through hipify-perl into oracle/_ref/src/ (git-ignored AND
// gpurun-ignored AND deleted when the recipe ends: the text never enters history, never travels, does not stay), checks that every changed line differs only by
// cuda* -> hip* API identifiers / the runtime header name, and compiles the result with hipcc for gfx950 against
// the image's QtCore 5.9.7.  The five kernels (dqn.cu:184-195, 275-319), the launch geometry, the allocation
// pattern and the host control flow of NeuralNetwork::forward / ::backpropagate (dqn.cu:199-260, 323-467) are
// byte-for-byte the reference's; only the runtime underneath is HIP instead of CUDA.  No header, library or tool
// is written as a stand-in.  The binary lands in oracle/_ref/xqref_nn and is run ON THE GPU BOX by
// oracle/gen_golden_nn.py, whose outputs are committed as tests/golden/ref_nn.npz.
//
// Only the public NeuralNetwork API is used (dqn.h:42-77): constructor, host_weights / host_biases /
// weightOffsets / biasOffsets (public members), copyToDevice, copyFromDevice, forward, backpropagate.
//
// What an execution can pin and what it cannot.  NeuralNetwork::backpropagate reads device memory it has already
// released (activations[l], l >= 1: freed at dqn.cu:371, read by updateWeightsBiasesKernel at :441) and indexes
// zs[l] past its end in hiddenLayerDeltaKernel (:420, results unused).  Outputs that depend on those reads —
// the updated WEIGHTS of layers >= 1 — are whatever the allocator left there; they are recorded under `ub_*`
// names and never asserted as golden.  Everything else (Q-values, every bias, the whole of layer 0's weights —
// which carry the as-written hidden delta of :406-423) is free of undefined behaviour and is the fixture.
// `probe` checks, without any undefined access, that the runtime sub-allocates small blocks out of one mapped
// 2-MB block (so the stale reads above stay inside mapped memory); gen_golden_nn.py refuses to run the rest
// if it does not.
#include "dqn.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// counter-based U[0,1): reproduced by tests/refnn.py with numpy uint64 arithmetic
inline double u01(uint64_t seed, uint64_t stream, uint64_t i) {
    return (double)(mix64(mix64(seed * 0x100000001B3ull + stream) + i) >> 11) * (1.0 / 9007199254740992.0);
}

FILE* g_out = nullptr;
void put(const char* name, int dtype /*0 f64, 1 i32, 2 i64*/, const void* data, uint64_t count) {
    uint32_t nl = (uint32_t)std::strlen(name), dt = (uint32_t)dtype;
    std::fwrite(&nl, 4, 1, g_out); std::fwrite(name, 1, nl, g_out); std::fwrite(&dt, 4, 1, g_out);
    std::fwrite(&count, 8, 1, g_out);
    std::fwrite(data, dtype == 1 ? 4 : 8, count, g_out);
}
void put_f64(const std::string& n, const std::vector<double>& v) { put(n.c_str(), 0, v.data(), v.size()); }
void put_i32(const std::string& n, const std::vector<int32_t>& v) { put(n.c_str(), 1, v.data(), v.size()); }
void put_i64(const std::string& n, const std::vector<int64_t>& v) { put(n.c_str(), 2, v.data(), v.size()); }

void fill_params(NeuralNetwork& nn, uint64_t seed) {
    for (size_t i = 0; i < nn.host_weights.size(); ++i) nn.host_weights[i] = (u01(seed, 1, i) - 0.5) * 0.1;
    for (size_t i = 0; i < nn.host_biases.size(); ++i) nn.host_biases[i] = (u01(seed, 2, i) - 0.5) * 0.02;
    nn.copyToDevice();
}

// a board-like one-hot input: `npieces` distinct squares, one of 14 planes each (chessai.cpp:268-289's index form
// sq*14 + plane); for inputs that are not 1260 wide: npieces distinct positions
std::vector<int32_t> onehot_indices(uint64_t seed, uint64_t k, int width, int npieces) {
    std::vector<int32_t> idx;
    if (width == 1260) {
        std::vector<char> used(90, 0);
        for (uint64_t t = 0; (int)idx.size() < npieces; ++t) {
            int sq = (int)(u01(seed, 100 + k, 2 * t) * 90), pl = (int)(u01(seed, 100 + k, 2 * t + 1) * 14);
            if (used[sq]) continue;
            used[sq] = 1; idx.push_back(sq * 14 + pl);
        }
    } else {
        std::vector<char> used(width, 0);
        if (npieces > width) npieces = width;
        for (uint64_t t = 0; (int)idx.size() < npieces; ++t) {
            int p = (int)(u01(seed, 100 + k, t) * width);
            if (used[p]) continue;
            used[p] = 1; idx.push_back(p);
        }
    }
    return idx;
}

int cmd_probe() {
    // no undefined access anywhere in here: addresses only
    char *x = nullptr, *a = nullptr, *b = nullptr, *c = nullptr, *d = nullptr;
    if (hipMalloc(&x, 69 * 1024) != hipSuccess) return 2;          // plays d_biases (lives as long as the net)
    if (hipMalloc(&a, 10080) != hipSuccess) return 2;
    if (hipMalloc(&b, 2048) != hipSuccess) return 2;
    if (hipMalloc(&c, 2048) != hipSuccess) return 2;
    const uintptr_t blk = (uintptr_t)x & ~(uintptr_t)((2u << 20) - 1);
    auto inblk = [&](const void* p) { return ((uintptr_t)p & ~(uintptr_t)((2u << 20) - 1)) == blk; };
    // carved out of one block, back to back (a fragment allocator), not three mappings of their own that happen to be neighbours
    const bool tight = (uintptr_t)a > (uintptr_t)x && (uintptr_t)a - (uintptr_t)x <= 80 * 1024 && (uintptr_t)b > (uintptr_t)a &&
                       (uintptr_t)b - (uintptr_t)a <= 16 * 1024 && (uintptr_t)c > (uintptr_t)b && (uintptr_t)c - (uintptr_t)b <= 8 * 1024;
    const bool same = inblk(a) && inblk(b) && inblk(c) && tight;
    (void)hipFree(b);
    if (hipMalloc(&d, 64800) != hipSuccess) return 2;
    std::printf("{\"x\": \"%p\", \"a\": \"%p\", \"b\": \"%p\", \"c\": \"%p\", \"d_after_free_b\": \"%p\", "
                "\"small_blocks_share_one_2mb_block\": %s}\n", (void*)x, (void*)a, (void*)b, (void*)c, (void*)d, same ? "true" : "false");
    (void)hipFree(a); (void)hipFree(c); (void)hipFree(d); (void)hipFree(x);
    return same ? 0 : 3;
}

// nn <out.bin> <seed> <L0> <L1> ... : forward + single-step backpropagate records for one topology
int cmd_nn(int argc, char** argv) {
    if (argc < 6) return 64;
    g_out = std::fopen(argv[2], "wb");
    if (!g_out) return 65;
    const uint64_t seed = std::strtoull(argv[3], nullptr, 10);
    std::vector<int> sizes;
    for (int i = 4; i < argc; ++i) sizes.push_back(std::atoi(argv[i]));
    const int nL = (int)sizes.size() - 1, IN = sizes.front(), OUT = sizes.back();

    NeuralNetwork nn(sizes);                                       // dqn.cu:14-57: random init, device upload

    // --- N8: what the constructor left (dqn.cu:96-146)
    {
        double wmin = 1e9, wmax = -1e9, wsum = 0, babs = 0;
        for (double v : nn.host_weights) { wmin = v < wmin ? v : wmin; wmax = v > wmax ? v : wmax; wsum += v; }
        for (double v : nn.host_biases) babs = (v < 0 ? -v : v) > babs ? (v < 0 ? -v : v) : babs;
        put_f64("init_w_min_max_mean", {wmin, wmax, wsum / (double)nn.host_weights.size()});
        put_f64("init_b_absmax", {babs});
        put_i64("counts", {(int64_t)nn.host_weights.size(), (int64_t)nn.host_biases.size()});
        std::vector<int64_t> wo(nn.weightOffsets.begin(), nn.weightOffsets.end()), bo(nn.biasOffsets.begin(), nn.biasOffsets.end());
        put_i64("weight_offsets", wo); put_i64("bias_offsets", bo);
        put_i32("sizes", std::vector<int32_t>(sizes.begin(), sizes.end()));
        // the freshly constructed net's device copy equals its host copy: forward of the zero vector = tanh chain of zero biases = 0
        std::vector<double> q0 = nn.forward(std::vector<double>(IN, 0.0));
        double q0abs = 0; for (double v : q0) q0abs = (v < 0 ? -v : v) > q0abs ? (v < 0 ? -v : v) : q0abs;
        put_f64("init_forward_of_zero_absmax", {q0abs});
    }

    fill_params(nn, seed);
    const std::vector<double> W = nn.host_weights, B = nn.host_biases;

    // sampled positions inside each layer's weights / the output vector
    const int NQ = OUT < 256 ? OUT : 256;
    std::vector<int32_t> qpos(NQ);
    for (int i = 0; i < NQ; ++i) qpos[i] = (i < 96 && i < OUT) ? i : (int)(u01(seed, 7, i) * OUT);
    put_i32("q_positions", qpos);

    // --- N3 / N4: forward (dqn.cu:199-260, kernel :184-195)
    const int NS = 6;
    for (int k = 0; k < NS; ++k) {
        std::vector<double> x(IN, 0.0);
        std::string tag = "fwd" + std::to_string(k);
        if (k < 4) {
            const int np = k == 0 ? 32 : (k == 1 ? 2 : 5 + (int)(u01(seed, 50, k) * 27));
            auto idx = onehot_indices(seed, k, IN, np);
            for (int i : idx) x[i] = 1.0;
            put_i32(tag + "_onehot", idx);
        } else {
            for (int i = 0; i < IN; ++i) x[i] = u01(seed, 60 + k, i) * 2.0 - 1.0;
            put_f64(tag + "_dense", x);
        }
        std::vector<double> q = nn.forward(x);
        std::vector<double> qs(NQ);
        for (int i = 0; i < NQ; ++i) qs[i] = q[qpos[i]];
        int am = 0; for (int i = 1; i < OUT; ++i) if (q[i] > q[am]) am = i;
        double sum = 0; for (double v : q) sum += v;
        put_f64(tag + "_q", qs);
        put_f64(tag + "_max_sum", {q[am], sum});
        put_i32(tag + "_argmax", {am});
    }

    // --- N5: one backpropagate() from the same parameters each time (dqn.cu:323-467, kernels :275-319)
    const int NU = 4;
    for (int u = 0; u < NU; ++u) {
        nn.host_weights = W; nn.host_biases = B; nn.copyToDevice();
        std::string tag = "bp" + std::to_string(u);
        std::vector<double> x(IN, 0.0);
        std::vector<int32_t> idx;
        if (u < 3) {
            idx = onehot_indices(seed, 20 + u, IN, u == 0 ? 32 : 3 + (int)(u01(seed, 51, u) * 29));
            for (int i : idx) x[i] = 1.0;
            put_i32(tag + "_onehot", idx);
        } else {
            for (int i = 0; i < IN; ++i) x[i] = u01(seed, 70, i) * 2.0 - 1.0;
            put_f64(tag + "_dense", x);
        }
        // the reference's own target construction (chessai.cpp:121-133): the net's Q with one entry replaced
        std::vector<double> target = nn.forward(x);
        // the action: a destination square (< 90); below the last hidden width where there is a hidden layer, because the
        // hidden delta as written (:406-423) sums over the first L[l+1] output deltas only — any other action would send nothing back
        int amax = OUT < 90 ? OUT : 90;
        if (nL >= 2 && sizes[nL - 1] < amax) amax = sizes[nL - 1];
        const int a = (int)(u01(seed, 52, u) * amax);
        const double y = (u & 1) ? (u01(seed, 53, u) * 2.0 - 1.0) : -19.0 + 4.0 * u;     // reference-scale reward / a value inside tanh's range
        const double lr = u == 2 ? 0.05 : 0.001;
        target[a] = y;
        put_i32(tag + "_action", {a});
        put_f64(tag + "_y_lr", {y, lr});
        nn.backpropagate(x, target, lr);
        nn.copyFromDevice();
        // every hidden bias and the select window of the output biases (free of undefined reads); the other output biases
        // move by lr * (a - target) * (1 - a^2) with a - target = the ulp difference of the two forward kernels: max reported
        {
            const size_t nhid = nn.biasOffsets[nL - 1];
            std::vector<double> hb(nn.host_biases.begin(), nn.host_biases.begin() + nhid);
            const int no = OUT < 96 ? OUT : 96;
            std::vector<double> ob(nn.host_biases.begin() + nhid, nn.host_biases.begin() + nhid + no);
            double rest = 0;
            for (int k = no; k < OUT; ++k) { double dv = nn.host_biases[nhid + k] - B[nhid + k]; rest = (dv < 0 ? -dv : dv) > rest ? (dv < 0 ? -dv : dv) : rest; }
            put_f64(tag + "_hidden_biases", hb);
            put_f64(tag + "_out_biases", ob);
            put_f64(tag + "_out_biases_rest_maxdiff", {rest});
        }
        // layer 0 weights: the columns of the active inputs for one-hot inputs (nothing else changes), sampled otherwise;
        // rows 0..15 and 16 sampled rows
        {
            const int H = sizes[1];
            std::vector<double> w0;
            std::vector<int32_t> cols, rows;
            if (!idx.empty()) cols = idx; else for (int i = 0; i < 24; ++i) cols.push_back((int)(u01(seed, 71, i) * IN));
            for (int j = 0; j < 32; ++j) rows.push_back(j < 16 ? (j < H ? j : H - 1) : (int)(u01(seed, 72, j) * H));
            for (int j : rows) for (int i : cols) w0.push_back(nn.host_weights[nn.weightOffsets[0] + (size_t)j * IN + i]);
            put_i32(tag + "_w0_rows", rows);
            put_i32(tag + "_w0_cols", cols);
            put_f64(tag + "_w0", w0);                               // [rows][cols]
            // untouched part of layer 0 must be bit-identical to the fill: count of changed entries outside those columns
            int64_t changed = 0;
            std::vector<char> iscol(IN, 0); for (int i : cols) iscol[i] = 1;
            if (!idx.empty())
                for (int j = 0; j < H; ++j) for (int i = 0; i < IN; ++i)
                    if (!iscol[i] && nn.host_weights[nn.weightOffsets[0] + (size_t)j * IN + i] != W[nn.weightOffsets[0] + (size_t)j * IN + i]) ++changed;
            put_i64(tag + "_w0_changed_elsewhere", {changed});
        }
        // layers >= 1: depend on the read of released activations — recorded, never golden
        for (int l = 1; l < nL; ++l) {
            std::vector<int64_t> pos(256);
            std::vector<double> val(256);
            for (int i = 0; i < 256; ++i) {
                // rows 0..95 of the layer (the rows a TD update can touch in the output layer) x any column
                const size_t row = (size_t)(u01(seed, 80 + l, 2 * i) * (sizes[l + 1] < 96 ? sizes[l + 1] : 96));
                const size_t col = (size_t)(u01(seed, 80 + l, 2 * i + 1) * sizes[l]);
                pos[i] = (int64_t)(row * sizes[l] + col);
                val[i] = nn.host_weights[nn.weightOffsets[l] + pos[i]];
            }
            put_i64(tag + "_ub_w" + std::to_string(l) + "_pos", pos);
            put_f64(tag + "_ub_w" + std::to_string(l), val);
        }
    }
    // --- N6: copyWeightsAndBiasesFrom (dqn.cu:507-515, what DQN::updateTargetNetwork calls) copies the HOST vectors, which
    // backpropagate never refreshes: the copy answers like the net BEFORE the update (the build copies the
    // trained device weights instead and documents the divergence)
    {
        nn.host_weights = W; nn.host_biases = B; nn.copyToDevice();
        std::vector<double> x(IN, 0.0);
        for (int i : onehot_indices(seed, 40, IN, 16)) x[i] = 1.0;
        const std::vector<double> q_pre = nn.forward(x);
        std::vector<double> target = q_pre;
        target[0] = -19.0;
        nn.backpropagate(x, target, 0.001);
        const std::vector<double> q_post = nn.forward(x);
        NeuralNetwork other(sizes);
        other.copyWeightsAndBiasesFrom(nn);
        const std::vector<double> q_copy = other.forward(x);
        double d_pre = 0, d_post = 0, d_move = 0;
        for (int i = 0; i < OUT; ++i) {
            auto ab = [](double v) { return v < 0 ? -v : v; };
            d_pre = ab(q_copy[i] - q_pre[i]) > d_pre ? ab(q_copy[i] - q_pre[i]) : d_pre;
            d_post = ab(q_copy[i] - q_post[i]) > d_post ? ab(q_copy[i] - q_post[i]) : d_post;
            d_move = ab(q_post[i] - q_pre[i]) > d_move ? ab(q_post[i] - q_pre[i]) : d_move;
        }
        put_f64("copy_vs_pre_vs_post_moved", {d_pre, d_post, d_move});
    }
    std::fclose(g_out);
    return 0;
}

// time <iters> <L0> <L1> ... : the NN work of one ply of ChessAI::train (chessai.cpp:121-133) — getQValues(state), getQValues(nextState),
// backpropagate(state, target) — as the reference's own runtime does it on THIS GPU: batch 1, a cudaMalloc / cudaFree / synchronize per
// layer.  Prints one JSON line.  (The read of released memory inside backpropagate is there as in every run of the reference.)
int cmd_time(int argc, char** argv) {
    if (argc < 5) return 64;
    const int iters = std::atoi(argv[2]);
    std::vector<int> sizes;
    for (int i = 3; i < argc; ++i) sizes.push_back(std::atoi(argv[i]));
    NeuralNetwork nn(sizes);
    fill_params(nn, 99);
    const int IN = sizes.front();
    std::vector<double> x(IN, 0.0), x2(IN, 0.0);
    for (int i : onehot_indices(99, 1, IN, 30)) x[i] = 1.0;
    for (int i : onehot_indices(99, 2, IN, 30)) x2[i] = 1.0;
    for (int w = 0; w < 3; ++w) { auto q = nn.forward(x); auto q2 = nn.forward(x2); q[w] = 0.1; nn.backpropagate(x, q, 0.001); }
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < iters; ++it) {
        std::vector<double> q = nn.forward(x);
        std::vector<double> q2 = nn.forward(x2);
        double m = q2[0]; for (double v : q2) m = v > m ? v : m;
        q[it % 90] = -1.0 + 0.99 * m;
        nn.backpropagate(x, q, 0.001);
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("{\"plies\": %d, \"seconds\": %.4f, \"plies_per_s\": %.1f}\n", iters, sec, iters / sec);
    return 0;
}

// seq <in.bin> <out.bin> <seed> <lr> <gamma> <L0> ... <Ln> : K plies of the NN half of ChessAI::train's loop body (chessai.cpp:121-133) on the
// reference's own runtime, the parameters carried from ply to ply:
//     targetQ = getQValues(state);  targetQ[action.to] = done ? r : r + gamma * max(getQValues(nextState));  backpropagate(state, targetQ, lr)
// in.bin: int32 K, then per ply 90 bytes state (piece codes 0..14 = plane + 1 of chessai.cpp:278-282), 90 bytes next state, int32 action.to,
// float64 reward, int32 done.  Recorded per ply: Q(state)[0..95] and max Q(nextState) as the reference computed them BEFORE the update,
// the target entry, every bias and a sample of layer 0's weights AFTER it (free of undefined reads), and 64 sampled weights of every layer
// >= 1 (`ub_*`: they depend on the read of released memory, dqn.cu:371 / :441 — the generator compares them with the "released block still
// holds the activation" model and records a flag per ply; a test asserts them only where the flag says the block was intact).
int cmd_seq(int argc, char** argv) {
    if (argc < 9) return 64;
    FILE* in = std::fopen(argv[2], "rb");
    if (!in) return 65;
    g_out = std::fopen(argv[3], "wb");
    if (!g_out) return 65;
    const uint64_t seed = std::strtoull(argv[4], nullptr, 10);
    const double lr = std::atof(argv[5]), gamma = std::atof(argv[6]);
    std::vector<int> sizes;
    for (int i = 7; i < argc; ++i) sizes.push_back(std::atoi(argv[i]));
    const int nL = (int)sizes.size() - 1, IN = sizes.front(), OUT = sizes.back();
    if (IN != 1260 || OUT < 96) return 64;
    int32_t K = 0;
    if (std::fread(&K, 4, 1, in) != 1 || K < 1 || K > 64) return 66;
    NeuralNetwork nn(sizes);
    fill_params(nn, seed);
    auto onehot = [&](const unsigned char* codes) {
        std::vector<double> x(IN, 0.0);
        for (int sq = 0; sq < 90; ++sq) if (codes[sq]) x[sq * 14 + codes[sq] - 1] = 1.0;
        return x;
    };
    const size_t nhid = nn.biasOffsets[nL - 1];
    for (int t = 0; t < K; ++t) {
        unsigned char s[90], s2[90]; int32_t a = 0, done = 0; double r = 0;
        if (std::fread(s, 1, 90, in) != 90 || std::fread(s2, 1, 90, in) != 90 || std::fread(&a, 4, 1, in) != 1 || std::fread(&r, 8, 1, in) != 1 ||
            std::fread(&done, 4, 1, in) != 1 || a < 0 || a >= 90) return 66;
        const std::string tag = "ply" + std::to_string(t);
        const std::vector<double> x = onehot(s), x2 = onehot(s2);
        std::vector<double> target = nn.forward(x);                              // chessai.cpp:121
        const std::vector<double> q2 = nn.forward(x2);                           // :126
        double m = q2[0]; for (double v : q2) m = v > m ? v : m;
        const double y = done ? r : r + gamma * m;                               // :122-128
        put_f64(tag + "_q", std::vector<double>(target.begin(), target.begin() + 96));
        put_f64(tag + "_maxq2_y", {m, y});
        target[a] = y;
        nn.backpropagate(x, target, lr);                                         // :133
        nn.copyFromDevice();
        put_f64(tag + "_hidden_biases", std::vector<double>(nn.host_biases.begin(), nn.host_biases.begin() + nhid));
        put_f64(tag + "_out_biases", std::vector<double>(nn.host_biases.begin() + nhid, nn.host_biases.begin() + nhid + 96));
        {
            // layer 0: rows 0..15 x the columns of the occupied (square, piece) pairs of this ply's state
            std::vector<double> w0;
            std::vector<int32_t> cols;
            for (int sq = 0; sq < 90; ++sq) if (s[sq]) cols.push_back(sq * 14 + s[sq] - 1);
            for (int j = 0; j < 16; ++j) for (int i : cols) w0.push_back(nn.host_weights[nn.weightOffsets[0] + (size_t)j * IN + i]);
            put_i32(tag + "_w0_cols", cols);
            put_f64(tag + "_w0", w0);                                            // [16][cols]
        }
        for (int l = 1; l < nL; ++l) {
            std::vector<double> val(64);
            for (int i = 0; i < 64; ++i) {                                       // tests/refnn.py::sample_positions
                const size_t row = (size_t)(u01(seed, 80 + l, 2 * i) * (sizes[l + 1] < 96 ? sizes[l + 1] : 96));
                const size_t col = (size_t)(u01(seed, 80 + l, 2 * i + 1) * sizes[l]);
                val[i] = nn.host_weights[nn.weightOffsets[l] + row * sizes[l] + col];
            }
            put_f64(tag + "_ub_w" + std::to_string(l), val);
        }
    }
    // the net after the last ply answers the first state once more (the whole trajectory in 96 numbers)
    {
        std::rewind(in);
        int32_t k2; unsigned char s[90];
        if (std::fread(&k2, 4, 1, in) != 1 || std::fread(s, 1, 90, in) != 90) return 66;
        const std::vector<double> q = nn.forward(onehot(s));
        put_f64("final_q_of_first_state", std::vector<double>(q.begin(), q.begin() + 96));
    }
    std::fclose(in);
    std::fclose(g_out);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc >= 2 && std::strcmp(argv[1], "seq") == 0) {
        try { return cmd_seq(argc, argv); }
        catch (const std::exception& e) { std::fprintf(stderr, "xqref_nn: %s\n", e.what()); return 70; }
    }
    if (argc >= 2 && std::strcmp(argv[1], "time") == 0) {
        try { return cmd_time(argc, argv); }
        catch (const std::exception& e) { std::fprintf(stderr, "xqref_nn: %s\n", e.what()); return 70; }
    }
    if (argc >= 2 && std::strcmp(argv[1], "probe") == 0) return cmd_probe();
    if (argc >= 2 && std::strcmp(argv[1], "nn") == 0) {
        try { return cmd_nn(argc, argv); }
        catch (const std::exception& e) { std::fprintf(stderr, "xqref_nn: %s\n", e.what()); return 70; }
    }
    std::fprintf(stderr, "usage: xqref_nn probe | nn <out.bin> <seed> <L0> <L1> ... <Ln> | seq <in.bin> <out.bin> <seed> <lr> <gamma> <L0> ... <Ln> | "
                         "time <iters> <L0> ... <Ln>\n");
    return 64;
}
