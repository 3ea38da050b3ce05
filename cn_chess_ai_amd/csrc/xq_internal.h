// xq_internal.h — handle layouts shared by the translation units of libxqhip (not part of the ABI).
#pragma once

#include "xq_common.h"

#include <vector>

// replay ring in HBM: structure of arrays, states as packed boards (48 B) — never 1260 floats
struct ReplayDev {
    uint32_t* boards = nullptr;       // [capacity][12]
    uint32_t* next_boards = nullptr;  // [capacity][12]
    int32_t* action_to = nullptr;     // [capacity]   action.to (0..89), -1 = empty slot (contributes no gradient)
    float* reward = nullptr;          // [capacity]
    uint8_t* done = nullptr;          // [capacity]
    int capacity = 0;
};

struct xq_replay {
    ReplayDev dev;
    int size = 0;
    int write_pos = 0;
    uint64_t total = 0;
    uint64_t seed = 0;
    uint64_t sample_calls = 0;
    int32_t* slots_dev = nullptr;     // last sample()
    int slots_cap = 0;
    int last_batch = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
};

struct xq_env {
    int n = 0;
    uint64_t seed = 0;
    uint32_t first_id = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t* boards = nullptr;        // [n][12]
    uint4* meta = nullptr;             // [n] {moveCount|player<<16, red|black<<16, plies, episodes}
    uint4* stats = nullptr;            // [n] {red wins, black wins, captures, explored}
    xq_step_result* results = nullptr; // [n]
    uint16_t* codes = nullptr;         // [n][128] scratch for legal_moves
    int32_t* counts = nullptr;         // [n]
    int32_t* actions = nullptr;        // [n] scratch for step()
    float* q90 = nullptr;              // [n][96] scratch for selfplay_step_host
    uint8_t* validmat = nullptr;       // [8100]
    xq_episode_record* ep_ring = nullptr;
    int ep_cap = 0;
    unsigned long long* ep_head = nullptr;   // device counter of finished episodes
    uint64_t ep_drained = 0;
};

namespace xq {
// env-side launchers used by the trainer
int env_selfplay_launch(xq_env* env, const float* q90_dev, int q_stride, uint32_t eps_u32, xq_step_result* results_dev,
                        xq_replay* replay);
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
}  // namespace xq
