// xq_common.h — shared host/device definitions for libxqhip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/xq_capi.h"

namespace xq {

// ---- error plumbing: never throw across the C boundary -------------------------------------------------------
inline std::string& last_error_slot() {
    static thread_local std::string s;
    return s;
}
inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_slot() = buf;
    return code;
}

#define XQ_HIP(call)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return ::xq::fail(XQ_ERR_RUNTIME, "HIP error: %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, \
                              __LINE__, #call);                                                            \
    } while (0)

#define XQ_TRY(call)               \
    do {                           \
        int rc_ = (call);          \
        if (rc_ != XQ_OK) return rc_; \
    } while (0)

// ---- board encoding ------------------------------------------------------------------------------------------
constexpr int kSquares = 90;
constexpr int kBoardWords = XQ_BOARD_WORDS;   // 12 x u32 = 96 nibbles, squares 90..95 stay 0
constexpr int kMaxMoves = XQ_MAX_MOVES;
constexpr int kStateSize = 1260;              // 90 * 14 (chessai.cpp:399)

enum : int { T_EMPTY = 0, T_GENERAL = 1, T_ADVISOR = 2, T_ELEPHANT = 3, T_HORSE = 4, T_CHARIOT = 5, T_CANNON = 6, T_SOLDIER = 7 };
enum : int { C_RED = 0, C_BLACK = 1, C_NONE = 2 };

__host__ __device__ inline int code_type(int c) { return c == 0 ? 0 : (c > 7 ? c - 7 : c); }
__host__ __device__ inline int code_color(int c) { return c == 0 ? C_NONE : (c > 7 ? C_BLACK : C_RED); }

// start position (chessboard.cpp:8-29), packed 8 squares per word, square s in nibble s&7 of word s>>3
inline void start_position(uint8_t sq[96]) {
    static const uint8_t back[9] = {5, 4, 3, 2, 1, 2, 3, 4, 5};
    memset(sq, 0, 96);
    for (int c = 0; c < 9; ++c) {
        sq[c] = back[c];
        sq[81 + c] = (uint8_t)(back[c] + 7);
    }
    sq[2 * 9 + 1] = sq[2 * 9 + 7] = 6;
    sq[7 * 9 + 1] = sq[7 * 9 + 7] = 13;
    for (int c = 0; c < 9; c += 2) {
        sq[3 * 9 + c] = 7;
        sq[6 * 9 + c] = 14;
    }
}
inline void pack_board(const uint8_t* sq90, uint32_t words[kBoardWords]) {
    for (int w = 0; w < kBoardWords; ++w) {
        uint32_t v = 0;
        for (int k = 0; k < 8; ++k) {
            int s = w * 8 + k;
            uint32_t c = s < kSquares ? (sq90[s] & 15u) : 0u;
            v |= c << (4 * k);
        }
        words[w] = v;
    }
}
inline void unpack_board(const uint32_t words[kBoardWords], uint8_t* sq90) {
    for (int s = 0; s < kSquares; ++s) sq90[s] = (uint8_t)((words[s >> 3] >> (4 * (s & 7))) & 15u);
}

// ---- Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) ---------
struct Philox4 {
    uint32_t v[4];
};
__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}

}  // namespace xq
