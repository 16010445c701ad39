// Bitboard game rules for the self-play engine (host + device).
//
// A position is (nx, pv, nm): stones of the side to move, stones of the
// previous mover, and the number of moves played.  That is exactly the
// information the reference renders for its network
// (alphazero/games/connectfour.py:112-120, alphazero/games/tictactoe.py:50-62:
// plane 0 = next player, plane 1 = previous player) and is sufficient for every
// rule query the search makes.
//
// Connect Four: bit (col*7 + row), row 0 = bottom, the 7th bit of each column is
// a guard so shifts never wrap between columns.  Tic-tac-toe: bit = square 0..8.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define AZ_HD __host__ __device__ __forceinline__
#else
#define AZ_HD inline
#endif

#define AZ_W_NONE (-2)
#define AZ_W_DRAW (-1)

AZ_HD int az_popc32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return __builtin_popcount(x);
#endif
}

// index of the k-th (0-based) set bit of mask; k must be < popcount(mask)
AZ_HD int az_nth_set_bit(uint32_t mask, int k) {
    for (int i = 0; i < k; ++i) mask &= mask - 1;
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffs(mask) - 1;
#else
    return __builtin_ffs((int)mask) - 1;
#endif
}

struct GameC4 {
    static constexpr int A = 7, LMAX = 42, H = 6, W = 7, ID = 1;

    // connectfour.py:80-81: column c is playable while it holds fewer than 6 stones
    AZ_HD static uint32_t legal(uint64_t nx, uint64_t pv) {
        const uint64_t m = nx | pv;
        uint32_t r = 0;
#pragma unroll
        for (int c = 0; c < 7; ++c) r |= (uint32_t)((~m >> (c * 7 + 5)) & 1ull) << c;
        return r;
    }
    // game.py:30-33 + the row bookkeeping of connectfour.py:114-119
    AZ_HD static void play(uint64_t& nx, uint64_t& pv, int& nm, int a) {
        const uint64_t m = nx | pv;
        const uint64_t stone = (m + (1ull << (a * 7))) & (0x3Full << (a * 7));
        const uint64_t mine = nx | stone;
        nx = pv;
        pv = mine;
        nm += 1;
    }
    AZ_HD static bool four(uint64_t b) {
        uint64_t t = b & (b >> 7);            // horizontal
        if (t & (t >> 14)) return true;
        t = b & (b >> 6);                      // diagonal
        if (t & (t >> 12)) return true;
        t = b & (b >> 8);                      // diagonal
        if (t & (t >> 16)) return true;
        t = b & (b >> 1);                      // vertical
        return (t & (t >> 2)) != 0;
    }
    // connectfour.py:92-110: only the previous mover can have completed a line
    AZ_HD static int winner(uint64_t nx, uint64_t pv, int nm) {
        (void)nx;
        if (nm < 7) return AZ_W_NONE;
        if (four(pv)) return 1 - (nm & 1);
        if (nm == 42) return AZ_W_DRAW;
        return AZ_W_NONE;
    }
    AZ_HD static bool cell(uint64_t b, int row, int col) { return (b >> (col * 7 + row)) & 1ull; }
};

struct GameTTT {
    static constexpr int A = 9, LMAX = 9, H = 3, W = 3, ID = 0;

    AZ_HD static uint32_t legal(uint64_t nx, uint64_t pv) { return (uint32_t)(~(nx | pv)) & 0x1FFu; }
    AZ_HD static void play(uint64_t& nx, uint64_t& pv, int& nm, int a) {
        const uint64_t mine = nx | (1ull << a);
        nx = pv;
        pv = mine;
        nm += 1;
    }
    AZ_HD static bool line(uint32_t b) {
        // tictactoe.py:14-23
        return ((b & 0x007) == 0x007) | ((b & 0x038) == 0x038) | ((b & 0x1C0) == 0x1C0) |
               ((b & 0x049) == 0x049) | ((b & 0x092) == 0x092) | ((b & 0x124) == 0x124) |
               ((b & 0x111) == 0x111) | ((b & 0x054) == 0x054);
    }
    // tictactoe.py:36-44: player 0's lines are tested before player 1's
    AZ_HD static int winner(uint64_t nx, uint64_t pv, int nm) {
        if (nm < 5) return AZ_W_NONE;
        const uint32_t p0 = (uint32_t)((nm & 1) ? pv : nx), p1 = (uint32_t)((nm & 1) ? nx : pv);
        if (line(p0)) return 0;
        if (line(p1)) return 1;
        if (nm == 9) return AZ_W_DRAW;
        return AZ_W_NONE;
    }
    AZ_HD static bool cell(uint64_t b, int row, int col) { return (b >> (row * 3 + col)) & 1ull; }
};

// game.py:42-47 evaluated for the previous mover (mcts.py:134)
AZ_HD float az_terminal_value_prev(int winner, int nm) {
    if (winner < 0) return 0.0f;
    return (winner == 1 - (nm & 1)) ? 1.0f : -1.0f;
}
