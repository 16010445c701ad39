// az_train_c4_dev.h -- the steps of the training pass for the Connect Four network with 32 trunk channels
// (alphazero/training.py:57-71 with alphazero/models.py:6-87, alphazero/games/connectfour.py:47-72), each written
// once as a device function for one board per workgroup and used by both forms of the pass in az_train_c4.hip: the
// chain of launches (one step per launch) and the one-launch pass (all steps in one resident grid).  The two forms
// differ only in how the boards' partial sums travel from one step to the next, which is the `EX` argument:
//   ex.gather<NQ>(tot, scratch)   adds the rows [board][64] that the previous step's workgroups posted, in board
//                                 order, in double, by NQ * 64 threads -> tot[64] (LDS); true = give up (one-launch
//                                 form only: a workgroup did not show up in time)
//   ex.post(v)                    threads 0..63: this board's row for the next step
// A step written for 256 threads (NT = TT) leaves the other threads of a 512-thread workgroup idle, so both forms add
// the same numbers in the same order and produce the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define TC 32                  // trunk channels
#define CELLS 42
#define PADCELLS 72            // the board with its zero halo: rows -1..6, columns -1..7
#define TT 256                 // threads of the small steps: thread = (channel, one of 8 cell groups)
#define NPOS 6                 // cells per thread there (8 groups x 6 >= 42)
#define TTC 512                // threads of the 3x3 conv steps: two waves per SIMD, 16 cell groups
#define NPC 3                  // cells per thread there (16 groups x 3 >= 42): cell = group + 16 k
#define TS 33                  // floats per cell in the conv steps' LDS tiles (32 channels + 1: lanes that read one
                               // channel of 32 cells hit different banks)
#define PBS 65                 // floats per channel row of the partial-result buffer of the conv steps
#define EPT 3                  // elements of a board tensor [42][32] per thread of a conv step: element tid + 512 j is
                               // (cell = group + 16 j, channel = tid & 31) -- the same cells the thread computes

// The thread's index as a value the compiler has to take as new each time: in the one-launch pass the steps sit in
// loops, and everything a step derives from the thread index (LDS addresses, cell numbers, masks) would otherwise be
// computed once in front of the loops and kept in registers for the whole kernel -- some hundreds of them.
__device__ __forceinline__ int step_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

__device__ __forceinline__ int pad_index(int cell) { return (cell / 7 + 1) * 9 + cell % 7 + 1; }
__device__ __forceinline__ int tap_offset(int tap) { return (tap / 3 - 1) * 9 + tap % 3 - 1; }

// Rows part[board][64] (two quantities x 32 channels, summed over the board's cells) added up in board order, in
// double, by the workgroup's first NQ * 64 threads: no atomics, the same bits in every workgroup and every run.
template <int NQ>
__device__ __forceinline__ void reduce_rows(const float* __restrict__ part, int boards, double* tot, double* scratch /* LDS [8][64] */) {
    const int col = threadIdx.x & 63, q = threadIdx.x >> 6;
    if (q < NQ) {
        double s = 0.0;
        for (int b0 = q; b0 < boards; b0 += 32 * NQ) {         // up to 32 loads in flight (one trip to memory for 128
            float v[32];                                       // boards), added in board order
#pragma unroll
            for (int j = 0; j < 32; ++j) v[j] = b0 + NQ * j < boards ? part[(size_t)(b0 + NQ * j) * 64 + col] : 0.0f;
#pragma unroll
            for (int j = 0; j < 32; ++j) s += (double)v[j];
        }
        scratch[q * 64 + col] = s;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        double t = 0.0;
        for (int k = 0; k < NQ; ++k) t += scratch[k * 64 + col];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
}

// launch boundaries as the synchronisation: rows of floats in global memory, written by one launch, read by the next
struct ChainExchange {
    const float* part_in;
    float* part_out;
    int boards;
    template <int NQ> __device__ __forceinline__ bool gather(double* tot, double* scratch) const {
        reduce_rows<NQ>(part_in, boards, tot, scratch);
        return false;
    }
    __device__ __forceinline__ void post(float v) const { part_out[(size_t)blockIdx.x * 64 + threadIdx.x] = v; }
    __device__ __forceinline__ long long* marks() const { return nullptr; }
};

// Inside one launch: a row is 64 pairs (value, tag) of eight bytes, stored and loaded as single agent-scope accesses.
// A pair is valid as soon as its tag is the expected one, so neither side needs a fence or a counter; consecutive
// steps use two row sets in turn (a workgroup can be at most one step ahead of the slowest one, because it needs
// every workgroup's row of the step before).  A reader gives up after POLL_LIMIT rounds of polling (0.4 s to a second)
// or as soon as another workgroup has: the launch then ends with an error instead of hanging.
#define POLL_LIMIT (1 << 20)
#define POLL_SLOTS 16         // pairs a thread has in flight per round of polling
#ifdef PASS_TIMING
#define STEP_MARK(ex, i) do { if ((ex).marks() && blockIdx.x == 0 && threadIdx.x == 0) (ex).marks()[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define STEP_MARK(ex, i) do { } while (0)
#endif
struct ResidentExchange {
    const unsigned long long* rows_in;
    unsigned long long* rows_out;
    uint32_t tag_in, tag_out;
    int boards;
    int* error_word;
    long long* stamp;             // timing builds: where workgroup 0 notes the time at which its sums were complete
    long long* detail;            // and marks inside the step
    template <int NQ> __device__ __forceinline__ bool gather(double* tot, double* scratch) const {
        const int col = threadIdx.x & 63, q = threadIdx.x >> 6;
        bool gave_up = false;
        if (q < NQ) {
            // the same additions in the same order as reduce_rows<NQ>, POLL_SLOTS boards at a time
            double s = 0.0;
            for (int b0 = q; b0 < boards && !gave_up; b0 += POLL_SLOTS * NQ) {
                // every round loads all of the thread's pairs again (no per-pair bookkeeping: the loop stays free of
                // branches); a pair beyond the batch reads the round's first board instead and counts as zero
                unsigned long long w[POLL_SLOTS];
                uint32_t at[POLL_SLOTS];
#pragma unroll
                for (int j = 0; j < POLL_SLOTS; ++j) at[j] = (uint32_t)((b0 + NQ * j < boards ? b0 + NQ * j : b0) * 64 + col);
                int polls = 0;
                for (;;) {
#pragma unroll
                    for (int j = 0; j < POLL_SLOTS; ++j)
                        w[j] = __hip_atomic_load(rows_in + at[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bool all = true;
#pragma unroll
                    for (int j = 0; j < POLL_SLOTS; ++j) all = all && (uint32_t)(w[j] >> 32) == tag_in;
                    if (all) break;
                    if ((++polls & 63) == 0 &&
                        (polls >= POLL_LIMIT || __hip_atomic_load(error_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        gave_up = true;
                        break;
                    }
                }
#pragma unroll
                for (int j = 0; j < POLL_SLOTS; ++j) s += b0 + NQ * j < boards ? (double)__uint_as_float((uint32_t)w[j]) : 0.0;
            }
            scratch[q * 64 + col] = s;
        }
        const int any = __syncthreads_or(gave_up ? 1 : 0);
        if (threadIdx.x < 64) {
            double t = 0.0;
            for (int k = 0; k < NQ; ++k) t += scratch[k * 64 + col];
            tot[threadIdx.x] = t;
        }
        __syncthreads();
        if (any && threadIdx.x == 0) __hip_atomic_store(error_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (stamp && blockIdx.x == 0 && threadIdx.x == 0) stamp[0] = (long long)__builtin_readcyclecounter();
        return any != 0;
    }
    __device__ __forceinline__ long long* marks() const { return detail; }
    __device__ __forceinline__ void post(float v) const {
#ifdef PASS_TEST_STALL
        // test builds: workgroup 1 never posts its row of the step tagged PASS_TEST_STALL, so every reader has to give up
        if (blockIdx.x == 1 && (tag_out & 0xFFu) == PASS_TEST_STALL) return;
#endif
        __hip_atomic_store(rows_out + (size_t)blockIdx.x * 64 + threadIdx.x, ((unsigned long long)tag_out << 32) | __float_as_uint(v),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// per board: the 32 (or fewer) column sums of two quantities over the board's cells, from per-thread values of the
// first NG * 32 threads (thread = (channel = tid & 31, group = tid >> 5)), posted as the board's row
template <int NG, class EX>
__device__ __forceinline__ void board_partials(float a, float b, float* red /* LDS [2][NG][32] */, const EX& ex) {
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    if (g < NG) {
        red[g * 32 + c] = a;
        red[NG * 32 + g * 32 + c] = b;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const float* r = red + (threadIdx.x >> 5) * NG * 32 + (threadIdx.x & 31);
        float s = 0.0f;
        for (int k = 0; k < NG; ++k) s += r[k * 32];
        ex.post(s);
    }
}

// statistics of a batch-norm input from the totals (sum, sum of squares): st[0..31] mean, [32..63] 1/sqrt(var + eps),
// [64..95] biased variance -- into LDS for this workgroup, and by workgroup 0 into `keep` for later
__device__ __forceinline__ void stats_from_totals(const double* tot, int nch, long n, float eps, float* st /* LDS [96] */, float* keep) {
    if ((int)threadIdx.x < nch) {
        const int c = threadIdx.x;
        const double mean = tot[c] / (double)n;
        double var = tot[32 + c] / (double)n - mean * mean;
        if (var < 0.0) var = 0.0;
        st[c] = (float)mean;
        st[32 + c] = (float)(1.0 / sqrt(var + (double)eps));
        st[64 + c] = (float)var;
        if (blockIdx.x == 0) { keep[c] = st[c]; keep[32 + c] = st[32 + c]; keep[64 + c] = st[64 + c]; }
    }
    __syncthreads();
}

// a batch-norm whose statistics are being made by this step: st = where this workgroup keeps them (LDS [96]),
// stat_keep = the copy in global memory for the steps that come later in other launches
struct BnRef { float* st; float* stat_keep; const float* gamma; const float* beta; };

// scale / shift of a batch-norm in training mode from its statistics: y = z * sc + sh
__device__ __forceinline__ void bn_scale_shift(const float* st, const float* gamma, const float* beta, float* sc, float* sh) {
    if (threadIdx.x < TC) {
        const int c = threadIdx.x;
        const float k = gamma[c] * st[32 + c];
        sc[c] = k;
        sh[c] = beta[c] - st[c] * k;
    }
}

// both of the above for a 32-channel batch-norm whose statistics are being made, by the same 32 threads, one barrier
__device__ __forceinline__ void stats_scale_shift(const double* tot, long n, float eps, const BnRef& bn, float* sc, float* sh) {
    if (threadIdx.x < TC) {
        const int c = threadIdx.x;
        const double mean = tot[c] / (double)n;
        double var = tot[32 + c] / (double)n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float m = (float)mean, inv = (float)(1.0 / sqrt(var + (double)eps)), v = (float)var;
        bn.st[c] = m; bn.st[32 + c] = inv; bn.st[64 + c] = v;
        if (blockIdx.x == 0) { bn.stat_keep[c] = m; bn.stat_keep[32 + c] = inv; bn.stat_keep[64 + c] = v; }
        const float k = bn.gamma[c] * inv;
        sc[c] = k;
        sh[c] = bn.beta[c] - m * k;
    }
    __syncthreads();
}

// The 3x3 conv of one board on the fp32 matrix instruction v_mfma_f32_16x16x4_f32 (D[16 out][16 cells] += W[out][4 k]
// X[4 k][cells]): the board is three tiles of 16 cells (42 of 48 exist; 32-wide tiles would carry 64), the output two
// halves of 16 channels.  Waves 0..5 take one (channel half, cell tile) each through taps 0..6, waves 6 and 7 the
// taps 7 and 8 of three of those each: 56 and 48 matrix instructions per wave, 104 to 112 per SIMD.  The forward conv
// (SIGN = +1, w = [tap][ci][co]) and the input gradient (SIGN = -1, w = [tap][co][ci]) are the same loop; in both the
// second index has its halves of 16 swapped in odd rows (k_train_prep), which spreads the four rows of an operand
// over the LDS banks.  A wave
// leaves its result in pbuf [first taps / last taps][out][PBS]; the caller adds the two up.  tile = [72 padded cells][TS].
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
template <int SIGN>
__device__ __forceinline__ void conv_mfma(const float* tile, const float* w, float* pbuf, int wave, int lane) {
    const int col = lane & 15, kq = lane >> 4;
    // (a wave's matrix instructions go to independent accumulators in turn: one that waits for its predecessor's
    // result leaves the pipe idle for more than half of the time)
    if (wave < 6) {
        const int ot = wave & 1, ct = wave >> 1;
        const int cell = ct * 16 + col;
        const int pi = pad_index(cell < CELLS ? cell : 0);
        f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc0;
        const float* wbase = w + kq * TC + ((ot * 16 + col) ^ ((kq & 1) * 16));
        const float* xbase = tile + pi * TS + kq;
#pragma unroll
        for (int t = 0; t < 6; t += 2) {
            const float* x0 = xbase + SIGN * tap_offset(t) * TS;
            const float* x1 = xbase + SIGN * tap_offset(t + 1) * TS;
            const float* w0 = wbase + t * TC * TC;
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                acc0 = MFMA16(w0[4 * s4 * TC], x0[4 * s4], acc0);
                acc1 = MFMA16(w0[TC * TC + 4 * s4 * TC], x1[4 * s4], acc1);
            }
        }
        {
            const float* x0 = xbase + SIGN * tap_offset(6) * TS;
            const float* w0 = wbase + 6 * TC * TC;
#pragma unroll
            for (int s4 = 0; s4 < 8; s4 += 2) {
                acc0 = MFMA16(w0[4 * s4 * TC], x0[4 * s4], acc0);
                acc1 = MFMA16(w0[4 * (s4 + 1) * TC], x0[4 * (s4 + 1)], acc1);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) pbuf[(ot * 16 + 4 * kq + r) * PBS + cell] = acc0[r] + acc1[r];
    } else {
        f32x4 acc[3];
        const float* xb[3];
        const float* wb[3];
        int cells[3], ots[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int id = (wave - 6) * 3 + n, ot = id & 1, ct = id >> 1;
            const int cell = ct * 16 + col;
            cells[n] = cell; ots[n] = ot;
            xb[n] = tile + pad_index(cell < CELLS ? cell : 0) * TS + kq;
            wb[n] = w + kq * TC + ((ot * 16 + col) ^ ((kq & 1) * 16));
            acc[n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int t = 7; t < 9; ++t) {
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
#pragma unroll
                for (int n = 0; n < 3; ++n)
                    acc[n] = MFMA16(wb[n][(t * TC + 4 * s4) * TC], xb[n][SIGN * tap_offset(t) * TS + 4 * s4], acc[n]);
            }
        }
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) pbuf[(TC + ots[n] * 16 + 4 * kq + r) * PBS + cells[n]] = acc[n][r];
    }
}
// the two partial results of (channel c, cell), added in order
__device__ __forceinline__ float conv_total(const float* pbuf, int c, int cell) {
    const float* p = pbuf + c * PBS + cell;
    return p[0] + p[TC * PBS];
}

// The board's share of a conv's weight gradient, dW[tap][ci][co] = sum over cells of A[cell + tap][ci] dZ[cell][co], on
// v_mfma_f32_16x16x4_f32: a unit is one 16 x 16 tile (ci half, co half) of one tap, its k the cells, eleven steps of
// four (the last two cells of the last step are zeros); the 36 units go round the eight waves (unit = wave + 8 n), so
// a wave has five or four.  out = [tap][ci][co].  WgOld = what earlier applications of the shared block have left in
// this wave's units (requested at the start of the step, used at its end).
#define WG_UNITS 5
struct WgOld { f32x4 u[WG_UNITS]; };
__device__ __forceinline__ WgOld wgrad_load_old(bool accum, const float* out, int wave, int lane) {
    WgOld o;
    const int col = lane & 15, r4 = (lane >> 4) * 4;
#pragma unroll
    for (int n = 0; n < WG_UNITS; ++n) {
        const int unit = wave + 8 * n, tap = unit >> 2, ci0 = (unit & 2) * 8, co0 = (unit & 1) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o.u[n][i] = (accum && unit < 36) ? out[(tap * TC + ci0 + r4 + i) * TC + co0 + col] : 0.0f;
    }
    return o;
}
__device__ __forceinline__ void wgrad_mfma(const float* at, const float* dzt, float* out, const WgOld& old, int wave, int lane) {
    const int col = lane & 15, kq = lane >> 4, r4 = kq * 4;
    // where this lane's cell of each of the eleven steps sits in a tile (the same for every unit)
    int pos[11];
#pragma unroll
    for (int s4 = 0; s4 < 11; ++s4) {
        const int cell = 4 * s4 + kq;
        pos[s4] = pad_index(cell < CELLS ? cell : 0) * TS + col;
    }
    const bool tail = kq >= 2;                         // cells 42 and 43 of the last step do not exist
    // the wave's units side by side, step by step: five independent accumulators keep the matrix pipe busy
    f32x4 acc[WG_UNITS];
    const float* ap[WG_UNITS];
    const float* dp[WG_UNITS];
#pragma unroll
    for (int n = 0; n < WG_UNITS; ++n) {
        const int unit = wave + 8 * n < 36 ? wave + 8 * n : wave;          // (the idle fifth unit repeats the first)
        const int tap = unit >> 2, ci0 = (unit & 2) * 8, co0 = (unit & 1) * 16;
        ap[n] = at + tap_offset(tap) * TS + ci0;
        dp[n] = dzt + co0;
        acc[n] = old.u[n];
    }
    const bool five = wave + 8 * (WG_UNITS - 1) < 36;
    // the operands of step s4 + 1 are requested before the matrix instructions of step s4 are issued (a wave issues
    // in order: reads placed behind them would wait until the pipe has taken them all)
    float a[2][WG_UNITS], d[2][WG_UNITS];
#pragma unroll
    for (int n = 0; n < WG_UNITS; ++n) { a[0][n] = ap[n][pos[0]]; d[0][n] = dp[n][pos[0]]; }
#pragma unroll
    for (int s4 = 0; s4 < 11; ++s4) {
        const int cur = s4 & 1, nxt = cur ^ 1;
        if (s4 < 10) {
#pragma unroll
            for (int n = 0; n < WG_UNITS; ++n) { a[nxt][n] = ap[n][pos[s4 + 1]]; d[nxt][n] = dp[n][pos[s4 + 1]]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < WG_UNITS; ++n) {
            if (n == WG_UNITS - 1 && !five) continue;
            float x = a[cur][n], y = d[cur][n];
            if (s4 == 10 && tail) { x = 0.0f; y = 0.0f; }
            acc[n] = MFMA16(x, y, acc[n]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int n = 0; n < WG_UNITS; ++n) {
        const int unit = wave + 8 * n;
        if (unit < 36) {
            const int tap = unit >> 2, ci0 = (unit & 2) * 8, co0 = (unit & 1) * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) out[(tap * TC + ci0 + r4 + i) * TC + co0 + col] = acc[n][i];
        }
    }
}

// the conv's 36 KB of weights, global -> registers (issued early) -> LDS
struct WRegs { float4 r[5]; };
__device__ __forceinline__ WRegs load_conv_weights(const float* __restrict__ w) {
    WRegs x;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int i = threadIdx.x + j * TTC;
        x.r[j] = i < 9 * TC * TC / 4 ? reinterpret_cast<const float4*>(w)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return x;
}
__device__ __forceinline__ void store_conv_weights(float* wl, const WRegs& x) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int i = threadIdx.x + j * TTC;
        if (i < 9 * TC * TC / 4) reinterpret_cast<float4*>(wl)[i] = x.r[j];
    }
}

// LDS of a 3x3 conv step (floats): forward uses wl, at, coef, red, pbuf; backward all of it
#define CONV_LDS_WL 0                                      // [9][32][32]
#define CONV_LDS_DZT (9 * TC * TC)                         // [72][TS] dZ
#define CONV_LDS_AT (CONV_LDS_DZT + PADCELLS * TS)         // [72][TS] A
#define CONV_LDS_COEF (CONV_LDS_AT + PADCELLS * TS)        // 7 x [32]: sc, sh, k1, m1, m2, mean, invstd
#define CONV_LDS_RED (CONV_LDS_COEF + 7 * TC)              // [1024]
#define CONV_LDS_PBUF (CONV_LDS_RED + 1024)                // [4][32][PBS] (the conv uses half of it, the heads the rest)
#define CONV_LDS_FLOATS (CONV_LDS_PBUF + 4 * TC * PBS)

// n floats from global memory into LDS by the whole workgroup, sixteen loads of a thread in flight
__device__ __forceinline__ void stage_floats(float* dst, const float* src, int n, int tid) {
    const int nt = blockDim.x;
    for (int i0 = tid; i0 < n; i0 += 16 * nt) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = i0 + u * nt < n ? src[i0 + u * nt] : 0.0f;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (i0 + u * nt < n) dst[i0 + u * nt] = v[u];
    }
}

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
// stem: conv 2 -> 32 from the rendered planes x [board][2][6][7] -> Z0 and the board's sums of Z0, Z0^2
template <class EX>
__device__ __forceinline__ void stem_fwd_step(const float* x, const float* ws, float* zout, const EX& ex, float* xt /* LDS [144] */,
                                              float* wl /* LDS [576] */, float* red /* LDS [512] */) {
    const int b = blockIdx.x, tid = step_tid(), co = tid & 31, g = tid >> 5;
    const bool on = tid < TT;
    if (on) {
        for (int i = tid; i < PADCELLS * 2; i += TT) xt[i] = 0.0f;
        for (int i = tid; i < 18 * TC; i += TT) wl[i] = ws[i];
    }
    __syncthreads();
    if (on)
        for (int i = tid; i < 2 * CELLS; i += TT) xt[pad_index(i % CELLS) * 2 + i / CELLS] = x[(size_t)b * 84 + i];
    __syncthreads();
    float s = 0.0f, q = 0.0f;
    if (on) {
#pragma unroll
        for (int k = 0; k < NPOS; ++k) {
            const int cell = g + 8 * k;
            if (cell >= CELLS) break;
            const int pi = pad_index(cell);
            float acc = 0.0f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int pj = (pi + tap_offset(tap)) * 2;
                acc = fmaf(xt[pj], wl[(tap * 2) * TC + co], acc);
                acc = fmaf(xt[pj + 1], wl[(tap * 2 + 1) * TC + co], acc);
            }
            zout[((size_t)b * CELLS + cell) * TC + co] = acc;
            s += acc; q = fmaf(acc, acc, q);
        }
    }
    board_partials<TT / 32>(s, q, red, ex);
}

// one 3x3 conv of the tower (512 threads): Zout = conv(A), A = ReLU(BN(Zin) (+ Xprev)); BN's statistics from the rows
// the previous step posted (the sums of Zin, Zin^2 per board); the board's sums of Zout, Zout^2 are posted.  Every
// global load of the step is issued before the first wait.  lds = CONV_LDS_FLOATS floats.
template <bool RES, bool WRITE_X, class EX>
__device__ __forceinline__ bool conv_fwd_step(const float* zin, BnRef bn, const float* xprev, float* xout, const float* wf, float* zout,
                                              int boards, float eps, const EX& ex, float* lds, double* tot, double* scratch) {
    float* wl = lds + CONV_LDS_WL;
    float* tile = lds + CONV_LDS_AT;
    float* sc = lds + CONV_LDS_COEF;
    float* sh = sc + TC;
    float* red = lds + CONV_LDS_RED;
    float* pbuf = lds + CONV_LDS_PBUF;
    const int b = blockIdx.x, tid = step_tid(), co = tid & 31, g = tid >> 5;
    const size_t base = (size_t)b * CELLS * TC;
    float zr[EPT], xr[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = tid + j * TTC;
        zr[j] = i < CELLS * TC ? zin[base + i] : 0.0f;
        xr[j] = (RES && i < CELLS * TC) ? xprev[base + i] : 0.0f;
    }
    const WRegs wr = load_conv_weights(wf);
    for (int i = tid; i < PADCELLS * TS; i += TTC) tile[i] = 0.0f;
    store_conv_weights(wl, wr);                        // (while the rows are on their way)
    if (ex.template gather<TTC / 64>(tot, scratch)) return true;
    stats_scale_shift(tot, (long)boards * CELLS, eps, bn, sc, sh);
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = tid + j * TTC;
        if (i < CELLS * TC) {
            float a = fmaf(zr[j], sc[co], sh[co]);
            if (RES) a += xr[j];
            a = fmaxf(a, 0.0f);
            if (WRITE_X) xout[base + i] = a;
            tile[pad_index(i >> 5) * TS + co] = a;
        }
    }
    __syncthreads();
    conv_mfma<1>(tile, wl, pbuf, tid >> 6, tid & 63);
    __syncthreads();
    float s = 0.0f, q = 0.0f;
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
        const int cell = g + 16 * k;
        if (cell < CELLS) {
            const float zv = conv_total(pbuf, co, cell);
            zout[base + cell * TC + co] = zv;
            s += zv; q = fmaf(zv, zv, q);
        }
    }
    board_partials<TTC / 32>(s, q, red, ex);
    return false;
}

// heads, first part (256 threads): the tower's output X = ReLU(BN(Zin) (+ Xprev)) (stored), the three 1x1 convs (two
// policy planes with bias, one value plane) -> zh [board][cell][4], the board's sums of the three planes posted
template <bool RES, class EX>
__device__ __forceinline__ bool head_conv_fwd_step(const float* zin, BnRef bn, const float* xprev, float* xout, const float* wp,
                                                   const float* bp, const float* wv, float* zh, int boards, float eps, const EX& ex,
                                                   float* tile /* LDS [72][32] */, float* sc, float* sh, float* wh /* LDS [96] */,
                                                   float* red /* LDS [512] */, double* tot, double* scratch) {
    const int b = blockIdx.x, tid = step_tid();
    const bool on = tid < TT;
    if (tid < 2 * TC) wh[tid] = wp[tid];
    if (tid < TC) wh[2 * TC + tid] = wv[tid];
    if (ex.template gather<TT / 64>(tot, scratch)) return true;
    stats_scale_shift(tot, (long)boards * CELLS, eps, bn, sc, sh);
    if (on)
        for (int i = tid; i < CELLS * TC; i += TT) {
            const int cell = i >> 5, c = i & 31;
            const size_t gi = (size_t)b * CELLS * TC + i;
            float a = fmaf(zin[gi], sc[c], sh[c]);
            if (RES) a += xprev[gi];
            a = fmaxf(a, 0.0f);
            xout[gi] = a;
            tile[pad_index(cell) * TC + c] = a;
        }
    __syncthreads();
    float s = 0.0f, q = 0.0f;
    if (tid < 3 * CELLS) {
        const int cell = tid % CELLS, h = tid / CELLS;
        const float* a = tile + pad_index(cell) * TC;
        float acc = 0.0f;
#pragma unroll 8
        for (int c = 0; c < TC; ++c) acc = fmaf(a[c], wh[h * TC + c], acc);
        if (h < 2) acc += bp[h];
        zh[((size_t)b * CELLS + cell) * 4 + h] = acc;
        s = acc; q = acc * acc;
    }
    // column sums over the 42 cells of each plane: thread (cell, h) -> red[h][cell], summed by six threads
    if (on) { red[tid] = s; red[256 + tid] = q; }
    __syncthreads();
    if (tid < 64) {
        float t = 0.0f;
        const int h = tid & 31;
        if (h < 3) {
            const float* r = red + (tid >= 32 ? 256 : 0) + h * CELLS;
            for (int k = 0; k < CELLS; ++k) t += r[k];
        }
        ex.post(t);
    }
    return false;
}

// heads, second part (256 threads): batch-norm + ReLU of the three planes, policy FC 84 -> 42 and the landing-cell
// gather (connectfour.py:56-71), value FC 42 -> 256 -> 1 + tanh
template <class EX>
__device__ __forceinline__ bool head_fc_fwd_step(const float* x, const float* zh, float* stat_h /* LDS [96] */, float* stat_h_keep,
                                                 const float* gp, const float* btp, const float* gv, const float* btv, const float* wpf,
                                                 const float* bpf, const float* w1, const float* b1, const float* w2, const float* b2,
                                                 float* hplanes /* [board][128]: 84 policy, 42 value */, float* h1 /* [board][256] */,
                                                 float* logits7, float* vhat, int boards, float eps, const EX& ex, float* hp /* LDS [128] */,
                                                 float* cl /* LDS [42] */, int* height /* LDS [8] */, float* red /* LDS [256] */,
                                                 float* w1_l /* LDS [256][42] */, float* wpf_l /* LDS [42][84] */, bool staged,
                                                 double* tot, double* scratch) {
    const int b = blockIdx.x, tid = step_tid();
    const bool on = tid < TT;
    // the fully connected layers' weights into LDS while the batch sums are on their way (a thread of the layers below
    // reads one row of a matrix: from global memory that is a chain of uncoalesced loads); staged: already there
    if (!staged) {
        stage_floats(w1_l, w1, 256 * CELLS, tid);
        stage_floats(wpf_l, wpf, CELLS * 84, tid);
    }
    if (ex.template gather<TT / 64>(tot, scratch)) return true;
    stats_from_totals(tot, 3, (long)boards * CELLS, eps, stat_h, stat_h_keep);
    if (tid < 3 * CELLS) {
        const int cell = tid % CELLS, h = tid / CELLS;
        const float gamma = h < 2 ? gp[h] : gv[0], beta = h < 2 ? btp[h] : btv[0];
        const float z = zh[((size_t)b * CELLS + cell) * 4 + h];
        const float y = fmaxf(fmaf((z - stat_h[h]) * stat_h[32 + h], gamma, beta), 0.0f);
        hp[h * CELLS + cell] = y;
        hplanes[(size_t)b * 128 + h * CELLS + cell] = y;
    }
    if (tid >= 128 && tid < 135) {
        const int col = tid - 128;
        float n = 0.0f;
        for (int k = 0; k < 12; ++k) n += x[(size_t)b * 84 + k * 7 + col];      // both planes, six rows
        height[col] = (int)(n + 0.5f);
    }
    __syncthreads();
    if (tid < CELLS) {
        float acc = bpf[tid];
        const float* w = wpf_l + tid * 84;
#pragma unroll 4
        for (int k = 0; k < 84; ++k) acc = fmaf(w[k], hp[k], acc);
        cl[tid] = acc;
    }
    if (on) {
        float acc = b1[tid];
        const float* w = w1_l + tid * CELLS;
#pragma unroll 6
        for (int c = 0; c < CELLS; ++c) acc = fmaf(w[c], hp[84 + c], acc);
        const float hid = fmaxf(acc, 0.0f);
        h1[(size_t)b * 256 + tid] = hid;
        red[tid] = hid * w2[tid];
    }
    __syncthreads();
    if (tid < 7) logits7[(size_t)b * 7 + tid] = height[tid] < 6 ? cl[height[tid] * 7 + tid] : 0.0f;
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) vhat[b] = tanhf(red[0] + b2[0]);
    return false;
}

// ---------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------
// heads (256 threads): from d loss / d logits7 and d loss / d v to the gradient at the three head planes' batch-norm
// outputs (ReLU mask applied) -> dyh [board][cell][4]; keeps dh1 [board][256], dpre2 [board], dcl [board][42] for the
// weight gradients; posts the partial sums of the head batch-norms' backward
template <class EX>
__device__ __forceinline__ void head_fc_bwd_step(const float* x, const float* dlogits7, const float* dv, const float* vhat,
                                                 const float* h1, const float* hplanes, const float* zh, const float* stat_h,
                                                 const float* wpf, const float* w1, const float* w2, float* dh1, float* dpre2, float* dcl,
                                                 float* dyh, const EX& ex, float* d1 /* LDS [256] */, float* dcl_s /* LDS [42] */,
                                                 int* height /* LDS [8] */, float* red /* LDS [512 + 256] */, float* w1_l /* LDS [256][42] */,
                                                 bool w1_staged /* w1_l and height are in place */) {
    const int b = blockIdx.x, tid = step_tid();
    const bool on = tid < TT;
    float* vpart = red + 512;                          // [6][42]
    if (!w1_staged) stage_floats(w1_l, w1, 256 * CELLS, tid);
    const float v = vhat[b];
    const float dp = dv[b] * (1.0f - v * v);
    if (on) {
        const float d = h1[(size_t)b * 256 + tid] > 0.0f ? dp * w2[tid] : 0.0f;
        d1[tid] = d;
        dh1[(size_t)b * 256 + tid] = d;
    }
    if (tid == 0) dpre2[b] = dp;
    if (tid < CELLS) dcl_s[tid] = 0.0f;
    if (!w1_staged && tid >= 128 && tid < 135) {       // (staged: the forward part of this step has left the heights too)
        const int col = tid - 128;
        float n = 0.0f;
        for (int k = 0; k < 12; ++k) n += x[(size_t)b * 84 + k * 7 + col];
        height[col] = (int)(n + 0.5f);
    }
    __syncthreads();
    if (tid < 7 && height[tid] < 6) dcl_s[height[tid] * 7 + tid] = dlogits7[(size_t)b * 7 + tid];
    // the value plane's gradient, sum over the 256 hidden units of d1[j] * w1[j][cell]: six threads per cell, each
    // every sixth unit
    if (tid < 6 * CELLS) {
        const int cell = tid % CELLS, r = tid / CELLS;
        float d = 0.0f;
#pragma unroll 8
        for (int j = r; j < 256; j += 6) d = fmaf(d1[j], w1_l[j * CELLS + cell], d);
        vpart[r * CELLS + cell] = d;
    }
    __syncthreads();
    if (tid < CELLS) dcl[(size_t)b * CELLS + tid] = dcl_s[tid];
    float s1 = 0.0f, s2 = 0.0f;
    if (tid < 3 * CELLS) {
        const int cell = tid % CELLS, h = tid / CELLS;
        float d = 0.0f;
        if (h < 2) {
            const int k = h * CELLS + cell;
            for (int col = 0; col < 7; ++col)
                if (height[col] < 6) d = fmaf(dcl_s[height[col] * 7 + col], wpf[(height[col] * 7 + col) * 84 + k], d);
        } else {
            d = ((((vpart[cell] + vpart[CELLS + cell]) + vpart[2 * CELLS + cell]) + vpart[3 * CELLS + cell]) + vpart[4 * CELLS + cell]) +
                vpart[5 * CELLS + cell];
        }
        if (!(hplanes[(size_t)b * 128 + h * CELLS + cell] > 0.0f)) d = 0.0f;
        dyh[((size_t)b * CELLS + cell) * 4 + h] = d;
        const float xh = (zh[((size_t)b * CELLS + cell) * 4 + h] - stat_h[h]) * stat_h[32 + h];
        s1 = d; s2 = d * xh;
    }
    if (on) { red[tid] = s1; red[256 + tid] = s2; }
    __syncthreads();
    if (tid < 64) {
        float t = 0.0f;
        const int h = tid & 31;
        if (h < 3) {
            const float* r = red + (tid >= 32 ? 256 : 0) + h * CELLS;
            for (int k = 0; k < CELLS; ++k) t += r[k];
        }
        ex.post(t);
    }
}

// Sum over the batch by 256 threads (t = 0..255) for 64 outputs: thread (output = t & 63, part = t >> 6) adds the
// boards part, part + 4, ... in that order, the four parts are added in order.  A board's term is fa(board) *
// fb(board); the factors of 32 boards are requested before the first is used (one trip to memory for 128 boards).
// One __syncthreads per call.
template <class FA, class FB>
__device__ __forceinline__ float batch_sum(int boards, int t, float* red /* LDS [256] */, FA fa, FB fb) {
    const int q = t >> 6;
    float s = 0.0f;
    for (int b0 = q; b0 < boards; b0 += 128) {
        float va[32], vb[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const bool in = b0 + 4 * j < boards;
            va[j] = in ? fa(b0 + 4 * j) : 0.0f;
            vb[j] = in ? fb(b0 + 4 * j) : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) s += va[j] * vb[j];
    }
    red[t] = s;
    __syncthreads();
    return red[t & 63] + red[64 + (t & 63)] + red[128 + (t & 63)] + red[192 + (t & 63)];
}
struct OneF { __device__ __forceinline__ float operator()(int) const { return 1.0f; } };

// weight gradients of the fully connected layers: sums over the batch, 64 weights per group of 256 threads
// (each range below is a multiple of 64 long or the last one, so a group never straddles two of them)
#define FC_WGRAD_GROUPS (256 * CELLS / 64 + 4 + 4 + 1 + 56 + 1)
__device__ __forceinline__ void fc_wgrad_group(int group, int t, float* red, const float* dh1, const float* dpre2, const float* dcl,
                                               const float* h1, const float* hplanes, int boards, float* g_w1, float* g_b1, float* g_w2,
                                               float* g_b2, float* g_wpf, float* g_bpf) {
    int i = group * 64 + (t & 63);
    const bool writer = t < 64;
    if (i < 256 * CELLS) {
        const int j = i / CELLS, c = i % CELLS;
        const float s = batch_sum(boards, t, red, [&](int b) { return dh1[(size_t)b * 256 + j]; }, [&](int b) { return hplanes[(size_t)b * 128 + 84 + c]; });
        if (writer) g_w1[i] = s;
        return;
    }
    i -= 256 * CELLS;
    if (i < 256) {
        const float s = batch_sum(boards, t, red, [&](int b) { return dh1[(size_t)b * 256 + i]; }, OneF());
        if (writer) g_b1[i] = s;
        return;
    }
    i -= 256;
    if (i < 256) {
        const float s = batch_sum(boards, t, red, [&](int b) { return dpre2[b]; }, [&](int b) { return h1[(size_t)b * 256 + i]; });
        if (writer) g_w2[i] = s;
        return;
    }
    i -= 256;
    if (i < 64) {                                          // one output: the value head's last bias
        const float s = batch_sum(boards, t, red, [&](int b) { return i == 0 ? dpre2[b] : 0.0f; }, OneF());
        if (writer && i == 0) g_b2[0] = s;
        return;
    }
    i -= 64;
    if (i < 56 * 64) {                                     // 3528 weights in 56 groups (the last one partly idle)
        const bool in = i < CELLS * 84;
        const int cell = in ? i / 84 : 0, k = in ? i % 84 : 0;
        const float s = batch_sum(boards, t, red, [&](int b) { return dcl[(size_t)b * CELLS + cell]; }, [&](int b) { return hplanes[(size_t)b * 128 + k]; });
        if (writer && in) g_wpf[i] = s;
        return;
    }
    i -= 56 * 64;
    {
        const bool in = i >= 0 && i < CELLS;
        const int cell = in ? i : 0;
        const float s = batch_sum(boards, t, red, [&](int b) { return dcl[(size_t)b * CELLS + cell]; }, OneF());
        if (writer && in) g_bpf[cell] = s;
    }
}

// batch-norm backward coefficients of a layer from its saved statistics and the batch sums of its incoming gradient
// (tot: sum dy, sum dy * xhat): dz = k1 * (dy - m1 - xhat * m2), xhat = (z - mean) * invstd
struct BnBack { const float* stat; const float* gamma; };
__device__ __forceinline__ void bn_back_coeff(const BnBack& bb, const double* tot, int nch, float inv_n, float* k1, float* m1,
                                              float* m2, float* mean, float* invstd) {
    if ((int)threadIdx.x < nch) {
        const int c = threadIdx.x;
        mean[c] = bb.stat[c];
        invstd[c] = bb.stat[32 + c];
        k1[c] = bb.gamma[c] * bb.stat[32 + c];
        m1[c] = (float)tot[c] * inv_n;
        m2[c] = (float)tot[32 + c] * inv_n;
    }
}
// the same sums are that batch-norm's own gradients (d beta = sum dy, d gamma = sum dy * xhat): workgroup 0 adds them
// to the batch-norm's accumulators (the shared block's batch-norms are applied num_blocks times)
__device__ __forceinline__ void accumulate_bn_grad(const double* tot, float* bn_acc /* [64] */) {
    if (blockIdx.x == 0 && threadIdx.x < 64) bn_acc[threadIdx.x] += (float)tot[threadIdx.x];
}

// heads, last part of the backward (256 threads): dzh from dyh, the gradient at the tower's output X (three planes
// back through the 1x1 convs), its ReLU mask -> dU [board][cell][32], partial sums of the last tower batch-norm's
// backward (x-hat from zprev), and the board's share of the 1x1 convs' weight and bias gradients -> pwh [board][128]
template <class EX>
__device__ __forceinline__ bool head_conv_bwd_step(const float* dyh, const float* zh, const float* stat_h, const float* gp, const float* gv,
                                                   float* g_gp, float* g_btp, float* g_gv, float* g_btv, const float* wp, const float* wv,
                                                   const float* xtop, const float* zprev, const float* stat_prev, float* du, float* pwh,
                                                   int boards, const EX& ex, float* dz /* LDS [168] */, float* wh /* LDS [96] */,
                                                   float* red /* LDS [512] */, double* tot, double* scratch) {
    const int b = blockIdx.x, tid = step_tid(), ci = tid & 31, g = tid >> 5;
    const bool on = tid < TT;
    const float inv_n = 1.0f / (float)((long)boards * CELLS);
    if (tid < 2 * TC) wh[tid] = wp[tid];
    if (tid < TC) wh[2 * TC + tid] = wv[tid];
    if (ex.template gather<TT / 64>(tot, scratch)) return true;
    if (b == 0 && tid < 3) {
        // the head batch-norms' own gradients: d beta = sum dy, d gamma = sum dy * x-hat
        if (tid < 2) { g_btp[tid] = (float)tot[tid]; g_gp[tid] = (float)tot[32 + tid]; }
        else { g_btv[0] = (float)tot[2]; g_gv[0] = (float)tot[34]; }
    }
    if (tid < 3 * CELLS) {
        const int cell = tid % CELLS, h = tid / CELLS;
        const float gamma = h < 2 ? gp[h] : gv[0];
        const float xh = (zh[((size_t)b * CELLS + cell) * 4 + h] - stat_h[h]) * stat_h[32 + h];
        dz[cell * 4 + h] = gamma * stat_h[32 + h] *
                           (dyh[((size_t)b * CELLS + cell) * 4 + h] - (float)tot[h] * inv_n - xh * ((float)tot[32 + h] * inv_n));
    }
    __syncthreads();
    float s1 = 0.0f, s2 = 0.0f;
    if (on) {
#pragma unroll
        for (int k = 0; k < NPOS; ++k) {
            const int cell = g + 8 * k;
            if (cell >= CELLS) break;
            const size_t gi = ((size_t)b * CELLS + cell) * TC + ci;
            float d = dz[cell * 4] * wh[ci];
            d = fmaf(dz[cell * 4 + 1], wh[TC + ci], d);
            d = fmaf(dz[cell * 4 + 2], wh[2 * TC + ci], d);
            if (!(xtop[gi] > 0.0f)) d = 0.0f;
            du[gi] = d;
            const float xh = (zprev[gi] - stat_prev[ci]) * stat_prev[32 + ci];
            s1 += d; s2 = fmaf(d, xh, s2);
        }
    }
    // the board's share of the head convs' weight gradients: thread (plane h, channel ci), bias gradients beside them
    if (tid < 3 * TC) {
        const int h = tid >> 5;
        float w = 0.0f, bsum = 0.0f;
        for (int cell = 0; cell < CELLS; ++cell) {
            w = fmaf(dz[cell * 4 + h], xtop[((size_t)b * CELLS + cell) * TC + ci], w);
            bsum += dz[cell * 4 + h];
        }
        pwh[(size_t)b * 128 + tid] = w;
        if (ci == 0) pwh[(size_t)b * 128 + 96 + h] = bsum;
    }
    board_partials<TT / 32>(s1, s2, red, ex);
    return false;
}

// Backward of one 3x3 conv layer of the tower (512 threads), one board per workgroup.
//   dyout, zout, ob   gradient at this layer's batch-norm output (ReLU mask applied), its conv output, its batch-norm;
//                     the boards' sums of dyout, dyout * xhat are the rows gathered; bn_acc_out: that batch-norm's
//                     gradient accumulators
//   A (this conv's input): MODE_T: ReLU(BN_in(zin)) with BN_in's saved statistics, recomputed; else the stored xin
//   wb                this conv's weights as [tap][co][ci]
//   res               (RES) added to the input gradient before the mask: the gradient that reaches the block input
//                     through the residual connection
//   dyprev            out: (input gradient (+ res)) masked by A > 0 = gradient at the previous batch-norm's output
//   zprev, stat_prev  that previous batch-norm's input and statistics (MODE_T: zin itself), for its x-hat sums
//   pw                the board's share of this conv's weight gradient [tap][ci][co]; accum: added to what an earlier
//                     application of the shared block left there
// The input gradient comes first and the row for the next step is posted as soon as it is complete; the weight
// gradient, which nothing in the next step needs, follows.
template <bool MODE_T, bool RES, class EX>
__device__ __forceinline__ bool conv_bwd_step(bool accum, const float* dyout, const float* zout, BnBack ob, float* bn_acc_out,
                                              const float* zin, const float* stat_in, const float* gamma_in, const float* beta_in,
                                              const float* xin, const float* wb, const float* res /* may be dyprev */, float* dyprev,
                                              const float* zprev, const float* stat_prev, float* pw, int boards, const EX& ex, float* lds,
                                              double* tot, double* scratch) {
    float* wl = lds + CONV_LDS_WL;                     // [9][32 co][32 ci]
    float* dzt = lds + CONV_LDS_DZT;
    float* at = lds + CONV_LDS_AT;
    float* sc = lds + CONV_LDS_COEF;
    float* sh = sc + TC;
    float* k1 = sh + TC;
    float* m1 = k1 + TC;
    float* m2 = m1 + TC;
    float* mean = m2 + TC;
    float* invstd = mean + TC;
    float* red = lds + CONV_LDS_RED;
    float* pbuf = lds + CONV_LDS_PBUF;
    const int b = blockIdx.x, tid = step_tid(), lane32 = tid & 31, g = tid >> 5;
    const float inv_n = 1.0f / (float)((long)boards * CELLS);
    const size_t base = (size_t)b * CELLS * TC;
    // every global load of the step, before the first wait (element tid + 512 j = (cell g + 16 j, channel lane32):
    // the cells this thread also finishes the input gradient of)
    float dyr[EPT], zor[EPT], ar[EPT], rr[EPT], zpr[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = tid + j * TTC;
        const bool in = i < CELLS * TC;
        dyr[j] = in ? dyout[base + i] : 0.0f;
        zor[j] = in ? zout[base + i] : 0.0f;
        ar[j] = in ? (MODE_T ? zin[base + i] : xin[base + i]) : 0.0f;
        rr[j] = (RES && in) ? res[base + i] : 0.0f;
        zpr[j] = in ? (MODE_T ? ar[j] : zprev[base + i]) : 0.0f;       // MODE_T: the previous batch-norm's input is zin
    }
    const WRegs wr = load_conv_weights(wb);
    const WgOld wold = wgrad_load_old(accum, pw + (size_t)b * 9 * TC * TC, tid >> 6, tid & 63);
    const float sp_mean = stat_prev[lane32], sp_inv = stat_prev[32 + lane32];
    for (int i = tid; i < 2 * PADCELLS * TS; i += TTC) dzt[i] = 0.0f;         // both tiles
    store_conv_weights(wl, wr);                        // (while the rows are on their way)
    if (MODE_T) bn_scale_shift(stat_in, gamma_in, beta_in, sc, sh);
    if (ex.template gather<TTC / 64>(tot, scratch)) return true;
    STEP_MARK(ex, 0);
    bn_back_coeff(ob, tot, TC, inv_n, k1, m1, m2, mean, invstd);
    accumulate_bn_grad(tot, bn_acc_out);
    __syncthreads();
    float a_own[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = tid + j * TTC;
        a_own[j] = 0.0f;
        if (i < CELLS * TC) {
            const int pi = pad_index(i >> 5);
            const float xh = (zor[j] - mean[lane32]) * invstd[lane32];
            dzt[pi * TS + lane32] = k1[lane32] * (dyr[j] - m1[lane32] - xh * m2[lane32]);
            a_own[j] = MODE_T ? fmaxf(fmaf(ar[j], sc[lane32], sh[lane32]), 0.0f) : ar[j];
            at[pi * TS + lane32] = a_own[j];
        }
    }
    __syncthreads();
    STEP_MARK(ex, 1);
    // ---- input gradient on the matrix cores (into pbuf)
    conv_mfma<-1>(dzt, wl, pbuf, tid >> 6, tid & 63);
    __syncthreads();
    STEP_MARK(ex, 2);
    // ---- mask, the previous batch-norm's sums: thread (ci, three cells)
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
        const int cell = g + 16 * k;
        if (cell < CELLS) {
            float d = conv_total(pbuf, lane32, cell);
            if (RES) d += rr[k];
            if (!(a_own[k] > 0.0f)) d = 0.0f;
            dyprev[base + cell * TC + lane32] = d;
            const float xh = (zpr[k] - sp_mean) * sp_inv;
            s1 += d; s2 = fmaf(d, xh, s2);
        }
    }
    board_partials<TTC / 32>(s1, s2, red, ex);
    STEP_MARK(ex, 3);
    // ---- the board's share of the weight gradient
    wgrad_mfma(at, dzt, pw + (size_t)b * 9 * TC * TC, wold, tid >> 6, tid & 63);
    STEP_MARK(ex, 4);
    return false;
}

// stem backward (256 threads): dZ0 from the gradient at the stem batch-norm's output; the board's share of the stem
// conv's weight gradient -> pws [board][18][32] (index (tap * 2 + plane) * 32 + co)
template <class EX>
__device__ __forceinline__ bool stem_bwd_step(const float* x, const float* dy0, const float* z0, BnBack ob, float* bn_acc_out, float* pws,
                                              int boards, const EX& ex, float* xt /* LDS [144] */, float* dzt /* LDS [42][32] */,
                                              float* coef /* LDS 5 x [32] */, double* tot, double* scratch) {
    float* k1 = coef;
    float* m1 = k1 + TC;
    float* m2 = m1 + TC;
    float* mean = m2 + TC;
    float* invstd = mean + TC;
    const int b = blockIdx.x, tid = step_tid(), co = tid & 31, g = tid >> 5;
    const bool on = tid < TT;
    const float inv_n = 1.0f / (float)((long)boards * CELLS);
    // every global load of the step before the wait for the rows
    float zr[NPOS], dr[NPOS];
    float xv = 0.0f;
    if (on) {
#pragma unroll
        for (int j = 0; j < NPOS; ++j) {
            const int i = tid + j * TT;
            const size_t gi = (size_t)b * CELLS * TC + i;
            zr[j] = i < CELLS * TC ? z0[gi] : 0.0f;
            dr[j] = i < CELLS * TC ? dy0[gi] : 0.0f;
        }
        if (tid < 2 * CELLS) xv = x[(size_t)b * 84 + tid];
        for (int i = tid; i < PADCELLS * 2; i += TT) xt[i] = 0.0f;
    }
    if (ex.template gather<TT / 64>(tot, scratch)) return true;
    bn_back_coeff(ob, tot, TC, inv_n, k1, m1, m2, mean, invstd);
    accumulate_bn_grad(tot, bn_acc_out);
    __syncthreads();
    if (on) {
        if (tid < 2 * CELLS) xt[pad_index(tid % CELLS) * 2 + tid / CELLS] = xv;
#pragma unroll
        for (int j = 0; j < NPOS; ++j) {
            const int i = tid + j * TT;
            if (i < CELLS * TC) {
                const int c = i & 31;
                const float xh = (zr[j] - mean[c]) * invstd[c];
                dzt[i] = k1[c] * (dr[j] - m1[c] - xh * m2[c]);
            }
        }
    }
    __syncthreads();
    // thread (co, g): the (tap, plane) combinations q = g, g + 8, g + 16 (< 18)
    if (on)
        for (int q = g; q < 18; q += 8) {
            const int off = tap_offset(q >> 1), plane = q & 1;
            float acc = 0.0f;
            for (int cell = 0; cell < CELLS; ++cell) acc = fmaf(xt[(pad_index(cell) + off) * 2 + plane], dzt[cell * TC + co], acc);
            pws[(size_t)b * 18 * TC + q * TC + co] = acc;
        }
    return false;
}

struct RunStat { float* rmean; float* rvar; long long* nbt; };

// Last step, in groups of 256 threads for 64 outputs each: the boards' shares of the conv weight gradients summed in
// board order and written in torch's layout [co][ci][ky][kx]; the tower batch-norms' gradients from their
// accumulators; the running statistics of every batch-norm updated once per application, in order (torch: running =
// (1 - m) * running + m * batch, the variance unbiased), and the application counters.  One __syncthreads per call.
#define FINISH_GROUPS ((2 * 9 * TC * TC + 18 * TC + 128) / 64 + 1)
struct FinishArgs {
    const float* pw1; const float* pw2; const float* pws; const float* pwh;
    int boards, num_blocks;
    float* g_c1; float* g_c2; float* g_stem; float* g_wp; float* g_bp; float* g_wv;
    const float* bn_acc;
    float* g_bn0_w; float* g_bn0_b; float* g_bn1_w; float* g_bn1_b; float* g_bn2_w; float* g_bn2_b;
    const float* stats /* [2 * blocks + 1][96] */; const float* stat_h;
    RunStat r0, r1, r2, rp, rv;
    float momentum;
};
__device__ __forceinline__ void finish_group(int group, int t, float* red, const FinishArgs& f) {
    int i = group * 64 + (t & 63);
    const bool writer = t < 64;
    const int boards = f.boards, num_blocks = f.num_blocks;
    const long n = (long)boards * CELLS;
    if (i < 2 * 9 * TC * TC) {
        const bool second = i >= 9 * TC * TC;
        const int j = second ? i - 9 * TC * TC : i;                 // [tap][ci][co]
        const float* p = (second ? f.pw2 : f.pw1) + j;
        const float s = batch_sum(num_blocks > 0 ? boards : 0, t, red, [&](int b) { return p[(size_t)b * 9 * TC * TC]; }, OneF());
        const int tap = j / (TC * TC), ci = (j / TC) % TC, co = j % TC;
        if (writer) (second ? f.g_c2 : f.g_c1)[(co * TC + ci) * 9 + tap] = s;
        return;
    }
    i -= 2 * 9 * TC * TC;
    if (i < 18 * TC) {
        const float s = batch_sum(boards, t, red, [&](int b) { return f.pws[(size_t)b * 18 * TC + i]; }, OneF());
        const int tap = i / (2 * TC), plane = (i / TC) % 2, co = i % TC;
        if (writer) f.g_stem[(co * 2 + plane) * 9 + tap] = s;
        return;
    }
    i -= 18 * TC;
    if (i < 128) {
        const float s = batch_sum(boards, t, red, [&](int b) { return i < 99 ? f.pwh[(size_t)b * 128 + i] : 0.0f; }, OneF());
        if (writer) {
            if (i < 2 * TC) f.g_wp[i] = s;
            else if (i < 3 * TC) f.g_wv[i - 2 * TC] = s;
            else if (i >= 96 && i < 98) f.g_bp[i - 96] = s;
        }
        return;
    }
    i -= 128;
    __syncthreads();
    if (t < TC && i < TC) {
        const int c = i;
        const float* bn_acc = f.bn_acc;
        const float* stats = f.stats;
        const float momentum = f.momentum;
        f.g_bn0_b[c] = bn_acc[c];           f.g_bn0_w[c] = bn_acc[32 + c];
        f.g_bn1_b[c] = bn_acc[64 + c];      f.g_bn1_w[c] = bn_acc[64 + 32 + c];
        f.g_bn2_b[c] = bn_acc[128 + c];     f.g_bn2_w[c] = bn_acc[128 + 32 + c];
        const float unbias = (float)n / (float)(n - 1);
        f.r0.rmean[c] = (1.0f - momentum) * f.r0.rmean[c] + momentum * stats[c];
        f.r0.rvar[c] = (1.0f - momentum) * f.r0.rvar[c] + momentum * (stats[64 + c] * unbias);
        float m1 = f.r1.rmean[c], v1 = f.r1.rvar[c], m2 = f.r2.rmean[c], v2 = f.r2.rvar[c];
        for (int k = 0; k < num_blocks; ++k) {
            const float* s1 = stats + (size_t)(1 + 2 * k) * 96;
            const float* s2 = stats + (size_t)(2 + 2 * k) * 96;
            m1 = (1.0f - momentum) * m1 + momentum * s1[c];
            v1 = (1.0f - momentum) * v1 + momentum * (s1[64 + c] * unbias);
            m2 = (1.0f - momentum) * m2 + momentum * s2[c];
            v2 = (1.0f - momentum) * v2 + momentum * (s2[64 + c] * unbias);
        }
        f.r1.rmean[c] = m1; f.r1.rvar[c] = v1; f.r2.rmean[c] = m2; f.r2.rvar[c] = v2;
        if (c < 3) {
            const RunStat r = c < 2 ? f.rp : f.rv;
            const int k = c < 2 ? c : 0;
            r.rmean[k] = (1.0f - momentum) * r.rmean[k] + momentum * f.stat_h[c];
            r.rvar[k] = (1.0f - momentum) * r.rvar[k] + momentum * (f.stat_h[64 + c] * unbias);
        }
        if (c == 0) {
            f.r0.nbt[0] += 1; f.r1.nbt[0] += num_blocks; f.r2.nbt[0] += num_blocks; f.rp.nbt[0] += 1; f.rv.nbt[0] += 1;
        }
    }
}
