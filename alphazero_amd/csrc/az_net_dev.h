// Device-side building blocks of the fused Connect Four network (types, LDS / blob layout, stem,
// 3x3 convolution, epilogues): shared by the stand-alone network kernel (az_net.hip) and by kernels that
// run the network inside a larger launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/azhip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define TB 16                     // boards in the LDS images: two per wave
#define PSTRIDE 80                // bytes per position
#define BOARD_POS 56              // positions per board (zero row + 6 rows of 8)
#define POS0 9                    // guard + zero row
#define BOARD_PAD 128             // bytes skipped before every board, see board_off
#define NPOS_TOTAL (1 + TB * BOARD_POS + 8 + 1)
#define ACT_BYTES (((NPOS_TOTAL * PSTRIDE + TB * BOARD_PAD) + 255) / 256 * 256)
#define NTHREADS 512
#define TILES_PER_WAVE 3          // TB*48/32 = 24 tiles of 32 positions over 8 waves

// LDS scratch after the two activation images
#define S_BIAS 0                              // float[3][32] conv biases + head bias [4]
#define S_B1 512                              // float[256] value FC1 bias
#define S_W2 (S_B1 + 1024)                    // float[256] value FC2 weights
#define S_BP (S_W2 + 1024)                    // float[64]  policy FC bias
#define S_PW (S_BP + 256)                     // per-wave head scratch, PW_BYTES each:
#define PW_HP 0                               //   bf16 [4][96] policy planes, k = ch*42 + cell
#define PW_HV 768                             //   bf16 [4][48] value plane
#define PW_BYTES 1152                         // all 8 together are reused as float [32][64] logits + [8][32] value sums
#define MAXIT 8                               // wave-tasks per wave per launch (n <= 32768)
#define S_TL (S_PW + 8 * PW_BYTES)            // int [8 waves][MAXIT][2] leaf index of each board
#define S_WT (S_TL + 8 * MAXIT * 2 * 4)       // int [16] scan scratch
#define S_DUMP (S_WT + 64)                    // 64 B per wave: landing place for results of zero cells
#define S_HG (S_DUMP + 8 * 64)                // int [32] leaf index of each head column
#define S_BYTES (S_HG + 128)
#define LDS_BYTES (2 * ACT_BYTES + S_BYTES)

// blob layout (bytes); must match alphazero_amd/net.py
#define OFF_SF 0                                   // stem      [2][64][8] bf16
#define OFF_WF (OFF_SF + 2 * 1024)                 // c1, c2    [2][18][64][8] bf16
#define OFF_HF (OFF_WF + 2 * 18 * 1024)            // 1x1 heads [2][64][8] bf16
#define OFF_F1 (OFF_HF + 2 * 1024)                 // value FC1 [8][3][64][8] bf16
#define OFF_PF (OFF_F1 + 8 * 3 * 1024)             // policy FC [2][6][64][8] bf16
#define OFF_BIAS (OFF_PF + 2 * 6 * 1024)           // float [3][32] stem, c1, c2 ; then [4] heads
#define OFF_BP (OFF_BIAS + 100 * 4)                // float [64] policy FC bias
#define OFF_B1 (OFF_BP + 64 * 4)                   // float [256]
#define OFF_W2 (OFF_B1 + 256 * 4)                  // float [256]
#define OFF_B2 (OFF_W2 + 256 * 4)                  // float [4]
#define BLOB_BYTES (OFF_B2 + 16)

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// Byte offset of cell-grid position o (0..47) of board b in an activation image.  A tile of 32
// positions can straddle two boards; going from one board to the next skips 8 positions (the zero
// row), i.e. 40 slots of 16 B = 8 mod 16, which would put the two halves of a ds_read_b128 lane
// group on the same slots.  128 bytes (8 slots) of never-written padding per board make the step a
// multiple of 16 slots again; the padding extends the zero row, so taps across it still read zeros.
__device__ __forceinline__ int board_off(int b, int o) { return (b * BOARD_POS + POS0 + o) * PSTRIDE + b * BOARD_PAD; }

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

struct Weights { bf16x8 f[18]; };

__device__ __forceinline__ const bf16x8* frag_ptr(const uint8_t* blob, int off, int lane) {
    return reinterpret_cast<const bf16x8*>(blob + off) + lane;
}

// accumulator = per-channel bias: register r holds channel (r&3) + 8*(r>>2) + 4*h
__device__ __forceinline__ f32x16 bias_acc(const float* bias, int h) {
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 8 * g + 4 * h);
        acc[4 * g] = b[0]; acc[4 * g + 1] = b[1]; acc[4 * g + 2] = b[2]; acc[4 * g + 3] = b[3];
    }
    return acc;
}

// bf16(ReLU(a)), bf16(ReLU(b)) packed in one dword: convert first, then clamp the two bit patterns
// as signed 16-bit integers (negative floats, -0 included, are negative integers) -- one
// v_cvt_pk_bf16_f32 and one v_pk_max_i16 per pair instead of two v_max_f32 per value
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t relu_bf16x2(float a, float b) {
    const f32x2 f = {a, b};
    const bf16x2 y = __builtin_convertvector(f, bf16x2);     // one v_cvt_pk_bf16_f32 for the pair
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, y), z));
}

// (+ residual from q) -> ReLU -> bf16 -> q, 4 channels per 8-byte store; q = the lane's position
// (channel 0) plus h * 8
template <bool RES>
__device__ __forceinline__ void store_tile(const f32x16& acc, uint8_t* q) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float x0 = acc[4 * g], x1 = acc[4 * g + 1], x2 = acc[4 * g + 2], x3 = acc[4 * g + 3];
        if (RES) {
            const uint2 r = *reinterpret_cast<const uint2*>(q + g * 16);
            x0 += __uint_as_float(r.x << 16); x1 += __uint_as_float(r.x & 0xFFFF0000u);
            x2 += __uint_as_float(r.y << 16); x3 += __uint_as_float(r.y & 0xFFFF0000u);
        }
        *reinterpret_cast<uint2*>(q + g * 16) = make_uint2(relu_bf16x2(x0, x1), relu_bf16x2(x2, x3));
    }
}

// fragments of one row of taps (3 taps x 2 channel halves) for a tile.  `base` points 9 positions
// before the tile's own position, so every tap is a non-negative immediate offset of the ds_read.
__device__ __forceinline__ void load_taps(bf16x8 (&f)[6], const uint8_t* base, int row) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int d = (row * 8 + i) * PSTRIDE;
        f[2 * i] = *reinterpret_cast<const bf16x8*>(base + d);
        f[2 * i + 1] = *reinterpret_cast<const bf16x8*>(base + d + 32);
    }
}

// One 3x3 conv over the wave's own two boards (3 tiles of 32 positions; 2 tiles when only the first
// board holds a leaf): src -> dst (+ residual
// read from dst when RES).  A wave's tiles only ever read positions of its own boards and the
// never-written zero rows, so the tower needs no workgroup barrier: LDS operations of one wave
// complete in issue order.  Loads run one row of taps ahead of the MFMAs that consume them.
template <bool RES>
__device__ __forceinline__ void conv3x3(const Weights& w, const uint8_t* src, uint8_t* dst, uint8_t* dump,
                                        const float* bias, int wave, int lane, bool two) {
    const int h = lane >> 5;
    int posb[TILES_PER_WAVE];
    bool keep[TILES_PER_WAVE];
    const uint8_t* base[TILES_PER_WAVE];
#pragma unroll
    for (int t = 0; t < TILES_PER_WAVE; ++t) {
        const int R = (wave * TILES_PER_WAVE + t) * 32 + (lane & 31);
        const int b = R / 48, o = R - b * 48;
        posb[t] = board_off(b, o);
        keep[t] = (o & 7) != 7;
        base[t] = src + posb[t] + (h * 16 - 9 * PSTRIDE);
    }
    // The epilogue of tile t-1 (vector ALU and LDS stores) is placed after the first MFMAs of tile t,
    // which need nothing from it, so it runs while the matrix pipe works; only the last tile's
    // epilogue stands alone.  (dst is never the source of this conv, so the early MFMAs of tile t do
    // not read what tile t-1 still has to store.)
    bf16x8 fa[6], fb[6];
    f32x16 prev;
    load_taps(fa, base[0], 0);
#pragma unroll
    for (int t = 0; t < TILES_PER_WAVE; ++t) {
        if (t == 2 && !two) break;                 // a single board ends inside the second tile
        f32x16 acc = bias_acc(bias, h);
        load_taps(fb, base[t], 1);
#pragma unroll
        for (int i = 0; i < 6; ++i) acc = MFMA(w.f[i], fa[i], acc);
        // the zero cell that closes each row is computed like a cell; its result goes to a dump row
        if (t > 0) store_tile<RES>(prev, (keep[t - 1] ? dst + posb[t - 1] : dump) + h * 8);
        load_taps(fa, base[t], 2);
#pragma unroll
        for (int i = 0; i < 6; ++i) acc = MFMA(w.f[6 + i], fb[i], acc);
        if (t + 1 < TILES_PER_WAVE) load_taps(fb, base[t + 1], 0);
#pragma unroll
        for (int i = 0; i < 6; ++i) acc = MFMA(w.f[12 + i], fa[i], acc);
        prev = acc;
        if (t + 1 < TILES_PER_WAVE) {
#pragma unroll
            for (int i = 0; i < 6; ++i) fa[i] = fb[i];
        }
    }
    const int tl = two ? 2 : 1;
    store_tile<RES>(prev, (keep[tl] ? dst + posb[tl] : dump) + h * 8);
}

// stem conv 2 -> C straight from bitboards: B[k][pos], k = tap*2 + plane (18 of 32 used)
__device__ __forceinline__ void stem(const uint8_t* blob, const uint64_t* leaf_bb, const int* games, uint8_t* dst,
                                     const float* bias, int wave, int lane, int ntiles) {
    const int h = lane >> 5;
    const bf16x8 a0 = frag_ptr(blob, OFF_SF, lane)[0], a1 = frag_ptr(blob, OFF_SF, lane)[64];
    // both boards' bitboards are requested before the first tile needs them: one memory latency per task
    uint64_t bbs[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 2; ++i)
        if (games[i] >= 0) {
            bbs[2 * i] = leaf_bb[2 * (size_t)games[i]] << 8;
            bbs[2 * i + 1] = leaf_bb[2 * (size_t)games[i] + 1] << 8;
        }
#pragma unroll 1
    for (int t = 0; t < ntiles; ++t) {
        const int R = (wave * TILES_PER_WAVE + t) * 32 + (lane & 31);
        const int b = R / 48, o = R - b * 48;
        const int y = o >> 3, x = o & 7;
        const bool second = b != 2 * wave;
        const uint64_t bb[2] = {second ? bbs[2] : bbs[0], second ? bbs[3] : bbs[1]};
        // after the shift, cell (xx, yy) with xx in -1..7, yy in -1..6 is bit (xx+1)*7 + yy + 1;
        // out-of-board neighbours fall on guard bits, which are zero.  Three rows of one window
        // column are three consecutive bits, so six 64-bit shifts fetch the whole 3x3x2 window;
        // m18 collects it as bit k = 2 * tap + plane, tap = 3 * row + column of the window.
        uint32_t m18 = 0;
#pragma unroll
        for (int plane = 0; plane < 2; ++plane)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const uint32_t c3 = (uint32_t)(bb[plane] >> ((x + tx) * 7 + y)) & 7u;       // rows ty = 0, 1, 2
                m18 |= ((c3 * 0x421u) & 0x1041u) << (2 * tx + plane);                      // to bits 0, 6, 12
            }
        // fragment s holds k = 16 s + 8 h + j, j = 0..7, as bf16 1.0 / 0.0: two per dword
        const uint32_t k8 = (m18 >> (8 * h)) & 0xFFu, k2 = h ? 0u : (m18 >> 16) & 3u;
        uint32_t d[5];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t t2 = (k8 >> (2 * i)) & 3u;
            d[i] = (t2 & 1u) * 0x3F80u + (t2 >> 1) * 0x3F800000u;
        }
        d[4] = (k2 & 1u) * 0x3F80u + (k2 >> 1) * 0x3F800000u;
        bf16x8 f[2];
        {
            const uint4 f0 = make_uint4(d[0], d[1], d[2], d[3]), f1 = make_uint4(d[4], 0u, 0u, 0u);
            f[0] = __builtin_bit_cast(bf16x8, f0);
            f[1] = __builtin_bit_cast(bf16x8, f1);
        }
        f32x16 acc = bias_acc(bias, h);
        acc = MFMA(a0, f[0], acc);
        acc = MFMA(a1, f[1], acc);
        if (x != 7) store_tile<false>(acc, dst + board_off(b, o) + h * 8);
    }
}

// The network for the leaves listed in the workgroup's task list (scr + S_TL: per wave MAXIT tasks of
// two entries, entry = leaf index + 1, 0 = none), `nrounds` rounds; the whole workgroup calls it
// (it contains workgroup barriers).  Writes p_out[leaf][7] and v_out[leaf].
__device__ __forceinline__ void net_rounds(const uint8_t* blob, int num_blocks, const uint64_t* leaf_bb,
                                           float* p_out, float* v_out, const Weights& w1, const Weights& w2,
                                           uint8_t* bufX, uint8_t* bufY, uint8_t* scr, int nrounds, int tid,
                                           int wave, int lane) {
    const int h = lane >> 5;
    float* sbias = reinterpret_cast<float*>(scr + S_BIAS);
    int* tlist = reinterpret_cast<int*>(scr + S_TL);
    // ---- Rounds.  In a round every wave takes up to two tasks (two boards each) through stem, tower
    // and the 1x1 head convs on its own -- its two boards of the LDS images, no workgroup barrier --
    // and leaves four boards of head planes in its scratch.  The fully connected heads then run once
    // for the whole workgroup: 8 waves x 4 boards are exactly the 32 columns of an MFMA, wave w owns
    // rows 32w..32w+31 of the value FC and waves 0, 1 the two row tiles of the policy FC, so each
    // head weight is fetched from L2 once per workgroup and round instead of once per wave and pass.
    uint8_t* pw = scr + S_PW + wave * PW_BYTES;
    __bf16* hp = reinterpret_cast<__bf16*>(pw + PW_HP);
    __bf16* hv = reinterpret_cast<__bf16*>(pw + PW_HV);
    const float* sb1 = reinterpret_cast<const float*>(scr + S_B1);
    const float* sw2 = reinterpret_cast<const float*>(scr + S_W2);
    const float* sbp = reinterpret_cast<const float*>(scr + S_BP);
    const int* mylist = tlist + wave * MAXIT * 2;
    int* hgl = reinterpret_cast<int*>(scr + S_HG);                 // leaf of each of the 32 head columns
    float* lg = reinterpret_cast<float*>(scr + S_PW);              // [32 boards][64] logits, over the planes
    float* vpart = reinterpret_cast<float*>(scr + S_PW + 32 * 64 * 4);   // [8 waves][32 boards] value partial sums
    uint8_t* dump = scr + S_DUMP + wave * 64;  // where results for the rows' zero cells go (never read back)
    for (int round = 0; round < nrounds; ++round) {
        int hg0 = -1, hg1 = -1, hg2 = -1, hg3 = -1;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int iter = 2 * round + half;
            const int games[2] = {mylist[2 * iter] - 1, mylist[2 * iter + 1] - 1};
            if (games[0] < 0) break;
            if (half) { hg2 = games[0]; hg3 = games[1]; } else { hg0 = games[0]; hg1 = games[1]; }
            const bool two = games[1] >= 0;
            const int ntiles = two ? TILES_PER_WAVE : 2;
            const uint8_t* sblob = blob;
            asm volatile("" : "+s"(sblob));
            stem(sblob, leaf_bb, games, bufY, sbias, wave, lane, ntiles);
            for (int blk = 0; blk < num_blocks; ++blk) {
                conv3x3<false>(w1, bufY, bufX, dump, sbias + 32, wave, lane, two);
                conv3x3<true>(w2, bufX, bufY, dump, sbias + 64, wave, lane, two);
            }
            // 1x1 head convs as one 2-step MFMA: rows 0,1 = policy planes, row 2 = value plane.  (Head
            // weights are fetched where they are used: hoisted above the loop they would be spilled,
            // the tower owns the register file; the empty asm keeps the loads below this point.)
            const uint8_t* hblob = blob;
            asm volatile("" : "+s"(hblob));
            const bf16x8 a0 = frag_ptr(hblob, OFF_HF, lane)[0], a1 = frag_ptr(hblob, OFF_HF, lane)[64];
#pragma unroll 1
            for (int t = 0; t < ntiles; ++t) {
                const int R = (wave * TILES_PER_WAVE + t) * 32 + (lane & 31);
                const int b = R / 48, o = R - b * 48;
                const uint8_t* base = bufY + board_off(b, o) + h * 16;
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
                acc = MFMA(a0, *reinterpret_cast<const bf16x8*>(base), acc);
                acc = MFMA(a1, *reinterpret_cast<const bf16x8*>(base + 32), acc);
                const int y = o >> 3, x = o & 7, bl = b - 2 * wave + 2 * half;
                if (h == 0 && x != 7) {
                    const int cell = y * 7 + x;
                    hp[bl * 96 + cell] = (__bf16)fmaxf(acc[0] + sbias[96], 0.0f);
                    hp[bl * 96 + 42 + cell] = (__bf16)fmaxf(acc[1] + sbias[97], 0.0f);
                    hv[bl * 48 + cell] = (__bf16)fmaxf(acc[2] + sbias[98], 0.0f);
                }
            }
        }
        if (lane < 4) hgl[wave * 4 + lane] = lane == 0 ? hg0 : lane == 1 ? hg1 : lane == 2 ? hg2 : hg3;
        __syncthreads();                                   // every wave's planes are in place

        const uint8_t* fblob = blob;
        asm volatile("" : "+s"(fblob));
        // Thread indices are re-derived from an opaque copy: every address of this phase is then
        // computed here, per round, instead of being kept (and spilled) across the tower.
        int ht = tid;
        asm volatile("" : "+v"(ht));
        const int hl = ht & 63, hh = hl >> 5, hw = uni(ht >> 6);
        const int c = hl & 31;                                 // head column c = board (c & 3) of wave (c >> 2)
        const uint8_t* cpl = scr + S_PW + (c >> 2) * PW_BYTES;
        const __bf16* chp = reinterpret_cast<const __bf16*>(cpl + PW_HP) + (c & 3) * 96;
        const __bf16* chv = reinterpret_cast<const __bf16*>(cpl + PW_HV) + (c & 3) * 48;
        // ---- value head: FC 42->256 (+ReLU), this wave's 32 hidden units, then their share of 256->1
        float part = 0.0f;
        {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const bf16x8 a = frag_ptr(fblob, OFF_F1 + (hw * 3 + s) * 1024, hl)[0];
                acc = MFMA(a, *reinterpret_cast<const bf16x8*>(chv + 16 * s + 8 * hh), acc);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(sb1 + 32 * hw + 8 * g + 4 * hh);
                const f32x4 wv = *reinterpret_cast<const f32x4*>(sw2 + 32 * hw + 8 * g + 4 * hh);
#pragma unroll
                for (int i = 0; i < 4; ++i) part = fmaf(fmaxf(acc[4 * g + i] + b1[i], 0.0f), wv[i], part);
            }
            part += __shfl_xor(part, 32);
        }
        // ---- policy head: FC 84->42, row tile `wave` (cells 0..31 / 32..41) on waves 0 and 1
        f32x16 pacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) pacc[i] = 0.0f;
        if (hw < 2) {
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const bf16x8 a = frag_ptr(fblob, OFF_PF + (hw * 6 + s) * 1024, hl)[0];
                pacc = MFMA(a, *reinterpret_cast<const bf16x8*>(chp + 16 * s + 8 * hh), pacc);
            }
        }
        __syncthreads();                                   // all planes consumed: logits may overwrite them
        if (hl < 32) vpart[hw * 32 + c] = part;
        if (hw < 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bp = *reinterpret_cast<const f32x4*>(sbp + 32 * hw + 8 * g + 4 * hh);
                f32x4 o4;
#pragma unroll
                for (int i = 0; i < 4; ++i) o4[i] = pacc[4 * g + i] + bp[i];
                *reinterpret_cast<f32x4*>(lg + c * 64 + 32 * hw + 8 * g + 4 * hh) = o4;
            }
        }
        __syncthreads();
        if (ht < 256) {
            // landing-cell gather (connectfour.py:56-71) + soft-max over legal columns (mcts.py:106)
            const int b = ht >> 3, col = ht & 7;
            const int gidx = hgl[b];
            uint64_t occ = 0;
            if (gidx >= 0) occ = leaf_bb[2 * (size_t)gidx] | leaf_bb[2 * (size_t)gidx + 1];
            const int height = col < 7 ? __popcll((occ >> (7 * col)) & 0x3Full) : 6;
            const bool legal = col < 7 && height < 6;
            const float logit = legal ? lg[b * 64 + height * 7 + col] : -INFINITY;
            float m = logit;
            m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
            const float e = legal ? __expf(logit - m) : 0.0f;
            float sum = e;
            sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
            if (col < 7 && gidx >= 0) p_out[(size_t)gidx * 7 + col] = legal ? e / sum : 0.0f;
        } else if (ht < 256 + 32) {
            const int b = ht - 256;
            const int gidx = hgl[b];
            float sv = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) sv += vpart[w * 32 + b];
            if (gidx >= 0) v_out[gidx] = tanhf(sv + reinterpret_cast<const float*>(fblob + OFF_B2)[0]);
        }
        if (round + 1 < nrounds) {
            // the planes (and their k-padding) must read as zero again before the next round fills them
            __syncthreads();
            for (int i = tid; i < 8 * PW_BYTES / 16; i += NTHREADS)
                reinterpret_cast<uint4*>(scr + S_PW)[i] = make_uint4(0, 0, 0, 0);
            __syncthreads();
        }
    }
}
