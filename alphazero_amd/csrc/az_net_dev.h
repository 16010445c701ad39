// Device-side building blocks of the fused Connect Four network (types, LDS / blob layout, stem,
// 3x3 convolution, epilogues): shared by the stand-alone network kernel (az_net.hip) and by kernels that
// run the network inside a larger launch.
//
// A wave takes two boards at a time through stem, tower and the 1x1 head convs with every activation in
// REGISTERS.  Contractions are v_mfma_f32_16x16x32_bf16, oriented D[out ch][pos] = sum_k Wt[out ch][k] X[k][pos]:
//   * the 16 positions of a tile are one board row of both boards: lanes p = 0..7 hold cells x = 0..7 of row y
//     of the first board, p = 8..15 those of the second (x = 7 is a pad cell that is kept at zero).  A tile's
//     B operand is ONE fragment: lane (p, q = lane / 16) holds channels 8q..8q+7 of position p -- all 32
//     channels of a tap are one MFMA k-step.  An activation tensor of the two boards is 6 such fragments
//     (24 registers).
//   * a 3x3 tap at dy = +-1 reads the neighbouring row's fragment (another register; rows -1 and 6 do not
//     exist, so those MFMAs are simply not issued), a tap at dx = +-1 reads the row's fragment shifted by one
//     lane inside its 16-lane DPP row (row_shr:1 / row_shl:1 with zero fill; the pad cell supplies the zero
//     between the two boards).  No activation is ever written to or read from LDS in the tower.
//   * output channels are split over two MFMAs per tap in the order that makes D land in B's layout: MFMA
//     "half" h computes channels 8g + 4h + i (g = 0..3, i = 0..3), so lane (p, q) receives channels 8q + 4h + i
//     -- after both halves exactly the eight channels its B fragment holds.  The epilogue (residual, ReLU,
//     bf16) is lane-local: no transposition, no cross-lane traffic.
//   * weights are the A operand: 18 fragments per conv (9 taps x 2 halves), kept in LDS (36 KB for both tower
//     convs, filled once per launch) and streamed one ds_read_b128 per fragment; a fragment feeds up to six
//     MFMAs (one per row).  The twelve accumulators of a conv (6 rows x 2 halves) are independent, so
//     consecutive MFMAs never wait for each other.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/azhip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define NTHREADS 512
#define ROWS 6                    // board rows = tiles of a two-board task

// LDS: tower weights, then scratch
#define W_BYTES (2 * 18 * 1024)               // [2 convs][9 taps][2 halves][64 lanes][8] bf16
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
#define S_HG (S_WT + 64)                      // int [32] leaf index of each head column
#define S_BYTES (S_HG + 128)
#define LDS_BYTES (W_BYTES + S_BYTES)

// blob layout (bytes); must match alphazero_amd/net.py
#define OFF_SF 0                                   // stem      [2 halves][64][8] bf16 (16x16x32 A fragments)
#define OFF_WF (OFF_SF + 2 * 1024)                 // c1, c2    [2][9 taps][2 halves][64][8] bf16
#define OFF_HF (OFF_WF + W_BYTES)                  // 1x1 heads [64][8] bf16 (rows 0,1 policy planes, 2 value plane)
#define OFF_F1 (OFF_HF + 1024)                     // value FC1 [8][3][64][8] bf16 (32x32x16 A fragments)
#define OFF_PF (OFF_F1 + 8 * 3 * 1024)             // policy FC [2][6][64][8] bf16
#define OFF_BIAS (OFF_PF + 2 * 6 * 1024)           // float [3][32] stem, c1, c2 ; then [4] heads
#define OFF_BP (OFF_BIAS + 100 * 4)                // float [64] policy FC bias
#define OFF_B1 (OFF_BP + 64 * 4)                   // float [256]
#define OFF_W2 (OFF_B1 + 256 * 4)                  // float [256]
#define OFF_B2 (OFF_W2 + 256 * 4)                  // float [4]
#define BLOB_BYTES (OFF_B2 + 16)

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

__device__ __forceinline__ const bf16x8* frag_ptr(const uint8_t* blob, int off, int lane) {
    return reinterpret_cast<const bf16x8*>(blob + off) + lane;
}

// the tower weights go from the blob to LDS once per launch; every thread of the workgroup takes part
__device__ __forceinline__ void load_tower_weights(const uint8_t* blob, uint8_t* wlds, int tid) {
    const uint4* src = reinterpret_cast<const uint4*>(blob + OFF_WF);
    uint4* dst = reinterpret_cast<uint4*>(wlds);
    for (int i = tid; i < W_BYTES / 16; i += NTHREADS) dst[i] = src[i];
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

// activations of a two-board task: fragment y = row y of both boards, lane (p, q) = channels 8q..8q+7 of cell p
struct Act { bf16x8 t[ROWS]; };

// the row's fragment as seen one cell to the left (LEFT: lane p reads lane p - 1) or to the right, zero at
// the ends of the 16-lane row
template <bool LEFT>
__device__ __forceinline__ bf16x8 shift_cell(const bf16x8& v) {
    constexpr int CTRL = LEFT ? 0x111 : 0x101;               // row_shr:1 / row_shl:1
    const uint4 u = __builtin_bit_cast(uint4, v);
    uint4 r;
    r.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u.x, CTRL, 0xF, 0xF, true);
    r.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u.y, CTRL, 0xF, 0xF, true);
    r.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u.z, CTRL, 0xF, 0xF, true);
    r.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u.w, CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(bf16x8, r);
}

// (+ residual) -> ReLU -> bf16, the two accumulator halves of one row -> its fragment; `keep` is all ones on
// cell lanes and zero on the pad lanes
template <bool RES>
__device__ __forceinline__ bf16x8 finish_row(f32x4 c0, f32x4 c1, const bf16x8& res, uint32_t keep) {
    if (RES) {
        const uint4 r = __builtin_bit_cast(uint4, res);
        c0[0] += __uint_as_float(r.x << 16); c0[1] += __uint_as_float(r.x & 0xFFFF0000u);
        c0[2] += __uint_as_float(r.y << 16); c0[3] += __uint_as_float(r.y & 0xFFFF0000u);
        c1[0] += __uint_as_float(r.z << 16); c1[1] += __uint_as_float(r.z & 0xFFFF0000u);
        c1[2] += __uint_as_float(r.w << 16); c1[3] += __uint_as_float(r.w & 0xFFFF0000u);
    }
    const uint4 o = make_uint4(relu_bf16x2(c0[0], c0[1]) & keep, relu_bf16x2(c0[2], c0[3]) & keep,
                               relu_bf16x2(c1[0], c1[1]) & keep, relu_bf16x2(c1[2], c1[3]) & keep);
    return __builtin_bit_cast(bf16x8, o);
}

// One 3x3 conv of the task: in -> out (+ residual taken from out's old contents when RES; in and out are
// different tensors).  wl = this conv's 18 weight fragments in LDS, bias = its 32 floats in LDS.
// Taps are visited column by column of the window (centre column first, so the first MFMA of every
// accumulator is the centre tap, whose C operand is the bias); the shifted fragments of a window column are
// formed once and serve its three taps.
template <bool RES>
__device__ __forceinline__ void conv3x3(const uint8_t* wl, const float* bias, const Act& in, Act& out, int lane,
                                        uint32_t keep) {
    const int q = lane >> 4;
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + 8 * q);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + 8 * q + 4);
    const bf16x8* wf = reinterpret_cast<const bf16x8*>(wl) + lane;
    f32x4 acc[ROWS][2];
#pragma unroll
    for (int kxi = 0; kxi < 3; ++kxi) {
        const int kx = kxi == 0 ? 1 : (kxi == 1 ? 0 : 2);
        // nothing of the next window column (its six weight fragments, its shifted rows) is started before
        // this one is done: hoisted, they would not fit in the register file
        if (kxi) __builtin_amdgcn_sched_barrier(0);
        bf16x8 s[ROWS];
#pragma unroll
        for (int y = 0; y < ROWS; ++y)
            s[y] = kx == 1 ? in.t[y] : (kx == 0 ? shift_cell<true>(in.t[y]) : shift_cell<false>(in.t[y]));
#pragma unroll
        for (int kyi = 0; kyi < 3; ++kyi) {
            const int ky = kyi == 0 ? 1 : (kyi == 1 ? 0 : 2);
            const int tap = ky * 3 + kx;
            const bf16x8 a0 = wf[(tap * 2) * 64], a1 = wf[(tap * 2 + 1) * 64];
#pragma unroll
            for (int y = 0; y < ROWS; ++y) {
                const int yy = y + ky - 1;                   // input row of this tap
                if (yy < 0 || yy >= ROWS) continue;
                if (kxi == 0 && kyi == 0) {
                    acc[y][0] = MFMA16(a0, s[yy], b0);
                    acc[y][1] = MFMA16(a1, s[yy], b1);
                } else {
                    acc[y][0] = MFMA16(a0, s[yy], acc[y][0]);
                    acc[y][1] = MFMA16(a1, s[yy], acc[y][1]);
                }
            }
        }
    }
#pragma unroll
    for (int y = 0; y < ROWS; ++y) out.t[y] = finish_row<RES>(acc[y][0], acc[y][1], out.t[y], keep);
}

// stem conv 2 -> C straight from the bitboards: the B fragment of row y holds k = tap*2 + plane (18 of 32
// used) as bf16 1.0 / 0.0, built per lane from the 3x3 window around its cell
__device__ __forceinline__ void stem(const uint8_t* blob, const uint64_t* leaf_bb, const int* games, Act& X,
                                     const float* bias, int lane, uint32_t keep) {
    const int p = lane & 15, q = lane >> 4;
    const int x = (p & 7) == 7 ? 0 : (p & 7);                    // pad lanes compute a cell's window, then drop it
    const bf16x8 a0 = frag_ptr(blob, OFF_SF, lane)[0], a1 = frag_ptr(blob, OFF_SF, lane)[64];
    const int g_first = games[0], g_second = games[1];
    const int g = (p >> 3) ? g_second : g_first;
    uint64_t bb[2] = {0, 0};
    if (g >= 0) {
        bb[0] = leaf_bb[2 * (size_t)g] << 8;
        bb[1] = leaf_bb[2 * (size_t)g + 1] << 8;
    }
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + 8 * q);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + 8 * q + 4);
#pragma unroll
    for (int y = 0; y < ROWS; ++y) {
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
        // this lane's eight k are bits 8q..8q+7 (nothing beyond bit 17): two per dword
        const uint32_t k8 = (m18 >> (8 * q)) & 0xFFu;
        uint32_t d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t t2 = (k8 >> (2 * i)) & 3u;
            d[i] = (t2 & 1u) * 0x3F80u + (t2 >> 1) * 0x3F800000u;
        }
        const bf16x8 f = __builtin_bit_cast(bf16x8, make_uint4(d[0], d[1], d[2], d[3]));
        const f32x4 c0 = MFMA16(a0, f, b0), c1 = MFMA16(a1, f, b1);
        X.t[y] = finish_row<false>(c0, c1, f, keep);
    }
}

// The network for the leaves listed in the workgroup's task list (scr + S_TL: per wave MAXIT tasks of
// two entries, entry = leaf index + 1, 0 = none), `nrounds` rounds; the whole workgroup calls it
// (it contains workgroup barriers).  Writes p_out[leaf][7] and v_out[leaf].
__device__ __forceinline__ void net_rounds(const uint8_t* blob, int num_blocks, const uint64_t* leaf_bb,
                                           float* p_out, float* v_out, const uint8_t* wlds, uint8_t* scr,
                                           int nrounds, int tid, int wave, int lane) {
    float* sbias = reinterpret_cast<float*>(scr + S_BIAS);
    int* tlist = reinterpret_cast<int*>(scr + S_TL);
    // ---- Rounds.  In a round every wave takes up to two tasks (two boards each) through stem, tower
    // and the 1x1 head convs on its own -- activations in registers, no workgroup barrier --
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
    const uint32_t keep = (lane & 7) == 7 ? 0u : 0xFFFFFFFFu;      // pad cell of each board row stays zero
    for (int round = 0; round < nrounds; ++round) {
        int hg0 = -1, hg1 = -1, hg2 = -1, hg3 = -1;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int iter = 2 * round + half;
            const int games[2] = {mylist[2 * iter] - 1, mylist[2 * iter + 1] - 1};
            if (games[0] < 0) break;
            if (half) { hg2 = games[0]; hg3 = games[1]; } else { hg0 = games[0]; hg1 = games[1]; }
            const uint8_t* sblob = blob;
            asm volatile("" : "+s"(sblob));
            Act X, Y;
            stem(sblob, leaf_bb, games, X, sbias, lane, keep);
#pragma unroll 1
            for (int blk = 0; blk < num_blocks; ++blk) {
                conv3x3<false>(wlds, sbias + 32, X, Y, lane, keep);
                conv3x3<true>(wlds + W_BYTES / 2, sbias + 64, Y, X, lane, keep);
            }
            // 1x1 head convs as one MFMA per row: rows 0,1 of D = policy planes, row 2 = value plane, which
            // land in the lanes q = 0 (registers 0..2) of each position
            const uint8_t* hblob = blob;
            asm volatile("" : "+s"(hblob));
            const bf16x8 ah = frag_ptr(hblob, OFF_HF, lane)[0];
            const int p = lane & 15, x = p & 7, bl = (p >> 3) + 2 * half;
#pragma unroll
            for (int y = 0; y < ROWS; ++y) {
                const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
                const f32x4 d = MFMA16(ah, X.t[y], z4);
                if (lane < 16 && x != 7) {
                    const int cell = y * 7 + x;
                    hp[bl * 96 + cell] = (__bf16)fmaxf(d[0] + sbias[96], 0.0f);
                    hp[bl * 96 + 42 + cell] = (__bf16)fmaxf(d[1] + sbias[97], 0.0f);
                    hv[bl * 48 + cell] = (__bf16)fmaxf(d[2] + sbias[98], 0.0f);
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
