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

// LDS: every MFMA weight fragment of the blob (its first OFF_BIAS bytes, same offsets), then scratch
#define W_BYTES (2 * 18 * 1024 + 2 * 1024)    // tower: [2 convs][9 taps][2 halves][64 lanes][8] bf16, then the identity
                                              // "tap" [2 halves][64][8] that adds the residual on the matrix cores
#define S_BIAS 0                              // float[3][32] conv biases, head biases [3], value FC2 bias [1]
#define S_B1 512                              // float[256] value FC1 bias
#define S_W2 (S_B1 + 1024)                    // float[256] value FC2 weights
#define S_BP (S_W2 + 1024)                    // float[64]  policy FC bias
#define S_PW (S_BP + 256)                     // per-wave head scratch, PW_BYTES each:
#define PW_HP 0                               //   bf16 [4][96] policy planes, k = ch*42 + cell
#define PW_HV 768                             //   bf16 [4][48] value plane
#define PW_BYTES 1152                         // (the k-padding of the planes is zeroed once per launch and never written)
#define MAXIT 8                               // wave-tasks per wave per launch (n <= 32768)
#define S_TL (S_PW + 8 * PW_BYTES)            // int [8 waves][MAXIT][2] leaf index of each board
#define S_WT (S_TL + 8 * MAXIT * 2 * 4)       // int [16] scan scratch
#define S_HG (S_WT + 64)                      // int [32] leaf index of each head column
#define S_LBB (S_HG + 128)                    // uint64 [64][2] leaf positions of the workgroup's own games (kernels
                                              // that produce the leaves themselves put them here)
#define S_LG (S_LBB + 1024)                   // float [32 boards][64] policy logits, then [8 waves][32] value partial sums
#define S_END (S_LG + 32 * 64 * 4 + 8 * 32 * 4)
#ifdef AZ_EXP_TIMING
#define S_DBG S_END                   // unsigned long long [8 waves][8] cycle counters (tuning builds only)
#define S_BYTES (S_DBG + 512)
#else
#define S_BYTES S_END
#endif

// blob layout (bytes); must match alphazero_amd/net.py
#define OFF_SF 0                                   // stem      [2 halves][64][8] bf16 (16x16x32 A fragments)
#define OFF_WF (OFF_SF + 2 * 1024)                 // c1, c2    [2][9 taps][2 halves][64][8] bf16 + identity [2][64][8]
#define OFF_HF (OFF_WF + W_BYTES)                  // 1x1 heads [64][8] bf16 (rows 0,1 policy planes, 2 value plane)
#define OFF_F1 (OFF_HF + 1024)                     // value FC1 [8][3][64][8] bf16 (32x32x16 A fragments)
#define OFF_PF (OFF_F1 + 8 * 3 * 1024)             // policy FC [2][6][64][8] bf16
#define OFF_BIAS (OFF_PF + 2 * 6 * 1024)           // float [3][32] stem, c1, c2 ; then [4] heads
#define OFF_BP (OFF_BIAS + 100 * 4)                // float [64] policy FC bias
#define OFF_B1 (OFF_BP + 64 * 4)                   // float [256]
#define OFF_W2 (OFF_B1 + 256 * 4)                  // float [256]
#define OFF_B2 (OFF_W2 + 256 * 4)                  // float [4]
#define BLOB_BYTES (OFF_B2 + 16)
#define LDS_BYTES (OFF_BIAS + S_BYTES)

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

__device__ __forceinline__ const bf16x8* frag_ptr(const uint8_t* blob, int off, int lane) {
    return reinterpret_cast<const bf16x8*>(blob + off) + lane;
}

// all weight fragments go from the blob to LDS once per launch (77 KB); every thread of the workgroup takes part
__device__ __forceinline__ void load_weights(const uint8_t* blob, uint8_t* wlds, int tid) {
    const uint4* src = reinterpret_cast<const uint4*>(blob);
    uint4* dst = reinterpret_cast<uint4*>(wlds);
    for (int i = tid; i < OFF_BIAS / 16; i += NTHREADS) dst[i] = src[i];
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

__device__ __forceinline__ uint32_t bf16x2_of(float a, float b) {
    const f32x2 f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
}

// ---------------------------------------------------------------------------------------------------------
// The trunk of one task (two boards), written as a software pipeline.  Register state:
//   X[y]      bf16 fragment of row y of the block input (lane (p, q) = channels 8q..8q+7 of cell p)
//   Y[y]      the same for the output of the block's first conv
//   acc[y][h] the twelve independent accumulators of the conv in flight (h = channel half: 8q+4h..8q+4h+3)
// A conv's epilogue (ReLU, bf16) is merged with the centre tap of the NEXT conv (which needs row y only): row
// by row, finish row y, then issue its two centre MFMAs, so the vector work of row y+1 runs under the matrix
// work of row y even when the wave is alone on its SIMD.  The residual connection is one more "tap" of the
// block's second conv: an identity matrix times X on the matrix cores (exact: 1.0 x bf16 in an fp32
// accumulator) instead of sixteen vector instructions per row.  Weight fragments are requested one tap ahead
// of the MFMAs that use them and the lane-shifted rows of a window column are formed during the previous
// column's taps; scheduling fences between taps keep the compiler from hoisting all requests (they would not
// fit in the register file) or sinking them next to their use (their latency would be exposed).
// The pad lane of each board row must read as zero to its neighbours: its bias is -1e30, which the ReLU
// returns to zero, so no masking instruction is needed.
// ---------------------------------------------------------------------------------------------------------
#define PAD_BIAS (-1.0e30f)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

struct Bias2 { f32x4 h0, h1; };
__device__ __forceinline__ Bias2 lane_bias(const float* bias, int lane) {
    const int q = lane >> 4;
    Bias2 b;
    b.h0 = *reinterpret_cast<const f32x4*>(bias + 8 * q);
    b.h1 = *reinterpret_cast<const f32x4*>(bias + 8 * q + 4);
    if ((lane & 7) == 7) {
        const f32x4 pad = {PAD_BIAS, PAD_BIAS, PAD_BIAS, PAD_BIAS};
        b.h0 = pad; b.h1 = pad;
    }
    return b;
}

struct WPair { bf16x8 a0, a1; };                             // the two channel halves of one tap's weights
// CONV 0, 1: the block's convs; CONV 2 with KY = KX = 0: the identity
template <int CONV, int KY, int KX>
__device__ __forceinline__ WPair wload(const uint8_t* wlds, int lane) {
    const bf16x8* wf = reinterpret_cast<const bf16x8*>(wlds + OFF_WF) + ((CONV * 9 + KY * 3 + KX) * 2) * 64 + lane;
    WPair w;
    w.a0 = wf[0]; w.a1 = wf[64];
    return w;
}

// A task is two boards (NT = 6 tiles: tile y = row y of both boards) or ONE board folded onto itself (NT = 3
// tiles: tile t = row t in lanes 0..7 and row t + 3 in lanes 8..15), which costs little more than half: a lone
// leaf does not pay for an empty partner.  Folded, the row above tile 0 is [nothing | row 2] and the row below
// tile 2 is [row 3 | nothing]: the neighbouring tile moved by eight lanes.
template <int CTRL>
__device__ __forceinline__ bf16x8 dpp_frag(const bf16x8& v) {
    const uint4 u = __builtin_bit_cast(uint4, v);
    uint4 r;
    r.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u.x, CTRL, 0xF, 0xF, true);
    r.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u.y, CTRL, 0xF, 0xF, true);
    r.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u.z, CTRL, 0xF, 0xF, true);
    r.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u.w, CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(bf16x8, r);
}
// the row's fragment as seen one cell to the left (LEFT: lane p reads lane p - 1, row_shr:1) or to the right
// (row_shl:1), zero at the ends of the 16-lane row
template <bool LEFT>
__device__ __forceinline__ bf16x8 shift_cell(const bf16x8& v) { return dpp_frag<LEFT ? 0x111 : 0x101>(v); }
// folded tasks: lanes 8..15 <- lanes 0..7 and zeros below (row_shr:8), or lanes 0..7 <- lanes 8..15 (row_shl:8)
__device__ __forceinline__ bf16x8 fold_up(const bf16x8& v) { return dpp_frag<0x118>(v); }
__device__ __forceinline__ bf16x8 fold_down(const bf16x8& v) { return dpp_frag<0x108>(v); }

// the MFMAs of one tap: output row y takes input row y + KY - 1 (rows -1 and 6 do not exist)
template <int KY, int NT>
__device__ __forceinline__ void tap_rows(f32x4 (&acc)[NT][2], const bf16x8 (&s)[NT], const WPair& w) {
    if (NT == ROWS) {
#pragma unroll
        for (int y = 0; y < NT; ++y) {
            const int yy = y + KY - 1;
            if (yy < 0 || yy >= NT) continue;
            acc[y][0] = MFMA16(w.a0, s[yy], acc[y][0]);
            acc[y][1] = MFMA16(w.a1, s[yy], acc[y][1]);
        }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tt = t + KY - 1;
            const bf16x8 b = tt < 0 ? fold_up(s[NT - 1]) : (tt >= NT ? fold_down(s[0]) : s[tt]);
            acc[t][0] = MFMA16(w.a0, b, acc[t][0]);
            acc[t][1] = MFMA16(w.a1, b, acc[t][1]);
        }
    }
}

// The eight taps around the centre of conv CONV over `in`, plus (RES) the identity tap over `res`.  Weights are
// requested TWO taps ahead of their MFMAs (one region of ten or twelve MFMAs does not cover the LDS latency when
// eight waves are reading): on entry w = weights of tap (0,1) of CONV and nx = those of tap (2,1), already
// requested; on exit w = weights of the centre tap of conv NEXT and nx = those of its tap (0,1).
template <int CONV, int NEXT, bool RES, int NT>
__device__ __forceinline__ void conv_rest(const uint8_t* wlds, const bf16x8 (&in)[NT], const bf16x8 (&res)[NT],
                                          f32x4 (&acc)[NT][2], WPair& w, WPair& nx, int lane) {
    WPair n;
    bf16x8 sl[NT], sr[NT];
    constexpr int H = NT / 2 + (NT & 1), T3 = NT / 3;
#define NEXT_TAP() do { w = nx; nx = n; FENCE(); } while (0)
    // centre column of the window; meanwhile the rows as seen from the right-hand neighbour (window column 0)
    n = RES ? wload<2, 0, 0>(wlds, lane) : wload<CONV, 1, 0>(wlds, lane); tap_rows<0, NT>(acc, in, w);
#pragma unroll
    for (int y = 0; y < H; ++y) sl[y] = shift_cell<true>(in[y]);
    NEXT_TAP();
    n = RES ? wload<CONV, 1, 0>(wlds, lane) : wload<CONV, 0, 0>(wlds, lane); tap_rows<2, NT>(acc, in, w);
#pragma unroll
    for (int y = H; y < NT; ++y) sl[y] = shift_cell<true>(in[y]);
    NEXT_TAP();
    if (RES) {
        n = wload<CONV, 0, 0>(wlds, lane); tap_rows<1, NT>(acc, res, w); NEXT_TAP();
    }
    // window column 0; meanwhile the rows as seen from the left-hand neighbour (window column 2)
    n = wload<CONV, 2, 0>(wlds, lane); tap_rows<1, NT>(acc, sl, w);
#pragma unroll
    for (int y = 0; y < T3; ++y) sr[y] = shift_cell<false>(in[y]);
    NEXT_TAP();
    n = wload<CONV, 1, 2>(wlds, lane); tap_rows<0, NT>(acc, sl, w);
#pragma unroll
    for (int y = T3; y < 2 * T3; ++y) sr[y] = shift_cell<false>(in[y]);
    NEXT_TAP();
    n = wload<CONV, 0, 2>(wlds, lane); tap_rows<2, NT>(acc, sl, w);
#pragma unroll
    for (int y = 2 * T3; y < NT; ++y) sr[y] = shift_cell<false>(in[y]);
    NEXT_TAP();
    // window column 2
    n = wload<CONV, 2, 2>(wlds, lane); tap_rows<1, NT>(acc, sr, w); NEXT_TAP();
    n = wload<NEXT, 1, 1>(wlds, lane); tap_rows<0, NT>(acc, sr, w); NEXT_TAP();
    n = wload<NEXT, 0, 1>(wlds, lane); tap_rows<2, NT>(acc, sr, w); NEXT_TAP();
#undef NEXT_TAP
}

// Between two convs: acc -> out (ReLU, bf16) and, when CENTRE, the centre tap of the next conv with its bias as
// the C operand.  On entry w = centre weights of conv NEXT and nx = weights of its tap (0,1), already requested;
// on exit w = weights of tap (0,1) and nx = those of tap (2,1).
template <int NEXT, bool CENTRE, int NT>
__device__ __forceinline__ void between_convs(const uint8_t* wlds, f32x4 (&acc)[NT][2], bf16x8 (&out)[NT],
                                              const Bias2& bias, WPair& w, WPair& nx, int lane) {
    const WPair n = wload<NEXT, 2, 1>(wlds, lane);
#pragma unroll
    for (int y = 0; y < NT; ++y) {
        const f32x4 c0 = acc[y][0], c1 = acc[y][1];
        out[y] = __builtin_bit_cast(bf16x8, make_uint4(relu_bf16x2(c0[0], c0[1]), relu_bf16x2(c0[2], c0[3]),
                                                       relu_bf16x2(c1[0], c1[1]), relu_bf16x2(c1[2], c1[3])));
        if (CENTRE) {
            acc[y][0] = MFMA16(w.a0, out[y], bias.h0);
            acc[y][1] = MFMA16(w.a1, out[y], bias.h1);
            FENCE();                                         // row by row: vector work of row y+1 under these MFMAs
        }
    }
    w = nx;
    nx = n;
    FENCE();
}

// stem conv 2 -> C straight from the bitboards.  The B fragment of a tile holds the 3x3x2 window around each
// cell as bf16 1.0 / 0.0 in the k order that is cheapest to extract: k = 9 * column + 3 * plane + row of the window
// (18 of 32 used; the stem's weights are packed in the same order, alphazero_amd/net.py).  Leaves the raw
// accumulators.
template <int NT>
__device__ __forceinline__ void stem(const uint8_t* wlds, const uint64_t* pos /* this lane's board, or null */,
                                     f32x4 (&acc)[NT][2], const Bias2& b0, int lane) {
    const int p = lane & 15, q = lane >> 4;
    const int x = (p & 7) == 7 ? 0 : (p & 7);                    // pad lanes compute a cell's window, then drop it
    const int row0 = NT == ROWS ? 0 : 3 * (p >> 3);              // folded: the upper lanes hold rows 3..5
    const bf16x8 a0 = frag_ptr(wlds, OFF_SF, lane)[0], a1 = frag_ptr(wlds, OFF_SF, lane)[64];
    // After a shift by 8, cell (xx, yy) with xx in -1..7, yy in -1..6 is bit (xx+1)*7 + yy + 1; out-of-board
    // neighbours fall on guard bits, which are zero.  The three window columns of this lane's cell are taken out
    // once per task, nine bits each (rows -1..7), side by side in one word per plane.
    uint32_t cols[2] = {0, 0};
    if (pos) {
#pragma unroll
        for (int plane = 0; plane < 2; ++plane) {
            const uint64_t b = pos[plane] << 8;
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) cols[plane] |= ((uint32_t)(b >> ((x + tx) * 7)) & 0x1FFu) << (9 * tx);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int y = t + row0;
        // rows y-1, y, y+1 of the three columns: bits 0..2, 9..11, 18..20; the second plane goes three bits up
        const uint32_t m = ((cols[0] >> y) & 0x1C0E07u) | (((cols[1] >> y) & 0x1C0E07u) << 3);
        // this lane's eight k are bits 8q..8q+7: two per dword
        const uint32_t k8 = (m >> (8 * q)) & 0xFFu;
        uint32_t d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t t2 = (k8 >> (2 * i)) & 3u;
            d[i] = (t2 & 1u) * 0x3F80u + (t2 >> 1) * 0x3F800000u;
        }
        const bf16x8 f = __builtin_bit_cast(bf16x8, make_uint4(d[0], d[1], d[2], d[3]));
        acc[t][0] = MFMA16(a0, f, b0.h0);
        acc[t][1] = MFMA16(a1, f, b0.h1);
    }
}

// One task from the bitboards to the head planes: stem, tower, 1x1 head convs.  pos = this lane's board (the
// lane's half of a two-board task, or the one board of a folded task), null when there is none; hp0 / hv0 (and
// hp1 / hv1 for the second board) receive the ReLU'd head-conv outputs: 2 x 42 policy-plane values, 42 value-plane
// values (bf16).
template <int NT>
__device__ __forceinline__ void trunk_task(const uint8_t* wlds, const float* sbias, int num_blocks, const uint64_t* pos,
                                           __bf16* hp0, __bf16* hv0, __bf16* hp1, __bf16* hv1, int lane) {
    bf16x8 X[NT], Y[NT];
    f32x4 acc[NT][2];
    const Bias2 b1 = lane_bias(sbias + 32, lane), b2 = lane_bias(sbias + 64, lane);
    {
        const Bias2 b0 = lane_bias(sbias, lane);
        stem<NT>(wlds, pos, acc, b0, lane);
    }
    WPair w = wload<0, 1, 1>(wlds, lane), nx = wload<0, 0, 1>(wlds, lane);
    if (num_blocks > 0) between_convs<0, true, NT>(wlds, acc, X, b1, w, nx, lane);
    else between_convs<0, false, NT>(wlds, acc, X, b1, w, nx, lane);
#pragma unroll 1
    for (int blk = 0; blk < num_blocks; ++blk) {
        conv_rest<0, 1, false, NT>(wlds, X, X, acc, w, nx, lane);
        between_convs<1, true, NT>(wlds, acc, Y, b2, w, nx, lane);
        conv_rest<1, 0, true, NT>(wlds, Y, X, acc, w, nx, lane);
        if (blk + 1 < num_blocks) between_convs<0, true, NT>(wlds, acc, X, b1, w, nx, lane);
        else between_convs<0, false, NT>(wlds, acc, X, b1, w, nx, lane);
    }
    // 1x1 head convs as one MFMA per tile: rows 0,1 of D = policy planes, row 2 = value plane, which land in the
    // lanes q = 0 (registers 0..2) of each position
    const bf16x8 ah = frag_ptr(wlds, OFF_HF, lane)[0];
    const int p = lane & 15, x = p & 7;
    const int row0 = NT == ROWS ? 0 : 3 * (p >> 3);
    __bf16* hp = (NT == ROWS && (p >> 3)) ? hp1 : hp0;          // this lane's board
    __bf16* hv = (NT == ROWS && (p >> 3)) ? hv1 : hv0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
        const f32x4 d = MFMA16(ah, X[t], z4);
        if (lane < 16 && x != 7) {
            const int cell = (t + row0) * 7 + x;
            hp[cell] = (__bf16)fmaxf(d[0] + sbias[96], 0.0f);
            hp[42 + cell] = (__bf16)fmaxf(d[1] + sbias[97], 0.0f);
            hv[cell] = (__bf16)fmaxf(d[2] + sbias[98], 0.0f);
        }
    }
}

// The fully connected heads of up to 32 boards at once, by ONE wave: value FC 42->256->1 + tanh and policy FC
// 84->42 + landing-cell gather + soft-max, the same arithmetic in the same order as the workgroup-wide head phase
// of net_rounds (each board is a column of the same 32x32x16 MFMAs), so the results are bit-identical.  A batch
// costs the wave what two boards cost (the weights are read and the MFMAs issued once for all 32 columns).
//   planes  LDS, one slot of PLANE_SLOT bytes per game: 96 bf16 policy planes (k = ch*42 + cell, 84 used) then
//           48 bf16 value plane (42 used); the padding stays zero
//   batch   LDS int[32]: local game index of column c < n;  gmap: LDS, leaf index (row of p_out / v_out) of a
//           local game;  lbb: LDS, the games' leaf positions.  The logits pass through lg (LDS, float [n boards][64]).
#define PLANE_SLOT 288
__device__ __forceinline__ void heads_batch(const uint8_t* wlds, uint8_t* scr, float* lg, const uint8_t* planes, const int* batch,
                                            int n, const int* gmap, const uint64_t* lbb, float* p_out, float* v_out,
                                            int lane) {
    const float* sbias = reinterpret_cast<const float*>(scr + S_BIAS);
    const float* sb1 = reinterpret_cast<const float*>(scr + S_B1);
    const float* sw2 = reinterpret_cast<const float*>(scr + S_W2);
    const float* sbp = reinterpret_cast<const float*>(scr + S_BP);
    const int hh = lane >> 5, c = lane & 31;
    const int my_game = batch[c < n ? c : 0];                  // (columns past n repeat column 0)
    const __bf16* chp = reinterpret_cast<const __bf16*>(planes + my_game * PLANE_SLOT);
    const __bf16* chv = chp + 96;
    // One wave alone has nobody to hide its LDS latency behind, so every group of operands is requested in one
    // batch, a scheduling fence keeps the batch together, and the arithmetic follows.
#define HEADS_FENCE() __builtin_amdgcn_sched_barrier(0)
    float sv = 0.0f;                                           // column c: sum over the eight row tiles, in order
    bf16x8 bv[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) bv[s] = *reinterpret_cast<const bf16x8*>(chv + 16 * s + 8 * hh);
    // two row tiles at a time: their MFMA chains are independent and hide each other's latency (four at a time are
    // no faster and cost 56 registers more)
#pragma unroll 1
    for (int h2 = 0; h2 < 8; h2 += 2) {
        bf16x8 wf[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 3; ++s) wf[t][s] = frag_ptr(wlds, OFF_F1 + ((h2 + t) * 3 + s) * 1024, lane)[0];
        f32x4 b1v[2][4], w2v[2][4];                            // biases / second-layer weights of tile t
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                b1v[t][g] = *reinterpret_cast<const f32x4*>(sb1 + 32 * (h2 + t) + 8 * g + 4 * hh);
                w2v[t][g] = *reinterpret_cast<const f32x4*>(sw2 + 32 * (h2 + t) + 8 * g + 4 * hh);
            }
        HEADS_FENCE();
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = MFMA(wf[t][s], bv[s], acc[t]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float part = 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b1 = b1v[t][g], wv = w2v[t][g];
#pragma unroll
                for (int i = 0; i < 4; ++i) part = fmaf(fmaxf(acc[t][4 * g + i] + b1[i], 0.0f), wv[i], part);
            }
            part += __shfl_xor(part, 32);
            sv += part;
        }
    }
    {
        // the policy FC, one row tile (cells 0..31 / 32..41) after the other
        bf16x8 bp8[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) bp8[s] = *reinterpret_cast<const bf16x8*>(chp + 16 * s + 8 * hh);
#pragma unroll 1
        for (int hn = 0; hn < 2; ++hn) {
            bf16x8 pf[6];
            f32x4 bpv[4];
#pragma unroll
            for (int s = 0; s < 6; ++s) pf[s] = frag_ptr(wlds, OFF_PF + (hn * 6 + s) * 1024, lane)[0];
#pragma unroll
            for (int g = 0; g < 4; ++g) bpv[g] = *reinterpret_cast<const f32x4*>(sbp + 32 * hn + 8 * g + 4 * hh);
            HEADS_FENCE();
            f32x16 pacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) pacc[i] = 0.0f;
#pragma unroll
            for (int s = 0; s < 6; ++s) pacc = MFMA(pf[s], bp8[s], pacc);
            if (c < n) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o4[i] = pacc[4 * g + i] + bpv[g][i];
                    *reinterpret_cast<f32x4*>(lg + c * 64 + 32 * hn + 8 * g + 4 * hh) = o4;
                }
            }
        }
    }
#undef HEADS_FENCE
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     // this wave's logits, before its own reads below
#pragma unroll 1
    for (int b = lane >> 3; b < n; b += 8) {
        // landing-cell gather (connectfour.py:56-71) + soft-max over legal columns (mcts.py:106): 8 lanes per board
        const int col = lane & 7, gl = batch[b];
        const uint64_t* pos = lbb + 2 * gl;
        const uint64_t occ = pos[0] | pos[1];
        const int height = col < 7 ? __popcll((occ >> (7 * col)) & 0x3Full) : 6;
        const bool legal = col < 7 && height < 6;
        const float logit = legal ? lg[b * 64 + height * 7 + col] : -INFINITY;
        float m = logit;
        m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
        const float e = legal ? __expf(logit - m) : 0.0f;
        float sum = e;
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
        if (col < 7) p_out[(size_t)gmap[gl] * 7 + col] = legal ? e / sum : 0.0f;
    }
    if (lane < n) v_out[gmap[batch[lane]]] = tanhf(sv + sbias[99]);
}

#ifdef AZ_EXP_TIMING     // tuning aid: shader cycles per part of the network phase, summed over waves
// 0 stem+first centre, 1 tower, 2 head convs, 3 wait for the other waves, 4 fully connected heads + soft-max,
// 5 tasks, 6 waves that had a task, 7 waves without one: their whole wait
__device__ unsigned long long g_net_dbg[8];
// accumulated per wave in LDS (scr + S_DBG), flushed to g_net_dbg once at the end of the kernel
#define NTICK(i) do { const long long _t = clock64(); if (lane == 0) ndbg[i] += (unsigned long long)(_t - nt0); nt0 = _t; } while (0)
#define NCOUNT(i) do { if (lane == 0) ndbg[i] += 1ull; } while (0)
__device__ __forceinline__ void net_dbg_flush(uint8_t* scr, int wave, int lane) {
    if (lane < 8) atomicAdd(&g_net_dbg[lane], reinterpret_cast<unsigned long long*>(scr + S_DBG)[wave * 8 + lane]);
}
#else
#define NTICK(i) do { } while (0)
#define NCOUNT(i) do { } while (0)
#endif

// The network for the leaves listed in the workgroup's task list (scr + S_TL: per wave MAXIT tasks of
// two entries, entry = leaf index + 1, 0 = none), `nrounds` rounds; the whole workgroup calls it
// (it contains workgroup barriers).  Writes p_out[leaf][7] and v_out[leaf].
// LDS_BB: the position of leaf g is read from the LDS table scr + S_LBB, entry g - g_base, instead of leaf_bb.
template <bool LDS_BB>
__device__ __forceinline__ void net_rounds(const uint8_t* blob, int num_blocks, const uint64_t* leaf_bb, int g_base,
                                           float* p_out, float* v_out, const uint8_t* wlds, uint8_t* scr,
                                           int nrounds, int tid, int wave, int lane) {
    const uint64_t* lbb = reinterpret_cast<const uint64_t*>(scr + S_LBB);
    auto position = [&](int g) -> const uint64_t* {
        return LDS_BB ? lbb + 2 * (g - g_base) : leaf_bb + 2 * (size_t)g;
    };
    float* sbias = reinterpret_cast<float*>(scr + S_BIAS);
    int* tlist = reinterpret_cast<int*>(scr + S_TL);
    // ---- Rounds.  In a round every wave takes up to two tasks (two boards each) through stem, tower
    // and the 1x1 head convs on its own -- activations in registers, no workgroup barrier --
    // and leaves four boards of head planes in its scratch.  The fully connected heads then run once
    // for the whole workgroup: 8 waves x 4 boards are exactly the 32 columns of an MFMA, wave w owns
    // rows 32w..32w+31 of the value FC and waves 0, 1 the two row tiles of the policy FC, so each
    // head weight is read (from LDS) once per workgroup and round instead of once per wave and pass.
    uint8_t* pw = scr + S_PW + wave * PW_BYTES;
    __bf16* hp = reinterpret_cast<__bf16*>(pw + PW_HP);
    __bf16* hv = reinterpret_cast<__bf16*>(pw + PW_HV);
    const float* sb1 = reinterpret_cast<const float*>(scr + S_B1);
    const float* sw2 = reinterpret_cast<const float*>(scr + S_W2);
    const float* sbp = reinterpret_cast<const float*>(scr + S_BP);
    const int* mylist = tlist + wave * MAXIT * 2;
    int* hgl = reinterpret_cast<int*>(scr + S_HG);                 // leaf of each of the 32 head columns
    float* lg = reinterpret_cast<float*>(scr + S_LG);              // [32 boards][64] logits
    float* vpart = reinterpret_cast<float*>(scr + S_LG + 32 * 64 * 4);   // [8 waves][32 boards] value partial sums
    for (int round = 0; round < nrounds; ++round) {
        int hg0 = -1, hg1 = -1, hg2 = -1, hg3 = -1;
#ifdef AZ_EXP_TIMING
        unsigned long long* ndbg = reinterpret_cast<unsigned long long*>(scr + S_DBG) + wave * 8;
        long long nt0 = clock64();
        bool worked = false;
#endif
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int iter = 2 * round + half;
            const int g_first = mylist[2 * iter] - 1, g_second = mylist[2 * iter + 1] - 1;
            if (g_first < 0) break;
#ifdef AZ_EXP_TIMING
            worked = true;
            NCOUNT(5);
#endif
            if (half) { hg2 = g_first; hg3 = g_second; } else { hg0 = g_first; hg1 = g_second; }
            if (g_second >= 0) {
                const int g_mine = (lane & 8) ? g_second : g_first;
                trunk_task<ROWS>(wlds, sbias, num_blocks, position(g_mine), hp + 2 * half * 96, hv + 2 * half * 48,
                                 hp + (2 * half + 1) * 96, hv + (2 * half + 1) * 48, lane);
            } else {
                trunk_task<ROWS / 2>(wlds, sbias, num_blocks, position(g_first), hp + 2 * half * 96, hv + 2 * half * 48,
                                     nullptr, nullptr, lane);
            }
            NTICK(1);
        }
#ifdef AZ_EXP_TIMING
        if (worked) { NTICK(2); NCOUNT(6); } else NCOUNT(7);
#endif
        if (lane < 4) hgl[wave * 4 + lane] = lane == 0 ? hg0 : lane == 1 ? hg1 : lane == 2 ? hg2 : hg3;
        __syncthreads();                                   // every wave's planes are in place
#ifdef AZ_EXP_TIMING
        if (worked) NTICK(3); else NTICK(7);
#endif

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
                const bf16x8 a = frag_ptr(wlds, OFF_F1 + (hw * 3 + s) * 1024, hl)[0];
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
                const bf16x8 a = frag_ptr(wlds, OFF_PF + (hw * 6 + s) * 1024, hl)[0];
                pacc = MFMA(a, *reinterpret_cast<const bf16x8*>(chp + 16 * s + 8 * hh), pacc);
            }
        }
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
            if (gidx >= 0) occ = position(gidx)[0] | position(gidx)[1];
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
            if (gidx >= 0) v_out[gidx] = tanhf(sv + sbias[99]);
        }
        NTICK(4);
        // The next round may start at once.  Its plane stores come after this round's barrier above, which every
        // wave passed after its last plane read; planes of board slots without a leaf keep stale (finite) values
        // whose results nobody reads; the planes' k-padding is never written and stays zero.
    }
}
