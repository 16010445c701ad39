// Device-side building blocks of the fp32-grade form of the fused Connect Four network: the same register-
// resident trunk as az_net_dev.h (board-row tiles on 16x16x32 MFMAs, dy taps = other registers, dx taps = DPP,
// permuted output channels, pad cell zeroed by its bias, residual as an identity tap), with every operand carried
// as TWO fp16 numbers instead of one bf16:
//
//     x = h + l / 2048        h = fp16(x),  l = fp16((x - h) * 2048)
//
// h holds the leading 11 bits of x and l the next 11 (x - h is exact in fp32; the factor 2048 keeps l a normal
// fp16 number whatever the size of x), so h + l / 2048 differs from the fp32 value x by less than 2^-22 |x|.
// A product of two such operands is three fp16 MFMAs with fp32 accumulation:
//
//     w x = wh xh + (wh xl + wl xh) / 2048 (+ wl xl / 2^22, dropped)
//
// the first into the main accumulator, the two cross terms into a second one that is folded in (times 1/2048,
// exact) when the conv's outputs are formed.  fp16 products are exact in fp32, so the result is an fp32-grade
// dot product: the network's outputs agree with the fp32 statement of alphazero/models.py:70-87 to 1e-5 relative
// (tests/test_gpu_net.py), at a third of the bf16 trunk's rate instead of the sixteenth of the fp32 matrix
// instructions.  Range: activations saturate at 65504 (the largest fp16 number); folded weights beyond it are
// refused by the packer (alphazero_amd/net.py pack_c4_blob_hx).
//
// A task is ONE board folded onto itself (three tiles, as the one-board tasks of az_net_dev.h): two operand
// halves double the registers of every tensor, two boards would not fit two waves per SIMD.  The heads of the
// board are run by the same wave right after its trunk, in fp32 on the vector pipe (14 k multiply-adds per
// board against 3 x 1.3 M in the trunk) with the fully connected weights read from L2.
#pragma once
#include "az_net_dev.h"

typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
#define MFMAH(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define HX_SCALE 2048.0f
#define HX_INV (1.0f / 2048.0f)
#define HX_MAX 65504.0f

// blob layout (bytes); must match alphazero_amd/net.py pack_c4_blob_hx.  "h, l" = the two operand halves.
#define HX_OFF_SF 0                                  // stem      [2 channel halves][h, l][64][8] fp16
#define HX_OFF_WF (HX_OFF_SF + 4 * 1024)             // c1, c2    [2][9 taps][2 channel halves][h, l][64][8] fp16
#define HX_OFF_ID (HX_OFF_WF + 2 * 9 * 4 * 1024)     // identity  [2 channel halves][64][8] fp16 (exact: no l)
#define HX_OFF_HF (HX_OFF_ID + 2 * 1024)             // 1x1 heads [h, l][64][8] fp16 (rows 0,1 policy planes, 2 value plane)
#define HX_W_BYTES (HX_OFF_HF + 2 * 1024)            // everything above lives in LDS during a launch (80 KB)
#define HX_OFF_BIAS HX_W_BYTES                       // float [3][32] stem, c1, c2; [3] head convs; [1] value FC2 bias
#define HX_OFF_BP (HX_OFF_BIAS + 100 * 4)            // float [64] policy FC bias (42 used)
#define HX_OFF_B1 (HX_OFF_BP + 64 * 4)               // float [256] value FC1 bias
#define HX_OFF_W2 (HX_OFF_B1 + 256 * 4)              // float [256] value FC2 weights
#define HX_OFF_W1 (HX_OFF_W2 + 256 * 4)              // float [42 cells][64 lanes][4]: value FC1 weight of unit 4 lane + u
#define HX_OFF_WP (HX_OFF_W1 + 42 * 256 * 4)         // float [42 cells][84] policy FC weights
#define HX_BLOB_BYTES (HX_OFF_WP + 42 * 84 * 4)
#define HX_STRIP_BYTES 512                           // per wave: float [84] policy planes, [42] value plane of its board

struct HFrag { h8 h, l; };                           // operand fragment of a tile: x = h + l / 2048
struct HAcc { f32x4 m, c; };                         // main and cross-term accumulators
struct HWPair { h8 h0, l0, h1, l1; };                // one tap's weights: the two channel halves, h and l each

template <int CONV, int KY, int KX>
__device__ __forceinline__ HWPair hx_wload(const uint8_t* wlds, int lane) {
    const h8* wf = reinterpret_cast<const h8*>(wlds + HX_OFF_WF) + ((CONV * 9 + KY * 3 + KX) * 4) * 64 + lane;
    HWPair w;
    w.h0 = wf[0]; w.l0 = wf[64]; w.h1 = wf[128]; w.l1 = wf[192];
    return w;
}
// the identity "tap" that adds the residual: no l halves (1.0 is exact); they are filled with the h halves so
// that the value is well defined, and never used
__device__ __forceinline__ HWPair hx_wload_id(const uint8_t* wlds, int lane) {
    const h8* wf = reinterpret_cast<const h8*>(wlds + HX_OFF_ID) + lane;
    HWPair w;
    w.h0 = wf[0]; w.h1 = wf[64]; w.l0 = w.h0; w.l1 = w.h1;
    return w;
}

template <int CTRL>
__device__ __forceinline__ h8 hx_dpp(const h8& v) {
    const uint4 u = __builtin_bit_cast(uint4, v);
    uint4 r;
    r.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u.x, CTRL, 0xF, 0xF, true);
    r.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u.y, CTRL, 0xF, 0xF, true);
    r.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u.z, CTRL, 0xF, 0xF, true);
    r.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u.w, CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(h8, r);
}
template <int CTRL>
__device__ __forceinline__ HFrag hx_dpp(const HFrag& v) {
    HFrag r;
    r.h = hx_dpp<CTRL>(v.h); r.l = hx_dpp<CTRL>(v.l);
    return r;
}
// one cell to the left / right inside the 16-lane row (zero at its ends), and the folds of a one-board task
template <bool LEFT>
__device__ __forceinline__ HFrag hx_shift(const HFrag& v) { return hx_dpp<LEFT ? 0x111 : 0x101>(v); }
__device__ __forceinline__ HFrag hx_fold_up(const HFrag& v) { return hx_dpp<0x118>(v); }
__device__ __forceinline__ HFrag hx_fold_down(const HFrag& v) { return hx_dpp<0x108>(v); }

// The MFMAs of one tap over the three tiles: output tile t takes input tile t + KY - 1 (the tile above tile 0 and
// the one below tile 2 are the folds).  Three passes -- h x h, h x l, l x h -- so that two MFMAs on the same
// accumulator are always five others apart.
template <int KY, bool IDENT>
__device__ __forceinline__ void hx_tap(HAcc (&acc)[3][2], const HFrag (&s)[3], const HWPair& w) {
    HFrag b[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int tt = t + KY - 1;
        b[t] = tt < 0 ? hx_fold_up(s[2]) : (tt >= 3 ? hx_fold_down(s[0]) : s[tt]);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        acc[t][0].m = MFMAH(w.h0, b[t].h, acc[t][0].m);
        acc[t][1].m = MFMAH(w.h1, b[t].h, acc[t][1].m);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        acc[t][0].c = MFMAH(w.h0, b[t].l, acc[t][0].c);
        acc[t][1].c = MFMAH(w.h1, b[t].l, acc[t][1].c);
    }
    if (!IDENT) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            acc[t][0].c = MFMAH(w.l0, b[t].h, acc[t][0].c);
            acc[t][1].c = MFMAH(w.l1, b[t].h, acc[t][1].c);
        }
    }
}

// The eight taps around the centre of conv CONV over `in`, plus (RES) the identity tap over `res`; same software
// pipeline as conv_rest of az_net_dev.h: weights requested two taps ahead, the lane-shifted rows of a window
// column formed during the previous column's taps.  On entry w = weights of tap (0,1) of CONV and nx = those of
// tap (2,1); on exit w = weights of the centre tap of conv NEXT and nx = those of its tap (0,1).
template <int CONV, int NEXT, bool RES>
__device__ __forceinline__ void hx_conv_rest(const uint8_t* wlds, const HFrag (&in)[3], const HFrag (&res)[3],
                                             HAcc (&acc)[3][2], HWPair& w, HWPair& nx, int lane) {
    HWPair n;
    HFrag sl[3], sr[3];
#define HX_NEXT_TAP() do { w = nx; nx = n; FENCE(); } while (0)
    n = RES ? hx_wload_id(wlds, lane) : hx_wload<CONV, 1, 0>(wlds, lane); hx_tap<0, false>(acc, in, w);
    sl[0] = hx_shift<true>(in[0]); sl[1] = hx_shift<true>(in[1]);
    HX_NEXT_TAP();
    n = RES ? hx_wload<CONV, 1, 0>(wlds, lane) : hx_wload<CONV, 0, 0>(wlds, lane); hx_tap<2, false>(acc, in, w);
    sl[2] = hx_shift<true>(in[2]);
    HX_NEXT_TAP();
    if (RES) {
        n = hx_wload<CONV, 0, 0>(wlds, lane); hx_tap<1, true>(acc, res, w); HX_NEXT_TAP();
    }
    n = hx_wload<CONV, 2, 0>(wlds, lane); hx_tap<1, false>(acc, sl, w);
    sr[0] = hx_shift<false>(in[0]);
    HX_NEXT_TAP();
    n = hx_wload<CONV, 1, 2>(wlds, lane); hx_tap<0, false>(acc, sl, w);
    sr[1] = hx_shift<false>(in[1]);
    HX_NEXT_TAP();
    n = hx_wload<CONV, 0, 2>(wlds, lane); hx_tap<2, false>(acc, sl, w);
    sr[2] = hx_shift<false>(in[2]);
    HX_NEXT_TAP();
    n = hx_wload<CONV, 2, 2>(wlds, lane); hx_tap<1, false>(acc, sr, w); HX_NEXT_TAP();
    n = hx_wload<NEXT, 1, 1>(wlds, lane); hx_tap<0, false>(acc, sr, w); HX_NEXT_TAP();
    n = hx_wload<NEXT, 0, 1>(wlds, lane); hx_tap<2, false>(acc, sr, w); HX_NEXT_TAP();
#undef HX_NEXT_TAP
}

// fp32 value of an accumulator pair, ReLU'd and clamped to the fp16 range: max(0, min(65504, m + c / 2048))
__device__ __forceinline__ float hx_value(float m, float c) {
    return __builtin_amdgcn_fmed3f(fmaf(c, HX_INV, m), 0.0f, HX_MAX);
}
// two fp32 values -> their h halves and their l halves, packed
__device__ __forceinline__ void hx_split2(float a, float b, uint32_t& hh, uint32_t& ll) {
    const f32x2 f = {a, b};
    const h2 h = __builtin_convertvector(f, h2);
    const f32x2 r = {(a - (float)h[0]) * HX_SCALE, (b - (float)h[1]) * HX_SCALE};
    const h2 l = __builtin_convertvector(r, h2);
    hh = __builtin_bit_cast(uint32_t, h);
    ll = __builtin_bit_cast(uint32_t, l);
}

// Between two convs: acc -> out (ReLU, split) and, when CENTRE, the centre tap of the next conv with its bias as
// the main accumulator's initial value.  On entry w = centre weights of conv NEXT and nx = weights of its tap
// (0,1); on exit w = weights of tap (0,1) and nx = those of tap (2,1).
template <int NEXT, bool CENTRE>
__device__ __forceinline__ void hx_between(const uint8_t* wlds, HAcc (&acc)[3][2], HFrag (&out)[3], const Bias2& bias,
                                           HWPair& w, HWPair& nx, int lane) {
    const HWPair n = hx_wload<NEXT, 2, 1>(wlds, lane);
    const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int y = 0; y < 3; ++y) {
        const HAcc a0 = acc[y][0], a1 = acc[y][1];
        uint4 hh, ll;
        hx_split2(hx_value(a0.m[0], a0.c[0]), hx_value(a0.m[1], a0.c[1]), hh.x, ll.x);
        hx_split2(hx_value(a0.m[2], a0.c[2]), hx_value(a0.m[3], a0.c[3]), hh.y, ll.y);
        hx_split2(hx_value(a1.m[0], a1.c[0]), hx_value(a1.m[1], a1.c[1]), hh.z, ll.z);
        hx_split2(hx_value(a1.m[2], a1.c[2]), hx_value(a1.m[3], a1.c[3]), hh.w, ll.w);
        out[y].h = __builtin_bit_cast(h8, hh);
        out[y].l = __builtin_bit_cast(h8, ll);
        if (CENTRE) {
            acc[y][0].m = MFMAH(w.h0, out[y].h, bias.h0);
            acc[y][1].m = MFMAH(w.h1, out[y].h, bias.h1);
            acc[y][0].c = MFMAH(w.h0, out[y].l, z4);
            acc[y][1].c = MFMAH(w.h1, out[y].l, z4);
            acc[y][0].c = MFMAH(w.l0, out[y].h, acc[y][0].c);
            acc[y][1].c = MFMAH(w.l1, out[y].h, acc[y][1].c);
            FENCE();
        }
    }
    w = nx;
    nx = n;
    FENCE();
}

// stem conv 2 -> C straight from the bitboards (see stem() of az_net_dev.h: same window extraction, same k
// order); the window is 0 / 1 and therefore exact in fp16, so only the weights have an l half
__device__ __forceinline__ void hx_stem(const uint8_t* wlds, const uint64_t* pos, HAcc (&acc)[3][2], const Bias2& b0, int lane) {
    const int p = lane & 15, q = lane >> 4;
    const int x = (p & 7) == 7 ? 0 : (p & 7);
    const int row0 = 3 * (p >> 3);
    const h8* sf = reinterpret_cast<const h8*>(wlds + HX_OFF_SF) + lane;
    const h8 a0h = sf[0], a0l = sf[64], a1h = sf[128], a1l = sf[192];
    uint32_t cols[2] = {0, 0};
#pragma unroll
    for (int plane = 0; plane < 2; ++plane) {
        const uint64_t b = pos[plane] << 8;
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) cols[plane] |= ((uint32_t)(b >> ((x + tx) * 7)) & 0x1FFu) << (9 * tx);
    }
    const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int y = t + row0;
        const uint32_t m = ((cols[0] >> y) & 0x1C0E07u) | (((cols[1] >> y) & 0x1C0E07u) << 3);
        const uint32_t k8 = (m >> (8 * q)) & 0xFFu;
        uint32_t d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t t2 = (k8 >> (2 * i)) & 3u;
            d[i] = (t2 & 1u) * 0x3C00u + (t2 >> 1) * 0x3C000000u;          // fp16 1.0
        }
        const h8 f = __builtin_bit_cast(h8, make_uint4(d[0], d[1], d[2], d[3]));
        acc[t][0].m = MFMAH(a0h, f, b0.h0);
        acc[t][1].m = MFMAH(a1h, f, b0.h1);
        acc[t][0].c = MFMAH(a0l, f, z4);
        acc[t][1].c = MFMAH(a1l, f, z4);
    }
}

// One board from its bitboards to its head planes: stem, tower, 1x1 head convs.  pos = the board (every lane the
// same); strip (LDS, this wave's own) receives the ReLU'd head-conv outputs in fp32: [84] policy planes
// (k = channel * 42 + cell), [42] value plane.
__device__ __forceinline__ void hx_trunk_task(const uint8_t* wlds, const float* sbias, int num_blocks, const uint64_t* pos,
                                              float* strip, int lane) {
    HFrag X[3], Y[3];
    HAcc acc[3][2];
    const Bias2 b1 = lane_bias(sbias + 32, lane), b2 = lane_bias(sbias + 64, lane);
    {
        const Bias2 b0 = lane_bias(sbias, lane);
        hx_stem(wlds, pos, acc, b0, lane);
    }
    HWPair w = hx_wload<0, 1, 1>(wlds, lane), nx = hx_wload<0, 0, 1>(wlds, lane);
    if (num_blocks > 0) hx_between<0, true>(wlds, acc, X, b1, w, nx, lane);
    else hx_between<0, false>(wlds, acc, X, b1, w, nx, lane);
#pragma unroll 1
    for (int blk = 0; blk < num_blocks; ++blk) {
        hx_conv_rest<0, 1, false>(wlds, X, X, acc, w, nx, lane);
        hx_between<1, true>(wlds, acc, Y, b2, w, nx, lane);
        hx_conv_rest<1, 0, true>(wlds, Y, X, acc, w, nx, lane);
        if (blk + 1 < num_blocks) hx_between<0, true>(wlds, acc, X, b1, w, nx, lane);
        else hx_between<0, false>(wlds, acc, X, b1, w, nx, lane);
    }
    // 1x1 head convs: rows 0,1 of D = policy planes, row 2 = value plane, in the lanes q = 0 of each position
    const h8* hf = reinterpret_cast<const h8*>(wlds + HX_OFF_HF) + lane;
    const h8 ahh = hf[0], ahl = hf[64];
    const int p = lane & 15, x = p & 7;
    const int row0 = 3 * (p >> 3);
    const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const f32x4 dm = MFMAH(ahh, X[t].h, z4);
        f32x4 dc = MFMAH(ahh, X[t].l, z4);
        dc = MFMAH(ahl, X[t].h, dc);
        if (lane < 16 && x != 7) {
            const int cell = (t + row0) * 7 + x;
            strip[cell] = fmaxf(fmaf(dc[0], HX_INV, dm[0]) + sbias[96], 0.0f);
            strip[42 + cell] = fmaxf(fmaf(dc[1], HX_INV, dm[1]) + sbias[97], 0.0f);
            strip[84 + cell] = fmaxf(fmaf(dc[2], HX_INV, dm[2]) + sbias[98], 0.0f);
        }
    }
}

// The fully connected heads of the wave's board in fp32 on the vector pipe: value FC 42 -> 256 (ReLU) -> 1 + tanh
// (alphazero/models.py:47-67), policy FC 84 -> 42 at the landing cells only (the other 35 logits are dropped by
// alphazero/games/connectfour.py:56-71) + soft-max over the legal columns (alphazero/mcts.py:106).  Weights come
// from the blob in global memory (57 KB, resident in L2).  Every sum has a fixed order, so a board's (p, v) are a
// pure function of its position.  p_row = float[7], v_out = float[1] of this board.
__device__ __forceinline__ void hx_heads_own(const uint8_t* blob, const float* sbias, const float* strip, const uint64_t* pos,
                                             float* p_row, float* v_out, int lane) {
    // ---- value head: lane owns hidden units 4 lane .. 4 lane + 3
    const f32x4* w1 = reinterpret_cast<const f32x4*>(blob + HX_OFF_W1) + lane;
    f32x4 hsum = reinterpret_cast<const f32x4*>(blob + HX_OFF_B1)[lane];
    const f32x4 w2 = reinterpret_cast<const f32x4*>(blob + HX_OFF_W2)[lane];
#ifdef HX_PARTIAL_WAIT
    // diagnostic build (tools/exp_lds_wait_probe.py): the loop as first written, waits left to the compiler
#pragma unroll 6
    for (int c = 0; c < 42; ++c) {
        const f32x4 wv = w1[c * 64];
        const float a = strip[84 + c];
#pragma unroll
        for (int u = 0; u < 4; ++u) hsum[u] = fmaf(wv[u], a, hsum[u]);
    }
#else
#pragma unroll 1
    for (int c0 = 0; c0 < 42; c0 += 21) {                  // two trips to L2 of 21 rows each
        f32x4 wv[21];
        float av[21];
#pragma unroll
        for (int j = 0; j < 21; ++j) wv[j] = w1[(c0 + j) * 64];
#pragma unroll
        for (int j = 0; j < 21; ++j) av[j] = strip[84 + c0 + j];
        // All plane values of a trip are waited for together.  Left to the compiler, six floats arrived as one 128-bit
        // and one 64-bit LDS read with a partial wait (lgkmcnt(1)) in between, and with the SIMD's other wave
        // issuing MFMAs the first read's data was seen stale in lanes 48..63 by the packed multiply-adds that
        // followed (one hidden unit in seven boards wrong, differently from run to run); with the full wait
        // every run gives the same, correct bits.
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5]), "+v"(av[6]), "+v"(av[7]),
                       "+v"(av[8]), "+v"(av[9]), "+v"(av[10]), "+v"(av[11]), "+v"(av[12]), "+v"(av[13]), "+v"(av[14]),
                       "+v"(av[15]), "+v"(av[16]), "+v"(av[17]), "+v"(av[18]), "+v"(av[19]), "+v"(av[20])
                     :: "memory");
#pragma unroll
        for (int j = 0; j < 21; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) hsum[u] = fmaf(wv[j][u], av[j], hsum[u]);
    }
#endif
    float part = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; ++u) part = fmaf(fmaxf(hsum[u], 0.0f), w2[u], part);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
    // ---- policy head: eight lanes per column share the 84 terms of that column's landing cell
    const int col = lane >> 3, sub = lane & 7;
    const uint64_t occ = pos[0] | pos[1];
    const int height = col < 7 ? __popcll((occ >> (7 * col)) & 0x3Full) : 6;
    const bool legal = height < 6;
    const int cell = legal ? height * 7 + col : 0;
    const float* wp = reinterpret_cast<const float*>(blob + HX_OFF_WP) + cell * 84;
    float s = 0.0f, hpv[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) hpv[i] = strip[sub + 8 * i];              // (k = 84..87 reads the value plane: unused)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hpv[0]), "+v"(hpv[1]), "+v"(hpv[2]), "+v"(hpv[3]), "+v"(hpv[4]), "+v"(hpv[5]),
                 "+v"(hpv[6]), "+v"(hpv[7]), "+v"(hpv[8]), "+v"(hpv[9]), "+v"(hpv[10]) :: "memory");     // (as above)
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        const int k = sub + 8 * i;
        if (k < 84) s = fmaf(wp[k], hpv[i], s);
    }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    const float logit = legal ? s + reinterpret_cast<const float*>(blob + HX_OFF_BP)[cell] : -INFINITY;
    float lg[7], m = -INFINITY;
#pragma unroll
    for (int c = 0; c < 7; ++c) { lg[c] = __shfl(logit, 8 * c); m = fmaxf(m, lg[c]); }
    float e[7], sum = 0.0f;
#pragma unroll
    for (int c = 0; c < 7; ++c) { e[c] = lg[c] == -INFINITY ? 0.0f : expf(lg[c] - m); sum += e[c]; }
    float mine = 0.0f;
#pragma unroll
    for (int c = 0; c < 7; ++c) mine = lane == c ? e[c] / sum : mine;
    if (lane < 7) p_row[lane] = mine;
    if (lane == 0) *v_out = tanhf(part + sbias[99]);
}

// weights of the trunk and the head convs: blob -> LDS, once per launch; every thread of the workgroup takes part
__device__ __forceinline__ void hx_load_weights(const uint8_t* blob, uint8_t* wlds, int tid, int nthreads) {
    const uint4* src = reinterpret_cast<const uint4*>(blob);
    uint4* dst = reinterpret_cast<uint4*>(wlds);
    for (int i = tid; i < HX_W_BYTES / 16; i += nthreads) dst[i] = src[i];
}
