// az_net128.hip -- 128-channel Connect Four network on bf16 MFMA (gfx950).
//
// The 10x128 network (alphazero/models.py:70-87 with block_channels = 128) has 295 KB of bf16
// weights per 3x3 conv: they fit neither registers nor LDS, so unlike the 32-channel kernel
// (az_net.hip) the weights stream from L2 through LDS while the activations of a wave's two boards
// stay in registers for the whole residual tower.  Per conv the algorithmic work is
// 2*9*128*128*42 = 12.4 MFLOP per leaf.
//
// Three launches: stem (bitboards -> bf16 activations in HBM), tower (in place), heads.
// Activation tensor: [board][42 cells, cell = 7*row + column][128 ch] bf16, dense, channels in
// their natural order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/azhip.h"
#include "az_host.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define CH 128
#define GB 8                        // boards per tower workgroup (the activation tensor is padded to whole workgroups)
#define BPOS 42                     // positions per board

// blob layout (bytes); must match alphazero_amd/net.py pack_c4_blob_128
#define G_W 0                                        // bf16 [2 convs][4 groups][4 chunks][9 taps][2 halves][64 lanes][8]
#define G_STEM (G_W + 2 * 9 * CH * CH * 2)           // f32 [18][128]  k = tap*2 + plane
#define G_BIAS (G_STEM + 18 * CH * 4)                // f32 [3][128]   stem, c1, c2
#define G_HW (G_BIAS + 3 * CH * 4)                   // f32 [3][128]   1x1 heads: policy 0, 1, value
#define G_HB (G_HW + 3 * CH * 4)                     // f32 [4]
#define G_WP (G_HB + 16)                             // f32 [42][84]
#define G_BP (G_WP + 42 * 84 * 4)                    // f32 [44]
#define G_W1T (G_BP + 44 * 4)                        // f32 [42][256]
#define G_B1 (G_W1T + 42 * 256 * 4)                  // f32 [256]
#define G_W2 (G_B1 + 256 * 4)                        // f32 [256]
#define G_B2 (G_W2 + 256 * 4)                        // f32 [4]
#define G_BLOB_BYTES (G_B2 + 16)

// bf16(ReLU(a)), bf16(ReLU(b)) packed in one dword: convert first, then clamp the two bit patterns
// as signed 16-bit integers (negative floats, -0 included, are negative integers)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t relu_bf16x2(float a, float b) {
    const f32x2 f = {a, b};
    const bf16x2 y = __builtin_convertvector(f, bf16x2);     // one v_cvt_pk_bf16_f32 for the pair
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, y), z));
}

// rows of the activation tensor: whole workgroups of GB boards
__host__ __device__ static inline size_t act_positions(int n) { return (size_t)((n + GB - 1) / GB * GB) * BPOS; }

// ---- which leaves need the network: list[rank] = leaf index, in leaf order --------------------
// One workgroup; thread t takes `per` consecutive leaves, a shuffle scan per wave and one pass over
// the 16 wave totals give every thread its first rank.  list[n] receives the count.
__global__ void __launch_bounds__(1024) k_compact128(const int32_t* __restrict__ leaf_info, int n, int32_t* __restrict__ list) {
    __shared__ int wtot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + 1023) / 1024, g0 = tid * per;
    unsigned long long mask = 0;                       // per <= 32 (n <= 32768)
    for (int j = 0; j < per; ++j) {
        const int g = g0 + j;
        if (g < n) {
            const int k = leaf_info[g] & 0xFF;
            if (k == AZ_LEAF_EVAL || k == AZ_LEAF_ROOT) mask |= 1ull << j;
        }
    }
    const int cnt = __popcll(mask);
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int t = wtot[w];
        before += (w < wave) ? t : 0;
        total += t;
    }
    int rank = before + incl - cnt;
    while (mask) {
        const int j = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        list[rank++] = g0 + j;
    }
    if (tid == 0) list[n] = total;
}

// number of boards to evaluate and the leaf behind board b (list == nullptr: every leaf, in order)
__device__ __forceinline__ int boards_to_run(const int32_t* list, int n) { return list ? list[n] : n; }
__device__ __forceinline__ long leaf_of(const int32_t* list, long b) { return list ? (long)list[b] : b; }

// ---- stem: 2 -> 128 channels straight from the bitboards (vector ALU, fp32 accumulate) ----------
__global__ void __launch_bounds__(256) k_stem128(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ leaf_bb,
                                                 const int32_t* __restrict__ list, int nleaves, __bf16* __restrict__ out) {
    const int n = boards_to_run(list, nleaves);
    __shared__ float w[18 * CH];
    __shared__ float bias[CH];
    for (int i = threadIdx.x; i < 18 * CH; i += 256) w[i] = reinterpret_cast<const float*>(blob + G_STEM)[i];
    if (threadIdx.x < CH) bias[threadIdx.x] = reinterpret_cast<const float*>(blob + G_BIAS)[threadIdx.x];
    __syncthreads();
    // thread = (board, cell o, 16-channel group)
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(t & 7);
    const long bo = t >> 3;
    const int o = (int)(bo % BPOS);
    const long b = bo / BPOS;
    const int y = o / 7, x = o - 7 * y;
    __bf16* dst = out + ((size_t)b * BPOS + o) * CH + cg * 16;
    if (b >= n) {                         // boards that pad the last workgroup of the tower
        if ((size_t)b * BPOS < act_positions(n)) {
            reinterpret_cast<uint4*>(dst)[0] = make_uint4(0, 0, 0, 0);
            reinterpret_cast<uint4*>(dst)[1] = make_uint4(0, 0, 0, 0);
        }
        return;
    }
    // bit of cell (row, column) is 7*column + row; shifted by 8 so that the 3x3 window never indexes
    // below bit 0, and row 6 / column 7 land on bits that are always clear
    const long leaf = leaf_of(list, b);
    const uint64_t bb0 = leaf_bb[2 * leaf] << 8, bb1 = leaf_bb[2 * leaf + 1] << 8;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = bias[cg * 16 + c];
    for (int tap = 0; tap < 9; ++tap) {
        const int sh = (x + tap % 3) * 7 + (y + tap / 3);
        if ((bb0 >> sh) & 1ull) {
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] += w[(2 * tap) * CH + cg * 16 + c];
        }
        if ((bb1 >> sh) & 1ull) {
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] += w[(2 * tap + 1) * CH + cg * 16 + c];
        }
    }
    bf16x8 lo, hi;
#pragma unroll
    for (int c = 0; c < 8; ++c) { lo[c] = (__bf16)fmaxf(acc[c], 0.0f); hi[c] = (__bf16)fmaxf(acc[8 + c], 0.0f); }
    reinterpret_cast<bf16x8*>(dst)[0] = lo;
    reinterpret_cast<bf16x8*>(dst)[1] = hi;
}

// ---- residual tower: 2 * num_blocks 3x3 convs 128 -> 128, activations resident in REGISTERS ------
// A wave (one per SIMD, up to 512 registers) owns two boards for the whole tower.  Tiles, operand layout and
// the tricks are those of the 32-channel trunk (az_net_dev.h): a tile is one board row of both boards (16
// positions, the eighth cell of each row is a zero pad), contractions are v_mfma_f32_16x16x32_bf16 with one
// 32-channel chunk of a tap per k-step, dy taps read the neighbouring row's register fragment, dx taps read the
// row shifted by one lane (DPP, zero fill), output channels are permuted inside each group of 32 so that an
// accumulator lands in the next layer's operand layout (lane-local ReLU / bf16 epilogue), the pad cell is kept at
// zero by a -1e30 bias, and the residual is an identity "tap" on the matrix cores.  With 128 channels a tensor of
// two boards is 4 chunks x 6 rows = 24 fragments (96 registers); X (block input) and Y (first conv's output)
// both stay in registers, 48 registers accumulate one group of 32 output channels at a time.
// Weights (295 KB per conv) stream L2 -> LDS -> registers: the 72 KB of one output group ([tap][chunk][half]
// fragments, contiguous in the blob) are fetched by the whole workgroup into the idle half of a double buffer
// while the other half is being consumed, one workgroup barrier per group (every 396 MFMAs per wave); the four
// waves of a workgroup (eight boards) share every fetched byte.  HBM sees one read and one write of the
// activations per tower.
#define CONV_BYTES (9 * CH * CH * 2)
#define TB GB                        // boards per workgroup: 4 waves x 2
#define TROWS 6                      // board rows = tiles of a wave's two boards
#define NT 256
#define CHUNK_BYTES (4 * 9 * 2 * 1024)          // one output group of one conv: [chunk 4][tap 9][half 2][64][8] bf16
#define HALF_BYTES (CHUNK_BYTES / 2)            // input chunks 0, 1 | 2, 3: each within reach of a ds_read immediate offset
#define T_BIAS_OFF (2 * CHUNK_BYTES)            // f32 [2][128] conv biases
#define T_LDS_BYTES (T_BIAS_OFF + 2 * CH * 4)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define PAD_BIAS (-1.0e30f)
typedef __attribute__((ext_vector_type(4))) float f32x4;

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
// the row as seen one cell to the left (lane p reads lane p - 1: row_shr:1) or to the right (row_shl:1)
template <bool LEFT>
__device__ __forceinline__ bf16x8 shift_cell(const bf16x8& v) { return dpp_frag<LEFT ? 0x111 : 0x101>(v); }

struct Bias2 { f32x4 h0, h1; };
__device__ __forceinline__ Bias2 lane_bias(const uint8_t* lds, uint32_t group_off, int lane) {
    const int q = lane >> 4;
    const float* bias = reinterpret_cast<const float*>(lds + group_off);
    Bias2 b;
    b.h0 = *reinterpret_cast<const f32x4*>(bias + 8 * q);
    b.h1 = *reinterpret_cast<const f32x4*>(bias + 8 * q + 4);
    if ((lane & 7) == 7) {
        const f32x4 pad = {PAD_BIAS, PAD_BIAS, PAD_BIAS, PAD_BIAS};
        b.h0 = pad; b.h1 = pad;
    }
    return b;
}

// An LDS byte offset the compiler must keep in one register and may not fold into the absolute addresses
// behind it: a ds_read immediate reaches 64 KB, the weight buffers span 144 KB, and every address past the reach
// would otherwise get (and keep, hoisted out of the block loop) a register of its own.
__device__ __forceinline__ uint32_t opaque(uint32_t off) {
    asm volatile("" : "+v"(off));
    return off;
}

// both halves' weight fragments of tap (KY, KX) of input chunk KC; `wb` = this lane's offset in the half buffer
struct WPair { bf16x8 a0, a1; };
template <int KY, int KX, int KC>
__device__ __forceinline__ WPair wload(const uint8_t* lds, uint32_t wb) {
    const bf16x8* wf = reinterpret_cast<const bf16x8*>(lds + wb) + (((KC & 1) * 9 + KY * 3 + KX) * 2) * 64;
    WPair w;
    w.a0 = wf[0]; w.a1 = wf[64];
    return w;
}

template <int KY>
__device__ __forceinline__ void tap_rows(f32x4 (&acc)[TROWS][2], const bf16x8 (&s)[TROWS], const WPair& w) {
#pragma unroll
    for (int y = 0; y < TROWS; ++y) {
        const int yy = y + KY - 1;
        if (yy < 0 || yy >= TROWS) continue;
        acc[y][0] = MFMA16(w.a0, s[yy], acc[y][0]);
        acc[y][1] = MFMA16(w.a1, s[yy], acc[y][1]);
    }
}

// The workgroup's copy of the next group's weights, global -> registers -> LDS, in STAGE_BATCHES bursts spread
// over the 36 tap regions of the group being computed: a batch (6 x 16 bytes per thread) is fetched at the start
// of its third of the group and written to LDS at the start of the next third (the last one after the last
// region), which gives the fetch 12 regions (about 2k cycles) to arrive and costs 24 registers.  Measured
// alternatives: one piece per region (trickle) and one burst per group are 5 % slower, LDS-DMA loads
// (global_load_lds_dwordx4, no registers) 1.5 % faster.  `src` is uniform or null,
// `dst_off` = opaque(buffer offset + 16 * tid).
#define NSTAGE (CHUNK_BYTES / 16 / NT)           // 18 pieces per thread and group
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define STAGE_BATCHES 3
#define STAGE_PER (NSTAGE / STAGE_BATCHES)
struct Stage {
    u32x4 v[STAGE_PER];
    const uint8_t* src;
    uint32_t tid16, dst_off;
};
// called at the start of every region; REGION == 36: after the last one
template <int REGION>
__device__ __forceinline__ void stage_tick(Stage& st, uint8_t* lds) {
    if (!st.src) return;
    if (REGION % (36 / STAGE_BATCHES) != 0) return;
    constexpr int B = REGION / (36 / STAGE_BATCHES);
    if (B > 0) {
#pragma unroll
        for (int j = 0; j < STAGE_PER; ++j)
            *reinterpret_cast<u32x4*>(lds + st.dst_off + ((B - 1) * STAGE_PER + j) * (NT * 16)) = st.v[j];
    }
    if (B < STAGE_BATCHES) {
#pragma unroll
        for (int j = 0; j < STAGE_PER; ++j)
            st.v[j] = *reinterpret_cast<const u32x4*>(st.src + (B * STAGE_PER + j) * (NT * 16) + st.tid16);
    }
}

// the nine taps of input chunk KC into the accumulators of the current output group.  On entry w = weights of
// the centre tap of chunk KC, on exit w = those of chunk KC + 1 (or, after the last chunk, whatever `last` loads).
template <int KC, class LastLoad>
__device__ __forceinline__ void chunk_taps(uint8_t* lds, uint32_t wb, const bf16x8 (&in)[TROWS],
                                           f32x4 (&acc)[TROWS][2], const Bias2& bias, WPair& w, Stage& st, LastLoad last) {
    WPair n;
    bf16x8 sl[TROWS], sr[TROWS];
    // centre column; the first tap of the first chunk starts the accumulators from the bias
    n = wload<0, 1, KC>(lds, wb);
#pragma unroll
    for (int y = 0; y < TROWS; ++y) {
        if (KC == 0) {
            acc[y][0] = MFMA16(w.a0, in[y], bias.h0);
            acc[y][1] = MFMA16(w.a1, in[y], bias.h1);
        } else {
            acc[y][0] = MFMA16(w.a0, in[y], acc[y][0]);
            acc[y][1] = MFMA16(w.a1, in[y], acc[y][1]);
        }
    }
    w = n; stage_tick<9 * KC + 1>(st, lds); FENCE();
    n = wload<2, 1, KC>(lds, wb); tap_rows<0>(acc, in, w);
#pragma unroll
    for (int y = 0; y < 3; ++y) sl[y] = shift_cell<true>(in[y]);
    w = n; stage_tick<9 * KC + 2>(st, lds); FENCE();
    n = wload<1, 0, KC>(lds, wb); tap_rows<2>(acc, in, w);
#pragma unroll
    for (int y = 3; y < TROWS; ++y) sl[y] = shift_cell<true>(in[y]);
    w = n; stage_tick<9 * KC + 3>(st, lds); FENCE();
    // window column 0; meanwhile the rows for window column 2
    n = wload<0, 0, KC>(lds, wb); tap_rows<1>(acc, sl, w);
    sr[0] = shift_cell<false>(in[0]); sr[1] = shift_cell<false>(in[1]);
    w = n; stage_tick<9 * KC + 4>(st, lds); FENCE();
    n = wload<2, 0, KC>(lds, wb); tap_rows<0>(acc, sl, w);
    sr[2] = shift_cell<false>(in[2]); sr[3] = shift_cell<false>(in[3]);
    w = n; stage_tick<9 * KC + 5>(st, lds); FENCE();
    n = wload<1, 2, KC>(lds, wb); tap_rows<2>(acc, sl, w);
    sr[4] = shift_cell<false>(in[4]); sr[5] = shift_cell<false>(in[5]);
    w = n; stage_tick<9 * KC + 6>(st, lds); FENCE();
    // window column 2
    n = wload<0, 2, KC>(lds, wb); tap_rows<1>(acc, sr, w); w = n; stage_tick<9 * KC + 7>(st, lds); FENCE();
    n = wload<2, 2, KC>(lds, wb); tap_rows<0>(acc, sr, w); w = n; stage_tick<9 * KC + 8>(st, lds); FENCE();
    n = last(); tap_rows<2>(acc, sr, w); w = n; stage_tick<9 * KC + 9>(st, lds); FENCE();
}

// a uniform byte offset the compiler recomputes where it is used instead of keeping every multiple live
__device__ __forceinline__ uint32_t opaque_uniform(uint32_t off) {
    asm volatile("" : "+s"(off));
    return off;
}

// One group of 32 output channels of one conv: the four input chunks (the next group's weights trickle into
// the other LDS buffer meanwhile, when `src` is not null), (RES) the identity tap over the residual's chunk,
// ReLU / bf16 into `out`.
template <bool RES>
__device__ __forceinline__ void out_group(uint8_t* lds, uint32_t wb, const WPair& ident,
                                          const bf16x8 (&in)[4][TROWS], const bf16x8 (&res)[TROWS],
                                          bf16x8 (&out)[TROWS], const Bias2& bias, const uint8_t* src, uint32_t tid16,
                                          uint32_t dst_off) {
    f32x4 acc[TROWS][2];
    Stage st;
    st.src = src; st.tid16 = tid16; st.dst_off = dst_off;
    const uint32_t wb2 = opaque(wb + HALF_BYTES);
    WPair w = wload<1, 1, 0>(lds, wb);
    stage_tick<0>(st, lds);
    chunk_taps<0>(lds, wb, in[0], acc, bias, w, st, [&] { return wload<1, 1, 1>(lds, wb); });
    chunk_taps<1>(lds, wb, in[1], acc, bias, w, st, [&] { return wload<1, 1, 2>(lds, wb2); });
    chunk_taps<2>(lds, wb2, in[2], acc, bias, w, st, [&] { return wload<1, 1, 3>(lds, wb2); });
    chunk_taps<3>(lds, wb2, in[3], acc, bias, w, st, [&] { return ident; });
    if (RES) {
#pragma unroll
        for (int y = 0; y < TROWS; ++y) {
            acc[y][0] = MFMA16(w.a0, res[y], acc[y][0]);
            acc[y][1] = MFMA16(w.a1, res[y], acc[y][1]);
        }
    }
#pragma unroll
    for (int y = 0; y < TROWS; ++y) {
        const f32x4 c0 = acc[y][0], c1 = acc[y][1];
        out[y] = __builtin_bit_cast(bf16x8, make_uint4(relu_bf16x2(c0[0], c0[1]), relu_bf16x2(c0[2], c0[3]),
                                                       relu_bf16x2(c1[0], c1[1]), relu_bf16x2(c1[2], c1[3])));
    }
}

__global__ void __launch_bounds__(NT, 1)
k_tower128(const uint8_t* __restrict__ wts /* [conv][group][tap][chunk][half] fragments */,
           const float* __restrict__ bias /* [3][128] */, __bf16* __restrict__ act, int num_blocks,
           const int32_t* __restrict__ list, int nleaves) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // two weight buffers, then f32 [2][128] biases
    if ((int)blockIdx.x * TB >= boards_to_run(list, nleaves)) return;   // whole workgroup, before any barrier
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int p = lane & 15, q = lane >> 4, x = p & 7;
    const bool cell = x != 7;

    // weights of the first group on their way while the activations are fetched
    const uint32_t tid16 = (uint32_t)tid * 16, lane16 = (uint32_t)lane * 16;
    u32x4 first[NSTAGE];
#pragma unroll
    for (int j = 0; j < NSTAGE; ++j) first[j] = *reinterpret_cast<const u32x4*>(wts + j * (NT * 16) + tid16);
    if (tid < 2 * CH) reinterpret_cast<float*>(lds + T_BIAS_OFF)[tid] = bias[CH + tid];   // conv biases

    // this lane's board and the two tensors; element (y, x, channel 32 kc + 8 q ..) of board b is 16 contiguous bytes
    const size_t board = (size_t)blockIdx.x * TB + wave * 2 + (p >> 3);
    __bf16* mine = act + (board * BPOS + x) * CH + 8 * q;
    bf16x8 X[4][TROWS], Y[4][TROWS];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
        for (int y = 0; y < TROWS; ++y) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (cell) v = *reinterpret_cast<const uint4*>(mine + (size_t)y * 7 * CH + 32 * kc);
            X[kc][y] = __builtin_bit_cast(bf16x8, v);
        }
    // identity fragments: row r of half h is channel 8 (r / 4) + 4 h + r % 4 of the group
    WPair ident;
    {
        const int r = lane & 15;
        uint32_t d[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ch = 8 * (r >> 2) + 4 * h + (r & 3);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k0 = 8 * q + 2 * i;
                d[h][i] = (k0 == ch ? 0x3F80u : 0u) | (k0 + 1 == ch ? 0x3F800000u : 0u);
            }
        }
        ident.a0 = __builtin_bit_cast(bf16x8, make_uint4(d[0][0], d[0][1], d[0][2], d[0][3]));
        ident.a1 = __builtin_bit_cast(bf16x8, make_uint4(d[1][0], d[1][1], d[1][2], d[1][3]));
    }
    {
        const uint32_t to = opaque(tid16);
#pragma unroll
        for (int j = 0; j < NSTAGE; ++j) *reinterpret_cast<u32x4*>(lds + to + j * (NT * 16)) = first[j];
    }
    __syncthreads();

#pragma unroll 1
    for (int blk = 0; blk < num_blocks; ++blk) {
        const bool more = blk + 1 < num_blocks;
        // step s = conv * 4 + group uses buffer s & 1 and fetches the weights of step s + 1 into the other
#define TOWER_STEP(S, RESFLAG, IN, RESCH, OUT)                                                              \
        {                                                                                                   \
            const uint8_t* src = ((S) < 7 || more) ? wts + opaque_uniform((((S) + 1) & 7) * CHUNK_BYTES) : nullptr; \
            const Bias2 bs = lane_bias(lds, opaque(T_BIAS_OFF + (((S) >> 2) * CH + 32 * ((S) & 3)) * 4), lane); \
            out_group<RESFLAG>(lds, opaque(((S) & 1) * CHUNK_BYTES + lane16), ident, IN, RESCH, OUT, bs, src, tid16, \
                               opaque((((S) + 1) & 1) * CHUNK_BYTES + tid16));                              \
            __syncthreads();                                                                                \
        }
        TOWER_STEP(0, false, X, X[0], Y[0])
        TOWER_STEP(1, false, X, X[1], Y[1])
        TOWER_STEP(2, false, X, X[2], Y[2])
        TOWER_STEP(3, false, X, X[3], Y[3])
        TOWER_STEP(4, true, Y, X[0], X[0])
        TOWER_STEP(5, true, Y, X[1], X[1])
        TOWER_STEP(6, true, Y, X[2], X[2])
        TOWER_STEP(7, true, Y, X[3], X[3])
#undef TOWER_STEP
    }
    if (cell) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc)
#pragma unroll
            for (int y = 0; y < TROWS; ++y)
                *reinterpret_cast<uint4*>(mine + (size_t)y * 7 * CH + 32 * kc) = __builtin_bit_cast(uint4, X[kc][y]);
    }
}

// ---- heads: 1x1 convs, FCs, landing-cell gather, soft-max (vector ALU, fp32) ---------------------
#define HB_ 8
__global__ void __launch_bounds__(256) k_heads128(const uint8_t* __restrict__ blob, const __bf16* __restrict__ act,
                                                  const uint64_t* __restrict__ leaf_bb, const int32_t* __restrict__ list,
                                                  int nleaves, float* __restrict__ p_out, float* __restrict__ v_out) {
    const int n = boards_to_run(list, nleaves);
    if ((long)blockIdx.x * HB_ >= n) return;
    __shared__ float hpv[HB_ * 3 * 42];
    __shared__ float hid[256 * HB_];
    const int tid = threadIdx.x;
    const long board0 = (long)blockIdx.x * HB_;
    const float* hw = reinterpret_cast<const float*>(blob + G_HW);
    const float* hb = reinterpret_cast<const float*>(blob + G_HB);
    // (board, cell) pairs, the three 128-long dot products of each
    for (int i = tid; i < HB_ * 42; i += 256) {
        const int b = i / 42, cell = i - b * 42;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        if (board0 + b < n) {
            const __bf16* xs = act + ((size_t)(board0 + b) * BPOS + cell) * CH;
            for (int c = 0; c < CH; c += 8) {
                const bf16x8 xv = *reinterpret_cast<const bf16x8*>(xs + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xf = (float)xv[j];
                    a0 = fmaf(hw[c + j], xf, a0);
                    a1 = fmaf(hw[CH + c + j], xf, a1);
                    a2 = fmaf(hw[2 * CH + c + j], xf, a2);
                }
            }
        }
        hpv[(b * 3 + 0) * 42 + cell] = fmaxf(a0 + hb[0], 0.0f);
        hpv[(b * 3 + 1) * 42 + cell] = fmaxf(a1 + hb[1], 0.0f);
        hpv[(b * 3 + 2) * 42 + cell] = fmaxf(a2 + hb[2], 0.0f);
    }
    __syncthreads();
    {
        const float* w1t = reinterpret_cast<const float*>(blob + G_W1T);
        float a[HB_];
#pragma unroll
        for (int b = 0; b < HB_; ++b) a[b] = 0.0f;
        for (int i = 0; i < 42; ++i) {
            const float wv = w1t[i * 256 + tid];
#pragma unroll
            for (int b = 0; b < HB_; ++b) a[b] = fmaf(wv, hpv[(b * 3 + 2) * 42 + i], a[b]);
        }
        const float b1 = reinterpret_cast<const float*>(blob + G_B1)[tid];
        const float w2v = reinterpret_cast<const float*>(blob + G_W2)[tid];
#pragma unroll
        for (int b = 0; b < HB_; ++b) hid[tid * HB_ + b] = fmaxf(a[b] + b1, 0.0f) * w2v;
    }
    __syncthreads();
    if (tid < HB_ * 8) {
        const int b = tid >> 3, c = tid & 7;
        const bool live = board0 + b < n;
        const long gidx = live ? leaf_of(list, board0 + b) : 0;
        uint64_t occ = 0;
        if (live) occ = leaf_bb[2 * gidx] | leaf_bb[2 * gidx + 1];
        const int height = c < 7 ? __popcll((occ >> (7 * c)) & 0x3Full) : 6;
        const bool legal = c < 7 && height < 6;
        float logit = -INFINITY;
        if (legal) {
            const int cell = height * 7 + c;
            const float* wrow = reinterpret_cast<const float*>(blob + G_WP) + cell * 84;
            float s = reinterpret_cast<const float*>(blob + G_BP)[cell];
            for (int j = 0; j < 84; ++j) s = fmaf(wrow[j], hpv[(b * 3 + j / 42) * 42 + j % 42], s);
            logit = s;
        }
        float m = logit;
        m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
        const float e = legal ? __expf(logit - m) : 0.0f;
        float sum = e;
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
        if (c < 7 && live) p_out[(size_t)gidx * 7 + c] = legal ? e / sum : 0.0f;
    } else if (tid >= 128 && tid < 128 + HB_) {
        const int b = tid - 128;
        float s = 0.0f;
        for (int u = 0; u < 256; ++u) s += hid[u * HB_ + b];
        if (board0 + b < n) v_out[leaf_of(list, board0 + b)] = tanhf(s + reinterpret_cast<const float*>(blob + G_B2)[0]);
    }
}

extern "C" {

int az_c4net128_blob_bytes(void) { return G_BLOB_BYTES; }

// one activation tensor, padded to whole workgroups of boards, then the list of leaves to evaluate
static size_t act_bytes(int32_t n) { return (act_positions(n) * CH * 2 + 255) / 256 * 256; }
long az_c4net128_workspace_bytes(int32_t n) { return n < 0 ? -1 : (long)(act_bytes(n) + ((size_t)n + 1) * 4); }

int az_c4net128_bf16_forward(const uint8_t* blob, int32_t num_blocks, const uint64_t* leaf_bb,
                             const int32_t* leaf_info, int32_t n, void* workspace, float* p, float* v, void* stream) {
    if (n < 0 || num_blocks < 0 || (n && (!blob || !leaf_bb || !workspace || !p || !v)))
        return az_fail("az_c4net128_bf16_forward: null pointer or negative count");
    if (n == 0) return 0;
    // k_compact128 keeps 32 flags per thread
    if (leaf_info && n > 32768) return az_fail("az_c4net128_bf16_forward: at most 32768 leaves per call with leaf_info");
    hipStream_t st = (hipStream_t)stream;
    __bf16* t0 = reinterpret_cast<__bf16*>(workspace);
    int32_t* list = leaf_info ? reinterpret_cast<int32_t*>(static_cast<uint8_t*>(workspace) + act_bytes(n)) : nullptr;
    if (list) hipLaunchKernelGGL(k_compact128, dim3(1), dim3(1024), 0, st, leaf_info, (int)n, list);
    const int lds = T_LDS_BYTES;
    static std::atomic<unsigned long long> attr_done{0};
    if (int rc = az_set_kernel_lds(reinterpret_cast<const void*>(k_tower128), lds, attr_done)) return rc;
    const int ngroups = (n + GB - 1) / GB;
    const long stem_threads = (long)act_positions(n) * 8;
    // grids cover all n leaves; workgroups past the compacted count leave at once
    hipLaunchKernelGGL(k_stem128, dim3((unsigned)((stem_threads + 255) / 256)), dim3(256), 0, st, blob, leaf_bb,
                       (const int32_t*)list, (int)n, t0);
    const float* bias = reinterpret_cast<const float*>(blob + G_BIAS);
    if (num_blocks > 0)
        hipLaunchKernelGGL(k_tower128, dim3(ngroups), dim3(NT), lds, st, blob + G_W, bias, t0, (int)num_blocks,
                           (const int32_t*)list, (int)n);
    hipLaunchKernelGGL(k_heads128, dim3((n + HB_ - 1) / HB_), dim3(256), 0, st, blob, t0, leaf_bb, (const int32_t*)list,
                       (int)n, p, v);
    AZ_LAUNCH_CHECK("k_stem128 / k_tower128 / k_heads128");
    return 0;
}

}  // extern "C"
