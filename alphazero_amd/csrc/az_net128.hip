// az_net128.hip -- 128-channel Connect Four network on bf16 MFMA (gfx950).
//
// The 10x128 network (alphazero/models.py:70-87 with block_channels = 128) has 295 KB of bf16
// weights per 3x3 conv: they fit neither registers nor LDS, so unlike the 32-channel kernel
// (az_net.hip) the weights stream from L2 while the activations of six boards stay in LDS for the
// whole residual tower.  Per conv the algorithmic work is 2*9*128*128*42 = 12.4 MFLOP per leaf.
//
// Three launches: stem (bitboards -> bf16 activations in HBM), tower (in place), heads.
// Activation tensor: [board][42 cells, cell = 7*row + column][128 ch] bf16, dense.  The tower keeps
// the same dense rows in LDS: 6 boards = 252 rows fill 8 MFMA tiles of 32 positions with 4 idle
// rows, and a 3x3 tap that leaves the board reads a shared all-zero row instead (a per-lane select
// from a 9-bit validity mask), so no matrix-core work is spent on padding.
// D[co][pos] orientation puts 4 consecutive channels of one position in 4 registers: the epilogue
// (bias, residual, ReLU, bf16) moves 8 bytes per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/azhip.h"
#include "az_host.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CH 128
#define GB 6                        // boards per workgroup
#define BPOS 42                     // positions per board
#define NROW (GB * BPOS)            // 252 real rows per slab
#define ZROW0 256                   // 16 all-zero rows, one per LDS slot residue (row mod 16)
#define ROWB 272                    // LDS row stride in bytes: 256 + 16 (17 slots of 16 B, odd)
#define NT 512

// blob layout (bytes); must match alphazero_amd/net.py pack_c4_blob_128
#define G_W 0                                        // bf16 [2 convs][9 taps][4 co tiles][8 k-steps][64 lanes][8]
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

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

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

// ---- residual tower: 2 * num_blocks 3x3 convs 128 -> 128, activations resident in LDS ----------
// Workgroup = 8 waves, 6 boards.  Two activation slabs live in LDS for the whole tower: the first
// conv of a block reads slab X and writes slab Y, the second reads Y, adds the block input (still in
// X) and overwrites X in place -- HBM sees one read and one write of the activations per tower, not
// per conv.  Wave (wc, wp) owns channels [32 wc, 32 wc + 32) of rows [128 wp, 128 wp + 128): four
// accumulators fed by one weight fragment.  Weights never touch LDS: they are stored in fragment
// order [conv][tap][32-channel tile][k-step][lane][8 bf16] and every lane reads its 16 bytes
// straight from L2 one tap ahead of use, alternating between two register buffers; the prefetch
// runs across conv boundaries, so the only bubble per conv is the epilogue and its barrier.
// The 18 taps of a block are unrolled, so tap geometry, buffer roles and epilogue kind are static.
#define CONV_BYTES (9 * CH * CH * 2)
#define NTILE 4
#define SLAB_ROWS (ZROW0 + 16)           // 252 real rows, 4 landing rows for the idle lanes, 16 zero rows
#undef SLAB_BYTES
#define SLAB_BYTES (SLAB_ROWS * ROWB)
#define BIAS_OFF (2 * SLAB_BYTES)

struct TowerCtx {
    const uint8_t* wts;
    uint8_t* lds;
    int qrow[NTILE];     // byte offset of the lane's row in a slab, per tile
    int vmask[NTILE];    // bit k: tap k of that row stays on the board (0 for idle lanes)
    int wc, h, lane;
};

template <int CONV, int TAP>
__device__ __forceinline__ void load_wfrag(uint4 (&buf)[8], const TowerCtx& c) {
    // uniform base + one 32-bit lane offset (kept opaque so that 18 taps of 64-bit addresses are not
    // precomputed outside the block loop); the k-step is the instruction's immediate offset
    const uint8_t* base = c.wts + (size_t)CONV * CONV_BYTES + TAP * (4 * 8 * 64 * 16);
    uint32_t voff = (uint32_t)(c.wc * 8 * 64 + c.lane) * 16;
    asm volatile("" : "+v"(voff));
#pragma unroll
    for (int s = 0; s < 8; ++s) buf[s] = *reinterpret_cast<const uint4*>(base + (s >> 2) * 4096 + voff + (s & 3) * 1024);
}

template <int TAP>
__device__ __forceinline__ void tap_offsets(int (&off)[NTILE], const TowerCtx& c) {
    // An off-board tap reads the zero row with the same residue mod 16 as the row it replaces: the
    // 16 lanes of a ds_read_b128 group then still fall on 16 different 16-byte slots (row stride =
    // 17 slots), which one shared zero row would break for every group with a border cell.
    // (Tiles start at multiples of 32 rows, so that residue is the same for the lane's four tiles.)
    constexpr int delta = (TAP / 3 - 1) * 7 + (TAP % 3 - 1);
    int col = c.lane;
    asm volatile("" : "+v"(col));        // recomputed per tap: hoisting 18 taps of offsets out of the block loop would spill
    const int zoff = (ZROW0 + ((col + delta) & 15)) * ROWB;
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
        int m = c.vmask[t];
        asm volatile("" : "+v"(m));
        off[t] = ((m >> TAP) & 1) ? c.qrow[t] + delta * ROWB : zoff;
    }
}

// One tap = 8 k-steps of 4 MFMAs.  Activation fragments are read one k-step ahead of the MFMAs that
// consume them (b0 / b1 alternate), and the last k-step already fetches the next tap's first
// fragments, so an LDS round trip is never on the matrix pipe's critical path.  On entry b0 holds
// k-step 0 of this tap.  Tap 0 starts the accumulators from zero; tap 8 ends with the epilogue
// (bias, residual on the second conv of a block, ReLU, bf16, write to the other slab) and a barrier.
template <int CONV, int TAP>
__device__ __forceinline__ void tower_tap(f32x16 (&acc)[NTILE], const uint4 (&cur)[8], uint4 (&nxt)[8], bf16x8 (&b0)[NTILE],
                                          const TowerCtx& c, bool more) {
    constexpr bool SECOND = CONV == 1;
    if (TAP < 8) load_wfrag<CONV, (TAP + 1) % 9>(nxt, c);
    else if (more) load_wfrag<1 - CONV, 0>(nxt, c);
    const uint8_t* src = c.lds + CONV * SLAB_BYTES + c.h * 16;
    int off[NTILE], offn[NTILE];
    tap_offsets<TAP>(off, c);
    if (TAP < 8) tap_offsets<(TAP + 1) % 9>(offn, c);
    bf16x8 b1[NTILE];
#pragma unroll
    for (int s = 0; s < 8; s += 2) {
        bf16x8 a;
#pragma unroll
        for (int t = 0; t < NTILE; ++t) b1[t] = *reinterpret_cast<const bf16x8*>(src + off[t] + (s + 1) * 32);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_memcpy(&a, &cur[s], 16);
        if (TAP == 0 && s == 0) {
            const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < NTILE; ++t) acc[t] = MFMA(a, b0[t], zero);
        } else {
#pragma unroll
            for (int t = 0; t < NTILE; ++t) acc[t] = MFMA(a, b0[t], acc[t]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < 8) {
#pragma unroll
            for (int t = 0; t < NTILE; ++t) b0[t] = *reinterpret_cast<const bf16x8*>(src + off[t] + (s + 2) * 32);
        } else if (TAP < 8) {
#pragma unroll
            for (int t = 0; t < NTILE; ++t) b0[t] = *reinterpret_cast<const bf16x8*>(src + offn[t]);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_memcpy(&a, &cur[s + 1], 16);
#pragma unroll
        for (int t = 0; t < NTILE; ++t) acc[t] = MFMA(a, b1[t], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (TAP < 8) return;

    // epilogue: register r = channel 32*wc + (r&3) + 8*(r>>2) + 4*h of the lane's row
    uint8_t* dst = c.lds + (SECOND ? 0 : SLAB_BYTES) + (32 * c.wc + 4 * c.h) * 2;
    const float* bp = reinterpret_cast<const float*>(c.lds + BIAS_OFF) + CONV * CH + 32 * c.wc + 4 * c.h;
    float4 bv[4];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) bv[gq] = *reinterpret_cast<const float4*>(bp + 8 * gq);
    uint2 r[NTILE][4];
    if (SECOND) {
#pragma unroll
        for (int t = 0; t < NTILE; ++t)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) r[t][gq] = *reinterpret_cast<const uint2*>(dst + c.qrow[t] + 16 * gq);
    }
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            float x0 = acc[t][4 * gq] + bv[gq].x, x1 = acc[t][4 * gq + 1] + bv[gq].y;
            float x2 = acc[t][4 * gq + 2] + bv[gq].z, x3 = acc[t][4 * gq + 3] + bv[gq].w;
            if (SECOND) {
                x0 += __uint_as_float(r[t][gq].x << 16); x1 += __uint_as_float(r[t][gq].x & 0xFFFF0000u);
                x2 += __uint_as_float(r[t][gq].y << 16); x3 += __uint_as_float(r[t][gq].y & 0xFFFF0000u);
            }
            *reinterpret_cast<uint2*>(dst + c.qrow[t] + 16 * gq) = make_uint2(relu_bf16x2(x0, x1), relu_bf16x2(x2, x3));
        }
    }
    __syncthreads();
    if (more) {                                // first fragments of the next conv, from the slab just written
        const uint8_t* src2 = c.lds + (SECOND ? 0 : SLAB_BYTES) + c.h * 16;
        tap_offsets<0>(off, c);
#pragma unroll
        for (int t = 0; t < NTILE; ++t) b0[t] = *reinterpret_cast<const bf16x8*>(src2 + off[t]);
    }
}

__global__ void __launch_bounds__(NT, 2)
k_tower128(const uint8_t* __restrict__ wts /* fragment order, two convs */, const float* __restrict__ bias /* [3][128] */,
           __bf16* __restrict__ act, int num_blocks, const int32_t* __restrict__ list, int nleaves) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // [2][SLAB_ROWS][ROWB], then f32 [2][128] biases
    if ((int)blockIdx.x * GB >= boards_to_run(list, nleaves)) return;   // whole workgroup, before any barrier
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 31;
    const int wp = wave >> 2;
    TowerCtx c;
    c.wts = wts; c.lds = lds; c.wc = wave & 3; c.h = lane >> 5; c.lane = lane;

    uint4 bufa[8], bufb[8];
    load_wfrag<0, 0>(bufa, c);
    if (tid < 2 * CH) reinterpret_cast<float*>(lds + BIAS_OFF)[tid] = bias[CH + tid];   // conv biases

    // stage the six boards: 252 rows of 16 pieces of 16 B, contiguous in HBM; every load is issued
    // before the first LDS write so the latencies overlap
    uint4* gsl = reinterpret_cast<uint4*>(act + (size_t)blockIdx.x * NROW * CH);
    {
        constexpr int TOTAL = NROW * 16;
        constexpr int NLD = (TOTAL + NT - 1) / NT;
        uint4 tmp[NLD];
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * NT;
            tmp[j] = i < TOTAL ? gsl[i] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * NT;
            if (i < TOTAL) *reinterpret_cast<uint4*>(lds + (i >> 4) * ROWB + (i & 15) * 16) = tmp[j];
        }
        // zero rows of both slabs: 2 x 16 rows x 16 pieces
        *reinterpret_cast<uint4*>(lds + (tid >> 8) * SLAB_BYTES + (ZROW0 + ((tid >> 4) & 15)) * ROWB + (tid & 15) * 16) = make_uint4(0, 0, 0, 0);
    }

    // this wave's four row tiles.  The four lanes past row 251 read zero rows for every tap and
    // write their results to the landing rows 252..255, so the epilogue needs no lane mask.
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
        const int R = (wp * NTILE + t) * 32 + col;
        const int o = R % BPOS, y = o / 7, x = o - 7 * y;
        int m = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            m |= (yy >= 0 && yy < 6 && xx >= 0 && xx < 7) << tap;
        }
        c.qrow[t] = R * ROWB;
        c.vmask[t] = R < NROW ? m : 0;
    }
    f32x16 acc[NTILE];
    __syncthreads();
    bf16x8 b0[NTILE];
    {
        int off[NTILE];
        tap_offsets<0>(off, c);
#pragma unroll
        for (int t = 0; t < NTILE; ++t) b0[t] = *reinterpret_cast<const bf16x8*>(lds + c.h * 16 + off[t]);
    }

#pragma unroll 1
    for (int blk = 0; blk < num_blocks; ++blk) {
        const bool more = blk + 1 < num_blocks;
        tower_tap<0, 0>(acc, bufa, bufb, b0, c, true);
        tower_tap<0, 1>(acc, bufb, bufa, b0, c, true);
        tower_tap<0, 2>(acc, bufa, bufb, b0, c, true);
        tower_tap<0, 3>(acc, bufb, bufa, b0, c, true);
        tower_tap<0, 4>(acc, bufa, bufb, b0, c, true);
        tower_tap<0, 5>(acc, bufb, bufa, b0, c, true);
        tower_tap<0, 6>(acc, bufa, bufb, b0, c, true);
        tower_tap<0, 7>(acc, bufb, bufa, b0, c, true);
        tower_tap<0, 8>(acc, bufa, bufb, b0, c, true);
        tower_tap<1, 0>(acc, bufb, bufa, b0, c, true);
        tower_tap<1, 1>(acc, bufa, bufb, b0, c, true);
        tower_tap<1, 2>(acc, bufb, bufa, b0, c, true);
        tower_tap<1, 3>(acc, bufa, bufb, b0, c, true);
        tower_tap<1, 4>(acc, bufb, bufa, b0, c, true);
        tower_tap<1, 5>(acc, bufa, bufb, b0, c, true);
        tower_tap<1, 6>(acc, bufb, bufa, b0, c, true);
        tower_tap<1, 7>(acc, bufa, bufb, b0, c, true);
        tower_tap<1, 8>(acc, bufb, bufa, b0, c, more);
    }

    for (int i = tid; i < NROW * 16; i += NT)
        gsl[i] = *reinterpret_cast<const uint4*>(lds + (i >> 4) * ROWB + (i & 15) * 16);
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
    const int lds = BIAS_OFF + 2 * CH * 4;
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
