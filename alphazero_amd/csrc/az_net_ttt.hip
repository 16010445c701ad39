// Tic-tac-toe policy/value network (alphazero/models.py:6-87 with shape_in [2,3,3], shape_out [9]:
// alphazero/config/tictactoe.yaml:4-11) in fp32, one launch from bitboards to soft-maxed priors and
// value.  The board is 3x3 and the trunk has 4..16 channels, far too small for the matrix cores: a
// lane owns one output channel of one board, a wavefront 64/C boards, and the 49 (cell, tap) pairs
// of a padded 3x3 convolution on a 3x3 board are unrolled as fused multiply-adds.  All weights sit in
// LDS for the whole launch; a board's activations live in a wave-private LDS strip, so there is no
// workgroup barrier after set-up.  Eval-mode batch-norm is folded into the weights by the host packer.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/azhip.h"
#include "az_host.h"

namespace {

constexpr int TT_THREADS = 256;
constexpr int TT_WAVES = TT_THREADS / 64;
constexpr int TP = 12;                      // floats per (channel, board) row: 9 cells + 3 of padding

// blob layout in floats (the host packer writes exactly this)
template <int C>
struct L {
    static constexpr int STEM = 0;                        // [2 planes][C co][TP taps]
    static constexpr int C1 = STEM + 2 * C * TP;          // [C ci][C co][TP taps]
    static constexpr int C2 = C1 + C * C * TP;
    static constexpr int BIAS = C2 + C * C * TP;          // [3][C]: stem, conv1, conv2 (folded BN shift)
    static constexpr int HW = BIAS + 3 * C;               // [3][C]: 1x1 convs: policy 0, policy 1, value
    static constexpr int HB = HW + 3 * C;                 // [4]
    static constexpr int PFC = HB + 4;                    // [9][20]: policy FC rows, 18 inputs + 2 pad
    static constexpr int PB = PFC + 9 * 20;               // [12]
    static constexpr int V1 = PB + 12;                    // [256][TP]: value FC1 rows, 9 inputs + 3 pad
    static constexpr int VB1 = V1 + 256 * TP;             // [256]
    static constexpr int V2 = VB1 + 256;                  // [256]
    static constexpr int VB2 = V2 + 256;                  // [4]
    static constexpr int TOTAL = VB2 + 4;
    static constexpr int BPW = 64 / C;                    // boards per wavefront
    static constexpr int STRIP = C * TP + 4;              // floats per board strip; +4 staggers the banks of the boards
};

__device__ __forceinline__ void lds_order() {
    // the strips are shared between the lanes of one wavefront only
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// out[cell] += sum over taps of w[tap] * a[neighbour]; zero padding outside the 3x3 board
__device__ __forceinline__ void taps3x3(const float (&w)[TP], const float (&a)[TP], float (&out)[9]) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
                for (int dc = -1; dc <= 1; ++dc)
                    if (r + dr >= 0 && r + dr < 3 && c + dc >= 0 && c + dc < 3)
                        out[r * 3 + c] = fmaf(w[(dr + 1) * 3 + dc + 1], a[(r + dr) * 3 + c + dc], out[r * 3 + c]);
}

__device__ __forceinline__ void load12(const float* src, float (&dst)[TP]) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    const float4 x = s4[0], y = s4[1], z = s4[2];
    dst[0] = x.x; dst[1] = x.y; dst[2] = x.z; dst[3] = x.w;
    dst[4] = y.x; dst[5] = y.y; dst[6] = y.z; dst[7] = y.w;
    dst[8] = z.x; dst[9] = z.y; dst[10] = z.z; dst[11] = z.w;
}

__device__ __forceinline__ void store_row(float* dst, const float (&x)[9]) {
    float4* d4 = reinterpret_cast<float4*>(dst);
    d4[0] = make_float4(x[0], x[1], x[2], x[3]);
    d4[1] = make_float4(x[4], x[5], x[6], x[7]);
    d4[2] = make_float4(x[8], 0.0f, 0.0f, 0.0f);
}

// one 3x3 convolution C -> C for this lane's (board, co): weights [ci][co][TP], input strip [ci][TP]
template <int C>
__device__ __forceinline__ void conv_cc(const float* w, const float* strip, int co, float (&out)[9]) {
#pragma unroll 4
    for (int ci = 0; ci < C; ++ci) {
        float wv[TP], av[TP];
        load12(w + (ci * C + co) * TP, wv);
        load12(strip + ci * TP, av);
        taps3x3(wv, av, out);
    }
}

template <int C>
__global__ void __launch_bounds__(TT_THREADS)
k_tttnet_f32(const float* __restrict__ blob, int num_blocks, const uint64_t* __restrict__ leaf_bb,
             const int32_t* __restrict__ leaf_info, int n, float* __restrict__ p_out, float* __restrict__ v_out) {
    using Lc = L<C>;
    constexpr int BPW = Lc::BPW, STRIP = Lc::STRIP;
    extern __shared__ __attribute__((aligned(16))) float tt_lds[];
    float* W = tt_lds;                                                   // the whole blob
    float* strips = tt_lds + ((Lc::TOTAL + 3) & ~3);                     // [wave][2][BPW][STRIP]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < Lc::TOTAL / 4; i += TT_THREADS)
        reinterpret_cast<float4*>(W)[i] = reinterpret_cast<const float4*>(blob)[i];
    for (int i = tid; i < TT_WAVES * 2 * BPW * STRIP; i += TT_THREADS) strips[i] = 0.0f;
    __syncthreads();

    const int co = lane % C, b = lane / C;
    float* sx = strips + ((wave * 2 + 0) * BPW + b) * STRIP;             // block input / output
    float* sy = strips + ((wave * 2 + 1) * BPW + b) * STRIP;             // conv1 output, then head scratch
    const float b0 = W[Lc::BIAS + co], b1 = W[Lc::BIAS + C + co], b2 = W[Lc::BIAS + 2 * C + co];
    const int tasks = (n + BPW - 1) / BPW;
    for (int task = (int)blockIdx.x * TT_WAVES + wave; task < tasks; task += (int)gridDim.x * TT_WAVES) {
        const int row = task * BPW + b;
        bool live = row < n;
        if (live && leaf_info) {
            const int kind = leaf_info[row] & 0xFF;
            live = kind == AZ_LEAF_EVAL || kind == AZ_LEAF_ROOT;
        }
        uint32_t nx = 0, pv = 0;
        if (live) { nx = (uint32_t)leaf_bb[2 * (size_t)row]; pv = (uint32_t)leaf_bb[2 * (size_t)row + 1]; }

        // stem: planes are the bitboards themselves (plane 0 = side to move, alphazero/games/tictactoe.py:52-62)
        float x[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) x[i] = b0;
        {
            float wv[TP], av[TP];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const uint32_t bits = pl ? pv : nx;
#pragma unroll
                for (int i = 0; i < TP; ++i) av[i] = (i < 9 && ((bits >> i) & 1u)) ? 1.0f : 0.0f;
                load12(W + Lc::STEM + (pl * C + co) * TP, wv);
                taps3x3(wv, av, x);
            }
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) x[i] = fmaxf(x[i], 0.0f);
        store_row(sx + co * TP, x);
        lds_order();

        // tower: the same residual block num_blocks times (alphazero/models.py:78)
        for (int blk = 0; blk < num_blocks; ++blk) {
            float y[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) y[i] = b1;
            conv_cc<C>(W + Lc::C1, sx, co, y);
#pragma unroll
            for (int i = 0; i < 9; ++i) y[i] = fmaxf(y[i], 0.0f);
            store_row(sy + co * TP, y);
            lds_order();
            float z[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) z[i] = b2;
            conv_cc<C>(W + Lc::C2, sy, co, z);
#pragma unroll
            for (int i = 0; i < 9; ++i) x[i] = fmaxf(z[i] + x[i], 0.0f);
            lds_order();                                   // every lane has read sx before it changes
            store_row(sx + co * TP, x);
            lds_order();
        }

        // heads.  1x1 convs: 18 policy planes values + 9 value plane values per board into sy[0..26]
        for (int o = co; o < 27; o += C) {
            const int j = o / 9, cell = o - 9 * j;
            float s = W[Lc::HB + j];
            for (int ci = 0; ci < C; ++ci) s = fmaf(W[Lc::HW + j * C + ci], sx[ci * TP + cell], s);
            sy[o] = fmaxf(s, 0.0f);
        }
        lds_order();
        // policy FC 18 -> 9 (flatten order plane*9 + cell), logits into sy[28..36]
        for (int a = co; a < 9; a += C) {
            float s = W[Lc::PB + a];
            for (int k = 0; k < 18; ++k) s = fmaf(W[Lc::PFC + a * 20 + k], sy[k], s);
            sy[28 + a] = s;
        }
        // value FC 9 -> 256 (ReLU) -> 1 (tanh): this lane's share of the hidden units
        float hv[TP];
        load12(sy + 16, hv);                               // sy[18..26] are elements 2..10 of this window
        float acc = 0.0f;
        for (int h = co; h < 256; h += C) {
            float wv[TP];
            load12(W + Lc::V1 + h * TP, wv);
            float t = W[Lc::VB1 + h];
#pragma unroll
            for (int i = 0; i < 9; ++i) t = fmaf(wv[i], hv[2 + i], t);
            acc = fmaf(W[Lc::V2 + h], fmaxf(t, 0.0f), acc);
        }
#pragma unroll
        for (int d = 1; d < C; d <<= 1) acc += __shfl_xor(acc, d);
        const float value = tanhf(acc + W[Lc::VB2]);
        lds_order();
        // soft-max over the empty squares (alphazero/mcts.py:101-106)
        const uint32_t legal = ~(nx | pv) & 0x1FFu;
        float m = -INFINITY;
#pragma unroll
        for (int a = 0; a < 9; ++a) if ((legal >> a) & 1u) m = fmaxf(m, sy[28 + a]);
        float sum = 0.0f;
#pragma unroll
        for (int a = 0; a < 9; ++a) if ((legal >> a) & 1u) sum += expf(sy[28 + a] - m);
        if (live) {
            for (int a = co; a < 9; a += C)
                p_out[(size_t)row * 9 + a] = ((legal >> a) & 1u) ? expf(sy[28 + a] - m) / sum : 0.0f;
            if (co == 0) v_out[row] = value;
        }
        lds_order();                                       // sy and sx are rewritten by the next task
    }
}

template <int C>
int launch(const float* blob, int num_blocks, const uint64_t* leaf_bb, const int32_t* leaf_info, int n, float* p,
           float* v, hipStream_t st) {
    using Lc = L<C>;
    const int lds = (((Lc::TOTAL + 3) & ~3) + TT_WAVES * 2 * Lc::BPW * Lc::STRIP) * (int)sizeof(float);
    static std::atomic<unsigned long long> attr_done{0};      // one bitmap per channel width
    if (int rc = az_set_kernel_lds(reinterpret_cast<const void*>(k_tttnet_f32<C>), lds, attr_done)) return rc;
    const int tasks = (n + Lc::BPW - 1) / Lc::BPW;
    int grid = (tasks + TT_WAVES - 1) / TT_WAVES;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL((k_tttnet_f32<C>), dim3(grid), dim3(TT_THREADS), lds, st, blob, num_blocks, leaf_bb, leaf_info,
                       n, p, v);
    AZ_LAUNCH_CHECK("k_tttnet_f32");
    return 0;
}

}  // namespace

extern "C" {

int az_tttnet_f32_blob_floats(int32_t channels) {
    switch (channels) {
        case 4: return L<4>::TOTAL;
        case 8: return L<8>::TOTAL;
        case 16: return L<16>::TOTAL;
        default: return -1;
    }
}

int az_tttnet_f32_forward(const float* blob, int32_t channels, int32_t num_blocks, const uint64_t* leaf_bb,
                          const int32_t* leaf_info, int32_t n, float* p, float* v, void* stream) {
    if (n < 0 || num_blocks < 0 || (n && (!blob || !leaf_bb || !p || !v)))
        return az_fail("az_tttnet_f32_forward: null pointer or negative count");
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    switch (channels) {
        case 4: return launch<4>(blob, num_blocks, leaf_bb, leaf_info, n, p, v, st);
        case 8: return launch<8>(blob, num_blocks, leaf_bb, leaf_info, n, p, v, st);
        case 16: return launch<16>(blob, num_blocks, leaf_bb, leaf_info, n, p, v, st);
        default: return az_fail("az_tttnet_f32_forward: channels must be 4, 8 or 16");
    }
}

}  // extern "C"
