// az_net.hip -- fused Connect Four policy/value network forward for gfx950.
//
// The whole network of alphazero/models.py:70-87 + alphazero/games/connectfour.py:54-72
// (stem conv, tower of one residual block applied num_blocks times, policy and value
// heads, landing-cell gather, soft-max over legal columns) runs in ONE kernel for up to
// 32 trunk channels.  A workgroup takes 16 leaf positions straight from their bitboards,
// keeps every activation in LDS from the first conv to the last, and writes only
// p[7] and v per leaf: HBM traffic is 16 B in and 32 B out per leaf, so the kernel is
// bounded by the MFMA rate, not by memory.
//
// Convolutions are implicit GEMMs on the matrix cores: D[co][pos] = sum_k Wt[co][k] X[k][pos]
// with k = (tap, ci).  v_mfma_f32_32x32x16_bf16: A = folded weights, held in registers for
// the whole kernel (a lane's fragment of both tower convs is 144 VGPRs); B = activations
// read from LDS with one ds_read_b128 per MFMA.  Batch-norm (eval mode) is folded into the
// weights and a per-channel bias on the host.
//
// LDS activation image, per board: 66 positions x 32 channels bf16 (64 B per position).
// Positions form an 8-wide grid with one zero column and zero rows above and below, so a
// 3x3 tap is a constant position offset and needs no bounds test; 48 of them (6 rows x 8)
// are computed per conv, the 8th column is masked back to zero.  The four 16-byte channel
// chunks of a position are XOR-swizzled with (pos >> 2) & 3 so the 16 lanes of a
// ds_read_b128 group always hit 16 different 16-byte bank slots.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/azhip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define TB 16                    // boards per workgroup
#define NPOS 66                  // padded positions per board
#define POS0 9                   // first computed position (1 guard + one zero row of 8)
#define NTHREADS 512
#define NWAVES 8
#define TILES_PER_WAVE 3         // TB*48/32 = 24 tiles of 32 positions
#define ACT_BYTES (TB * NPOS * 64)

// blob layout (bytes); must match alphazero_amd/net.py
#define OFF_WF 0                                  // [3][18][64][8] bf16
#define OFF_HF (OFF_WF + 3 * 18 * 64 * 16)        // [2][64][8] bf16
#define OFF_BIAS (OFF_HF + 2 * 64 * 16)           // [3][32] f32
#define OFF_HB (OFF_BIAS + 3 * 32 * 4)            // [4] f32
#define OFF_WP (OFF_HB + 16)                      // [42][84] f32
#define OFF_BP (OFF_WP + 42 * 84 * 4)             // [44] f32 (42 used)
#define OFF_W1T (OFF_BP + 44 * 4)                 // [42][256] f32
#define OFF_B1 (OFF_W1T + 42 * 256 * 4)           // [256] f32
#define OFF_W2 (OFF_B1 + 256 * 4)                 // [256] f32
#define OFF_B2 (OFF_W2 + 256 * 4)                 // [4] f32
#define BLOB_BYTES (OFF_B2 + 16)

__device__ __forceinline__ int act_addr(int pos, int chunk) {
    return pos * 64 + ((chunk ^ ((pos >> 2) & 3)) << 4);
}

struct Weights { bf16x8 f[18]; };

__device__ __forceinline__ void load_weights(Weights& w, const uint8_t* blob, int conv, int lane) {
    const bf16x8* src = reinterpret_cast<const bf16x8*>(blob + OFF_WF) + (size_t)conv * 18 * 64 + lane;
#pragma unroll
    for (int s = 0; s < 18; ++s) w.f[s] = src[s * 64];
}

// one 3x3 conv over the workgroup's 16 boards: src -> dst (+ residual from dst itself when RES)
template <bool RES>
__device__ __forceinline__ void conv3x3(const Weights& w, const uint8_t* src, uint8_t* dst, const float* bias,
                                        int wave, int lane) {
    const int h = lane >> 5;
#pragma unroll 1
    for (int t = 0; t < TILES_PER_WAVE; ++t) {
        const int R = (wave * TILES_PER_WAVE + t) * 32 + (lane & 31);
        const int b = R / 48, o = R - b * 48;
        const int pos = b * NPOS + POS0 + o;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int p2 = pos + (tap / 3 - 1) * 8 + (tap % 3 - 1);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(src + act_addr(p2, h));
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(src + act_addr(p2, 2 + h));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.f[2 * tap], b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.f[2 * tap + 1], b1, acc, 0, 0, 0);
        }
        // D: column = position (lane & 31), register r = channel (r&3) + 8*(r>>2) + 4*h
        if ((o & 7) != 7) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bi = *reinterpret_cast<const float4*>(bias + 8 * g + 4 * h);
                float x0 = acc[4 * g] + bi.x, x1 = acc[4 * g + 1] + bi.y, x2 = acc[4 * g + 2] + bi.z,
                      x3 = acc[4 * g + 3] + bi.w;
                uint8_t* q = dst + act_addr(pos, g) + 8 * h;
                if (RES) {
                    const bf16x4 r = *reinterpret_cast<const bf16x4*>(q);
                    x0 += (float)r[0]; x1 += (float)r[1]; x2 += (float)r[2]; x3 += (float)r[3];
                }
                bf16x4 y;
                y[0] = (__bf16)fmaxf(x0, 0.0f); y[1] = (__bf16)fmaxf(x1, 0.0f);
                y[2] = (__bf16)fmaxf(x2, 0.0f); y[3] = (__bf16)fmaxf(x3, 0.0f);
                *reinterpret_cast<bf16x4*>(q) = y;
            }
        }
    }
}

__global__ void __launch_bounds__(NTHREADS, 2)
k_c4net_bf16(const uint8_t* __restrict__ blob, int num_blocks, const uint64_t* __restrict__ leaf_bb,
             const int32_t* __restrict__ leaf_info, int n, float* __restrict__ p_out, float* __restrict__ v_out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* bufX = lds;                       // conv1 output / scratch
    uint8_t* bufY = lds + ACT_BYTES;           // block input and output
    float* sbias = reinterpret_cast<float*>(lds + 2 * ACT_BYTES);   // [3][32] + head bias [4]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int board0 = blockIdx.x * TB;

    // ---- zero both images, stage biases ------------------------------------------------
    {
        uint4* z = reinterpret_cast<uint4*>(lds);
        for (int i = tid; i < 2 * ACT_BYTES / 16; i += NTHREADS) z[i] = make_uint4(0, 0, 0, 0);
        if (tid < 100) sbias[tid] = reinterpret_cast<const float*>(blob + OFF_BIAS)[tid];
    }
    Weights w1, w2;
    load_weights(w1, blob, 0, lane);           // stem first
    load_weights(w2, blob, 2, lane);
    __syncthreads();

    // ---- input planes from bitboards: channel 0 = side to move, 1 = previous mover ------
    {
        const int b = tid >> 5, j = tid & 31;
        const int gidx = board0 + b;
        uint64_t nx = 0, pv = 0;
        if (gidx < n) { nx = leaf_bb[2 * (size_t)gidx]; pv = leaf_bb[2 * (size_t)gidx + 1]; }
        for (int cell = j; cell < 42; cell += 32) {
            const int y = cell / 7, x = cell - 7 * y;
            const int bit = x * 7 + y;
            const int pos = b * NPOS + POS0 + y * 8 + x;
            const uint32_t two = (((nx >> bit) & 1ull) ? 0x3F80u : 0u) | (((pv >> bit) & 1ull) ? 0x3F800000u : 0u);
            *reinterpret_cast<uint32_t*>(bufX + act_addr(pos, 0)) = two;
        }
    }
    __syncthreads();

    // ---- stem, then the tower (one block, num_blocks times) -------------------------------
    conv3x3<false>(w1, bufX, bufY, sbias, wave, lane);
    load_weights(w1, blob, 1, lane);
    __syncthreads();
    for (int blk = 0; blk < num_blocks; ++blk) {
        conv3x3<false>(w1, bufY, bufX, sbias + 32, wave, lane);
        __syncthreads();
        conv3x3<true>(w2, bufX, bufY, sbias + 64, wave, lane);
        __syncthreads();
    }

    // ---- heads: both 1x1 convs as one 2-step MFMA; rows 0,1 = policy planes, 2 = value ------
    float* hbuf = reinterpret_cast<float*>(bufX);          // [TB][3][48] f32 (bufX is free now)
    {
        const bf16x8* hf = reinterpret_cast<const bf16x8*>(blob + OFF_HF) + lane;
        const bf16x8 a0 = hf[0], a1 = hf[64];
        const int h = lane >> 5;
#pragma unroll 1
        for (int t = 0; t < TILES_PER_WAVE; ++t) {
            const int R = (wave * TILES_PER_WAVE + t) * 32 + (lane & 31);
            const int b = R / 48, o = R - b * 48;
            const int pos = b * NPOS + POS0 + o;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(bufY + act_addr(pos, h));
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(bufY + act_addr(pos, 2 + h));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
            if (h == 0) {
                // all waves finished reading bufX before the last barrier; hbuf aliases it
                hbuf[(b * 3 + 0) * 48 + o] = fmaxf(acc[0] + sbias[96], 0.0f);
                hbuf[(b * 3 + 1) * 48 + o] = fmaxf(acc[1] + sbias[97], 0.0f);
                hbuf[(b * 3 + 2) * 48 + o] = fmaxf(acc[2] + sbias[98], 0.0f);
            }
        }
    }
    __syncthreads();

    float* hid = reinterpret_cast<float*>(bufY);           // [256][TB] value hidden layer
    if (tid < 256) {
        // value head FC 42 -> 256 (+ReLU): thread = hidden unit, 16 boards at once
        const float* w1t = reinterpret_cast<const float*>(blob + OFF_W1T);
        float a[TB];
#pragma unroll
        for (int b = 0; b < TB; ++b) a[b] = 0.0f;
        for (int i = 0; i < 42; ++i) {
            const float wv = w1t[i * 256 + tid];
            const int o = (i / 7) * 8 + (i % 7);
#pragma unroll
            for (int b = 0; b < TB; ++b) a[b] = fmaf(wv, hbuf[(b * 3 + 2) * 48 + o], a[b]);
        }
        const float b1 = reinterpret_cast<const float*>(blob + OFF_B1)[tid];
        const float w2v = reinterpret_cast<const float*>(blob + OFF_W2)[tid];
#pragma unroll
        for (int b = 0; b < TB; ++b) hid[tid * TB + b] = fmaxf(a[b] + b1, 0.0f) * w2v;
    } else if (tid < 256 + TB * 8) {
        // policy head: FC 84 -> 42 evaluated only at the 7 landing cells, soft-max over legal columns
        const int idx = tid - 256, b = idx >> 3, c = idx & 7;
        const int gidx = board0 + b;
        uint64_t occ = 0;
        if (gidx < n) occ = leaf_bb[2 * (size_t)gidx] | leaf_bb[2 * (size_t)gidx + 1];
        const int height = c < 7 ? __popcll((occ >> (7 * c)) & 0x3Full) : 6;
        const bool legal = c < 7 && height < 6;
        float logit = -INFINITY;
        if (legal) {
            const int cell = height * 7 + c;
            const float* wrow = reinterpret_cast<const float*>(blob + OFF_WP) + cell * 84;
            float s = reinterpret_cast<const float*>(blob + OFF_BP)[cell];
            for (int ch = 0; ch < 2; ++ch)
                for (int i = 0; i < 42; ++i)
                    s = fmaf(wrow[ch * 42 + i], hbuf[(b * 3 + ch) * 48 + (i / 7) * 8 + (i % 7)], s);
            logit = s;
        }
        float m = logit;
        m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
        const float e = legal ? __expf(logit - m) : 0.0f;
        float sum = e;
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
        if (c < 7 && gidx < n) p_out[(size_t)gidx * 7 + c] = legal ? e / sum : 0.0f;
    }
    __syncthreads();
    if (tid < TB * 16) {
        // value head FC 256 -> 1 + tanh: 16 lanes per board
        const int b = tid >> 4, part = tid & 15;
        float s = 0.0f;
        for (int u = part; u < 256; u += 16) s += hid[u * TB + b];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        const int gidx = board0 + b;
        if (part == 0 && gidx < n) v_out[gidx] = tanhf(s + reinterpret_cast<const float*>(blob + OFF_B2)[0]);
    }
    (void)leaf_info;
}

static thread_local std::string g_net_err;
extern "C" {

int az_c4net_blob_bytes(void) { return BLOB_BYTES; }

int az_c4net_bf16_forward(const uint8_t* blob, int32_t num_blocks, const uint64_t* leaf_bb,
                          const int32_t* leaf_info, int32_t n, float* p, float* v, void* stream) {
    if (n < 0 || num_blocks < 0 || (n && (!blob || !leaf_bb || !p || !v))) return -1;
    if (n == 0) return 0;
    const int lds_bytes = 2 * ACT_BYTES + 128 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_c4net_bf16), hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_bytes) != hipSuccess)
            return -2;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_c4net_bf16, dim3((n + TB - 1) / TB), dim3(NTHREADS), lds_bytes, (hipStream_t)stream, blob,
                       (int)num_blocks, leaf_bb, leaf_info, (int)n, p, v);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // extern "C"
