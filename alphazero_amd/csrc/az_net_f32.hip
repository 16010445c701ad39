// az_net_f32.hip -- the fused Connect Four network forward in full fp32 (gfx950).
//
// Same network and same LDS-resident structure as az_net.hip, but every value (weights,
// activations, accumulators) is float32 and every contraction of the trunk runs on
// v_mfma_f32_32x32x2_f32, whose result is bit-for-bit a k-ordered fp32 fmaf chain.  This is the
// parity path: outputs agree with the fp32 torch statement of the network
// (alphazero/models.py:70-87, alphazero/games/connectfour.py:54-72) to ~1e-6; it runs at the fp32
// vector rate (1/16 of the bf16 matrix rate) and is not the throughput path.
//
// Workgroup = 4 waves = 8 boards, wave w owns boards 2w, 2w+1 for the whole tower (no cross-wave
// data dependencies inside the tower); the current conv's weights are staged in LDS for all waves,
// which is the only reason for the two barriers per conv.
// LDS image: position stride 33 floats (32 channels + 1 pad) so that 32 lanes reading one channel
// of 32 consecutive positions, or 32 channels of one position, hit 32 different banks.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/azhip.h"
#include "az_host.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

#define FB 8                         // boards per workgroup
#define FSTRIDE 33                   // floats per position
#define FBOARD_POS 56
#define FPOS0 9
#define FNPOS (1 + FB * FBOARD_POS + 8 + 1)
#define FACT_FLOATS (FNPOS * FSTRIDE)
#define FTHREADS 256
#define FTILES 3                     // per wave: 2 boards * 48 / 32

// blob layout in floats; must match alphazero_amd/net.py pack_c4_blob_f32
#define FO_STEM 0                    // [18][32]   k = tap*2 + plane
#define FO_C1 (FO_STEM + 18 * 32)    // [288][32]  k = tap*32 + ci
#define FO_C2 (FO_C1 + 288 * 32)
#define FO_BIAS (FO_C2 + 288 * 32)   // [3][32]
#define FO_HW (FO_BIAS + 96)         // [3][32]    1x1 head convs: policy 0, policy 1, value
#define FO_HB (FO_HW + 96)           // [4]
#define FO_WP (FO_HB + 4)            // [42][84]
#define FO_BP (FO_WP + 42 * 84)      // [44]
#define FO_W1T (FO_BP + 44)          // [42][256]
#define FO_B1 (FO_W1T + 42 * 256)    // [256]
#define FO_W2 (FO_B1 + 256)          // [256]
#define FO_B2 (FO_W2 + 256)          // [4]
#define F_BLOB_FLOATS (FO_B2 + 4)

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x16 f_bias_acc(const float* bias, int h) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias[(r & 3) + 8 * (r >> 2) + 4 * h];
    return acc;
}

template <bool RES>
__device__ __forceinline__ void f_store_tile(const f32x16& acc, float* dst, int pos, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float* q = dst + pos * FSTRIDE + (r & 3) + 8 * (r >> 2) + 4 * h;
        float x = acc[r];
        if (RES) x += *q;
        *q = fmaxf(x, 0.0f);
    }
}

// conv over the wave's three tiles; wl = staged weights [288][32]
template <bool RES>
__device__ __forceinline__ void f_conv3x3(const float* wl, const float* src, float* dst, const float* bias, int wave,
                                          int lane) {
    const int h = lane >> 5, col = lane & 31;
#pragma unroll 1
    for (int t = 0; t < FTILES; ++t) {
        const int R = (wave * FTILES + t) * 32 + col;
        const int b = R / 48, o = R - b * 48;
        const int pos = b * FBOARD_POS + FPOS0 + o;
        f32x16 acc = f_bias_acc(bias, h);
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const float* xs = src + (pos + (tap / 3 - 1) * 8 + (tap % 3 - 1)) * FSTRIDE + h;
            const float* ws = wl + (tap * 32 + h) * 32 + col;
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = MFMA32(ws[2 * s * 32], xs[2 * s], acc);
        }
        if ((o & 7) != 7) f_store_tile<RES>(acc, dst, pos, h);
    }
}

__global__ void __launch_bounds__(FTHREADS, 1)
k_c4net_f32(const float* __restrict__ blob, int num_blocks, const uint64_t* __restrict__ leaf_bb, int n,
            float* __restrict__ p_out, float* __restrict__ v_out) {
    extern __shared__ __attribute__((aligned(16))) float fl[];
    float* sbias = fl;                             // [3][32] + head bias [4]
    float* bufX = fl + 128;
    float* bufY = bufX + FACT_FLOATS;
    float* wl = bufY + FACT_FLOATS;                // [288][32] staged weights
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, h = lane >> 5, col = lane & 31;
    const int board0 = blockIdx.x * FB;

    for (int i = tid; i < 2 * FACT_FLOATS; i += FTHREADS) bufX[i] = 0.0f;
    if (tid < 96) sbias[tid] = blob[FO_BIAS + tid];
    if (tid < 4) sbias[96 + tid] = blob[FO_HB + tid];
    for (int i = tid; i < 18 * 32; i += FTHREADS) wl[i] = blob[FO_STEM + i];
    __syncthreads();

    // ---- stem: K = 18 = 9 taps x 2 planes; k-step s = tap, lane half h = plane
#pragma unroll 1
    for (int t = 0; t < FTILES; ++t) {
        const int R = (wave * FTILES + t) * 32 + col;
        const int b = R / 48, o = R - b * 48;
        const int pos = b * FBOARD_POS + FPOS0 + o;
        const int y = o >> 3, x = o & 7;
        const int gidx = board0 + b;
        uint64_t bb = 0;
        if (gidx < n) bb = leaf_bb[2 * (size_t)gidx + h] << 8;      // guard bits make out-of-board taps read 0
        f32x16 acc = f_bias_acc(sbias, h);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int sh = (x + tap % 3) * 7 + (y + tap / 3);
            const float bit = (float)((bb >> sh) & 1ull);
            acc = MFMA32(wl[(2 * tap + h) * 32 + col], bit, acc);
        }
        if (x != 7) f_store_tile<false>(acc, bufY, pos, h);
    }

    for (int blk = 0; blk < num_blocks; ++blk) {
        __syncthreads();
        for (int i = tid; i < 288 * 32; i += FTHREADS) wl[i] = blob[FO_C1 + i];
        __syncthreads();
        f_conv3x3<false>(wl, bufY, bufX, sbias + 32, wave, lane);
        __syncthreads();
        for (int i = tid; i < 288 * 32; i += FTHREADS) wl[i] = blob[FO_C2 + i];
        __syncthreads();
        f_conv3x3<true>(wl, bufX, bufY, sbias + 64, wave, lane);
    }
    __syncthreads();

    // ---- heads (plain fp32 on the vector ALU; bufX and wl are free now)
    float* hpv = bufX;                              // [FB][3][42]
    float* hid = wl;                                // [256][FB]
    for (int i = tid; i < FB * 42; i += FTHREADS) {
        const int b = i / 42, cell = i - b * 42;
        const float* xs = bufY + (b * FBOARD_POS + FPOS0 + (cell / 7) * 8 + cell % 7) * FSTRIDE;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int c = 0; c < 32; ++c) {
            const float xv = xs[c];
            a0 = fmaf(blob[FO_HW + c], xv, a0);
            a1 = fmaf(blob[FO_HW + 32 + c], xv, a1);
            a2 = fmaf(blob[FO_HW + 64 + c], xv, a2);
        }
        hpv[(b * 3 + 0) * 42 + cell] = fmaxf(a0 + sbias[96], 0.0f);
        hpv[(b * 3 + 1) * 42 + cell] = fmaxf(a1 + sbias[97], 0.0f);
        hpv[(b * 3 + 2) * 42 + cell] = fmaxf(a2 + sbias[98], 0.0f);
    }
    __syncthreads();
    {   // value FC 42 -> 256 (+ReLU), thread = hidden unit
        float a[FB];
#pragma unroll
        for (int b = 0; b < FB; ++b) a[b] = 0.0f;
        for (int i = 0; i < 42; ++i) {
            const float wv = blob[FO_W1T + i * 256 + tid];
#pragma unroll
            for (int b = 0; b < FB; ++b) a[b] = fmaf(wv, hpv[(b * 3 + 2) * 42 + i], a[b]);
        }
        const float b1 = blob[FO_B1 + tid], w2v = blob[FO_W2 + tid];
#pragma unroll
        for (int b = 0; b < FB; ++b) hid[tid * FB + b] = fmaxf(a[b] + b1, 0.0f) * w2v;
    }
    __syncthreads();
    if (tid < FB * 8) {
        // policy FC at the landing cells, soft-max over legal columns
        const int b = tid >> 3, c = tid & 7;
        const int gidx = board0 + b;
        uint64_t occ = 0;
        if (gidx < n) occ = leaf_bb[2 * (size_t)gidx] | leaf_bb[2 * (size_t)gidx + 1];
        const int height = c < 7 ? __popcll((occ >> (7 * c)) & 0x3Full) : 6;
        const bool legal = c < 7 && height < 6;
        float logit = -INFINITY;
        if (legal) {
            const int cell = height * 7 + c;
            float s = blob[FO_BP + cell];
            for (int j = 0; j < 84; ++j) s = fmaf(blob[FO_WP + cell * 84 + j], hpv[(b * 3 + j / 42) * 42 + j % 42], s);
            logit = s;
        }
        float m = logit;
        m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
        const float e = legal ? expf(logit - m) : 0.0f;
        float sum = e;
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
        if (c < 7 && gidx < n) p_out[(size_t)gidx * 7 + c] = legal ? e / sum : 0.0f;
    } else if (tid >= 128 && tid < 128 + FB) {
        const int b = tid - 128, gidx = board0 + b;
        float s = 0.0f;
        for (int u = 0; u < 256; ++u) s += hid[u * FB + b];
        if (gidx < n) v_out[gidx] = tanhf(s + blob[FO_B2]);
    }
}

extern "C" {

int az_c4net_f32_blob_floats(void) { return F_BLOB_FLOATS; }

int az_c4net_f32_forward(const float* blob, int32_t num_blocks, const uint64_t* leaf_bb, int32_t n, float* p,
                         float* v, void* stream) {
    if (n < 0 || num_blocks < 0 || (n && (!blob || !leaf_bb || !p || !v)))
        return az_fail("az_c4net_f32_forward: null pointer or negative count");
    if (n == 0) return 0;
    const int lds = (2 * FACT_FLOATS + 288 * 32 + 128) * 4;
    static std::atomic<unsigned long long> attr_done{0};
    if (int rc = az_set_kernel_lds(reinterpret_cast<const void*>(k_c4net_f32), lds, attr_done)) return rc;
    hipLaunchKernelGGL(k_c4net_f32, dim3((n + FB - 1) / FB), dim3(FTHREADS), lds, (hipStream_t)stream, blob,
                       (int)num_blocks, leaf_bb, (int)n, p, v);
    AZ_LAUNCH_CHECK("k_c4net_f32");
    return 0;
}

}  // extern "C"
