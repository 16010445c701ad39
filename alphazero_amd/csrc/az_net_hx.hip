// az_net_hx.hip -- the fused Connect Four policy/value network at fp32 grade on the fp16 matrix cores.
//
// Same network and same interface as az_net.hip (alphazero/models.py:70-87 + alphazero/games/connectfour.py:54-72:
// stem, tower of one residual block applied num_blocks times, both heads, landing-cell gather, soft-max over the
// legal columns; leaves given as bitboards, batch-norm folded on the host), but every operand of the trunk is
// carried as two fp16 halves and every product is three MFMAs (az_net_hx_dev.h): outputs within 1e-5 relative of
// the fp32 torch statement, like az_c4net_f32_forward, at about five times its rate.  A wave takes one leaf at a
// time through trunk and heads on its own; the waves of a launch share nothing but the weights in LDS, so a
// leaf's (p, v) do not depend on the batch it is evaluated in.
#include "az_net_hx_dev.h"
#include "az_pack_dev.h"
#include "az_host.h"

#define HXK_SBIAS HX_W_BYTES                         // LDS: float[128] biases
#define HXK_STRIPS (HXK_SBIAS + 512)                 // LDS: per-wave head planes
#define HXK_LDS_BYTES (HXK_STRIPS + 8 * HX_STRIP_BYTES)

__global__ void __launch_bounds__(NTHREADS, 2)
k_c4net_hx(const uint8_t* __restrict__ blob, int num_blocks, const uint64_t* __restrict__ leaf_bb,
           const int32_t* __restrict__ leaf_info, int n, float* __restrict__ p_out, float* __restrict__ v_out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    float* sbias = reinterpret_cast<float*>(lds + HXK_SBIAS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = uni(tid >> 6);
    if (tid < 100) sbias[tid] = reinterpret_cast<const float*>(blob + HX_OFF_BIAS)[tid];
    hx_load_weights(blob, lds, tid, NTHREADS);
    __syncthreads();
    float* strip = reinterpret_cast<float*>(lds + HXK_STRIPS + wave * HX_STRIP_BYTES);
    // leaf i goes to wave slot i % (waves of the launch), workgroups first: few leaves spread over many CUs
    const int nslots = (int)gridDim.x * 8;
#pragma unroll 1
    for (int i = wave * (int)gridDim.x + (int)blockIdx.x; i < n; i += nslots) {
        if (leaf_info) {
            const int kind = leaf_info[i] & 0xFF;
            if (kind != AZ_LEAF_EVAL && kind != AZ_LEAF_ROOT) continue;
        }
        const uint64_t* pos = leaf_bb + 2 * (size_t)i;
        hx_trunk_task(lds, sbias, num_blocks, pos, strip, lane);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");        // the strip, before this wave reads it back
        hx_heads_own(blob, sbias, strip, pos, p_out + (size_t)i * 7, v_out + i, lane);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");        // ... and before the next leaf overwrites it
    }
}

// ---- the module's parameters -> the packed weights of the fp16-pair kernels, in one launch --------------------
// (the statement of this layout in tensor operations is net.py:_pack_c4_blob_hx_torch; same bytes): batch-norm folded
// in double, rounded once to float, then split: h = fp16(w), l = fp16((w - h) * 2048).  A weight beyond the fp16
// range sets *range_error.
#define HX_SPLIT_SCALE 2048.0f
__global__ void __launch_bounds__(256) k_pack_c4_hx(const PackArgs a, uint8_t* __restrict__ blob, int* __restrict__ range_error) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int nh = HX_W_BYTES / 2;
    if (e < nh) {
        // one fp16 element of a 16x16x32 fragment [64 lanes][8]: element j of lane l is M[row = l & 15][k = 8 (l >> 4) + j]
        const int byte = 2 * e, l = (e >> 3) & 63, j = e & 7, k = 8 * (l >> 4) + j;
        double v;
        int low;                                           // 0: the leading half h, 1: the following half l
        if (byte < HX_OFF_ID) {
            // stem and conv taps: [fragment][2 channel halves][h, l][64][8]
            low = (e >> 9) & 1;
            v = trunk_weight(a, e >> 11, half_row((e >> 10) & 1, l & 15), k);
        } else if (byte < HX_OFF_HF) {
            low = 0;                                       // the identity is exact: [2 channel halves][64][8]
            v = trunk_weight(a, 19, half_row(((e - HX_OFF_ID / 2) >> 9) & 1, l & 15), k);
        } else {
            low = ((e - HX_OFF_HF / 2) >> 9) & 1;          // 1x1 heads: [h, l][64][8]
            v = head_conv_weight(a, l & 15, k);
        }
        const float w = (float)v;
        if (!(fabsf(w) < 6.0e4f)) *range_error = 1;
        const _Float16 h = (_Float16)w;
        const _Float16 out = low ? (_Float16)((w - (float)h) * HX_SPLIT_SCALE) : h;
        reinterpret_cast<_Float16*>(blob)[e] = out;
        return;
    }
    const int f = e - nh;
    if (f >= (HX_BLOB_BYTES - HX_OFF_BIAS) / 4) return;
    double v;
    if (f < 99) v = folded_bias(a, f);
    else if (f == 99) v = (double)a.val_fc2_b[0];
    else if (f < 100 + 64) v = f - 100 < 42 ? (double)a.pol_fc_b[f - 100] : 0.0;
    else if (f < 100 + 64 + 256) v = (double)a.val_fc1_b[f - 164];
    else if (f < 100 + 64 + 512) v = (double)a.val_fc2_w[f - 420];
    else if (f < 676 + 42 * 256) {
        // [cell][lane][u] = W1[4 lane + u][cell]
        const int q = f - 676, cell = q / 256, unit = q % 256;
        v = (double)a.val_fc1_w[unit * 42 + cell];
    } else {
        v = (double)a.pol_fc_w[f - 676 - 42 * 256];
    }
    reinterpret_cast<float*>(blob + HX_OFF_BIAS)[f] = (float)v;
}

extern "C" {

int az_c4net_hx_blob_bytes(void) { return HX_BLOB_BYTES; }

int az_c4net_pack_hx(const az_c4_train_net* net, int32_t channels, double bn_eps, uint8_t* blob, int32_t* range_error_dev,
                     void* stream) {
    if (!net || !blob || !range_error_dev || channels < 1 || channels > 32 || net->num_blocks < 0)
        return az_fail("az_c4net_pack_hx: null pointer, or not 1..32 trunk channels");
    const PackArgs a = make_pack_args(net, channels, bn_eps);
    const int n = HX_W_BYTES / 2 + (HX_BLOB_BYTES - HX_OFF_BIAS) / 4;
    hipLaunchKernelGGL(k_pack_c4_hx, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, blob, (int*)range_error_dev);
    AZ_LAUNCH_CHECK("k_pack_c4_hx");
    return 0;
}

int az_c4net_hx_forward(const uint8_t* blob, int32_t num_blocks, const uint64_t* leaf_bb, const int32_t* leaf_info,
                        int32_t n, float* p, float* v, void* stream) {
    if (n < 0 || num_blocks < 0 || (n && (!blob || !leaf_bb || !p || !v)))
        return az_fail("az_c4net_hx_forward: null pointer or negative count");
    if (n == 0) return 0;
    static std::atomic<unsigned long long> attr_done{0};
    if (int rc = az_set_kernel_lds(reinterpret_cast<const void*>(k_c4net_hx), HXK_LDS_BYTES, attr_done)) return rc;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v_ = 0;
        if (hipDeviceGetAttribute(&v_, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v_ > 0) cus = v_;
    }
    const int want = (n + 7) / 8;
    const int grid = want < cus ? want : cus;
    hipLaunchKernelGGL(k_c4net_hx, dim3(grid), dim3(NTHREADS), HXK_LDS_BYTES, (hipStream_t)stream, blob, (int)num_blocks,
                       leaf_bb, leaf_info, (int)n, p, v);
    AZ_LAUNCH_CHECK("k_c4net_hx");
    return 0;
}

}  // extern "C"
