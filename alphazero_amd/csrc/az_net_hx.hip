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

extern "C" {

int az_c4net_hx_blob_bytes(void) { return HX_BLOB_BYTES; }

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
