// az_net.hip -- fused Connect Four policy/value network forward for gfx950.
//
// The whole network of alphazero/models.py:70-87 + alphazero/games/connectfour.py:54-72
// (stem conv, tower of one residual block applied num_blocks times, policy and value
// heads, landing-cell gather, soft-max over legal columns) runs in ONE kernel for up to
// 32 trunk channels.  A workgroup (one per CU, 8 waves) takes its leaf positions straight
// from their bitboards, two boards per wave at a time, keeps every activation of the trunk in
// registers (az_net_dev.h) and writes only p[7] and v per leaf: HBM traffic is 16 B in
// and 32 B out per leaf, so the kernel is bounded by the matrix-core rate, not by memory.
//
// Batch-norm (eval mode) is folded into weights and a per-channel bias on the host; the
// bias is the accumulator's initial value.  The fully connected heads run once per round for
// the whole workgroup on v_mfma_f32_32x32x16_bf16 from small planes in LDS.
#include "az_net_dev.h"
#include "az_pack_dev.h"
#include "az_host.h"

__global__ void __launch_bounds__(NTHREADS, 2)
k_c4net_bf16(const uint8_t* __restrict__ blob, int num_blocks, const uint64_t* __restrict__ leaf_bb,
             const int32_t* __restrict__ leaf_info, int n, float* __restrict__ p_out, float* __restrict__ v_out,
             int grid_shift) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* wlds = lds;                       // tower weights
    uint8_t* scr = lds + OFF_BIAS;
    float* sbias = reinterpret_cast<float*>(scr + S_BIAS);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    // zero the scratch once: the k-padding of the head planes is never written again, and an empty
    // entry of the task list is 0 (entries hold leaf index + 1)
    {
        uint4* z = reinterpret_cast<uint4*>(scr);
        for (int i = tid; i < S_BYTES / 16; i += NTHREADS) z[i] = make_uint4(0, 0, 0, 0);
    }
    // Memory requests go out in the order in which their data is needed: the small head / bias tables,
    // then this thread's slice of leaf_info, and only then the workgroup's 36 KB of tower weights (loads
    // return in order, so whatever is waited for first must not queue behind the weights).
    float t_bias = 0.0f, t_b1 = 0.0f, t_w2 = 0.0f, t_bp = 0.0f;
    if (tid < 99) t_bias = reinterpret_cast<const float*>(blob + OFF_BIAS)[tid];
    if (tid == 99) t_bias = reinterpret_cast<const float*>(blob + OFF_B2)[0];      // value FC2 bias rides in the pad slot
    if (tid < 256) {
        t_b1 = reinterpret_cast<const float*>(blob + OFF_B1)[tid];
        t_w2 = reinterpret_cast<const float*>(blob + OFF_W2)[tid];
    }
    if (tid < 64) t_bp = reinterpret_cast<const float*>(blob + OFF_BP)[tid];
    // thread t scans games [t*per, t*per + per); `per` is rounded up to a multiple of 4 so the kinds
    // arrive as 16-byte loads; the first 16 games of the slice are requested here
    const int per = ((n + NTHREADS - 1) / NTHREADS + 3) & ~3;
    const int g0 = tid * per;
    auto load4 = [&](int g) {
        int4 k4 = make_int4(0, 0, 0, 0);
        if (g + 3 < n) k4 = *reinterpret_cast<const int4*>(leaf_info + g);
        else {
            if (g < n) k4.x = leaf_info[g];
            if (g + 1 < n) k4.y = leaf_info[g + 1];
            if (g + 2 < n) k4.z = leaf_info[g + 2];
        }
        return k4;
    };
    int4 kfirst[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) kfirst[q] = (leaf_info && 4 * q < per) ? load4(g0 + 4 * q) : make_int4(0, 0, 0, 0);
    load_weights(blob, wlds, tid);
    __syncthreads();
    if (tid < 100) sbias[tid] = t_bias;
    if (tid < 256) {
        reinterpret_cast<float*>(scr + S_B1)[tid] = t_b1;
        reinterpret_cast<float*>(scr + S_W2)[tid] = t_w2;
    }
    if (tid < 64) reinterpret_cast<float*>(scr + S_BP)[tid] = t_bp;

    // ---- which leaves does each wave evaluate?  Only leaves that need the network (kind EVAL or
    // ROOT in leaf_info) count.  The E of them are dealt out in rank order to SIMDs first: waves w and
    // w + 4 of a workgroup share a SIMD, and that pair (unit u = workgroup + gridDim * w, w < 4) takes
    // L = base + (u < rem) consecutive ranks (base = E / units, rem = E % units).  Inside the pair
    // wave w + 4 takes an even number b = 2 * ((L + 1) / 4) of them and wave w the rest, so at most one
    // of the pair's tasks is a single board (2 tiles for one leaf instead of 3 for two).  Every workgroup
    // scans the (L2-resident) leaf_info array itself: no extra launch, no atomics, deterministic.
    int* tlist = reinterpret_cast<int*>(scr + S_TL);
    int* wtot = reinterpret_cast<int*>(scr + S_WT);
    const int NU = (int)gridDim.x * 4;
    int E = n;                                 // leaves to evaluate in this launch
    if (!leaf_info) {
        // dense mode: every leaf is evaluated, rank = leaf index
        if (tid < 8 * MAXIT * 2) {
            const int w = tid / (MAXIT * 2), pos = tid % (MAXIT * 2);
            const int unit = (int)blockIdx.x + (int)gridDim.x * (w & 3);
            const int base = n / NU, rem = n - base * NU;
            const int start = unit * base + min(unit, rem), len = base + (unit < rem);
            const int nb = 2 * ((len + 1) >> 2), na = len - nb;
            if (w < 4 ? pos < na : pos < nb) tlist[tid] = start + (w < 4 ? 0 : na) + pos + 1;
        }
    } else {
        // the "needs eval" flags of the slice stay in a register mask: up to 64 games per thread (n <= 32768)
        auto flags4 = [&](const int4& k4) {
            const int ks[4] = {k4.x & 0xFF, k4.y & 0xFF, k4.z & 0xFF, k4.w & 0xFF};
            unsigned long long m = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (ks[i] == AZ_LEAF_EVAL || ks[i] == AZ_LEAF_ROOT) m |= 1ull << i;
            return m;
        };
        unsigned long long mask = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) mask |= flags4(kfirst[q]) << (4 * q);
        for (int j = 16; j < per; j += 4) mask |= flags4(load4(g0 + j)) << j;
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
        for (int w = 0; w < 8; ++w) {
            const int t = wtot[w];
            before += (w < wave) ? t : 0;
            total += t;
        }
        E = total;
        // ranks [rank, rank + cnt) belong to this thread: find the unit of the first one with one
        // division, then step through units as the ranks run out of them
        if (cnt) {
            const int rank = before + incl - cnt;
            const int base = total / NU, rem = total - base * NU;
            const int big = rem * (base + 1);           // ranks below this sit in units of base + 1
            int unit, pos;
            if (rank < big) { unit = rank / (base + 1); pos = rank - unit * (base + 1); }
            else { const int q = (rank - big) / base; unit = rem + q; pos = rank - big - q * base; }
            int len = base + (unit < rem);
            while (mask) {
                const int j = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                if ((unit & ((int)gridDim.x - 1)) == (int)blockIdx.x) {
                    const int na = len - 2 * ((len + 1) >> 2);
                    const int w = (unit >> grid_shift) + (pos < na ? 0 : 4), q = pos < na ? pos : pos - na;
                    if (q < MAXIT * 2) tlist[w * MAXIT * 2 + q] = g0 + j + 1;
                }
                if (++pos == len) { ++unit; pos = 0; len = base + (unit < rem); }
            }
        }
    }
    __syncthreads();

    // the longest list of this workgroup is wave 0's (unit = blockIdx is its lowest unit): 4 leaves per round
    int nrounds;
    {
        const int base = E / NU, rem = E - base * NU;
        const int len = base + ((int)blockIdx.x < rem), na = len - 2 * ((len + 1) >> 2);
        nrounds = (min(na, MAXIT * 2) + 3) >> 2;
    }

    net_rounds<false>(blob, num_blocks, leaf_bb, 0, p_out, v_out, wlds, scr, nrounds, tid, wave, lane);
}

// ---- the module's parameters -> the packed weights, in one launch ---------------------------------------------
// (the statement of this layout in tensor operations is net.py:_pack_c4_blob_torch; same bytes): batch-norm folded in
// double (az_pack_dev.h), then double -> float -> bf16, round to nearest even both times.
__device__ __forceinline__ uint16_t to_bf16(double x) {
    const float f = (float)x;
    uint32_t u = __float_as_uint(f);
    if (f != f) return 0x7FC0;
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

__global__ void __launch_bounds__(256) k_pack_c4_bf16(const PackArgs a, uint8_t* __restrict__ blob) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int nbf = OFF_BIAS / 2;
    if (e < nbf) {
        // one bf16 element; inside a 16x16x32 fragment [64 lanes][8]: element j of lane l is M[row = l & 15][k = 8 (l >> 4) + j],
        // inside a 32x32x16 one [k-step][64][8]: M[row = l & 31][k = 16 s + 8 (l >> 5) + j]
        const int byte = 2 * e, l = (e >> 3) & 63, j = e & 7;
        double v = 0.0;
        if (byte < OFF_HF) {
            // trunk fragments [..][2 halves][64][8]: stem, nine taps of each conv, identity
            v = trunk_weight(a, e >> 10, half_row((e >> 9) & 1, l & 15), 8 * (l >> 4) + j);
        } else if (byte < OFF_F1) {
            v = head_conv_weight(a, l & 15, 8 * (l >> 4) + j);
        } else if (byte < OFF_PF) {
            const int q = e - OFF_F1 / 2, tile = q / 1536, s = (q / 512) % 3;
            const int row = 32 * tile + (l & 31), k = 16 * s + 8 * (l >> 5) + j;
            if (k < 42) v = (double)a.val_fc1_w[row * 42 + k];
        } else {
            const int q = e - OFF_PF / 2, tile = q / 3072, s = (q / 512) % 6;
            const int row = 32 * tile + (l & 31), k = 16 * s + 8 * (l >> 5) + j;
            if (row < 42 && k < 84) v = (double)a.pol_fc_w[row * 84 + k];
        }
        reinterpret_cast<uint16_t*>(blob)[e] = to_bf16(v);
        return;
    }
    const int f = e - nbf;
    if (f >= (BLOB_BYTES - OFF_BIAS) / 4) return;
    double v;
    if (f < 100) v = folded_bias(a, f);
    else if (f < 100 + 64) v = f - 100 < 42 ? (double)a.pol_fc_b[f - 100] : 0.0;
    else if (f < 100 + 64 + 256) v = (double)a.val_fc1_b[f - 164];
    else if (f < 100 + 64 + 512) v = (double)a.val_fc2_w[f - 420];
    else v = f == 676 ? (double)a.val_fc2_b[0] : 0.0;
    reinterpret_cast<float*>(blob + OFF_BIAS)[f] = (float)v;
}

extern "C" {

int az_c4net_blob_bytes(void) { return BLOB_BYTES; }

int az_c4net_pack_bf16(const az_c4_train_net* net, int32_t channels, double bn_eps, uint8_t* blob, void* stream) {
    if (!net || !blob || channels < 1 || channels > 32 || net->num_blocks < 0)
        return az_fail("az_c4net_pack_bf16: null pointer, or not 1..32 trunk channels");
    const PackArgs a = make_pack_args(net, channels, bn_eps);
    const int n = OFF_BIAS / 2 + (BLOB_BYTES - OFF_BIAS) / 4;
    hipLaunchKernelGGL(k_pack_c4_bf16, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, blob);
    AZ_LAUNCH_CHECK("k_pack_c4_bf16");
    return 0;
}

int az_c4net_bf16_forward(const uint8_t* blob, int32_t num_blocks, const uint64_t* leaf_bb,
                          const int32_t* leaf_info, int32_t n, float* p, float* v, void* stream) {
    if (n < 0 || num_blocks < 0 || (n && (!blob || !leaf_bb || !p || !v)))
        return az_fail("az_c4net_bf16_forward: null pointer or negative count");
    if (n == 0) return 0;
    // MAXIT wave-tasks per wave, 64 leaves per scanning thread
    if (n > 32768) return az_fail("az_c4net_bf16_forward: at most 32768 leaves per call");
    static std::atomic<unsigned long long> attr_done{0};
    if (int rc = az_set_kernel_lds(reinterpret_cast<const void*>(k_c4net_bf16), LDS_BYTES, attr_done)) return rc;
    // one persistent workgroup per CU at most; a power of two so slots map to waves by bit masks; with
    // fewer leaves than wave slots they are spread over the CUs first (slot s = WG s % grid)
    const int ntasks = (n + 1) / 2;
    int grid_shift = 0;
    while ((1 << grid_shift) < ntasks && grid_shift < 8) ++grid_shift;
    const int grid = 1 << grid_shift;
    hipLaunchKernelGGL(k_c4net_bf16, dim3(grid), dim3(NTHREADS), LDS_BYTES, (hipStream_t)stream, blob, (int)num_blocks,
                       leaf_bb, leaf_info, (int)n, p, v, grid_shift);
    AZ_LAUNCH_CHECK("k_c4net_bf16");
    return 0;
}

}  // extern "C"
