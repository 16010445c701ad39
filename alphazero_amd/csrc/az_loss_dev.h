// az_loss_dev.h -- AlphaZeroLoss for one sample (alphazero/loss.py:29-63): the masked cross-entropy of the policy
// and the squared error of the value, with their gradients with respect to the network's outputs already scaled by
// 1 / batch (both losses are batch means).  Shared by the stand-alone loss kernel (az_train.hip) and the one-launch
// training pass (az_train_c4.hip), which therefore produce the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

template <int A>
__device__ __forceinline__ void loss_sample(const float* logits, const float* v_hat, const float* target, const float* z,
                                            const uint8_t* valid, int b, float invB, float* dlogits, float* dv, float& ce_out,
                                            float& sq_out, float& dev_out) {
    // (everything is loaded before the first use: one trip to memory, not one per branch)
    float lg[A], tg[A];
    bool ok[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {
        lg[a] = logits[(size_t)b * A + a];
        tg[a] = target[(size_t)b * A + a];
        ok[a] = valid[(size_t)b * A + a] != 0;
    }
    const float vh = v_hat[b], zz = z[b];
    float m = -INFINITY;
#pragma unroll
    for (int a = 0; a < A; ++a) if (ok[a]) m = fmaxf(m, lg[a]);
    float sum = 0.0f, tsum = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        if (ok[a]) sum += expf(lg[a] - m);
        tsum += tg[a];
    }
    const float lse = m + logf(sum);
    float ce = 0.0f, tvalid = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a)
        if (ok[a] && tg[a] != 0.0f) { ce -= tg[a] * (lg[a] - lse); tvalid += tg[a]; }      // loss.py:29-47
#pragma unroll
    for (int a = 0; a < A; ++a) {
        const float q = ok[a] ? expf(lg[a] - lse) : 0.0f;
        dlogits[(size_t)b * A + a] = ok[a] ? (q * tvalid - (tg[a] != 0.0f ? tg[a] : 0.0f)) * invB : 0.0f;
    }
    const float d = vh - zz;
    dv[b] = 2.0f * d * invB;                                                               // loss.py:57
    ce_out = ce;
    sq_out = d * d;
    dev_out = fabsf(tsum - 1.0f);
}

// The batch totals of the three per-sample quantities, by one workgroup's first 256 threads: thread t holds the sums
// (maximum for the third) of samples t, t + 256, ...; red = LDS [3][256].  Thread 0 returns with the totals in
// red[0][0], red[1][0], red[2][0].  Every thread of the workgroup must call it.
__device__ __forceinline__ void loss_totals(float pl, float vl, float dev, float (*red)[256]) {
    if (threadIdx.x < 256) { red[0][threadIdx.x] = pl; red[1][threadIdx.x] = vl; red[2][threadIdx.x] = dev; }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
            red[2][threadIdx.x] = fmaxf(red[2][threadIdx.x], red[2][threadIdx.x + s]);
        }
        __syncthreads();
    }
}
