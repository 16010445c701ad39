// az_train.hip -- the two hand-written pieces of the training step (alphazero/training.py:42-75):
//   * az_loss_fwd_bwd : AlphaZeroLoss (alphazero/loss.py:29-63: masked cross-entropy of the policy + MSE of the
//                       value, both batch means) together with its gradients with respect to the network's
//                       outputs, in one launch;
//   * az_sgd_momentum : torch.optim.SGD's update (momentum, weight decay, no dampening, no Nesterov --
//                       training.py:27-29) over ONE flat parameter / gradient / momentum buffer, in one launch
//                       instead of three per tensor.
//   * az_sample_window: a batch drawn from the device-resident replay window (ReplayBufferSampler,
//                       alphazero/replay_buffer.py:175-195) and gathered, in one launch.
// The network's forward and backward: az_train_c4.hip for the 32-channel Connect Four network, PyTorch otherwise.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/azhip.h"
#include "az_host.h"
#include "az_loss_dev.h"

// One workgroup, thread = sample (strided): deterministic sums, and the batches of this path are small
// (config/connectfour.yaml: 128).  A <= AZ_MAX_ACTIONS.
template <int A>
__global__ void __launch_bounds__(256)
k_loss(const float* __restrict__ logits, const float* __restrict__ v_hat, const float* __restrict__ target,
       const float* __restrict__ z, const uint8_t* __restrict__ valid, int B, float* __restrict__ out,
       float* __restrict__ dlogits, float* __restrict__ dv) {
    __shared__ float red[3][256];
    float pl = 0.0f, vl = 0.0f, dev = 0.0f;
    const float invB = 1.0f / (float)B;
    for (int b = threadIdx.x; b < B; b += 256) {
        float ce, sq, dv1;
        loss_sample<A>(logits, v_hat, target, z, valid, b, invB, dlogits, dv, ce, sq, dv1);
        pl += ce; vl += sq;
        dev = fmaxf(dev, dv1);
    }
    loss_totals(pl, vl, dev, red);
    if (threadIdx.x == 0) { out[0] = red[0][0] * invB; out[1] = red[1][0] * invB; out[2] = red[2][0]; }
}

// torch.optim.SGD, one element per thread: d = g + wd * p; buf = first ? d : momentum * buf + d; p -= lr * buf
__global__ void __launch_bounds__(256)
k_sgd(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n, float lr, float momentum,
      float wd, int first) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float d = fmaf(wd, p[i], g[i]);
    const float b = first ? d : buf[i] * momentum + d;
    buf[i] = b;
    p[i] = fmaf(-lr, b, p[i]);
}

// the same with the learning rate read from device memory, for launches recorded once and replayed (a HIP graph
// bakes scalar arguments in): buf must start as zeros, which makes the first step's buf = d without a flag
__global__ void __launch_bounds__(256)
k_sgd_lr_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n,
             const float* __restrict__ lr_dev, float momentum, float wd) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float lr = lr_dev[0];
    const float d = fmaf(wd, p[i], g[i]);
    const float b = buf[i] * momentum + d;
    buf[i] = b;
    p[i] = fmaf(-lr, b, p[i]);
}

// k_sgd_lr_dev behind a guard that is also read from device memory: when !(guard[0] < limit) -- a failed check of
// the batch, NaN included -- nothing is updated and withheld[0] counts the step
__global__ void __launch_bounds__(256)
k_sgd_guarded(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n,
              const float* __restrict__ lr_dev, float momentum, float wd, const float* __restrict__ guard, float limit,
              int* __restrict__ withheld) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (!(guard[0] < limit)) {
        if (i == 0) withheld[0] += 1;
        return;
    }
    if (i >= n) return;
    const float lr = lr_dev[0];
    const float d = fmaf(wd, p[i], g[i]);
    const float b = buf[i] * momentum + d;
    buf[i] = b;
    p[i] = fmaf(-lr, b, p[i]);
}

// ---- a training batch drawn from the device-resident replay window, in one launch ------------------------------
__device__ __forceinline__ uint32_t mix32(uint64_t seed, uint64_t draw, uint32_t i, uint32_t attempt) {
    // splitmix64 over a key that is unique per (seed, draw, sample, attempt)
    uint64_t zz = seed + 0x9E3779B97F4A7C15ull * (draw * 0x100000001B3ull + ((uint64_t)i << 20) + attempt + 1);
    zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ull;
    zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBull;
    zz ^= zz >> 31;
    return (uint32_t)(zz >> 32);
}

#define SAMPLE_MAX_WINDOW (1 << 18)          // states: one bit each in LDS
#define SAMPLE_MAX_BATCH 2048
#define SAMPLE_THREADS 1024
__global__ void __launch_bounds__(SAMPLE_THREADS)
k_sample_window(const float* __restrict__ wx, const float* __restrict__ wp, const float* __restrict__ wz,
                const uint8_t* __restrict__ wvalid, int x_floats, int A, const long long* __restrict__ count_dev, int capacity,
                unsigned long long seed, long long* __restrict__ draw_dev, int batch, float* __restrict__ x, float* __restrict__ p,
                float* __restrict__ z, uint8_t* __restrict__ valid) {
    __shared__ uint32_t taken[SAMPLE_MAX_WINDOW / 32];
    __shared__ int pick[SAMPLE_MAX_BATCH];
    __shared__ uint32_t first[SAMPLE_MAX_BATCH];       // every sample's first draw
    const int tid = threadIdx.x;
    const long long count = count_dev[0];
    const int n = count < capacity ? (int)count : capacity;
    const unsigned long long draw = (unsigned long long)draw_dev[0];
    __shared__ int repeated;
    // Uniform without replacement while the window holds at least a batch (each draw is rejected while it repeats an
    // earlier one), with replacement otherwise (alphazero/replay_buffer.py:188-195).  Every sample's first draw is made
    // in parallel; only when two of them coincide (one batch in six for 128 of 50 000 states) does thread 0 go through
    // the samples in order, which is what defines the result in both cases.
    const bool distinct = batch <= n;
    for (int i = tid; i < (n + 31) / 32; i += SAMPLE_THREADS) taken[i] = 0u;
    if (tid == 0) repeated = 0;
    __syncthreads();
    for (int i = tid; i < batch; i += SAMPLE_THREADS) {
        const uint32_t r = (uint32_t)(((uint64_t)mix32(seed, draw, (uint32_t)i, 0) * (uint64_t)n) >> 32);
        if (distinct && ((atomicOr(&taken[r >> 5], 1u << (r & 31)) >> (r & 31)) & 1u)) repeated = 1;
        first[i] = r;
        pick[i] = (int)(((long long)r + (count - n)) % capacity);           // the i-th oldest state sits at (count - n + i) % capacity
    }
    __syncthreads();
    if (repeated) {
        for (int i = tid; i < (n + 31) / 32; i += SAMPLE_THREADS) taken[i] = 0u;
        __syncthreads();
        if (tid == 0) {
            for (int i = 0; i < batch; ++i) {
                uint32_t r = first[i], attempt = 1;
                while ((taken[r >> 5] >> (r & 31)) & 1u)
                    r = (uint32_t)(((uint64_t)mix32(seed, draw, (uint32_t)i, attempt++) * (uint64_t)n) >> 32);
                taken[r >> 5] |= 1u << (r & 31);
                pick[i] = (int)(((long long)r + (count - n)) % capacity);
            }
        }
    }
    if (tid == 0) draw_dev[0] = (long long)(draw + 1);
    __syncthreads();
    // the gather, eight loads of a thread in flight (each is a trip to memory: one at a time they took 20 us)
    for (int i0 = tid; i0 < batch * x_floats; i0 += 8 * SAMPLE_THREADS) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * SAMPLE_THREADS;
            v[u] = i < batch * x_floats ? wx[(size_t)pick[i / x_floats] * x_floats + i % x_floats] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * SAMPLE_THREADS;
            if (i < batch * x_floats) x[i] = v[u];
        }
    }
    for (int i0 = tid; i0 < batch * A; i0 += 8 * SAMPLE_THREADS) {
        float v[8];
        uint8_t m[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * SAMPLE_THREADS;
            const bool in = i < batch * A;
            v[u] = in ? wp[(size_t)pick[i / A] * A + i % A] : 0.0f;
            m[u] = in ? wvalid[(size_t)pick[i / A] * A + i % A] : (uint8_t)0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * SAMPLE_THREADS;
            if (i < batch * A) { p[i] = v[u]; valid[i] = m[u]; }
        }
    }
    for (int i = tid; i < batch; i += SAMPLE_THREADS) z[i] = wz[pick[i]];
}

extern "C" {

int az_sample_window(const float* wx, const float* wp, const float* wz, const uint8_t* wvalid, int32_t x_floats,
                     int32_t num_actions, const int64_t* count_dev, int32_t capacity, uint64_t seed, int64_t* draw_dev,
                     int32_t batch, float* x, float* p, float* z, uint8_t* valid, void* stream) {
    if (!wx || !wp || !wz || !wvalid || !count_dev || !draw_dev || !x || !p || !z || !valid)
        return az_fail("az_sample_window: null pointer");
    if (batch <= 0 || batch > SAMPLE_MAX_BATCH || capacity <= 0 || capacity > SAMPLE_MAX_WINDOW || x_floats <= 0 || num_actions <= 0)
        return az_fail("az_sample_window: batch <= 2048 states from a window of at most 262144");
    hipLaunchKernelGGL(k_sample_window, dim3(1), dim3(SAMPLE_THREADS), 0, (hipStream_t)stream, wx, wp, wz, wvalid, (int)x_floats,
                       (int)num_actions, reinterpret_cast<const long long*>(count_dev), (int)capacity,
                       (unsigned long long)seed, reinterpret_cast<long long*>(draw_dev), (int)batch, x, p, z, valid);
    AZ_LAUNCH_CHECK("k_sample_window");
    return 0;
}

int az_loss_fwd_bwd(int32_t num_actions, const float* logits, const float* v_hat, const float* target, const float* z,
                    const uint8_t* valid, int32_t batch, float* out3, float* dlogits, float* dv, void* stream) {
    if (batch <= 0 || !logits || !v_hat || !target || !z || !valid || !out3 || !dlogits || !dv)
        return az_fail("az_loss_fwd_bwd: null pointer or empty batch");
    hipStream_t st = (hipStream_t)stream;
    if (num_actions == 7)
        hipLaunchKernelGGL((k_loss<7>), dim3(1), dim3(256), 0, st, logits, v_hat, target, z, valid, (int)batch, out3, dlogits, dv);
    else if (num_actions == 9)
        hipLaunchKernelGGL((k_loss<9>), dim3(1), dim3(256), 0, st, logits, v_hat, target, z, valid, (int)batch, out3, dlogits, dv);
    else
        return az_fail("az_loss_fwd_bwd: 7 (Connect Four) or 9 (tic-tac-toe) actions");
    AZ_LAUNCH_CHECK("k_loss");
    return 0;
}

int az_sgd_momentum(float* params, const float* grads, float* momentum_buf, int64_t n, float lr, float momentum,
                    float weight_decay, int32_t first_step, void* stream) {
    if (n < 0 || (n && (!params || !grads || !momentum_buf))) return az_fail("az_sgd_momentum: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_sgd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads,
                       momentum_buf, (long)n, lr, momentum, weight_decay, (int)first_step);
    AZ_LAUNCH_CHECK("k_sgd");
    return 0;
}

int az_sgd_momentum_lr_dev(float* params, const float* grads, float* momentum_buf, int64_t n, const float* lr_dev,
                           float momentum, float weight_decay, void* stream) {
    if (n < 0 || !lr_dev || (n && (!params || !grads || !momentum_buf))) return az_fail("az_sgd_momentum_lr_dev: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_sgd_lr_dev, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads,
                       momentum_buf, (long)n, lr_dev, momentum, weight_decay);
    AZ_LAUNCH_CHECK("k_sgd_lr_dev");
    return 0;
}

int az_sgd_momentum_guarded(float* params, const float* grads, float* momentum_buf, int64_t n, const float* lr_dev,
                            float momentum, float weight_decay, const float* guard_dev, float guard_limit,
                            int32_t* withheld_dev, void* stream) {
    if (n < 0 || !lr_dev || !guard_dev || !withheld_dev || (n && (!params || !grads || !momentum_buf)))
        return az_fail("az_sgd_momentum_guarded: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_sgd_guarded, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads,
                       momentum_buf, (long)n, lr_dev, momentum, weight_decay, guard_dev, guard_limit, withheld_dev);
    AZ_LAUNCH_CHECK("k_sgd_guarded");
    return 0;
}

}  // extern "C"
