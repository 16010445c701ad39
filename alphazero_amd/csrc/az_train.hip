// az_train.hip -- the two hand-written pieces of the training step (alphazero/training.py:42-75):
//   * az_loss_fwd_bwd : AlphaZeroLoss (alphazero/loss.py:29-63: masked cross-entropy of the policy + MSE of the
//                       value, both batch means) together with its gradients with respect to the network's
//                       outputs, in one launch;
//   * az_sgd_momentum : torch.optim.SGD's update (momentum, weight decay, no dampening, no Nesterov --
//                       training.py:27-29) over ONE flat parameter / gradient / momentum buffer, in one launch
//                       instead of three per tensor.
// The convolutions' forward and backward in between stay with the vendor library that PyTorch calls.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/azhip.h"
#include "az_host.h"

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
        float m = -INFINITY;
        for (int a = 0; a < A; ++a) if (valid[(size_t)b * A + a]) m = fmaxf(m, logits[(size_t)b * A + a]);
        float sum = 0.0f, tsum = 0.0f;
        for (int a = 0; a < A; ++a) {
            if (valid[(size_t)b * A + a]) sum += expf(logits[(size_t)b * A + a] - m);
            tsum += target[(size_t)b * A + a];
        }
        const float lse = m + logf(sum);
        float ce = 0.0f, tvalid = 0.0f;
        for (int a = 0; a < A; ++a) {
            const bool ok = valid[(size_t)b * A + a];
            const float t = target[(size_t)b * A + a];
            if (ok && t != 0.0f) { ce -= t * (logits[(size_t)b * A + a] - lse); tvalid += t; }   // loss.py:29-47
        }
        for (int a = 0; a < A; ++a) {
            const bool ok = valid[(size_t)b * A + a];
            const float t = target[(size_t)b * A + a];
            const float q = ok ? expf(logits[(size_t)b * A + a] - lse) : 0.0f;
            dlogits[(size_t)b * A + a] = ok ? (q * tvalid - (t != 0.0f ? t : 0.0f)) * invB : 0.0f;
        }
        const float d = v_hat[b] - z[b];
        dv[b] = 2.0f * d * invB;                                                               // loss.py:57
        pl += ce; vl += d * d;
        dev = fmaxf(dev, fabsf(tsum - 1.0f));
    }
    red[0][threadIdx.x] = pl; red[1][threadIdx.x] = vl; red[2][threadIdx.x] = dev;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
            red[2][threadIdx.x] = fmaxf(red[2][threadIdx.x], red[2][threadIdx.x + s]);
        }
        __syncthreads();
    }
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

extern "C" {

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
