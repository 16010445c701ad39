// exp_ll_sync.hip -- how long does one batch-wide exchange take inside a resident grid?  W workgroups of 512 threads;
// per round every workgroup publishes 64 floats as (value, tag) pairs of eight bytes (agent-scope stores: a pair is
// valid as soon as its tag is the round's, no fences) and adds up the rows of all workgroups.  Prints microseconds per
// round.  Build: hipcc -O3 --offload-arch=gfx950 tools/exp_ll_sync.hip -o tools/bin/exp_ll_sync
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ bool gather(const unsigned long long* buf, int boards, uint32_t tag, double* tot, double* scratch,
                                       int* abort_word) {
    const int col = threadIdx.x & 63, q = threadIdx.x >> 6;
    float v[32];
    uint32_t pending = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) { v[j] = 0.0f; if (q + 8 * j < boards) pending |= 1u << j; }
    int spins = 0;
    bool timed_out = false;
    while (pending) {
        unsigned long long w[32];
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if (pending >> j & 1) w[j] = __hip_atomic_load(buf + (size_t)(q + 8 * j) * 64 + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if ((pending >> j & 1) && (uint32_t)(w[j] >> 32) == tag) { v[j] = __uint_as_float((uint32_t)w[j]); pending &= ~(1u << j); }
        if (pending && ++spins > (1 << 18)) { timed_out = true; break; }
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 32; ++j) s += (double)v[j];
    scratch[q * 64 + col] = s;
    const int any = __syncthreads_or(timed_out);
    if (threadIdx.x < 64) {
        double t = 0.0;
        for (int k = 0; k < 8; ++k) t += scratch[k * 64 + col];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (any && threadIdx.x == 0) atomicExch(abort_word, 1);
    return any != 0;
}

__global__ void __launch_bounds__(512) k_rounds(unsigned long long* bufs, int rounds, const int* epoch, double* out, int* abort_word, int work) {
    __shared__ double tot[64], scratch[512];
    const int b = blockIdx.x, W = gridDim.x;
    const uint32_t e = (uint32_t)epoch[0] * 1024u;
    float x = (float)(b + 1);
    double check = 0.0;
    for (int r = 1; r <= rounds; ++r) {
        unsigned long long* buf = bufs + (size_t)(r & 1) * W * 64;
        // some dependent arithmetic between the rounds, like a layer would do
        for (int k = 0; k < work; ++k) x = fmaf(x, 1.0000001f, 1e-9f);
        if (threadIdx.x < 64)
            __hip_atomic_store(buf + (size_t)b * 64 + threadIdx.x, ((unsigned long long)(e + r) << 32) | __float_as_uint(x + threadIdx.x),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gather(buf, W, e + r, tot, scratch, abort_word)) break;
        check += tot[threadIdx.x & 63];
        x = (float)(tot[0] / (double)W);
    }
    if (threadIdx.x == 0) out[b] = check;
}

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 128, rounds = argc > 2 ? atoi(argv[2]) : 40, work = argc > 3 ? atoi(argv[3]) : 0;
    if (W < 1 || W > 256 || rounds < 1 || rounds > 1000) { printf("W <= 256, rounds <= 1000\n"); return 1; }
    unsigned long long* bufs; int* epoch; double* out; int* abort_word;
    CHECK(hipMalloc(&bufs, (size_t)2 * W * 64 * 8));
    CHECK(hipMemset(bufs, 0, (size_t)2 * W * 64 * 8));
    CHECK(hipMalloc(&epoch, 4)); CHECK(hipMalloc(&out, W * 8)); CHECK(hipMalloc(&abort_word, 4));
    CHECK(hipMemset(abort_word, 0, 4));
    hipEvent_t a, c;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&c));
    double first[2] = {0, 0};
    for (int rep = 0; rep < 6; ++rep) {
        int ep = rep + 1;
        CHECK(hipMemcpy(epoch, &ep, 4, hipMemcpyHostToDevice));
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k_rounds, dim3(W), dim3(512), 0, 0, bufs, rounds, epoch, out, abort_word, work);
        CHECK(hipEventRecord(c, 0));
        CHECK(hipEventSynchronize(c));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, a, c));
        int ab = 0; CHECK(hipMemcpy(&ab, abort_word, 4, hipMemcpyDeviceToHost));
        double h[256]; CHECK(hipMemcpy(h, out, W * 8, hipMemcpyDeviceToHost));
        bool same = true; for (int i = 1; i < W; ++i) same = same && h[i] == h[0];
        printf("rep %d: %d workgroups, %d rounds: %.1f us total, %.2f us per round; abort %d; all workgroups agree: %d (%.6g)\n", rep, W, rounds,
               ms * 1e3, ms * 1e3 / rounds, ab, (int)same, h[0]);
        if (rep == 0) first[0] = h[0];
        if (ab) return 2;
    }
    return 0;
}
