// Cycles per v_mfma_f32_16x16x32_bf16 issued back to back by ONE wave per SIMD, twelve independent accumulators,
// with the accumulators in architectural VGPRs or in accumulation registers, and with the A operand in either.
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_probe tools/mfma_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ void __launch_bounds__(256, 1) k(long long* out, int iters) {
    f32x4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = f32x4{0, 0, 0, 0};
    bf16x8 a0, a1, b[6];
    for (int j = 0; j < 8; ++j) { a0[j] = (__bf16)(threadIdx.x * 0.001f); a1[j] = (__bf16)0.5f; }
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(i + j);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int y = 0; y < 6; ++y) {
            if (MODE == 0) {        // everything in VGPRs
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[2 * y]) : "v"(a0), "v"(b[y]));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[2 * y + 1]) : "v"(a1), "v"(b[y]));
            } else if (MODE == 1) { // accumulators in AGPRs
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[2 * y]) : "v"(a0), "v"(b[y]));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[2 * y + 1]) : "v"(a1), "v"(b[y]));
            } else if (MODE == 2) { // A operand in AGPRs
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[2 * y]) : "a"(a0), "v"(b[y]));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[2 * y + 1]) : "a"(a1), "v"(b[y]));
            } else {                // accumulators and A in AGPRs
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[2 * y]) : "a"(a0), "v"(b[y]));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[2 * y + 1]) : "a"(a1), "v"(b[y]));
            }
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][3];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[MODE] = t1 - t0; }
    if (s == 123.456f) out[7] = 1;
}

int main() {
    long long* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, iters);
        hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, iters);
        hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, iters);
        hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, d, iters);
        hipDeviceSynchronize();
    }
    long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    const char* names[4] = {"acc VGPR, A VGPR", "acc AGPR, A VGPR", "acc VGPR, A AGPR", "acc AGPR, A AGPR"};
    for (int m = 0; m < 4; ++m) printf("%s: %.2f cycles per MFMA (one wave per SIMD)\n", names[m], (double)h[m] / (12.0 * iters));
    return 0;
}
