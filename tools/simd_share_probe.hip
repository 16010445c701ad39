// What a wave that streams v_mfma_f32_16x16x32_bf16 (twelve independent accumulators and one packed VALU op per
// MFMA, like a network task of the move kernel) loses to the OTHER wave of its SIMD, by what that wave does:
// nothing, the same MFMA stream, polling LDS between sleeps, f64 divisions (the PUCT score), f32 VALU work,
// a dependent chain of global loads (a tree walk), LDS permutes.  512 threads = two waves per SIMD; waves 0..3
// stream, waves 4..7 are the partners and stop when the streaming waves are done; then the roles are swapped (the
// SIMD issues the older wave first), and the experiment is repeated with v_mfma_f32_32x32x16_bf16.
// Measured on an MI355X: alone 17.4 cycles per MFMA; two streams 18.7 (older wave) and 33.4 (younger); a younger
// partner of any kind costs the stream <= 3 %; an OLDER partner doing f64 divisions or an f32 chain slows it to 37-41
// cycles per MFMA; polling, pointer chasing and LDS permutes cost nothing either way.  32x32x16: 34.1 alone, 40.1 /
// 68.6 shared, 68-78 under an older vector-heavy partner.
// Build: hipcc --offload-arch=gfx950 -O2 -o simd_share_probe tools/simd_share_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) short s16x2;

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

typedef __attribute__((ext_vector_type(16))) float f32x16;
// the same with v_mfma_f32_32x32x16_bf16 (twice the work per instruction, four independent accumulators)
template <int PARTNER, bool SWAP>
__global__ void __launch_bounds__(512, 2) k32(long long* out, const int* chase, int iters) {
    __shared__ int flag[64];
    const int hw_wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave = SWAP ? (hw_wave ^ 4) : hw_wave;
    if (threadIdx.x < 64) flag[threadIdx.x] = 0;
    __syncthreads();
    const int simd = (int)__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));
    if (wave < 4 || PARTNER == 1) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;
        bf16x8 a0, b[4];
        for (int j = 0; j < 8; ++j) a0[j] = (__bf16)(threadIdx.x * 0.001f);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(i + j);
        s16x2 m = {1, 2};
        const long long t0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[y]) : "v"(a0), "v"(b[y]));
                asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(m) : "v"(y));
                asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(m) : "v"(y));
            }
        }
        const long long t1 = clock64();
        float s = (float)m[0];
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
        if (lane == 0 && wave < 4) __hip_atomic_fetch_add(&flag[0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (blockIdx.x == 0 && lane == 0) { out[8 + 2 * hw_wave] = 3 * (t1 - t0); out[9 + 2 * hw_wave] = simd; }   // x3: 4 per iteration, the report divides by 12
        if (s == 123.456f) out[7] = 1;
        return;
    }
    if (blockIdx.x == 0 && lane == 0) { out[8 + 2 * hw_wave] = 0; out[9 + 2 * hw_wave] = simd; }
    if (PARTNER == 0) return;
    long long work = 0;
    double x = 1.0 + lane, y = 3.0;
    float f = 1.0f + lane;
    while (uni(__hip_atomic_load(&flag[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < 4) {
        if (PARTNER == 3) {
            for (int i = 0; i < 8; ++i) { x = y / (x + 1.0); y = y + 1.0; }
            work += 8;
        } else if (PARTNER == 4) {
            for (int i = 0; i < 64; ++i) f = f * 1.0001f + 0.5f;
            work += 64;
        }
    }
    if (x == 0.123 || f == 0.123f) out[7] = 2;
    if (blockIdx.x == 0 && lane == 0 && hw_wave == (SWAP ? 0 : 4)) out[3] = work;
}

template <int PARTNER, bool SWAP>
__global__ void __launch_bounds__(512, 2) k(long long* out, const int* chase, int iters) {
    __shared__ int flag[64];
    const int hw_wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave = SWAP ? (hw_wave ^ 4) : hw_wave;      // SWAP: the streaming waves are the YOUNGER ones of each SIMD
    if (threadIdx.x < 64) flag[threadIdx.x] = 0;
    __syncthreads();
    const int simd = (int)__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));      // HW_ID bits 5:4
    if (wave < 4 || PARTNER == 1) {
        f32x4 acc[12];
        for (int i = 0; i < 12; ++i) acc[i] = f32x4{0, 0, 0, 0};
        bf16x8 a0, a1, b[6];
        for (int j = 0; j < 8; ++j) { a0[j] = (__bf16)(threadIdx.x * 0.001f); a1[j] = (__bf16)0.5f; }
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(i + j);
        s16x2 m = {1, 2};
        const long long t0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int y = 0; y < 6; ++y) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[2 * y]) : "v"(a0), "v"(b[y]));
                asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(m) : "v"(y));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[2 * y + 1]) : "v"(a1), "v"(b[y]));
                asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(m) : "v"(y));
            }
        }
        const long long t1 = clock64();
        float s = (float)m[0];
        for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][3];
        if (lane == 0 && wave < 4) __hip_atomic_fetch_add(&flag[0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (blockIdx.x == 0 && lane == 0) { out[8 + 2 * hw_wave] = t1 - t0; out[9 + 2 * hw_wave] = simd; }
        if (s == 123.456f) out[7] = 1;
        return;
    }
    if (blockIdx.x == 0 && lane == 0) { out[8 + 2 * hw_wave] = 0; out[9 + 2 * hw_wave] = simd; }
    if (PARTNER == 0) return;
    long long work = 0;
    double x = 1.0 + lane, y = 3.0;
    float f = 1.0f + lane;
    int at = lane;
    while (uni(__hip_atomic_load(&flag[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < 4) {
        if (PARTNER == 2) {              // an idle worker: sleep, then read a dozen queue words
            __builtin_amdgcn_s_sleep(2);
            int s = 0;
            for (int i = 1; i < 12; ++i) s += uni(__hip_atomic_load(&flag[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            work += s + 1;
        } else if (PARTNER == 3) {       // f64 divisions
            for (int i = 0; i < 8; ++i) { x = y / (x + 1.0); y = y + 1.0; }
            work += 8;
        } else if (PARTNER == 4) {       // f32 VALU
            for (int i = 0; i < 64; ++i) f = f * 1.0001f + 0.5f;
            work += 64;
        } else if (PARTNER == 5) {       // dependent global loads
            for (int i = 0; i < 4; ++i) at = chase[at];
            work += 4;
        } else if (PARTNER == 6) {       // LDS permutes
            for (int i = 0; i < 16; ++i) at = __shfl(at, (lane + 1) & 63);
            work += 16;
        }
    }
    if (x == 0.123 || f == 0.123f || at == -7) out[7] = 2;
    if (blockIdx.x == 0 && lane == 0 && wave == 4) out[3] = work;
}

template <int PARTNER, bool SWAP>
static void run32(const char* name, long long* d, const int* chase, int iters) {
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(d, 0, 256);
        hipLaunchKernelGGL((k32<PARTNER, SWAP>), dim3(256), dim3(512), 0, 0, d, chase, iters);
        hipDeviceSynchronize();
    }
    long long h[32]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("32x32x16, %s partner %-24s:", SWAP ? "older  " : "younger", name);
    for (int w = 0; w < 8; ++w) printf("  w%d@simd%lld %.1f", w, h[9 + 2 * w], (double)h[8 + 2 * w] / (12.0 * iters));
    printf("  (cycles per MFMA; partner loop items %lld)\n", h[3]);
}

template <int PARTNER, bool SWAP>
static void run(const char* name, long long* d, const int* chase, int iters) {
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(d, 0, 256);
        hipLaunchKernelGGL((k<PARTNER, SWAP>), dim3(256), dim3(512), 0, 0, d, chase, iters);
        hipDeviceSynchronize();
    }
    long long h[32]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("%s partner %-24s:", SWAP ? "older  " : "younger", name);
    for (int w = 0; w < 8; ++w) printf("  w%d@simd%lld %.1f", w, h[9 + 2 * w], (double)h[8 + 2 * w] / (12.0 * iters));
    printf("  (cycles per MFMA; partner loop items %lld)\n", h[3]);
}

int main() {
    long long* d; hipMalloc(&d, 256);
    const int n = 1 << 22;                  // 16 MB pointer chase: every load misses L1
    int* hc = new int[n];
    for (int i = 0; i < n; ++i) hc[i] = (int)(((long long)i * 1048583 + 12345) % n);
    int* chase; hipMalloc(&chase, n * sizeof(int)); hipMemcpy(chase, hc, n * sizeof(int), hipMemcpyHostToDevice);
    const int iters = 2000;
    run<0, false>("none (exits)", d, chase, iters);
    run<1, false>("the same MFMA stream", d, chase, iters);
    run<2, false>("sleep + poll LDS", d, chase, iters);
    run<3, false>("f64 divisions", d, chase, iters);
    run<4, false>("f32 VALU chain", d, chase, iters);
    run<5, false>("dependent global loads", d, chase, iters);
    run<6, false>("LDS permutes", d, chase, iters);
    run<2, true>("sleep + poll LDS", d, chase, iters);
    run<3, true>("f64 divisions", d, chase, iters);
    run<4, true>("f32 VALU chain", d, chase, iters);
    run<5, true>("dependent global loads", d, chase, iters);
    run<6, true>("LDS permutes", d, chase, iters);
    run32<0, false>("none (exits)", d, chase, iters);
    run32<1, false>("the same MFMA stream", d, chase, iters);
    run32<3, false>("f64 divisions", d, chase, iters);
    run32<4, false>("f32 VALU chain", d, chase, iters);
    run32<3, true>("f64 divisions", d, chase, iters);
    run32<4, true>("f32 VALU chain", d, chase, iters);
    return 0;
}
