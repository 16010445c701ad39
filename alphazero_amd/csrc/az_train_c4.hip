// az_train_c4.hip -- forward and backward of the training step for the Connect Four network with 32 trunk
// channels (alphazero/training.py:57-71 with alphazero/models.py:6-87, alphazero/games/connectfour.py:47-72 and
// alphazero/loss.py:29-63), hand-written: no vendor convolution, no autograd.  Everything is fp32, like the
// reference's training.
//
// Batch-norm runs in training mode (statistics of the batch), so every conv is followed by a reduction over the
// whole batch before the next layer can start, and every layer's backward by another one.  The step is therefore
// a chain of 4 * blocks + 11 launches (35 for the shipped 6 x 32 network) with the launch boundaries as the only
// synchronisation: a workgroup owns one board (42 cells x 32 channels; the conv's halo is the zero padding, so
// workgroups share nothing) and ends by writing its board's partial sums; every workgroup of the NEXT launch begins
// by adding all boards' partial sums up for itself (32 KB from L2, fixed order, double precision: deterministic, no
// atomics, no fences).  (A first version let the last workgroup of a launch do that sum behind a device-scope
// counter: 18 of each launch's 30 us went into it.)
//   * activations: channel-last fp32 [board][cell][32]; kept per layer: the raw conv outputs Z (batch-norm input)
//     and the block inputs X; everything else (ReLU(BN(Z)), x-hat, ReLU masks) is recomputed from those while a
//     tile is staged into LDS;
//   * forward conv: the board's 42 x 32 x 288 contraction on v_mfma_f32_32x32x2_f32, input tile and weights in LDS,
//     eight waves = two cell tiles x four tap groups;
//   * backward of a conv layer in ONE launch: dZ from the incoming gradient by the batch-norm backward formula
//     (while staging), input gradient (the conv with flipped taps and transposed weights), ReLU mask, the partial
//     sums of the previous batch-norm's backward, and the board's share of the weight gradient (summed over the
//     boards by the last launch, and over the applications of the shared block on the way: the tower applies ONE
//     block num_blocks times, alphazero/models.py:78);
//   * heads: 1x1 convs + batch-norm + fully connected layers per board, their weight gradients as sums over the
//     batch by one thread per weight.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/azhip.h"
#include "az_host.h"

#define TC 32                  // trunk channels
#define CELLS 42
#define PADCELLS 72            // the board with its zero halo: rows -1..6, columns -1..7
#define TT 256                 // threads per workgroup of the small kernels: thread = (channel, one of 8 cell groups)
#define NPOS 6                 // cells per thread there (8 groups x 6 >= 42)
#define TTC 512                // threads per workgroup of the 3x3 conv kernels: two waves per SIMD, 16 cell groups
#define NPC 3                  // cells per thread there (16 groups x 3 >= 42): cell = group + 16 k
#define TS 33                  // floats per cell in the conv kernels' LDS tiles (32 channels + 1: lanes that read one
                               // channel of 32 cells hit different banks)
#define PBS 65                 // floats per channel row of the partial-result buffer of the conv kernels
#define EPT 3                  // elements of a board tensor [42][32] per thread of a conv kernel: element tid + 512 j is
                               // (cell = group + 16 j, channel = tid & 31) -- the same cells the thread computes

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int pad_index(int cell) { return (cell / 7 + 1) * 9 + cell % 7 + 1; }
__device__ __forceinline__ int tap_offset(int tap) { return (tap / 3 - 1) * 9 + tap % 3 - 1; }

// The workgroups of the PREVIOUS launch have each written a row part[board][64] (two quantities x 32 channels, summed
// over the board's cells).  Every workgroup of this launch adds the rows up for itself, in board order, in double:
// 32 KB of L2 reads, no atomics, no fences, the same bits in every workgroup and every run.  tot = LDS [64].
__device__ __forceinline__ void reduce_partials(const float* __restrict__ part, int boards, double* tot,
                                                double* scratch /* LDS [8][64] */) {
    const int col = threadIdx.x & 63, q = threadIdx.x >> 6, nq = blockDim.x >> 6;
    double s = 0.0;
    for (int b0 = q; b0 < boards; b0 += 32 * nq) {         // up to 32 loads in flight (one trip to memory for 128
        float v[32];                                       // boards), added in board order
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = b0 + nq * j < boards ? part[(size_t)(b0 + nq * j) * 64 + col] : 0.0f;
#pragma unroll
        for (int j = 0; j < 32; ++j) s += (double)v[j];
    }
    scratch[q * 64 + col] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        double t = 0.0;
        for (int k = 0; k < nq; ++k) t += scratch[k * 64 + col];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
}

// per board: the 32 (or fewer) column sums of two quantities over the board's cells, from per-thread values
// (thread = (channel = tid & 31, group = tid >> 5)), into part[board][0..31] and part[board][32..63]
__device__ __forceinline__ void board_partials(float a, float b, float* red /* LDS [2][groups][32] */, float* part_row) {
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5, ng = blockDim.x >> 5;
    red[g * 32 + c] = a;
    red[ng * 32 + g * 32 + c] = b;
    __syncthreads();
    if (threadIdx.x < 64) {
        const float* r = red + (threadIdx.x >> 5) * ng * 32 + (threadIdx.x & 31);
        float s = 0.0f;
        for (int k = 0; k < ng; ++k) s += r[k * 32];
        part_row[threadIdx.x] = s;
    }
}

// statistics of a batch-norm input from the totals (sum, sum of squares): st[0..31] mean, [32..63] 1/sqrt(var + eps),
// [64..95] biased variance -- into LDS for this workgroup, and by workgroup 0 into `keep` for the launches to come
__device__ __forceinline__ void stats_from_totals(const double* tot, int nch, long n, float eps, float* st /* LDS [96] */,
                                                  float* keep) {
    if ((int)threadIdx.x < nch) {
        const int c = threadIdx.x;
        const double mean = tot[c] / (double)n;
        double var = tot[32 + c] / (double)n - mean * mean;
        if (var < 0.0) var = 0.0;
        st[c] = (float)mean;
        st[32 + c] = (float)(1.0 / sqrt(var + (double)eps));
        st[64 + c] = (float)var;
        if (blockIdx.x == 0) { keep[c] = st[c]; keep[32 + c] = st[32 + c]; keep[64 + c] = st[64 + c]; }
    }
    __syncthreads();
}

// a batch-norm whose statistics are being made by this launch (from the previous launch's partial sums) and kept
// in stat_keep [96] for later launches
struct BnRef { float* stat_keep; const float* gamma; const float* beta; };

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
// weights of the 3x3 convs in the orders the kernels read them: wf [tap][ci][co] (forward), wb [tap][co][ci]
// (input gradient); stem: ws [tap * 2 + plane][co].  Also clears what the step accumulates into.
__global__ void __launch_bounds__(TT)
k_train_prep(const float* __restrict__ c1, const float* __restrict__ c2, const float* __restrict__ stem, float* __restrict__ wf1,
             float* __restrict__ wb1, float* __restrict__ wf2, float* __restrict__ wb2, float* __restrict__ ws,
             float* __restrict__ bn_acc /* [3][64] */) {
    const int i = blockIdx.x * TT + threadIdx.x;
    if (i < 9 * TC * TC) {
        const int tap = i / (TC * TC), ci = (i / TC) % TC, co = i % TC;
        const float a = c1[(co * TC + ci) * 9 + tap], b = c2[(co * TC + ci) * 9 + tap];
        wf1[i] = a; wf2[i] = b;
        wb1[(tap * TC + co) * TC + ci] = a; wb2[(tap * TC + co) * TC + ci] = b;
    }
    if (i < 18 * TC) {
        const int tap = i / (2 * TC), plane = (i / TC) % 2, co = i % TC;
        ws[i] = stem[(co * 2 + plane) * 9 + tap];
    }
    if (i < 3 * 64) bn_acc[i] = 0.0f;
}

// stem: conv 2 -> 32 from the rendered planes x [board][2][6][7] -> Z0 and the board's sums of Z0, Z0^2
__global__ void __launch_bounds__(TT)
k_train_stem_fwd(const float* __restrict__ x, const float* __restrict__ ws, float* __restrict__ zout, float* __restrict__ part_out) {
    __shared__ float xt[PADCELLS * 2];
    __shared__ float wl[18 * TC];
    __shared__ float red[512];
    const int b = blockIdx.x, tid = threadIdx.x, co = tid & 31, g = tid >> 5;
    for (int i = tid; i < PADCELLS * 2; i += TT) xt[i] = 0.0f;
    for (int i = tid; i < 18 * TC; i += TT) wl[i] = ws[i];
    __syncthreads();
    for (int i = tid; i < 2 * CELLS; i += TT) xt[pad_index(i % CELLS) * 2 + i / CELLS] = x[(size_t)b * 84 + i];
    __syncthreads();
    float s = 0.0f, q = 0.0f;
#pragma unroll
    for (int k = 0; k < NPOS; ++k) {
        const int cell = g + 8 * k;
        if (cell >= CELLS) break;
        const int pi = pad_index(cell);
        float acc = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int pj = (pi + tap_offset(tap)) * 2;
            acc = fmaf(xt[pj], wl[(tap * 2) * TC + co], acc);
            acc = fmaf(xt[pj + 1], wl[(tap * 2 + 1) * TC + co], acc);
        }
        zout[((size_t)b * CELLS + cell) * TC + co] = acc;
        s += acc; q = fmaf(acc, acc, q);
    }
    board_partials(s, q, red, part_out + (size_t)b * 64);
}

// scale / shift of a batch-norm in training mode from its statistics (LDS): y = z * sc + sh
__device__ __forceinline__ void bn_scale_shift(const float* st, const float* gamma, const float* beta, float* sc, float* sh) {
    if (threadIdx.x < TC) {
        const int c = threadIdx.x;
        const float k = gamma[c] * st[32 + c];
        sc[c] = k;
        sh[c] = beta[c] - st[c] * k;
    }
}

// the board's input activation into the padded LDS tile, for the head kernel (256 threads):
// A = ReLU(BN(Zin) (+ Xprev)); stored to xout when given
template <bool RES, bool WRITE_X>
__device__ __forceinline__ void stage_activation(float* tile, const float* zin, const float* xprev, float* xout,
                                                 const float* sc, const float* sh, int b) {
    const int tid = threadIdx.x;
    for (int i = tid; i < CELLS * TC; i += TT) {
        const int cell = i >> 5, c = i & 31;
        const size_t gi = (size_t)b * CELLS * TC + i;
        float a = fmaf(zin[gi], sc[c], sh[c]);
        if (RES) a += xprev[gi];
        a = fmaxf(a, 0.0f);
        if (WRITE_X) xout[gi] = a;
        tile[pad_index(cell) * TC + c] = a;
    }
}

// The 3x3 conv of one board on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32: D[32 out][32 cells] += W[out][2 k]
// X[2 k][cells]), by the workgroup's eight waves: wave = (cell tile: cells 0..31 / 32..63 of which 42 exist, tap
// group: taps 0-2 / 3-4 / 5-6 / 7-8).  The forward conv (SIGN = +1, w = [tap][ci][co]) and the input gradient
// (SIGN = -1, w = [tap][co][ci]) are the same loop.  A wave leaves its partial result in pbuf [tap group][out][PBS];
// the caller adds the four groups up in order.  tile = [72 padded cells][TS].
template <int SIGN>
__device__ __forceinline__ void conv_mfma(const float* tile, const float* w, float* pbuf, int wave, int lane) {
    const int col = lane & 31, h = lane >> 5, pt = wave & 1, kq = wave >> 1;
    const int cell = pt * 32 + col;
    const int pi = pad_index(cell < CELLS ? cell : 0);
    const int t0 = kq == 0 ? 0 : 1 + 2 * kq, nt = kq == 0 ? 3 : 2;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll 1
    for (int t = t0; t < t0 + nt; ++t) {
        const float* xs = tile + (pi + SIGN * tap_offset(t)) * TS + h;
        const float* ws = w + (t * TC + h) * TC + col;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) acc = MFMA32(ws[2 * s2 * TC], xs[2 * s2], acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) pbuf[(kq * TC + (r & 3) + 8 * (r >> 2) + 4 * h) * PBS + cell] = acc[r];
}
// the four tap groups' partial results of (channel c, cell), added in order
__device__ __forceinline__ float conv_total(const float* pbuf, int c, int cell) {
    const float* p = pbuf + c * PBS + cell;
    return ((p[0] + p[TC * PBS]) + p[2 * TC * PBS]) + p[3 * TC * PBS];
}

// The board's share of a conv's weight gradient on the same instruction: dW[tap][ci][co] = sum over cells of
// A[cell + tap][ci] dZ[cell][co] = D[32 ci][32 co] with the cells as k, 21 steps of two; wave w takes tap w (wave 0
// also the ninth).  out = [tap][ci][co].
// What earlier applications of the shared block have left in this wave's part of the board's share (requested at
// the start of the launch, used at its end).
struct WgOld { f32x16 a, b; };
template <bool ACCUM>
__device__ __forceinline__ WgOld wgrad_load_old(const float* out, int wave, int lane) {
    WgOld o;
    const int col = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        o.a[r] = ACCUM ? out[(wave * TC + row) * TC + col] : 0.0f;
        o.b[r] = (ACCUM && wave == 0) ? out[(8 * TC + row) * TC + col] : 0.0f;
    }
    return o;
}
template <bool ACCUM>
__device__ __forceinline__ void wgrad_mfma(const float* at, const float* dzt, float* out, const WgOld& old, int wave, int lane) {
    const int col = lane & 31, h = lane >> 5;
#pragma unroll 1
    for (int t = wave; t < 9; t += 8) {
        const int off = tap_offset(t);
        f32x16 acc = t == wave ? old.a : old.b;
#pragma unroll 3
        for (int s2 = 0; s2 < CELLS / 2; ++s2) {
            const int pi = pad_index(2 * s2 + h);
            acc = MFMA32(at[(pi + off) * TS + col], dzt[pi * TS + col], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(t * TC + (r & 3) + 8 * (r >> 2) + 4 * h) * TC + col] = acc[r];
    }
}

// the conv's 36 KB of weights, global -> registers (issued early) -> LDS
struct WRegs { float4 r[5]; };
__device__ __forceinline__ WRegs load_conv_weights(const float* __restrict__ w) {
    WRegs x;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int i = threadIdx.x + j * TTC;
        x.r[j] = i < 9 * TC * TC / 4 ? reinterpret_cast<const float4*>(w)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return x;
}
__device__ __forceinline__ void store_conv_weights(float* wl, const WRegs& x) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int i = threadIdx.x + j * TTC;
        if (i < 9 * TC * TC / 4) reinterpret_cast<float4*>(wl)[i] = x.r[j];
    }
}

// one 3x3 conv of the tower: Zout = conv(A), A = ReLU(BN(Zin) (+ Xprev)); BN's statistics from part_in (the sums of
// Zin, Zin^2 per board); the board's sums of Zout, Zout^2 -> part_out.  Every global load of the launch is issued
// before the first wait: what the previous launch wrote comes from beyond this chip's L2 (2 us), once.
template <bool RES, bool WRITE_X>
__global__ void __launch_bounds__(TTC)
k_train_conv_fwd(const float* __restrict__ zin, const float* __restrict__ part_in, BnRef bn, const float* __restrict__ xprev,
                 float* __restrict__ xout, const float* __restrict__ wf, float* __restrict__ zout, float* __restrict__ part_out,
                 int boards, float eps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;                                   // [9][32][32]
    float* tile = lds + 9 * TC * TC;                   // [72][TS]
    float* sc = tile + PADCELLS * TS;                  // [32], [32]
    float* sh = sc + TC;
    float* red = sh + TC;                              // [1024]
    float* pbuf = red + 1024;                          // [4][32][PBS]
    __shared__ double tot[64], scratch[512];
    __shared__ float st[96];
    const int b = blockIdx.x, tid = threadIdx.x, co = tid & 31, g = tid >> 5;
    const size_t base = (size_t)b * CELLS * TC;
    float zr[EPT], xr[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = tid + j * TTC;
        zr[j] = i < CELLS * TC ? zin[base + i] : 0.0f;
        xr[j] = (RES && i < CELLS * TC) ? xprev[base + i] : 0.0f;
    }
    const WRegs wr = load_conv_weights(wf);
    for (int i = tid; i < PADCELLS * TS; i += TTC) tile[i] = 0.0f;
    reduce_partials(part_in, boards, tot, scratch);
    store_conv_weights(wl, wr);
    stats_from_totals(tot, TC, (long)boards * CELLS, eps, st, bn.stat_keep);
    bn_scale_shift(st, bn.gamma, bn.beta, sc, sh);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = tid + j * TTC;
        if (i < CELLS * TC) {
            float a = fmaf(zr[j], sc[co], sh[co]);
            if (RES) a += xr[j];
            a = fmaxf(a, 0.0f);
            if (WRITE_X) xout[base + i] = a;
            tile[pad_index(i >> 5) * TS + co] = a;
        }
    }
    __syncthreads();
    conv_mfma<1>(tile, wl, pbuf, tid >> 6, tid & 63);
    __syncthreads();
    float s = 0.0f, q = 0.0f;
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
        const int cell = g + 16 * k;
        if (cell < CELLS) {
            const float zv = conv_total(pbuf, co, cell);
            zout[base + cell * TC + co] = zv;
            s += zv; q = fmaf(zv, zv, q);
        }
    }
    board_partials(s, q, red, part_out + (size_t)b * 64);
}

// heads, first part: the tower's output X = ReLU(BN(Zin) (+ Xprev)) (stored), the three 1x1 convs (two policy
// planes with bias, one value plane) -> zh [board][cell][4], the board's sums of the three planes -> part_out
template <bool RES>
__global__ void __launch_bounds__(TT)
k_train_head_conv_fwd(const float* __restrict__ zin, const float* __restrict__ part_in, BnRef bn, const float* __restrict__ xprev,
                      float* __restrict__ xout, const float* __restrict__ wp, const float* __restrict__ bp,
                      const float* __restrict__ wv, float* __restrict__ zh, float* __restrict__ part_out, int boards, float eps) {
    __shared__ __attribute__((aligned(16))) float tile[PADCELLS * TC];
    __shared__ float sc[TC], sh[TC], wh[3 * TC], red[512], st[96];
    __shared__ double tot[64], scratch[512];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < 2 * TC) wh[tid] = wp[tid];
    if (tid < TC) wh[2 * TC + tid] = wv[tid];
    reduce_partials(part_in, boards, tot, scratch);
    stats_from_totals(tot, TC, (long)boards * CELLS, eps, st, bn.stat_keep);
    bn_scale_shift(st, bn.gamma, bn.beta, sc, sh);
    __syncthreads();
    stage_activation<RES, true>(tile, zin, xprev, xout, sc, sh, b);
    __syncthreads();
    float s = 0.0f, q = 0.0f;
    if (tid < 3 * CELLS) {
        const int cell = tid % CELLS, h = tid / CELLS;
        const float* a = tile + pad_index(cell) * TC;
        float acc = 0.0f;
#pragma unroll 8
        for (int c = 0; c < TC; ++c) acc = fmaf(a[c], wh[h * TC + c], acc);
        if (h < 2) acc += bp[h];
        zh[((size_t)b * CELLS + cell) * 4 + h] = acc;
        s = acc; q = acc * acc;
    }
    // column sums over the 42 cells of each plane: thread (cell, h) -> red[h][cell], summed by six threads
    red[tid] = s; red[256 + tid] = q;
    __syncthreads();
    if (tid < 64) {
        float t = 0.0f;
        const int h = tid & 31;
        if (h < 3) {
            const float* r = red + (tid >= 32 ? 256 : 0) + h * CELLS;
            for (int k = 0; k < CELLS; ++k) t += r[k];
        }
        part_out[(size_t)b * 64 + tid] = t;
    }
}

// heads, second part, one board per workgroup: batch-norm + ReLU of the three planes, policy FC 84 -> 42 and the
// landing-cell gather (connectfour.py:56-71), value FC 42 -> 256 -> 1 + tanh
__global__ void __launch_bounds__(TT)
k_train_head_fc_fwd(const float* __restrict__ x, const float* __restrict__ zh, const float* __restrict__ part_in,
                    float* __restrict__ stat_h_keep, const float* __restrict__ gp, const float* __restrict__ btp,
                    const float* __restrict__ gv, const float* __restrict__ btv, const float* __restrict__ wpf,
                    const float* __restrict__ bpf, const float* __restrict__ w1, const float* __restrict__ b1,
                    const float* __restrict__ w2, const float* __restrict__ b2,
                    float* __restrict__ hplanes /* [board][128]: 84 policy, 42 value */, float* __restrict__ h1 /* [board][256] */,
                    float* __restrict__ logits7, float* __restrict__ vhat, int boards, float eps) {
    __shared__ float hp[128], cl[CELLS], red[TT], stat_h[96];
    __shared__ int height[8];
    __shared__ double tot[64], scratch[512];
    const int b = blockIdx.x, tid = threadIdx.x;
    reduce_partials(part_in, boards, tot, scratch);
    stats_from_totals(tot, 3, (long)boards * CELLS, eps, stat_h, stat_h_keep);
    if (tid < 3 * CELLS) {
        const int cell = tid % CELLS, h = tid / CELLS;
        const float gamma = h < 2 ? gp[h] : gv[0], beta = h < 2 ? btp[h] : btv[0];
        const float z = zh[((size_t)b * CELLS + cell) * 4 + h];
        const float y = fmaxf(fmaf((z - stat_h[h]) * stat_h[32 + h], gamma, beta), 0.0f);
        hp[h * CELLS + cell] = y;
        hplanes[(size_t)b * 128 + h * CELLS + cell] = y;
    }
    if (tid >= 128 && tid < 135) {
        const int col = tid - 128;
        float n = 0.0f;
        for (int k = 0; k < 12; ++k) n += x[(size_t)b * 84 + k * 7 + col];      // both planes, six rows
        height[col] = (int)(n + 0.5f);
    }
    __syncthreads();
    if (tid < CELLS) {
        float acc = bpf[tid];
        const float* w = wpf + tid * 84;
#pragma unroll 4
        for (int k = 0; k < 84; ++k) acc = fmaf(w[k], hp[k], acc);
        cl[tid] = acc;
    }
    float hid;
    {
        float acc = b1[tid];
        const float* w = w1 + tid * CELLS;
#pragma unroll 6
        for (int c = 0; c < CELLS; ++c) acc = fmaf(w[c], hp[84 + c], acc);
        hid = fmaxf(acc, 0.0f);
        h1[(size_t)b * 256 + tid] = hid;
        red[tid] = hid * w2[tid];
    }
    __syncthreads();
    if (tid < 7) logits7[(size_t)b * 7 + tid] = height[tid] < 6 ? cl[height[tid] * 7 + tid] : 0.0f;
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) vhat[b] = tanhf(red[0] + b2[0]);
}

// ---------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------
// heads, one board per workgroup: from d loss / d logits7 and d loss / d v to the gradient at the three head
// planes' batch-norm outputs (ReLU mask applied) -> dyh [board][cell][4]; keeps dh1 [board][256], dpre2 [board],
// dcl [board][42] for the weight gradients; partial sums of the head batch-norms' backward
__global__ void __launch_bounds__(TT)
k_train_head_fc_bwd(const float* __restrict__ x, const float* __restrict__ dlogits7, const float* __restrict__ dv,
                    const float* __restrict__ vhat, const float* __restrict__ h1, const float* __restrict__ hplanes,
                    const float* __restrict__ zh, const float* __restrict__ stat_h, const float* __restrict__ wpf,
                    const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ dh1, float* __restrict__ dpre2,
                    float* __restrict__ dcl, float* __restrict__ dyh, float* __restrict__ part_out) {
    __shared__ float d1[TT], dcl_s[CELLS], red[512];
    __shared__ int height[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float v = vhat[b];
    const float dp = dv[b] * (1.0f - v * v);
    {
        const float d = h1[(size_t)b * 256 + tid] > 0.0f ? dp * w2[tid] : 0.0f;
        d1[tid] = d;
        dh1[(size_t)b * 256 + tid] = d;
    }
    if (tid == 0) dpre2[b] = dp;
    if (tid < CELLS) dcl_s[tid] = 0.0f;
    if (tid >= 128 && tid < 135) {
        const int col = tid - 128;
        float n = 0.0f;
        for (int k = 0; k < 12; ++k) n += x[(size_t)b * 84 + k * 7 + col];
        height[col] = (int)(n + 0.5f);
    }
    __syncthreads();
    if (tid < 7 && height[tid] < 6) dcl_s[height[tid] * 7 + tid] = dlogits7[(size_t)b * 7 + tid];
    __syncthreads();
    if (tid < CELLS) dcl[(size_t)b * CELLS + tid] = dcl_s[tid];
    float s1 = 0.0f, s2 = 0.0f;
    if (tid < 3 * CELLS) {
        const int cell = tid % CELLS, h = tid / CELLS;
        float d = 0.0f;
        if (h < 2) {
            const int k = h * CELLS + cell;
            for (int col = 0; col < 7; ++col)
                if (height[col] < 6) d = fmaf(dcl_s[height[col] * 7 + col], wpf[(height[col] * 7 + col) * 84 + k], d);
        } else {
#pragma unroll 8
            for (int j = 0; j < 256; ++j) d = fmaf(d1[j], w1[j * CELLS + cell], d);
        }
        if (!(hplanes[(size_t)b * 128 + h * CELLS + cell] > 0.0f)) d = 0.0f;
        dyh[((size_t)b * CELLS + cell) * 4 + h] = d;
        const float xh = (zh[((size_t)b * CELLS + cell) * 4 + h] - stat_h[h]) * stat_h[32 + h];
        s1 = d; s2 = d * xh;
    }
    red[tid] = s1; red[256 + tid] = s2;
    __syncthreads();
    if (tid < 64) {
        float t = 0.0f;
        const int h = tid & 31;
        if (h < 3) {
            const float* r = red + (tid >= 32 ? 256 : 0) + h * CELLS;
            for (int k = 0; k < CELLS; ++k) t += r[k];
        }
        part_out[(size_t)b * 64 + tid] = t;
    }
}

// Sum over the batch by a workgroup of 256 threads for 64 outputs: thread (output = tid & 63, part = tid >> 6) adds
// the boards part, part + 4, ... (eight loads in flight), the four parts are added in order.  f(board) = the term.
template <class F>
__device__ __forceinline__ float batch_sum(int boards, float* red /* LDS [256] */, F f) {
    const int q = threadIdx.x >> 6;
    float s = 0.0f;
    for (int b0 = q; b0 < boards; b0 += 32) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = b0 + 4 * j < boards ? f(b0 + 4 * j) : 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    return red[threadIdx.x & 63] + red[64 + (threadIdx.x & 63)] + red[128 + (threadIdx.x & 63)] + red[192 + (threadIdx.x & 63)];
}

// weight gradients of the fully connected layers: sums over the batch, 64 weights per workgroup
__global__ void __launch_bounds__(TT)
k_train_fc_wgrad(const float* __restrict__ dh1, const float* __restrict__ dpre2, const float* __restrict__ dcl,
                 const float* __restrict__ h1, const float* __restrict__ hplanes, int boards, float* __restrict__ g_w1,
                 float* __restrict__ g_b1, float* __restrict__ g_w2, float* __restrict__ g_b2, float* __restrict__ g_wpf,
                 float* __restrict__ g_bpf) {
    __shared__ float red[TT];
    int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const bool writer = threadIdx.x < 64;
    // (each range below is a multiple of 64 long or the last one, so a workgroup never straddles two of them)
    if (i < 256 * CELLS) {
        const int j = i / CELLS, c = i % CELLS;
        const float s = batch_sum(boards, red, [&](int b) { return dh1[(size_t)b * 256 + j] * hplanes[(size_t)b * 128 + 84 + c]; });
        if (writer) g_w1[i] = s;
        return;
    }
    i -= 256 * CELLS;
    if (i < 256) {
        const float s = batch_sum(boards, red, [&](int b) { return dh1[(size_t)b * 256 + i]; });
        if (writer) g_b1[i] = s;
        return;
    }
    i -= 256;
    if (i < 256) {
        const float s = batch_sum(boards, red, [&](int b) { return dpre2[b] * h1[(size_t)b * 256 + i]; });
        if (writer) g_w2[i] = s;
        return;
    }
    i -= 256;
    if (i < 64) {                                          // one output: the value head's last bias
        const float s = batch_sum(boards, red, [&](int b) { return i == 0 ? dpre2[b] : 0.0f; });
        if (writer && i == 0) g_b2[0] = s;
        return;
    }
    i -= 64;
    if (i < 56 * 64) {                                     // 3528 weights in 56 workgroups (the last one partly idle)
        const bool in = i < CELLS * 84;
        const int cell = in ? i / 84 : 0, k = in ? i % 84 : 0;
        const float s = batch_sum(boards, red, [&](int b) { return dcl[(size_t)b * CELLS + cell] * hplanes[(size_t)b * 128 + k]; });
        if (writer && in) g_wpf[i] = s;
        return;
    }
    i -= 56 * 64;
    {
        const bool in = i >= 0 && i < CELLS;
        const int cell = in ? i : 0;
        const float s = batch_sum(boards, red, [&](int b) { return dcl[(size_t)b * CELLS + cell]; });
        if (writer && in) g_bpf[cell] = s;
    }
}
#define FC_WGRAD_GROUPS (256 * CELLS / 64 + 4 + 4 + 1 + 56 + 1)

// batch-norm backward coefficients of a layer from its saved statistics and the batch sums of its incoming gradient
// (tot: sum dy, sum dy * xhat, just added up from the previous launch's partial sums):
// dz = k1 * (dy - m1 - xhat * m2), xhat = (z - mean) * invstd
struct BnBack { const float* stat; const float* gamma; };
__device__ __forceinline__ void bn_back_coeff(const BnBack& bb, const double* tot, int nch, float inv_n, float* k1, float* m1,
                                              float* m2, float* mean, float* invstd) {
    if ((int)threadIdx.x < nch) {
        const int c = threadIdx.x;
        mean[c] = bb.stat[c];
        invstd[c] = bb.stat[32 + c];
        k1[c] = bb.gamma[c] * bb.stat[32 + c];
        m1[c] = (float)tot[c] * inv_n;
        m2[c] = (float)tot[32 + c] * inv_n;
    }
}
// the same sums are that batch-norm's own gradients (d beta = sum dy, d gamma = sum dy * xhat): workgroup 0 adds them
// to the batch-norm's accumulators (the shared block's batch-norms are applied num_blocks times)
__device__ __forceinline__ void accumulate_bn_grad(const double* tot, float* bn_acc /* [64] */) {
    if (blockIdx.x == 0 && threadIdx.x < 64) bn_acc[threadIdx.x] += (float)tot[threadIdx.x];
}

// heads, last part of the backward: dzh from dyh, the gradient at the tower's output X (three planes back through
// the 1x1 convs), its ReLU mask -> dU [board][cell][32], partial sums of the last tower batch-norm's backward
// (x-hat from zprev), and the board's share of the 1x1 convs' weight and bias gradients -> pwh [board][128]
__global__ void __launch_bounds__(TT)
k_train_head_conv_bwd(const float* __restrict__ dyh, const float* __restrict__ zh, const float* __restrict__ stat_h,
                      const float* __restrict__ part_in, const float* __restrict__ gp, const float* __restrict__ gv,
                      float* __restrict__ g_gp, float* __restrict__ g_btp, float* __restrict__ g_gv, float* __restrict__ g_btv,
                      const float* __restrict__ wp, const float* __restrict__ wv, const float* __restrict__ xtop,
                      const float* __restrict__ zprev, const float* __restrict__ stat_prev, float* __restrict__ du,
                      float* __restrict__ pwh, float* __restrict__ part_out, int boards) {
    __shared__ float dz[CELLS * 4], wh[3 * TC], red[512];
    __shared__ double tot[64], scratch[512];
    const int b = blockIdx.x, tid = threadIdx.x, ci = tid & 31, g = tid >> 5;
    const float inv_n = 1.0f / (float)((long)boards * CELLS);
    if (tid < 2 * TC) wh[tid] = wp[tid];
    if (tid < TC) wh[2 * TC + tid] = wv[tid];
    reduce_partials(part_in, boards, tot, scratch);
    if (b == 0 && tid < 3) {
        // the head batch-norms' own gradients: d beta = sum dy, d gamma = sum dy * x-hat
        if (tid < 2) { g_btp[tid] = (float)tot[tid]; g_gp[tid] = (float)tot[32 + tid]; }
        else { g_btv[0] = (float)tot[2]; g_gv[0] = (float)tot[34]; }
    }
    if (tid < 3 * CELLS) {
        const int cell = tid % CELLS, h = tid / CELLS;
        const float gamma = h < 2 ? gp[h] : gv[0];
        const float xh = (zh[((size_t)b * CELLS + cell) * 4 + h] - stat_h[h]) * stat_h[32 + h];
        dz[cell * 4 + h] = gamma * stat_h[32 + h] *
                           (dyh[((size_t)b * CELLS + cell) * 4 + h] - (float)tot[h] * inv_n - xh * ((float)tot[32 + h] * inv_n));
    }
    __syncthreads();
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int k = 0; k < NPOS; ++k) {
        const int cell = g + 8 * k;
        if (cell >= CELLS) break;
        const size_t gi = ((size_t)b * CELLS + cell) * TC + ci;
        float d = dz[cell * 4] * wh[ci];
        d = fmaf(dz[cell * 4 + 1], wh[TC + ci], d);
        d = fmaf(dz[cell * 4 + 2], wh[2 * TC + ci], d);
        if (!(xtop[gi] > 0.0f)) d = 0.0f;
        du[gi] = d;
        const float xh = (zprev[gi] - stat_prev[ci]) * stat_prev[32 + ci];
        s1 += d; s2 = fmaf(d, xh, s2);
    }
    // the board's share of the head convs' weight gradients: thread (plane h, channel ci), bias gradients beside them
    if (tid < 3 * TC) {
        const int h = tid >> 5;
        float w = 0.0f, bsum = 0.0f;
        for (int cell = 0; cell < CELLS; ++cell) {
            w = fmaf(dz[cell * 4 + h], xtop[((size_t)b * CELLS + cell) * TC + ci], w);
            bsum += dz[cell * 4 + h];
        }
        pwh[(size_t)b * 128 + tid] = w;
        if (ci == 0) pwh[(size_t)b * 128 + 96 + h] = bsum;
    }
    board_partials(s1, s2, red, part_out + (size_t)b * 64);
}

// Backward of one 3x3 conv layer of the tower, one board per workgroup.
//   dyout, zout, ob   gradient at this layer's batch-norm output (ReLU mask applied), its conv output, its batch-norm;
//   part_in           the boards' sums of dyout, dyout * xhat from the launch before; bn_acc_out: that batch-norm's
//                     gradient accumulators
//   A (this conv's input): MODE_T: ReLU(BN_in(zin)) with BN_in's saved statistics, recomputed; else the stored xin
//   wb                this conv's weights as [tap][co][ci]
//   res               (RES) added to the input gradient before the mask: the gradient that reaches the block input
//                     through the residual connection
//   dyprev            out: (input gradient (+ res)) masked by A > 0 = gradient at the previous batch-norm's output
//   zprev, stat_prev  that previous batch-norm's input and statistics (MODE_T: zin itself), for its x-hat sums
//   pw                the board's share of this conv's weight gradient [tap][ci][co]; ACCUM: added to what an earlier
//                     application of the shared block left there
template <bool MODE_T, bool RES, bool ACCUM>
__global__ void __launch_bounds__(TTC)
k_train_conv_bwd(const float* __restrict__ dyout, const float* __restrict__ zout, BnBack ob, const float* __restrict__ part_in,
                 float* __restrict__ bn_acc_out, const float* __restrict__ zin, const float* __restrict__ stat_in,
                 const float* __restrict__ gamma_in, const float* __restrict__ beta_in, const float* __restrict__ xin,
                 const float* __restrict__ wb, const float* res /* may be dyprev */, float* dyprev,
                 const float* __restrict__ zprev, const float* __restrict__ stat_prev, float* __restrict__ pw,
                 float* __restrict__ part_out, int boards) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;                                   // [9][32 co][32 ci]
    float* dzt = lds + 9 * TC * TC;                    // [72][TS] dZ
    float* at = dzt + PADCELLS * TS;                   // [72][TS] A
    float* sc = at + PADCELLS * TS;                    // 7 x [32]
    float* sh = sc + TC;
    float* k1 = sh + TC;
    float* m1 = k1 + TC;
    float* m2 = m1 + TC;
    float* mean = m2 + TC;
    float* invstd = mean + TC;
    float* red = invstd + TC;                          // [1024]
    float* pbuf = red + 1024;                          // [4][32][PBS]
    __shared__ double tot[64], scratch[512];
    const int b = blockIdx.x, tid = threadIdx.x, lane32 = tid & 31, g = tid >> 5;
    const float inv_n = 1.0f / (float)((long)boards * CELLS);
    const size_t base = (size_t)b * CELLS * TC;
    // every global load of the launch, before the first wait (element tid + 512 j = (cell g + 16 j, channel lane32):
    // the cells this thread also finishes the input gradient of)
    float dyr[EPT], zor[EPT], ar[EPT], rr[EPT], zpr[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = tid + j * TTC;
        const bool in = i < CELLS * TC;
        dyr[j] = in ? dyout[base + i] : 0.0f;
        zor[j] = in ? zout[base + i] : 0.0f;
        ar[j] = in ? (MODE_T ? zin[base + i] : xin[base + i]) : 0.0f;
        rr[j] = (RES && in) ? res[base + i] : 0.0f;
        zpr[j] = in ? (MODE_T ? ar[j] : zprev[base + i]) : 0.0f;       // MODE_T: the previous batch-norm's input is zin
    }
    const WRegs wr = load_conv_weights(wb);
    const WgOld wold = wgrad_load_old<ACCUM>(pw + (size_t)b * 9 * TC * TC, tid >> 6, tid & 63);
    const float sp_mean = stat_prev[lane32], sp_inv = stat_prev[32 + lane32];
    for (int i = tid; i < 2 * PADCELLS * TS; i += TTC) dzt[i] = 0.0f;         // both tiles
    reduce_partials(part_in, boards, tot, scratch);
    store_conv_weights(wl, wr);
    bn_back_coeff(ob, tot, TC, inv_n, k1, m1, m2, mean, invstd);
    accumulate_bn_grad(tot, bn_acc_out);
    if (MODE_T) bn_scale_shift(stat_in, gamma_in, beta_in, sc, sh);
    __syncthreads();
    float a_own[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = tid + j * TTC;
        a_own[j] = 0.0f;
        if (i < CELLS * TC) {
            const int pi = pad_index(i >> 5);
            const float xh = (zor[j] - mean[lane32]) * invstd[lane32];
            dzt[pi * TS + lane32] = k1[lane32] * (dyr[j] - m1[lane32] - xh * m2[lane32]);
            a_own[j] = MODE_T ? fmaxf(fmaf(ar[j], sc[lane32], sh[lane32]), 0.0f) : ar[j];
            at[pi * TS + lane32] = a_own[j];
        }
    }
    __syncthreads();
    // ---- input gradient (into pbuf) and the board's share of the weight gradient, both on the matrix cores
    conv_mfma<-1>(dzt, wl, pbuf, tid >> 6, tid & 63);
    wgrad_mfma<ACCUM>(at, dzt, pw + (size_t)b * 9 * TC * TC, wold, tid >> 6, tid & 63);
    __syncthreads();
    // ---- mask, the previous batch-norm's sums: thread (ci, three cells)
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
        const int cell = g + 16 * k;
        if (cell < CELLS) {
            float d = conv_total(pbuf, lane32, cell);
            if (RES) d += rr[k];
            if (!(a_own[k] > 0.0f)) d = 0.0f;
            dyprev[base + cell * TC + lane32] = d;
            const float xh = (zpr[k] - sp_mean) * sp_inv;
            s1 += d; s2 = fmaf(d, xh, s2);
        }
    }
    board_partials(s1, s2, red, part_out + (size_t)b * 64);
}

// stem backward: dZ0 from the gradient at the stem batch-norm's output; the board's share of the stem conv's weight
// gradient -> pws [board][18][32] (index (tap * 2 + plane) * 32 + co)
__global__ void __launch_bounds__(TT)
k_train_stem_bwd(const float* __restrict__ x, const float* __restrict__ dy0, const float* __restrict__ z0, BnBack ob,
                 const float* __restrict__ part_in, float* __restrict__ bn_acc_out, float* __restrict__ pws, int boards) {
    __shared__ float xt[PADCELLS * 2], dzt[CELLS * TC], k1[TC], m1[TC], m2[TC], mean[TC], invstd[TC];
    __shared__ double tot[64], scratch[512];
    const int b = blockIdx.x, tid = threadIdx.x, co = tid & 31, g = tid >> 5;
    const float inv_n = 1.0f / (float)((long)boards * CELLS);
    reduce_partials(part_in, boards, tot, scratch);
    bn_back_coeff(ob, tot, TC, inv_n, k1, m1, m2, mean, invstd);
    accumulate_bn_grad(tot, bn_acc_out);
    for (int i = tid; i < PADCELLS * 2; i += TT) xt[i] = 0.0f;
    __syncthreads();
    for (int i = tid; i < 2 * CELLS; i += TT) xt[pad_index(i % CELLS) * 2 + i / CELLS] = x[(size_t)b * 84 + i];
    for (int i = tid; i < CELLS * TC; i += TT) {
        const int c = i & 31;
        const size_t gi = (size_t)b * CELLS * TC + i;
        const float xh = (z0[gi] - mean[c]) * invstd[c];
        dzt[i] = k1[c] * (dy0[gi] - m1[c] - xh * m2[c]);
    }
    __syncthreads();
    // thread (co, g): the (tap, plane) combinations q = g, g + 8, g + 16 (< 18)
    for (int q = g; q < 18; q += 8) {
        const int off = tap_offset(q >> 1), plane = q & 1;
        float acc = 0.0f;
        for (int cell = 0; cell < CELLS; ++cell) acc = fmaf(xt[(pad_index(cell) + off) * 2 + plane], dzt[cell * TC + co], acc);
        pws[(size_t)b * 18 * TC + q * TC + co] = acc;
    }
}

struct RunStat { float* rmean; float* rvar; long long* nbt; };

// Last launch: the boards' shares of the conv weight gradients summed in board order and written in torch's layout
// [co][ci][ky][kx]; the tower batch-norms' gradients from their accumulators; the running statistics of every
// batch-norm updated once per application, in order (torch: running = (1 - m) * running + m * batch, the variance
// unbiased), and the application counters.
__global__ void __launch_bounds__(TT)
k_train_finish(const float* __restrict__ pw1, const float* __restrict__ pw2, const float* __restrict__ pws,
               const float* __restrict__ pwh, int boards, int num_blocks, float* __restrict__ g_c1, float* __restrict__ g_c2,
               float* __restrict__ g_stem, float* __restrict__ g_wp, float* __restrict__ g_bp, float* __restrict__ g_wv,
               const float* __restrict__ bn_acc, float* __restrict__ g_bn0_w, float* __restrict__ g_bn0_b,
               float* __restrict__ g_bn1_w, float* __restrict__ g_bn1_b, float* __restrict__ g_bn2_w, float* __restrict__ g_bn2_b,
               const float* __restrict__ stats /* [2 * blocks + 1][96] */, const float* __restrict__ stat_h, RunStat r0, RunStat r1,
               RunStat r2, RunStat rp, RunStat rv, float momentum) {
    __shared__ float red[TT];
    int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const bool writer = threadIdx.x < 64;
    const long n = (long)boards * CELLS;
    if (i < 2 * 9 * TC * TC) {
        const bool second = i >= 9 * TC * TC;
        const int j = second ? i - 9 * TC * TC : i;                 // [tap][ci][co]
        const float* p = (second ? pw2 : pw1) + j;
        const float s = batch_sum(num_blocks > 0 ? boards : 0, red, [&](int b) { return p[(size_t)b * 9 * TC * TC]; });
        const int tap = j / (TC * TC), ci = (j / TC) % TC, co = j % TC;
        if (writer) (second ? g_c2 : g_c1)[(co * TC + ci) * 9 + tap] = s;
        return;
    }
    i -= 2 * 9 * TC * TC;
    if (i < 18 * TC) {
        const float s = batch_sum(boards, red, [&](int b) { return pws[(size_t)b * 18 * TC + i]; });
        const int tap = i / (2 * TC), plane = (i / TC) % 2, co = i % TC;
        if (writer) g_stem[(co * 2 + plane) * 9 + tap] = s;
        return;
    }
    i -= 18 * TC;
    if (i < 128) {
        const float s = batch_sum(boards, red, [&](int b) { return i < 99 ? pwh[(size_t)b * 128 + i] : 0.0f; });
        if (writer) {
            if (i < 2 * TC) g_wp[i] = s;
            else if (i < 3 * TC) g_wv[i - 2 * TC] = s;
            else if (i >= 96 && i < 98) g_bp[i - 96] = s;
        }
        return;
    }
    i -= 128;
    if (threadIdx.x < TC && i < TC) {
        const int c = i;
        g_bn0_b[c] = bn_acc[c];           g_bn0_w[c] = bn_acc[32 + c];
        g_bn1_b[c] = bn_acc[64 + c];      g_bn1_w[c] = bn_acc[64 + 32 + c];
        g_bn2_b[c] = bn_acc[128 + c];     g_bn2_w[c] = bn_acc[128 + 32 + c];
        const float unbias = (float)n / (float)(n - 1);
        r0.rmean[c] = (1.0f - momentum) * r0.rmean[c] + momentum * stats[c];
        r0.rvar[c] = (1.0f - momentum) * r0.rvar[c] + momentum * (stats[64 + c] * unbias);
        float m1 = r1.rmean[c], v1 = r1.rvar[c], m2 = r2.rmean[c], v2 = r2.rvar[c];
        for (int k = 0; k < num_blocks; ++k) {
            const float* s1 = stats + (size_t)(1 + 2 * k) * 96;
            const float* s2 = stats + (size_t)(2 + 2 * k) * 96;
            m1 = (1.0f - momentum) * m1 + momentum * s1[c];
            v1 = (1.0f - momentum) * v1 + momentum * (s1[64 + c] * unbias);
            m2 = (1.0f - momentum) * m2 + momentum * s2[c];
            v2 = (1.0f - momentum) * v2 + momentum * (s2[64 + c] * unbias);
        }
        r1.rmean[c] = m1; r1.rvar[c] = v1; r2.rmean[c] = m2; r2.rvar[c] = v2;
        if (c < 3) {
            RunStat r = c < 2 ? rp : rv;
            const int k = c < 2 ? c : 0;
            r.rmean[k] = (1.0f - momentum) * r.rmean[k] + momentum * stat_h[c];
            r.rvar[k] = (1.0f - momentum) * r.rvar[k] + momentum * (stat_h[64 + c] * unbias);
        }
        if (c == 0) {
            r0.nbt[0] += 1; r1.nbt[0] += num_blocks; r2.nbt[0] += num_blocks; rp.nbt[0] += 1; rv.nbt[0] += 1;
        }
    }
}
#define FINISH_GROUPS ((2 * 9 * TC * TC + 18 * TC + 128) / 64 + 1)

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
namespace {
struct Ws {                        // the workspace, in floats
    size_t act;                    // one activation tensor [boards][42][32]
    size_t z, x, g0, g1, zh, dyh, hplanes, h1, dh1, dpre2, dcl, logits7, vhat, dlogits7, dv, out3, part, stats, stat_h,
        part2, bn_acc, wf1, wb1, wf2, wb2, ws, pw1, pw2, pws, pwh, total;
};
Ws layout(int boards, int blocks) {
    Ws w;
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o += (n + 63) & ~(size_t)63; return at; };
    w.act = (size_t)boards * CELLS * TC;
    w.z = take(w.act * (2 * blocks + 1));
    w.x = take(w.act * (blocks + 1));
    w.g0 = take(w.act);
    w.g1 = take(w.act);
    w.zh = take((size_t)boards * CELLS * 4);
    w.dyh = take((size_t)boards * CELLS * 4);
    w.hplanes = take((size_t)boards * 128);
    w.h1 = take((size_t)boards * 256);
    w.dh1 = take((size_t)boards * 256);
    w.dpre2 = take(boards);
    w.dcl = take((size_t)boards * CELLS);
    w.logits7 = take((size_t)boards * 7);
    w.vhat = take(boards);
    w.dlogits7 = take((size_t)boards * 7);
    w.dv = take(boards);
    w.out3 = take(4);
    w.part = take((size_t)boards * 64);
    w.stats = take((size_t)(2 * blocks + 1) * 96);
    w.stat_h = take(96);
    w.part2 = take((size_t)boards * 64);
    w.bn_acc = take(3 * 64);
    w.wf1 = take(9 * TC * TC); w.wb1 = take(9 * TC * TC); w.wf2 = take(9 * TC * TC); w.wb2 = take(9 * TC * TC);
    w.ws = take(18 * TC);
    w.pw1 = take((size_t)boards * 9 * TC * TC);
    w.pw2 = take((size_t)boards * 9 * TC * TC);
    w.pws = take((size_t)boards * 18 * TC);
    w.pwh = take((size_t)boards * 128);
    w.total = o;
    return w;
}
constexpr int FWD_LDS = (9 * TC * TC + PADCELLS * TS + 2 * TC + 1024 + 4 * TC * PBS) * 4;
constexpr int BWD_LDS = (9 * TC * TC + 2 * PADCELLS * TS + 7 * TC + 1024 + 4 * TC * PBS) * 4;
std::atomic<unsigned long long> g_attr[8];
template <class K> int big_lds(K kernel, int bytes, int slot) {
    return az_set_kernel_lds(reinterpret_cast<const void*>(kernel), bytes, g_attr[slot]);
}
}  // namespace

extern "C" {

long az_c4_train_workspace_bytes(int32_t batch, int32_t num_blocks) {
    if (batch <= 0 || num_blocks < 0) return -1;
    return (long)(layout(batch, num_blocks).total * sizeof(float));
}

int az_c4_train_fwd_bwd(const az_c4_train_net* net, const float* x, const uint8_t* valid, const float* target_p,
                        const float* z, int32_t batch, void* workspace, float* out3, float* logits7_out, float* vhat_out,
                        void* stream) {
    if (!net || !x || !valid || !target_p || !z || !workspace || !out3 || batch <= 1)
        return az_fail("az_c4_train_fwd_bwd: null pointer or a batch of fewer than two boards");
    const int nb = net->num_blocks, B = batch;
    if (nb < 0) return az_fail("az_c4_train_fwd_bwd: num_blocks");
    hipStream_t st = (hipStream_t)stream;
    float* W = static_cast<float*>(workspace);
    const Ws w = layout(B, nb);
    const float eps = net->bn_eps, mom = net->bn_momentum;
    float* Z = W + w.z;
    float* X = W + w.x;
    float* stats = W + w.stats;
    // the boards' partial sums go from each launch to the next through two buffers used in turn
    float* parts[2] = {W + w.part, W + w.part2};
    int turn = 0;
    auto Pout = [&]() { return parts[turn]; };
    auto Pin = [&]() { return parts[turn ^ 1]; };
    auto next = [&]() { turn ^= 1; };
    auto Zl = [&](int l) { return Z + (size_t)l * w.act; };
    auto Xk = [&](int k) { return X + (size_t)k * w.act; };
    auto St = [&](int l) { return stats + (size_t)l * 96; };
    if (int rc = big_lds(k_train_conv_fwd<false, true>, FWD_LDS, 0)) return rc;
    if (int rc = big_lds(k_train_conv_fwd<false, false>, FWD_LDS, 1)) return rc;
    if (int rc = big_lds(k_train_conv_fwd<true, true>, FWD_LDS, 2)) return rc;
    if (int rc = big_lds(k_train_conv_bwd<true, false, false>, BWD_LDS, 3)) return rc;
    if (int rc = big_lds(k_train_conv_bwd<true, false, true>, BWD_LDS, 4)) return rc;
    if (int rc = big_lds(k_train_conv_bwd<false, true, false>, BWD_LDS, 5)) return rc;
    if (int rc = big_lds(k_train_conv_bwd<false, true, true>, BWD_LDS, 6)) return rc;

    const BnRef bn0{St(0), net->stem_bn.weight, net->stem_bn.bias};
    const float* nof = nullptr;
    // ---- forward
    hipLaunchKernelGGL(k_train_prep, dim3((9 * TC * TC + TT - 1) / TT), dim3(TT), 0, st, net->c1_w, net->c2_w, net->stem_w,
                       W + w.wf1, W + w.wb1, W + w.wf2, W + w.wb2, W + w.ws, W + w.bn_acc);
    hipLaunchKernelGGL(k_train_stem_fwd, dim3(B), dim3(TT), 0, st, x, W + w.ws, Zl(0), Pout());
    next();
    for (int k = 0; k < nb; ++k) {
        const BnRef bn1{St(1 + 2 * k), net->bn1.weight, net->bn1.bias};
        if (k == 0)
            hipLaunchKernelGGL((k_train_conv_fwd<false, true>), dim3(B), dim3(TTC), FWD_LDS, st, Zl(0), Pin(), bn0, nof, Xk(0),
                               W + w.wf1, Zl(1), Pout(), B, eps);
        else {
            const BnRef bn2p{St(2 * k), net->bn2.weight, net->bn2.bias};
            hipLaunchKernelGGL((k_train_conv_fwd<true, true>), dim3(B), dim3(TTC), FWD_LDS, st, Zl(2 * k), Pin(), bn2p,
                               (const float*)Xk(k - 1), Xk(k), W + w.wf1, Zl(1 + 2 * k), Pout(), B, eps);
        }
        next();
        hipLaunchKernelGGL((k_train_conv_fwd<false, false>), dim3(B), dim3(TTC), FWD_LDS, st, Zl(1 + 2 * k), Pin(), bn1, nof,
                           (float*)nullptr, W + w.wf2, Zl(2 + 2 * k), Pout(), B, eps);
        next();
    }
    if (nb == 0)
        hipLaunchKernelGGL((k_train_head_conv_fwd<false>), dim3(B), dim3(TT), 0, st, Zl(0), Pin(), bn0, nof, Xk(0),
                           net->pol_conv_w, net->pol_conv_b, net->val_conv_w, W + w.zh, Pout(), B, eps);
    else {
        const BnRef bn2l{St(2 * nb), net->bn2.weight, net->bn2.bias};
        hipLaunchKernelGGL((k_train_head_conv_fwd<true>), dim3(B), dim3(TT), 0, st, Zl(2 * nb), Pin(), bn2l,
                           (const float*)Xk(nb - 1), Xk(nb), net->pol_conv_w, net->pol_conv_b, net->val_conv_w, W + w.zh, Pout(),
                           B, eps);
    }
    next();
    hipLaunchKernelGGL(k_train_head_fc_fwd, dim3(B), dim3(TT), 0, st, x, W + w.zh, Pin(), W + w.stat_h, net->pol_bn.weight,
                       net->pol_bn.bias, net->val_bn.weight, net->val_bn.bias, net->pol_fc_w, net->pol_fc_b, net->val_fc1_w,
                       net->val_fc1_b, net->val_fc2_w, net->val_fc2_b, W + w.hplanes, W + w.h1, W + w.logits7, W + w.vhat, B,
                       eps);
    AZ_LAUNCH_CHECK("training forward");
    // ---- loss and its gradients (alphazero/loss.py:29-63)
    if (int rc = az_loss_fwd_bwd(7, W + w.logits7, W + w.vhat, target_p, z, valid, B, out3, W + w.dlogits7, W + w.dv, stream))
        return rc;
    // ---- backward
    hipLaunchKernelGGL(k_train_head_fc_bwd, dim3(B), dim3(TT), 0, st, x, W + w.dlogits7, W + w.dv, W + w.vhat, W + w.h1,
                       W + w.hplanes, W + w.zh, W + w.stat_h, net->pol_fc_w, net->val_fc1_w, net->val_fc2_w, W + w.dh1,
                       W + w.dpre2, W + w.dcl, W + w.dyh, Pout());
    next();
    {
        hipLaunchKernelGGL(k_train_fc_wgrad, dim3(FC_WGRAD_GROUPS), dim3(TT), 0, st, W + w.dh1, W + w.dpre2, W + w.dcl,
                           W + w.h1, W + w.hplanes, B, net->val_fc1_w_grad, net->val_fc1_b_grad, net->val_fc2_w_grad,
                           net->val_fc2_b_grad, net->pol_fc_w_grad, net->pol_fc_b_grad);
    }
    float* g0 = W + w.g0;              // gradient at a block's output batch-norm (dU) / at the stem's
    float* g1 = W + w.g1;              // gradient at a block's first batch-norm
    float* acc0 = W + w.bn_acc, * acc1 = acc0 + 64, * acc2 = acc0 + 128;
    hipLaunchKernelGGL(k_train_head_conv_bwd, dim3(B), dim3(TT), 0, st, W + w.dyh, W + w.zh, W + w.stat_h, Pin(),
                       net->pol_bn.weight, net->val_bn.weight, net->pol_bn.weight_grad, net->pol_bn.bias_grad,
                       net->val_bn.weight_grad, net->val_bn.bias_grad, net->pol_conv_w, net->val_conv_w, (const float*)Xk(nb),
                       (const float*)Zl(2 * nb), (const float*)St(2 * nb), g0, W + w.pwh, Pout(), B);
    next();
    for (int k = nb - 1; k >= 0; --k) {
        const bool first = k == nb - 1;                 // first application to be differentiated: its share is stored
        const BnBack ob2{St(2 + 2 * k), net->bn2.weight};
        // second conv of block k: A = ReLU(BN1(Z1)), previous batch-norm = BN1 over Z1
        if (first)
            hipLaunchKernelGGL((k_train_conv_bwd<true, false, false>), dim3(B), dim3(TTC), BWD_LDS, st, g0, Zl(2 + 2 * k), ob2, Pin(),
                               acc2, Zl(1 + 2 * k), St(1 + 2 * k), net->bn1.weight, net->bn1.bias, nof, W + w.wb2, nof, g1,
                               Zl(1 + 2 * k), St(1 + 2 * k), W + w.pw2, Pout(), B);
        else
            hipLaunchKernelGGL((k_train_conv_bwd<true, false, true>), dim3(B), dim3(TTC), BWD_LDS, st, g0, Zl(2 + 2 * k), ob2, Pin(),
                               acc2, Zl(1 + 2 * k), St(1 + 2 * k), net->bn1.weight, net->bn1.bias, nof, W + w.wb2, nof, g1,
                               Zl(1 + 2 * k), St(1 + 2 * k), W + w.pw2, Pout(), B);
        next();
        // first conv of block k: A = X_k (stored), residual gradient g0 added, previous batch-norm = BN2 of block
        // k - 1 over its Z2, or the stem's over Z0
        const BnBack ob1{St(1 + 2 * k), net->bn1.weight};
        if (first)
            hipLaunchKernelGGL((k_train_conv_bwd<false, true, false>), dim3(B), dim3(TTC), BWD_LDS, st, g1, Zl(1 + 2 * k), ob1, Pin(),
                               acc1, nof, nof, nof, nof, (const float*)Xk(k), W + w.wb1, (const float*)g0, g0, Zl(2 * k), St(2 * k),
                               W + w.pw1, Pout(), B);
        else
            hipLaunchKernelGGL((k_train_conv_bwd<false, true, true>), dim3(B), dim3(TTC), BWD_LDS, st, g1, Zl(1 + 2 * k), ob1, Pin(),
                               acc1, nof, nof, nof, nof, (const float*)Xk(k), W + w.wb1, (const float*)g0, g0, Zl(2 * k), St(2 * k),
                               W + w.pw1, Pout(), B);
        next();
    }
    {
        const BnBack ob0{St(0), net->stem_bn.weight};
        hipLaunchKernelGGL(k_train_stem_bwd, dim3(B), dim3(TT), 0, st, x, g0, Zl(0), ob0, Pin(), acc0, W + w.pws, B);
    }
    {
        const RunStat r0{net->stem_bn.running_mean, net->stem_bn.running_var, (long long*)net->stem_bn.num_batches_tracked};
        const RunStat r1{net->bn1.running_mean, net->bn1.running_var, (long long*)net->bn1.num_batches_tracked};
        const RunStat r2{net->bn2.running_mean, net->bn2.running_var, (long long*)net->bn2.num_batches_tracked};
        const RunStat rp{net->pol_bn.running_mean, net->pol_bn.running_var, (long long*)net->pol_bn.num_batches_tracked};
        const RunStat rv{net->val_bn.running_mean, net->val_bn.running_var, (long long*)net->val_bn.num_batches_tracked};
        hipLaunchKernelGGL(k_train_finish, dim3(FINISH_GROUPS), dim3(TT), 0, st, W + w.pw1, W + w.pw2, W + w.pws, W + w.pwh,
                           B, nb, net->c1_w_grad, net->c2_w_grad, net->stem_w_grad, net->pol_conv_w_grad, net->pol_conv_b_grad,
                           net->val_conv_w_grad, W + w.bn_acc, net->stem_bn.weight_grad, net->stem_bn.bias_grad,
                           net->bn1.weight_grad, net->bn1.bias_grad, net->bn2.weight_grad, net->bn2.bias_grad, stats,
                           W + w.stat_h, r0, r1, r2, rp, rv, mom);
    }
    AZ_LAUNCH_CHECK("training backward");
    if (logits7_out) AZ_HIPCHK(hipMemcpyAsync(logits7_out, W + w.logits7, (size_t)B * 7 * 4, hipMemcpyDeviceToDevice, st));
    if (vhat_out) AZ_HIPCHK(hipMemcpyAsync(vhat_out, W + w.vhat, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    return 0;
}

}  // extern "C"
