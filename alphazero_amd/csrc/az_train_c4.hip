// az_train_c4.hip -- forward and backward of the training step for the Connect Four network with 32 trunk
// channels (alphazero/training.py:57-71 with alphazero/models.py:6-87, alphazero/games/connectfour.py:47-72 and
// alphazero/loss.py:29-63), hand-written: no vendor convolution, no autograd.  Everything is fp32, like the
// reference's training.
//
// Batch-norm runs in training mode (statistics of the batch), so every conv is followed by a reduction over the
// whole batch before the next layer can start, and every layer's backward by another one: the pass is a sequence of
// 4 * blocks + 10 steps with a batch-wide exchange between them.  A workgroup owns one board (42 cells x 32 channels;
// the conv's halo is the zero padding, so workgroups share nothing else), ends a step by posting its board's partial
// sums and begins the next one by adding all boards' partial sums up for itself (fixed order, double precision:
// deterministic, no atomics).  The steps themselves are in az_train_c4_dev.h.  Two forms:
//   * one launch (az_c4_train_fwd_bwd, batches of up to one board per compute unit): all steps in one resident grid;
//     the partial sums travel as tagged pairs (ResidentExchange), 4.4 us per exchange instead of a launch boundary;
//     per-sample loss in the same launch; the batch sums of the weight gradients at the end, behind the pass's only
//     full barrier;
//   * a chain of 4 * blocks + 11 launches (az_c4_train_fwd_bwd_chain, any batch size) with the launch boundaries as
//     the synchronisation.
// Both run the same step functions with the same thread-to-data mapping and give the same bits.
//   * activations: channel-last fp32 [board][cell][32]; kept per layer: the raw conv outputs Z (batch-norm input)
//     and the block inputs X; everything else (ReLU(BN(Z)), x-hat, ReLU masks) is recomputed from those while a
//     tile is staged into LDS;
//   * forward conv: the board's 42 x 32 x 288 contraction on v_mfma_f32_32x32x2_f32, input tile and weights in LDS,
//     eight waves = two cell tiles x four tap groups;
//   * backward of a conv layer in ONE step: dZ from the incoming gradient by the batch-norm backward formula
//     (while staging), input gradient (the conv with flipped taps and transposed weights), ReLU mask, the partial
//     sums of the previous batch-norm's backward, and the board's share of the weight gradient (summed over the
//     boards by the last step, and over the applications of the shared block on the way: the tower applies ONE
//     block num_blocks times, alphazero/models.py:78);
//   * heads: 1x1 convs + batch-norm + fully connected layers per board, their weight gradients as sums over the
//     batch by one thread per weight.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/azhip.h"
#include "az_host.h"
#include "az_loss_dev.h"
#include "az_train_c4_dev.h"

namespace {
struct Ws {                        // the workspace, in floats
    size_t act;                    // one activation tensor [boards][42][32]
    size_t z, x, g0, g1, zh, dyh, hplanes, h1, dh1, dpre2, dcl, logits7, vhat, dlogits7, dv, out3, part, stats, stat_h,
        part2, bn_acc, wf1, wb1, wf2, wb2, ws, pw1, pw2, pws, pwh, rows, ctl, lossb, stamps, total;
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
    w.rows = take((size_t)2 * boards * 64 * 2);        // two sets of rows of 64 eight-byte pairs (one-launch form)
    w.ctl = take(64);                                  // int32: [0] launches so far, [1] error
    w.lossb = take((size_t)boards * 4);                // per sample: cross-entropy, squared error, |sum(target) - 1|
    w.stamps = take(512);                              // (-DPASS_TIMING builds: 256 timestamps of workgroup 0)
    w.total = o;
    return w;
}
}  // namespace

// weights of the 3x3 convs in the orders the steps read them: wf [tap][ci][co'] (forward), wb [tap][co][ci']
// (input gradient), the second index with its halves swapped in odd rows; stem: ws [tap * 2 + plane][co].  Also clears what the pass accumulates into, and the error word
// of the one-launch form.
__global__ void __launch_bounds__(TT)
k_train_prep(const float* __restrict__ c1, const float* __restrict__ c2, const float* __restrict__ stem, float* __restrict__ wf1,
             float* __restrict__ wb1, float* __restrict__ wf2, float* __restrict__ wb2, float* __restrict__ ws,
             float* __restrict__ bn_acc /* [3][64] */, int* __restrict__ ctl, unsigned long long* __restrict__ rows, int num_rows) {
    const int i = blockIdx.x * TT + threadIdx.x;
    // the one-launch form's rows start every pass without a valid tag, whatever the workspace held before
    for (int j = i; j < num_rows * 64; j += (int)gridDim.x * TT) rows[j] = 0ull;
    if (i < 9 * TC * TC) {
        // (within a row the two halves of 16 change places when the row's index is odd: the four rows that one matrix
        // instruction's operand is read from then lie in different LDS banks, conv_mfma)
        const int tap = i / (TC * TC), ci = (i / TC) % TC, co = i % TC;
        const float a = c1[(co * TC + ci) * 9 + tap], b = c2[(co * TC + ci) * 9 + tap];
        const int f = (tap * TC + ci) * TC + (co ^ ((ci & 1) * 16)), g = (tap * TC + co) * TC + (ci ^ ((co & 1) * 16));
        wf1[f] = a; wf2[f] = b;
        wb1[g] = a; wb2[g] = b;
    }
    if (i < 18 * TC) {
        const int tap = i / (2 * TC), plane = (i / TC) % 2, co = i % TC;
        ws[i] = stem[(co * 2 + plane) * 9 + tap];
    }
    if (i < 3 * 64) bn_acc[i] = 0.0f;
    if (i == 0) ctl[1] = 0;
}

// ---------------------------------------------------------------------------------------------------------
// the chain of launches: one step per kernel
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TT)
k_train_stem_fwd(const float* __restrict__ x, const float* __restrict__ ws, float* __restrict__ zout, float* __restrict__ part_out) {
    __shared__ float xt[PADCELLS * 2];
    __shared__ float wl[18 * TC];
    __shared__ float red[512];
    stem_fwd_step(x, ws, zout, ChainExchange{nullptr, part_out, 0}, xt, wl, red);
}

template <bool RES, bool WRITE_X>
__global__ void __launch_bounds__(TTC)
k_train_conv_fwd(const float* __restrict__ zin, const float* __restrict__ part_in, float* stat_keep, const float* __restrict__ gamma,
                 const float* __restrict__ beta, const float* __restrict__ xprev, float* __restrict__ xout,
                 const float* __restrict__ wf, float* __restrict__ zout, float* __restrict__ part_out, int boards, float eps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double tot[64], scratch[512];
    __shared__ float st[96];
    conv_fwd_step<RES, WRITE_X>(zin, BnRef{st, stat_keep, gamma, beta}, xprev, xout, wf, zout, boards, eps,
                                ChainExchange{part_in, part_out, boards}, lds, tot, scratch);
}

template <bool RES>
__global__ void __launch_bounds__(TT)
k_train_head_conv_fwd(const float* __restrict__ zin, const float* __restrict__ part_in, float* stat_keep, const float* __restrict__ gamma,
                      const float* __restrict__ beta, const float* __restrict__ xprev, float* __restrict__ xout,
                      const float* __restrict__ wp, const float* __restrict__ bp, const float* __restrict__ wv,
                      float* __restrict__ zh, float* __restrict__ part_out, int boards, float eps) {
    __shared__ __attribute__((aligned(16))) float tile[PADCELLS * TC];
    __shared__ float sc[TC], sh[TC], wh[3 * TC], red[512], st[96];
    __shared__ double tot[64], scratch[512];
    head_conv_fwd_step<RES>(zin, BnRef{st, stat_keep, gamma, beta}, xprev, xout, wp, bp, wv, zh, boards, eps,
                            ChainExchange{part_in, part_out, boards}, tile, sc, sh, wh, red, tot, scratch);
}

__global__ void __launch_bounds__(TT)
k_train_head_fc_fwd(const float* __restrict__ x, const float* __restrict__ zh, const float* __restrict__ part_in,
                    float* __restrict__ stat_h_keep, const float* __restrict__ gp, const float* __restrict__ btp,
                    const float* __restrict__ gv, const float* __restrict__ btv, const float* __restrict__ wpf,
                    const float* __restrict__ bpf, const float* __restrict__ w1, const float* __restrict__ b1,
                    const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ hplanes, float* __restrict__ h1,
                    float* __restrict__ logits7, float* __restrict__ vhat, int boards, float eps) {
    __shared__ float hp[128], cl[CELLS], red[TT], stat_h[96], w1_l[256 * CELLS], wpf_l[CELLS * 84];
    __shared__ int height[8];
    __shared__ double tot[64], scratch[512];
    head_fc_fwd_step(x, zh, stat_h, stat_h_keep, gp, btp, gv, btv, wpf, bpf, w1, b1, w2, b2, hplanes, h1, logits7, vhat, boards, eps,
                     ChainExchange{part_in, nullptr, boards}, hp, cl, height, red, w1_l, wpf_l, false, tot, scratch);
}

__global__ void __launch_bounds__(TT)
k_train_head_fc_bwd(const float* __restrict__ x, const float* __restrict__ dlogits7, const float* __restrict__ dv,
                    const float* __restrict__ vhat, const float* __restrict__ h1, const float* __restrict__ hplanes,
                    const float* __restrict__ zh, const float* __restrict__ stat_h, const float* __restrict__ wpf,
                    const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ dh1, float* __restrict__ dpre2,
                    float* __restrict__ dcl, float* __restrict__ dyh, float* __restrict__ part_out) {
    __shared__ float d1[TT], dcl_s[CELLS], red[512 + 256], w1_l[256 * CELLS];
    __shared__ int height[8];
    head_fc_bwd_step(x, dlogits7, dv, vhat, h1, hplanes, zh, stat_h, wpf, w1, w2, dh1, dpre2, dcl, dyh,
                     ChainExchange{nullptr, part_out, 0}, d1, dcl_s, height, red, w1_l, false);
}

__global__ void __launch_bounds__(TT)
k_train_fc_wgrad(const float* __restrict__ dh1, const float* __restrict__ dpre2, const float* __restrict__ dcl,
                 const float* __restrict__ h1, const float* __restrict__ hplanes, int boards, float* __restrict__ g_w1,
                 float* __restrict__ g_b1, float* __restrict__ g_w2, float* __restrict__ g_b2, float* __restrict__ g_wpf,
                 float* __restrict__ g_bpf) {
    __shared__ float red[TT];
    fc_wgrad_group(blockIdx.x, threadIdx.x, red, dh1, dpre2, dcl, h1, hplanes, boards, g_w1, g_b1, g_w2, g_b2, g_wpf, g_bpf);
}

__global__ void __launch_bounds__(TT)
k_train_head_conv_bwd(const float* __restrict__ dyh, const float* __restrict__ zh, const float* __restrict__ stat_h,
                      const float* __restrict__ part_in, const float* __restrict__ gp, const float* __restrict__ gv,
                      float* __restrict__ g_gp, float* __restrict__ g_btp, float* __restrict__ g_gv, float* __restrict__ g_btv,
                      const float* __restrict__ wp, const float* __restrict__ wv, const float* __restrict__ xtop,
                      const float* __restrict__ zprev, const float* __restrict__ stat_prev, float* __restrict__ du,
                      float* __restrict__ pwh, float* __restrict__ part_out, int boards) {
    __shared__ float dz[CELLS * 4], wh[3 * TC], red[512];
    __shared__ double tot[64], scratch[512];
    head_conv_bwd_step(dyh, zh, stat_h, gp, gv, g_gp, g_btp, g_gv, g_btv, wp, wv, xtop, zprev, stat_prev, du, pwh, boards,
                       ChainExchange{part_in, part_out, boards}, dz, wh, red, tot, scratch);
}

template <bool MODE_T, bool RES, bool ACCUM>
__global__ void __launch_bounds__(TTC)
k_train_conv_bwd(const float* __restrict__ dyout, const float* __restrict__ zout, BnBack ob, const float* __restrict__ part_in,
                 float* __restrict__ bn_acc_out, const float* __restrict__ zin, const float* __restrict__ stat_in,
                 const float* __restrict__ gamma_in, const float* __restrict__ beta_in, const float* __restrict__ xin,
                 const float* __restrict__ wb, const float* res /* may be dyprev */, float* dyprev,
                 const float* __restrict__ zprev, const float* __restrict__ stat_prev, float* __restrict__ pw,
                 float* __restrict__ part_out, int boards) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double tot[64], scratch[512];
    conv_bwd_step<MODE_T, RES>(ACCUM, dyout, zout, ob, bn_acc_out, zin, stat_in, gamma_in, beta_in, xin, wb, res, dyprev, zprev,
                               stat_prev, pw, boards, ChainExchange{part_in, part_out, boards}, lds, tot, scratch);
}

__global__ void __launch_bounds__(TT)
k_train_stem_bwd(const float* __restrict__ x, const float* __restrict__ dy0, const float* __restrict__ z0, BnBack ob,
                 const float* __restrict__ part_in, float* __restrict__ bn_acc_out, float* __restrict__ pws, int boards) {
    __shared__ float xt[PADCELLS * 2], dzt[CELLS * TC], coef[5 * TC];
    __shared__ double tot[64], scratch[512];
    stem_bwd_step(x, dy0, z0, ob, bn_acc_out, pws, boards, ChainExchange{part_in, nullptr, boards}, xt, dzt, coef, tot, scratch);
}

__global__ void __launch_bounds__(TT) k_train_finish(const FinishArgs f) {
    __shared__ float red[TT];
    finish_group(blockIdx.x, threadIdx.x, red, f);
}

// ---------------------------------------------------------------------------------------------------------
// the whole pass in one launch
// ---------------------------------------------------------------------------------------------------------
struct PassArgs {
    az_c4_train_net net;
    const float* x; const uint8_t* valid; const float* target_p; const float* z;
    float* W;
    Ws w;
    int boards;
    float* out3; float* logits7_out; float* vhat_out;
};
#define PASS_SMALL (CONV_LDS_FLOATS)               // [1024]: the heads' vectors
#define PASS_STATS (PASS_SMALL + 1024)             // [2 blocks + 2][96]: every batch-norm's statistics, the heads' last
static inline int pass_lds_bytes(int blocks) { return (PASS_STATS + (2 * blocks + 2) * 96) * 4; }

// the steps in order; true = an exchange gave up
__device__ __forceinline__ bool pass_steps(const PassArgs& A, float* lds, double* tot, double* scratch) {
    const az_c4_train_net& net = A.net;
    const Ws& w = A.w;
    float* W = A.W;
    const int b = blockIdx.x, tid = threadIdx.x, B = A.boards, nb = net.num_blocks;
    const float eps = net.bn_eps;
    float* small = lds + PASS_SMALL;
    float* st_all = lds + PASS_STATS;
    float* coef = lds + CONV_LDS_COEF;
    float* red = lds + CONV_LDS_RED;
    int* ctl = reinterpret_cast<int*>(W + w.ctl);
    unsigned long long* rows = reinterpret_cast<unsigned long long*>(W + w.rows);
    const uint32_t tag0 = ((uint32_t)(ctl[0] & 0xFFFFF) + 1u) << 8;          // never 0: a cleared row matches no step
    // step p adds up the rows tagged p (set p & 1) and posts rows tagged p + 1 (the other set)
#ifdef PASS_TIMING
    // workgroup 0 notes the time at which each step's sums were complete (2 p) and at which the step ended (2 p + 1)
    long long* stamps = reinterpret_cast<long long*>(W + w.stamps);
#else
    long long* stamps = nullptr;
#endif
    auto EX = [&](int p) {
        return ResidentExchange{rows + (size_t)(p & 1) * B * 64, rows + (size_t)((p + 1) & 1) * B * 64, tag0 + (uint32_t)p,
                                tag0 + (uint32_t)p + 1u, B, ctl + 1, stamps ? stamps + 2 * p : nullptr,
                                stamps ? stamps + 128 : nullptr};
    };
    auto stamp_end = [&](int p) {
        if (stamps && b == 0 && tid == 0) stamps[2 * p + 1] = (long long)__builtin_readcyclecounter();
    };
    auto Zl = [&](int l) { return W + w.z + (size_t)l * w.act; };
    auto Xk = [&](int k) { return W + w.x + (size_t)k * w.act; };
    auto St = [&](int l) { return W + w.stats + (size_t)l * 96; };
    auto Sl = [&](int l) { return st_all + l * 96; };
    float* stat_h = st_all + (2 * nb + 1) * 96;
    const float* nof = nullptr;
    int p = 0;
    // ---- forward
    stem_fwd_step(A.x, W + w.ws, Zl(0), EX(p), lds + CONV_LDS_DZT, lds + CONV_LDS_AT, red);
    stamp_end(p);
    ++p;
    __syncthreads();
    const BnRef bn0{Sl(0), St(0), net.stem_bn.weight, net.stem_bn.bias};
    for (int k = 0; k < nb; ++k) {
        const BnRef bn1{Sl(1 + 2 * k), St(1 + 2 * k), net.bn1.weight, net.bn1.bias};
        bool bad;
        if (k == 0)
            bad = conv_fwd_step<false, true>(Zl(0), bn0, nof, Xk(0), W + w.wf1, Zl(1), B, eps, EX(p), lds, tot, scratch);
        else
            bad = conv_fwd_step<true, true>(Zl(2 * k), BnRef{Sl(2 * k), St(2 * k), net.bn2.weight, net.bn2.bias}, Xk(k - 1), Xk(k),
                                            W + w.wf1, Zl(1 + 2 * k), B, eps, EX(p), lds, tot, scratch);
        if (bad) return true;
        stamp_end(p);
        ++p;
        __syncthreads();
        if (conv_fwd_step<false, false>(Zl(1 + 2 * k), bn1, nof, (float*)nullptr, W + w.wf2, Zl(2 + 2 * k), B, eps, EX(p), lds, tot, scratch))
            return true;
        stamp_end(p);
        ++p;
        __syncthreads();
    }
    int* height = reinterpret_cast<int*>(small + 320);
    float* w1_l = lds + CONV_LDS_WL;                   // [256][42]: the conv weights' and the dZ tile's place
    float* wpf_l = lds + CONV_LDS_PBUF;                // [42][84]
    {
        // the fully connected layers' weights go to LDS here, one exchange ahead of their use: the 1x1 convs need
        // none of that space
        stage_floats(w1_l, net.val_fc1_w, 256 * CELLS, tid);
        stage_floats(wpf_l, net.pol_fc_w, CELLS * 84, tid);
        bool bad;
        if (nb == 0)
            bad = head_conv_fwd_step<false>(Zl(0), bn0, nof, Xk(0), net.pol_conv_w, net.pol_conv_b, net.val_conv_w, W + w.zh, B, eps,
                                            EX(p), lds + CONV_LDS_AT, coef, coef + TC, small, red, tot, scratch);
        else
            bad = head_conv_fwd_step<true>(Zl(2 * nb), BnRef{Sl(2 * nb), St(2 * nb), net.bn2.weight, net.bn2.bias}, Xk(nb - 1), Xk(nb),
                                           net.pol_conv_w, net.pol_conv_b, net.val_conv_w, W + w.zh, B, eps, EX(p),
                                           lds + CONV_LDS_AT, coef, coef + TC, small, red, tot, scratch);
        if (bad) return true;
        stamp_end(p);
        ++p;
        __syncthreads();
    }
    if (head_fc_fwd_step(A.x, W + w.zh, stat_h, W + w.stat_h, net.pol_bn.weight, net.pol_bn.bias, net.val_bn.weight, net.val_bn.bias,
                         net.pol_fc_w, net.pol_fc_b, net.val_fc1_w, net.val_fc1_b, net.val_fc2_w, net.val_fc2_b, W + w.hplanes,
                         W + w.h1, W + w.logits7, W + w.vhat, B, eps, EX(p), small + 128, small + 256, height, red, w1_l, wpf_l, true, tot,
                         scratch))
        return true;
    __syncthreads();
    // ---- this sample's loss and its gradients (alphazero/loss.py:29-63)
    if (tid == 0) {
        float ce, sq, dev;
        loss_sample<7>(W + w.logits7, W + w.vhat, A.target_p, A.z, A.valid, b, 1.0f / (float)B, W + w.dlogits7, W + w.dv, ce, sq, dev);
        float* lb = W + w.lossb + (size_t)b * 4;
        lb[0] = ce; lb[1] = sq; lb[2] = dev;
        if (A.vhat_out) A.vhat_out[b] = W[w.vhat + b];
    }
    if (tid < 7 && A.logits7_out) A.logits7_out[(size_t)b * 7 + tid] = W[w.logits7 + (size_t)b * 7 + tid];
    __syncthreads();
    // ---- backward
    head_fc_bwd_step(A.x, W + w.dlogits7, W + w.dv, W + w.vhat, W + w.h1, W + w.hplanes, W + w.zh, stat_h, wpf_l, net.val_fc1_w,
                     net.val_fc2_w, W + w.dh1, W + w.dpre2, W + w.dcl, W + w.dyh, EX(p), small + 640, small + 384, height, red, w1_l, true);
    stamp_end(p);
    ++p;
    __syncthreads();
    float* g0 = W + w.g0;              // gradient at a block's output batch-norm (dU) / at the stem's
    float* g1 = W + w.g1;              // gradient at a block's first batch-norm
    float* acc0 = W + w.bn_acc, * acc1 = acc0 + 64, * acc2 = acc0 + 128;
    if (head_conv_bwd_step(W + w.dyh, W + w.zh, stat_h, net.pol_bn.weight, net.val_bn.weight, net.pol_bn.weight_grad,
                           net.pol_bn.bias_grad, net.val_bn.weight_grad, net.val_bn.bias_grad, net.pol_conv_w, net.val_conv_w, Xk(nb),
                           Zl(2 * nb), Sl(2 * nb), g0, W + w.pwh, B, EX(p), small + 448, small, red, tot, scratch))
        return true;
    stamp_end(p);
    ++p;
    __syncthreads();
    for (int k = nb - 1; k >= 0; --k) {
        const bool accum = k != nb - 1;                 // the first application to be differentiated stores its share
        // second conv of block k: A = ReLU(BN1(Z1)), previous batch-norm = BN1 over Z1
        if (conv_bwd_step<true, false>(accum, g0, Zl(2 + 2 * k), BnBack{Sl(2 + 2 * k), net.bn2.weight}, acc2, Zl(1 + 2 * k), Sl(1 + 2 * k),
                                       net.bn1.weight, net.bn1.bias, nof, W + w.wb2, nof, g1, Zl(1 + 2 * k), Sl(1 + 2 * k), W + w.pw2, B,
                                       EX(p), lds, tot, scratch))
            return true;
        stamp_end(p);
        ++p;
        __syncthreads();
        // first conv of block k: A = X_k (stored), residual gradient g0 added, previous batch-norm = BN2 of block
        // k - 1 over its Z2, or the stem's over Z0
        if (conv_bwd_step<false, true>(accum, g1, Zl(1 + 2 * k), BnBack{Sl(1 + 2 * k), net.bn1.weight}, acc1, nof, nof, nof, nof, Xk(k),
                                       W + w.wb1, g0, g0, Zl(2 * k), Sl(2 * k), W + w.pw1, B, EX(p), lds, tot, scratch))
            return true;
        stamp_end(p);
        ++p;
        __syncthreads();
    }
    if (stem_bwd_step(A.x, g0, Zl(0), BnBack{Sl(0), net.stem_bn.weight}, acc0, W + w.pws, B, EX(p), lds + CONV_LDS_AT,
                      lds + CONV_LDS_DZT, coef, tot, scratch))
        return true;
    // ---- the only full barrier: every board's shares of the weight gradients must be visible to all workgroups
    __syncthreads();
    if (tid < 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        EX(p).post(0.0f);
    }
    stamp_end(p);
    ++p;
    if (EX(p).template gather<TTC / 64>(tot, scratch)) return true;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    return false;
}

__global__ void __launch_bounds__(TTC) k_train_pass(const PassArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double tot[64], scratch[512];
    const az_c4_train_net& net = A.net;
    const Ws& w = A.w;
    float* W = A.W;
    const int b = blockIdx.x, tid = threadIdx.x, B = A.boards, nb = net.num_blocks;
    int* ctl = reinterpret_cast<int*>(W + w.ctl);
    const int launches = ctl[0];
#ifdef PASS_TIMING
    if (b == 0 && tid == 0) reinterpret_cast<long long*>(W + w.stamps)[254] = (long long)__builtin_readcyclecounter();
#endif
    if (pass_steps(A, lds, tot, scratch)) {
        // a workgroup did not arrive: nothing of this pass may be used
        if (b == 0 && tid == 0) {
            A.out3[0] = NAN; A.out3[1] = NAN; A.out3[2] = INFINITY;
            ctl[0] = launches + 1;
        }
        return;
    }
    // ---- sums over the batch, 64 outputs per half workgroup: the fully connected layers' weight gradients, the conv
    // weight gradients in torch's layout, batch-norm gradients and running statistics
    FinishArgs f;
    f.pw1 = W + w.pw1; f.pw2 = W + w.pw2; f.pws = W + w.pws; f.pwh = W + w.pwh;
    f.boards = B; f.num_blocks = nb;
    f.g_c1 = net.c1_w_grad; f.g_c2 = net.c2_w_grad; f.g_stem = net.stem_w_grad; f.g_wp = net.pol_conv_w_grad;
    f.g_bp = net.pol_conv_b_grad; f.g_wv = net.val_conv_w_grad;
    f.bn_acc = W + w.bn_acc;
    f.g_bn0_w = net.stem_bn.weight_grad; f.g_bn0_b = net.stem_bn.bias_grad;
    f.g_bn1_w = net.bn1.weight_grad; f.g_bn1_b = net.bn1.bias_grad; f.g_bn2_w = net.bn2.weight_grad; f.g_bn2_b = net.bn2.bias_grad;
    f.stats = W + w.stats; f.stat_h = W + w.stat_h;
    f.r0 = RunStat{net.stem_bn.running_mean, net.stem_bn.running_var, (long long*)net.stem_bn.num_batches_tracked};
    f.r1 = RunStat{net.bn1.running_mean, net.bn1.running_var, (long long*)net.bn1.num_batches_tracked};
    f.r2 = RunStat{net.bn2.running_mean, net.bn2.running_var, (long long*)net.bn2.num_batches_tracked};
    f.rp = RunStat{net.pol_bn.running_mean, net.pol_bn.running_var, (long long*)net.pol_bn.num_batches_tracked};
    f.rv = RunStat{net.val_bn.running_mean, net.val_bn.running_var, (long long*)net.val_bn.num_batches_tracked};
    f.momentum = net.bn_momentum;
    const int half = tid >> 8, t = tid & 255;
    float* red = lds + CONV_LDS_RED + half * 256;
#ifdef PASS_TIMING
    long long* marks = reinterpret_cast<long long*>(W + w.stamps) + 140;
    int mark = 0;
    if (b == 0 && tid == 0) marks[mark] = (long long)__builtin_readcyclecounter();
    ++mark;
#endif
    for (int g0 = 0; g0 < FC_WGRAD_GROUPS + FINISH_GROUPS; g0 += 2 * (int)gridDim.x) {
        const int group = g0 + 2 * b + half;
        if (group < FC_WGRAD_GROUPS)
            fc_wgrad_group(group, t, red, W + w.dh1, W + w.dpre2, W + w.dcl, W + w.h1, W + w.hplanes, B, net.val_fc1_w_grad,
                           net.val_fc1_b_grad, net.val_fc2_w_grad, net.val_fc2_b_grad, net.pol_fc_w_grad, net.pol_fc_b_grad);
        else if (group < FC_WGRAD_GROUPS + FINISH_GROUPS)
            finish_group(group - FC_WGRAD_GROUPS, t, red, f);
        else
            __syncthreads();                           // (every group does exactly one)
        __syncthreads();                               // red is used again
#ifdef PASS_TIMING
        if (b == 0 && tid == 0) marks[mark] = (long long)__builtin_readcyclecounter();
        ++mark;
#endif
    }
    // ---- the losses: batch means of the per-sample values, added as the stand-alone loss kernel adds them
    if (b == (int)gridDim.x - 1) {
        float pl = 0.0f, vl = 0.0f, dev = 0.0f;
        if (tid < 256)
            for (int s = tid; s < B; s += 256) {
                const float* lb = W + w.lossb + (size_t)s * 4;
                pl += lb[0]; vl += lb[1];
                dev = fmaxf(dev, lb[2]);
            }
        loss_totals(pl, vl, dev, reinterpret_cast<float(*)[256]>(lds + CONV_LDS_RED));
        if (tid == 0) {
            const float invB = 1.0f / (float)B;
            const float(*r)[256] = reinterpret_cast<const float(*)[256]>(lds + CONV_LDS_RED);
            A.out3[0] = r[0][0] * invB; A.out3[1] = r[1][0] * invB; A.out3[2] = r[2][0];
        }
    }
    if (b == 0 && tid == 0) ctl[0] = launches + 1;
#ifdef PASS_TIMING
    if (b == 0 && tid == 0) reinterpret_cast<long long*>(W + w.stamps)[255] = (long long)__builtin_readcyclecounter();
#endif
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int CONV_LDS = CONV_LDS_FLOATS * 4;
std::atomic<unsigned long long> g_attr[8];
template <class K> int big_lds(K kernel, int bytes, int slot) {
    return az_set_kernel_lds(reinterpret_cast<const void*>(kernel), bytes, g_attr[slot]);
}
int check_args(const az_c4_train_net* net, const float* x, const uint8_t* valid, const float* target_p, const float* z,
               int32_t batch, void* workspace, float* out3, const char* who) {
    if (!net || !x || !valid || !target_p || !z || !workspace || !out3 || batch <= 1 || net->num_blocks < 0) {
        char msg[160];
        snprintf(msg, sizeof msg, "%s: null pointer, num_blocks < 0 or a batch of fewer than two boards", who);
        return az_fail(msg);
    }
    return 0;
}
int launch_prep(const az_c4_train_net* net, float* W, const Ws& w, int boards, hipStream_t st) {
    hipLaunchKernelGGL(k_train_prep, dim3((9 * TC * TC + TT - 1) / TT), dim3(TT), 0, st, net->c1_w, net->c2_w, net->stem_w,
                       W + w.wf1, W + w.wb1, W + w.wf2, W + w.wb2, W + w.ws, W + w.bn_acc, reinterpret_cast<int*>(W + w.ctl),
                       reinterpret_cast<unsigned long long*>(W + w.rows), 2 * boards);
    AZ_LAUNCH_CHECK("k_train_prep");
    return 0;
}
// boards that can be resident at once: one 512-thread workgroup per compute unit
int resident_boards(int* out) {
    int dev = 0, cus = 0;
    AZ_HIPCHK(hipGetDevice(&dev));
    AZ_HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    *out = cus;
    return 0;
}
}  // namespace

extern "C" {

long az_c4_train_workspace_bytes(int32_t batch, int32_t num_blocks) {
    if (batch <= 0 || num_blocks < 0) return -1;
    return (long)(layout(batch, num_blocks).total * sizeof(float));
}

int az_c4_train_one_launch_max_batch(int32_t num_blocks, int32_t* max_batch) {
    if (!max_batch || num_blocks < 0) return az_fail("az_c4_train_one_launch_max_batch: null pointer or num_blocks < 0");
    *max_batch = 0;
    if (4 * num_blocks + 8 >= 256 || pass_lds_bytes(num_blocks) > 160 * 1024) return 0;      // step tags have eight bits
    int cus = 0;
    if (int rc = resident_boards(&cus)) return rc;
    *max_batch = cus;
    return 0;
}

int az_c4_train_fwd_bwd_chain(const az_c4_train_net* net, const float* x, const uint8_t* valid, const float* target_p,
                              const float* z, int32_t batch, void* workspace, float* out3, float* logits7_out, float* vhat_out,
                              void* stream) {
    if (int rc = check_args(net, x, valid, target_p, z, batch, workspace, out3, "az_c4_train_fwd_bwd_chain")) return rc;
    const int nb = net->num_blocks, B = batch;
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
    if (int rc = big_lds(k_train_conv_fwd<false, true>, CONV_LDS, 0)) return rc;
    if (int rc = big_lds(k_train_conv_fwd<false, false>, CONV_LDS, 1)) return rc;
    if (int rc = big_lds(k_train_conv_fwd<true, true>, CONV_LDS, 2)) return rc;
    if (int rc = big_lds(k_train_conv_bwd<true, false, false>, CONV_LDS, 3)) return rc;
    if (int rc = big_lds(k_train_conv_bwd<true, false, true>, CONV_LDS, 4)) return rc;
    if (int rc = big_lds(k_train_conv_bwd<false, true, false>, CONV_LDS, 5)) return rc;
    if (int rc = big_lds(k_train_conv_bwd<false, true, true>, CONV_LDS, 6)) return rc;

    const float* nof = nullptr;
    // ---- forward
    if (int rc = launch_prep(net, W, w, B, st)) return rc;
    hipLaunchKernelGGL(k_train_stem_fwd, dim3(B), dim3(TT), 0, st, x, W + w.ws, Zl(0), Pout());
    next();
    for (int k = 0; k < nb; ++k) {
        if (k == 0)
            hipLaunchKernelGGL((k_train_conv_fwd<false, true>), dim3(B), dim3(TTC), CONV_LDS, st, Zl(0), Pin(), St(0),
                               net->stem_bn.weight, net->stem_bn.bias, nof, Xk(0), W + w.wf1, Zl(1), Pout(), B, eps);
        else
            hipLaunchKernelGGL((k_train_conv_fwd<true, true>), dim3(B), dim3(TTC), CONV_LDS, st, Zl(2 * k), Pin(), St(2 * k),
                               net->bn2.weight, net->bn2.bias, (const float*)Xk(k - 1), Xk(k), W + w.wf1, Zl(1 + 2 * k), Pout(), B, eps);
        next();
        hipLaunchKernelGGL((k_train_conv_fwd<false, false>), dim3(B), dim3(TTC), CONV_LDS, st, Zl(1 + 2 * k), Pin(), St(1 + 2 * k),
                           net->bn1.weight, net->bn1.bias, nof, (float*)nullptr, W + w.wf2, Zl(2 + 2 * k), Pout(), B, eps);
        next();
    }
    if (nb == 0)
        hipLaunchKernelGGL((k_train_head_conv_fwd<false>), dim3(B), dim3(TT), 0, st, Zl(0), Pin(), St(0), net->stem_bn.weight,
                           net->stem_bn.bias, nof, Xk(0), net->pol_conv_w, net->pol_conv_b, net->val_conv_w, W + w.zh, Pout(), B, eps);
    else
        hipLaunchKernelGGL((k_train_head_conv_fwd<true>), dim3(B), dim3(TT), 0, st, Zl(2 * nb), Pin(), St(2 * nb), net->bn2.weight,
                           net->bn2.bias, (const float*)Xk(nb - 1), Xk(nb), net->pol_conv_w, net->pol_conv_b, net->val_conv_w,
                           W + w.zh, Pout(), B, eps);
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
    hipLaunchKernelGGL(k_train_fc_wgrad, dim3(FC_WGRAD_GROUPS), dim3(TT), 0, st, W + w.dh1, W + w.dpre2, W + w.dcl, W + w.h1,
                       W + w.hplanes, B, net->val_fc1_w_grad, net->val_fc1_b_grad, net->val_fc2_w_grad, net->val_fc2_b_grad,
                       net->pol_fc_w_grad, net->pol_fc_b_grad);
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
            hipLaunchKernelGGL((k_train_conv_bwd<true, false, false>), dim3(B), dim3(TTC), CONV_LDS, st, g0, Zl(2 + 2 * k), ob2, Pin(),
                               acc2, Zl(1 + 2 * k), St(1 + 2 * k), net->bn1.weight, net->bn1.bias, nof, W + w.wb2, nof, g1,
                               Zl(1 + 2 * k), St(1 + 2 * k), W + w.pw2, Pout(), B);
        else
            hipLaunchKernelGGL((k_train_conv_bwd<true, false, true>), dim3(B), dim3(TTC), CONV_LDS, st, g0, Zl(2 + 2 * k), ob2, Pin(),
                               acc2, Zl(1 + 2 * k), St(1 + 2 * k), net->bn1.weight, net->bn1.bias, nof, W + w.wb2, nof, g1,
                               Zl(1 + 2 * k), St(1 + 2 * k), W + w.pw2, Pout(), B);
        next();
        // first conv of block k: A = X_k (stored), residual gradient g0 added, previous batch-norm = BN2 of block
        // k - 1 over its Z2, or the stem's over Z0
        const BnBack ob1{St(1 + 2 * k), net->bn1.weight};
        if (first)
            hipLaunchKernelGGL((k_train_conv_bwd<false, true, false>), dim3(B), dim3(TTC), CONV_LDS, st, g1, Zl(1 + 2 * k), ob1, Pin(),
                               acc1, nof, nof, nof, nof, (const float*)Xk(k), W + w.wb1, (const float*)g0, g0, Zl(2 * k), St(2 * k),
                               W + w.pw1, Pout(), B);
        else
            hipLaunchKernelGGL((k_train_conv_bwd<false, true, true>), dim3(B), dim3(TTC), CONV_LDS, st, g1, Zl(1 + 2 * k), ob1, Pin(),
                               acc1, nof, nof, nof, nof, (const float*)Xk(k), W + w.wb1, (const float*)g0, g0, Zl(2 * k), St(2 * k),
                               W + w.pw1, Pout(), B);
        next();
    }
    {
        const BnBack ob0{St(0), net->stem_bn.weight};
        hipLaunchKernelGGL(k_train_stem_bwd, dim3(B), dim3(TT), 0, st, x, g0, Zl(0), ob0, Pin(), acc0, W + w.pws, B);
    }
    {
        FinishArgs f;
        f.pw1 = W + w.pw1; f.pw2 = W + w.pw2; f.pws = W + w.pws; f.pwh = W + w.pwh;
        f.boards = B; f.num_blocks = nb;
        f.g_c1 = net->c1_w_grad; f.g_c2 = net->c2_w_grad; f.g_stem = net->stem_w_grad; f.g_wp = net->pol_conv_w_grad;
        f.g_bp = net->pol_conv_b_grad; f.g_wv = net->val_conv_w_grad;
        f.bn_acc = W + w.bn_acc;
        f.g_bn0_w = net->stem_bn.weight_grad; f.g_bn0_b = net->stem_bn.bias_grad;
        f.g_bn1_w = net->bn1.weight_grad; f.g_bn1_b = net->bn1.bias_grad;
        f.g_bn2_w = net->bn2.weight_grad; f.g_bn2_b = net->bn2.bias_grad;
        f.stats = stats; f.stat_h = W + w.stat_h;
        f.r0 = RunStat{net->stem_bn.running_mean, net->stem_bn.running_var, (long long*)net->stem_bn.num_batches_tracked};
        f.r1 = RunStat{net->bn1.running_mean, net->bn1.running_var, (long long*)net->bn1.num_batches_tracked};
        f.r2 = RunStat{net->bn2.running_mean, net->bn2.running_var, (long long*)net->bn2.num_batches_tracked};
        f.rp = RunStat{net->pol_bn.running_mean, net->pol_bn.running_var, (long long*)net->pol_bn.num_batches_tracked};
        f.rv = RunStat{net->val_bn.running_mean, net->val_bn.running_var, (long long*)net->val_bn.num_batches_tracked};
        f.momentum = mom;
        hipLaunchKernelGGL(k_train_finish, dim3(FINISH_GROUPS), dim3(TT), 0, st, f);
    }
    AZ_LAUNCH_CHECK("training backward");
    if (logits7_out) AZ_HIPCHK(hipMemcpyAsync(logits7_out, W + w.logits7, (size_t)B * 7 * 4, hipMemcpyDeviceToDevice, st));
    if (vhat_out) AZ_HIPCHK(hipMemcpyAsync(vhat_out, W + w.vhat, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    return 0;
}

int az_c4_train_fwd_bwd(const az_c4_train_net* net, const float* x, const uint8_t* valid, const float* target_p,
                        const float* z, int32_t batch, void* workspace, float* out3, float* logits7_out, float* vhat_out,
                        void* stream) {
    if (int rc = check_args(net, x, valid, target_p, z, batch, workspace, out3, "az_c4_train_fwd_bwd")) return rc;
    int32_t max_batch = 0;
    if (int rc = az_c4_train_one_launch_max_batch(net->num_blocks, &max_batch)) return rc;
    static const bool chain_only = [] { const char* e = getenv("AZ_TRAIN_CHAIN"); return e && e[0] == '1'; }();
    if (batch > max_batch || chain_only)
        return az_c4_train_fwd_bwd_chain(net, x, valid, target_p, z, batch, workspace, out3, logits7_out, vhat_out, stream);
    hipStream_t st = (hipStream_t)stream;
    float* W = static_cast<float*>(workspace);
    PassArgs a;
    a.net = *net;
    a.x = x; a.valid = valid; a.target_p = target_p; a.z = z;
    a.W = W;
    a.w = layout(batch, net->num_blocks);
    a.boards = batch;
    a.out3 = out3; a.logits7_out = logits7_out; a.vhat_out = vhat_out;
    const int lds_bytes = pass_lds_bytes(net->num_blocks);
    // (the attribute only ever grows: a deeper network raises it again)
    static std::atomic<int> lds_set[64];
    int dev = 0;
    AZ_HIPCHK(hipGetDevice(&dev));
    if (dev >= 64 || lds_set[dev].load(std::memory_order_acquire) < lds_bytes) {
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (st && hipStreamIsCapturing(st, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone)
            return az_fail("az_c4_train_fwd_bwd: the first call for a device (or a deeper network) sets a kernel attribute and "
                           "cannot be recorded in a HIP graph: call it once outside the capture");
        AZ_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_train_pass), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        if (dev < 64) lds_set[dev].store(lds_bytes, std::memory_order_release);
    }
    if (int rc = launch_prep(net, W, a.w, batch, st)) return rc;
    hipLaunchKernelGGL(k_train_pass, dim3(batch), dim3(TTC), lds_bytes, st, a);
    AZ_LAUNCH_CHECK("k_train_pass");
    return 0;
}

}  // extern "C"
