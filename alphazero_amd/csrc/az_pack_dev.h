// az_pack_dev.h -- what the weight-packing kernels share (az_net.hip: bf16 fragments; az_net_hx.hip: fp16 pairs):
// the module's parameters as plain pointers and the eval-mode batch-norm folding, in double as the packing in
// tensor operations (alphazero_amd/net.py) always did it.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/azhip.h"

// Eval-mode batch-norm folded into the conv in double, as the packing always did: s = gamma / sqrt(var + eps),
// w' = w * s, b' = beta - mean * s (+ conv bias * s); then double -> float -> bf16, round to nearest even both times.
struct FoldBn { const float* gamma; const float* beta; const float* mean; const float* var; };
__device__ __forceinline__ double fold_scale(const FoldBn& bn, int c, double eps) {
    return (double)bn.gamma[c] / sqrt((double)bn.var[c] + eps);
}
struct PackArgs {
    int channels, num_blocks;
    double eps;
    const float* stem_w; FoldBn stem_bn;
    const float* c1_w; FoldBn bn1;
    const float* c2_w; FoldBn bn2;
    const float* pol_conv_w; const float* pol_conv_b; FoldBn pol_bn;
    const float* val_conv_w; FoldBn val_bn;
    const float* val_fc1_w; const float* val_fc1_b; const float* val_fc2_w; const float* val_fc2_b;
    const float* pol_fc_w; const float* pol_fc_b;
};
// output channel computed by row r of MFMA half h in the trunk (the MFMA result lands in the lanes' own B-fragment layout)
__device__ __forceinline__ int half_row(int h, int r) { return 8 * (r / 4) + 4 * h + r % 4; }


inline PackArgs make_pack_args(const az_c4_train_net* net, int channels, double bn_eps) {
    auto bn = [](const az_bn& b) { return FoldBn{b.weight, b.bias, b.running_mean, b.running_var}; };
    PackArgs a;
    a.channels = channels; a.num_blocks = net->num_blocks; a.eps = bn_eps;
    a.stem_w = net->stem_w; a.stem_bn = bn(net->stem_bn);
    a.c1_w = net->c1_w; a.bn1 = bn(net->bn1);
    a.c2_w = net->c2_w; a.bn2 = bn(net->bn2);
    a.pol_conv_w = net->pol_conv_w; a.pol_conv_b = net->pol_conv_b; a.pol_bn = bn(net->pol_bn);
    a.val_conv_w = net->val_conv_w; a.val_bn = bn(net->val_bn);
    a.val_fc1_w = net->val_fc1_w; a.val_fc1_b = net->val_fc1_b; a.val_fc2_w = net->val_fc2_w; a.val_fc2_b = net->val_fc2_b;
    a.pol_fc_w = net->pol_fc_w; a.pol_fc_b = net->pol_fc_b;
    return a;
}

// A trunk weight as the packed layouts index it: fragment `frag` (0 = stem, 1..9 / 10..18 = the taps of the two convs,
// 19 = the identity), output channel co, contraction index k; folded, in double.
__device__ __forceinline__ double trunk_weight(const PackArgs& a, int frag, int co, int k) {
    const int C = a.channels;
    if (frag == 0) {
        if (co < C && k < 27 && k % 9 < 6) {                 // k = 9 kx + 3 plane + ky: 18 of the 32 are used
            const int kx = k / 9, plane = (k % 9) / 3, ky = k % 3;
            return (double)a.stem_w[((co * 2 + plane) * 3 + ky) * 3 + kx] * fold_scale(a.stem_bn, co, a.eps);
        }
        return 0.0;
    }
    if (frag <= 18) {
        const int conv = (frag - 1) / 9, tap = (frag - 1) % 9;
        if (a.num_blocks > 0 && co < C && k < C)
            return (double)(conv ? a.c2_w : a.c1_w)[(co * C + k) * 9 + tap] * fold_scale(conv ? a.bn2 : a.bn1, co, a.eps);
        return 0.0;
    }
    return co == k ? 1.0 : 0.0;
}
// row 0, 1: the policy planes' 1x1 conv, row 2: the value plane's; folded
__device__ __forceinline__ double head_conv_weight(const PackArgs& a, int row, int k) {
    if (k >= a.channels) return 0.0;
    if (row < 2) return (double)a.pol_conv_w[row * a.channels + k] * fold_scale(a.pol_bn, row, a.eps);
    if (row == 2) return (double)a.val_conv_w[k] * fold_scale(a.val_bn, 0, a.eps);
    return 0.0;
}
// the hundred floats both layouts start their fp32 part with: conv biases [3][32], head conv biases [3], one more
__device__ __forceinline__ double folded_bias(const PackArgs& a, int f) {
    if (f < 96) {
        const int layer = f / 32, c = f % 32;
        if (c < a.channels && (layer == 0 || a.num_blocks > 0)) {
            const FoldBn& bn = layer == 0 ? a.stem_bn : layer == 1 ? a.bn1 : a.bn2;
            return (double)bn.beta[c] - (double)bn.mean[c] * fold_scale(bn, c, a.eps);
        }
        return 0.0;
    }
    if (f < 98) {
        const int c = f - 96;
        const double s = fold_scale(a.pol_bn, c, a.eps);
        return ((double)a.pol_bn.beta[c] - (double)a.pol_bn.mean[c] * s) + (double)a.pol_conv_b[c] * s;
    }
    if (f == 98) return (double)a.val_bn.beta[0] - (double)a.val_bn.mean[0] * fold_scale(a.val_bn, 0, a.eps);
    return 0.0;
}
