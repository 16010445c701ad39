"""Fused HIP forward of the Connect Four policy/value network (csrc/az_net.hip).

`FusedC4Net.from_module(AlphaZeroC4)` folds eval-mode batch-norm into the convolutions,
packs the weights into the MFMA fragment order the kernel wants and evaluates engine
leaves (bitboards) directly: `net(leaf_bb, leaf_info) -> (p float32[n,7], v float32[n])`,
p soft-maxed over the legal columns (alphazero/mcts.py:94-107 for a whole batch).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr

CMAX = 32          # trunk channels supported by the fused kernel


def _fold_bn(w, bn, conv_bias=None):
    """Eval-mode BN folded into the preceding conv: returns (w * s[co], shift[co]) in float64."""
    s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
    shift = bn.bias.double() - bn.running_mean.double() * s
    if conv_bias is not None:
        shift = shift + conv_bias.double() * s
    return w.double() * s.view(-1, 1, 1, 1), shift


def _bf16_bits(t):
    return t.float().to(torch.bfloat16).view(torch.int16).numpy()


def _lane_k(s):
    """(row, k) held by element j of lane l in k-step s of a 32x32x16 MFMA A fragment:
    row = l & 31, k = 16*s + 8*(l >> 5) + j.  Returns index tensors [64, 8]."""
    lane = torch.arange(64).view(64, 1)
    j = torch.arange(8).view(1, 8)
    return (lane & 31).expand(64, 8), (16 * s + 8 * (lane >> 5) + j).expand(64, 8)


def _gemm_frags(mat, ksteps):
    """Dense matrix [rows <= 32][K] -> fragments [ksteps][64][8] of Wt[row][k], zero padded."""
    rows, K = mat.shape
    pad = torch.zeros(32, 16 * ksteps, dtype=torch.float64)
    pad[:rows, :K] = mat
    out = torch.zeros(ksteps, 64, 8, dtype=torch.float64)
    for s in range(ksteps):
        r, k = _lane_k(s)
        out[s] = pad[r, k]
    return _bf16_bits(out)


def _conv_frags(w):
    """Folded 3x3 conv weight [co <= 32][ci <= 32][3][3] -> [18][64][8]; k-step s covers tap s >> 1 and
    input channels 16*(s & 1) .. +15, i.e. k = tap*32 + ci."""
    co, ci = w.shape[:2]
    mat = torch.zeros(32, 9 * 32, dtype=torch.float64)
    for tap in range(9):
        mat[:co, tap * 32:tap * 32 + ci] = w[:, :, tap // 3, tap % 3]
    return _gemm_frags(mat, 18)


def pack_c4_blob(model):
    """AlphaZeroC4 module -> uint8 blob in the layout of csrc/az_net.hip."""
    C_ = model.block_channels
    if C_ > CMAX:
        raise ValueError(f'fused kernel supports up to {CMAX} trunk channels, got {C_}')
    if tuple(model.shape_in) != (2, 6, 7):
        raise ValueError('fused kernel is for Connect Four (2x6x7 input)')
    lib = _lib.load()
    blob = np.zeros(lib.az_c4net_blob_bytes(), np.uint8)
    off = 0

    def put(arr):
        nonlocal off
        a = np.ascontiguousarray(arr).reshape(-1).view(np.uint8)
        blob[off:off + a.size] = a
        off += a.size

    def putf(arr, pad_to=None):
        a = np.ascontiguousarray(arr, np.float32).reshape(-1)
        if pad_to:
            a = np.concatenate([a, np.zeros(pad_to - a.size, np.float32)])
        put(a)

    with torch.no_grad():
        bias = np.zeros((3, 32), np.float32)
        # stem: K index = tap*2 + plane
        w, b = _fold_bn(model._cbr[0].weight.cpu(), model._cbr[1].cpu())
        mat = torch.zeros(C_, 18, dtype=torch.float64)
        for tap in range(9):
            mat[:, 2 * tap] = w[:, 0, tap // 3, tap % 3]
            mat[:, 2 * tap + 1] = w[:, 1, tap // 3, tap % 3]
        put(_gemm_frags(mat, 2))
        bias[0, :C_] = b.float().numpy()
        if len(model._tower):
            blk = model._tower[0]
            for i, (cv, bn) in enumerate(((blk._c1, blk._bn1), (blk._c2, blk._bn2))):
                w, b = _fold_bn(cv.weight.cpu(), bn.cpu())
                put(_conv_frags(w))
                bias[1 + i, :C_] = b.float().numpy()
        else:
            off += 2 * 18 * 1024
        # heads: 1x1 convs -> rows 0,1 (policy planes) and 2 (value plane) of one [32][32] matrix
        ph, vh = model._policy_head, model._value_head
        wpol, bpol = _fold_bn(ph._cbr[0].weight.cpu(), ph._cbr[1].cpu(), ph._cbr[0].bias.cpu())
        wval, bval = _fold_bn(vh._cbr[0].weight.cpu(), vh._cbr[1].cpu())
        mat = torch.zeros(3, 32, dtype=torch.float64)
        mat[0:2, :C_] = wpol[:, :, 0, 0]
        mat[2, :C_] = wval[0, :, 0, 0]
        put(_gemm_frags(mat, 2))
        # value FC1 [256][42]: 8 row tiles x 3 k-steps
        w1 = vh._fc[0].weight.cpu().double()
        for t in range(8):
            put(_gemm_frags(w1[32 * t:32 * t + 32], 3))
        # policy FC [42][84]: 2 row tiles x 6 k-steps
        wp = ph._fc.weight.cpu().double()
        put(_gemm_frags(wp[0:32], 6))
        put(_gemm_frags(wp[32:42], 6))
        putf(np.concatenate([bias.reshape(-1), [float(bpol[0]), float(bpol[1]), float(bval[0]), 0.0]]))
        putf(ph._fc.bias.cpu().numpy(), 64)
        putf(vh._fc[0].bias.cpu().numpy())
        putf(vh._fc[2].weight.cpu().numpy())
        putf(np.array([vh._fc[2].bias.item(), 0, 0, 0]))
    assert off == blob.size, (off, blob.size)
    return blob


class FusedC4Net:
    def __init__(self, blob, num_blocks, device):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.num_blocks = int(num_blocks)
        self.blob = torch.from_numpy(blob).to(self.device)
        self._p = self._v = None

    @classmethod
    def from_module(cls, model, device=None):
        device = device if device is not None else next(model.parameters()).device
        return cls(pack_c4_blob(model), model.num_blocks, device)

    def load_module(self, model):
        """Refresh the packed weights from `model` (after a weight broadcast / optimiser step)."""
        self.blob.copy_(torch.from_numpy(pack_c4_blob(model)))
        self.num_blocks = model.num_blocks

    def __call__(self, leaf_bb, leaf_info):
        n = leaf_bb.shape[0]
        if self._p is None or self._p.shape[0] != n:
            self._p = torch.empty(n, 7, dtype=torch.float32, device=self.device)
            self._v = torch.empty(n, dtype=torch.float32, device=self.device)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self.lib.az_c4net_bf16_forward(ptr(self.blob), self.num_blocks, ptr(leaf_bb), ptr(leaf_info),
                                             n, ptr(self._p), ptr(self._v), st))
        return self._p, self._v
