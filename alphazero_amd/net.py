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


def _frags(w):
    """Folded conv weight [32][32][3][3] -> MFMA A fragments [18][64][8] (bf16 as int16):
    lane l holds Wt[co = l & 31][k = 8*(l >> 5) + j] of k-step s, with tap = s >> 1 and
    input channel ci = 16*(s & 1) + 8*(l >> 5) + j."""
    out = torch.zeros(18, 64, 8, dtype=torch.float32)
    lane = torch.arange(64)
    co, h = lane & 31, lane >> 5
    for s in range(18):
        tap = s >> 1
        ty, tx = tap // 3, tap % 3
        for j in range(8):
            ci = 16 * (s & 1) + 8 * h + j
            out[s, :, j] = w[co, ci, ty, tx]
    return out.to(torch.bfloat16).view(torch.int16).numpy()


def pack_c4_blob(model):
    """AlphaZeroC4 module -> uint8 blob in the layout of csrc/az_net.hip."""
    C_ = model.block_channels
    if C_ > CMAX:
        raise ValueError(f'fused kernel supports up to {CMAX} trunk channels, got {C_}')
    if tuple(model.shape_in) != (2, 6, 7):
        raise ValueError('fused kernel is for Connect Four (2x6x7 input)')
    lib = _lib.load()
    blob = np.zeros(lib.az_c4net_blob_bytes(), np.uint8)
    with torch.no_grad():
        convs = []
        w, b = _fold_bn(model._cbr[0].weight.cpu(), model._cbr[1].cpu())
        convs.append((w, b))
        blk = model._tower[0] if len(model._tower) else None
        if blk is not None:
            convs.append(_fold_bn(blk._c1.weight.cpu(), blk._bn1.cpu()))
            convs.append(_fold_bn(blk._c2.weight.cpu(), blk._bn2.cpu()))
        else:
            z = torch.zeros(C_, C_, 3, 3, dtype=torch.float64)
            convs += [(z, torch.zeros(C_, dtype=torch.float64))] * 2
        off = 0
        bias = np.zeros((3, 32), np.float32)
        for i, (w, b) in enumerate(convs):
            wp = torch.zeros(32, 32, 3, 3, dtype=torch.float64)
            wp[:w.shape[0], :w.shape[1]] = w
            fr = _frags(wp.float())
            blob[off:off + fr.nbytes] = fr.view(np.uint8).reshape(-1)
            off += fr.nbytes
            bias[i, :b.shape[0]] = b.float().numpy()
        # heads: 1x1 convs -> rows 0,1 (policy) and 2 (value) of one 32x32 weight matrix
        ph, vh = model._policy_head, model._value_head
        wpol, bpol = _fold_bn(ph._cbr[0].weight.cpu(), ph._cbr[1].cpu(), ph._cbr[0].bias.cpu())
        wval, bval = _fold_bn(vh._cbr[0].weight.cpu(), vh._cbr[1].cpu())
        wh = torch.zeros(32, 32, 3, 3, dtype=torch.float64)
        wh[0:2, :C_, 1, 1] = wpol[:, :, 0, 0]
        wh[2, :C_, 1, 1] = wval[0, :, 0, 0]
        hf = _frags(wh.float())[8:10]            # centre tap = k-steps 8, 9
        blob[off:off + hf.nbytes] = hf.view(np.uint8).reshape(-1)
        off += hf.nbytes
        blob[off:off + bias.nbytes] = bias.view(np.uint8).reshape(-1)
        off += bias.nbytes
        hb = np.array([bpol[0], bpol[1], bval[0], 0.0], np.float32)
        blob[off:off + 16] = hb.view(np.uint8)
        off += 16

        def put(arr, pad_to=None):
            nonlocal off
            a = np.ascontiguousarray(arr, np.float32).reshape(-1)
            if pad_to:
                a = np.concatenate([a, np.zeros(pad_to - a.size, np.float32)])
            blob[off:off + a.nbytes] = a.view(np.uint8)
            off += a.nbytes
        put(ph._fc.weight.cpu().numpy())                       # [42][84]
        put(ph._fc.bias.cpu().numpy(), 44)
        put(vh._fc[0].weight.cpu().numpy().T)                  # [42][256]
        put(vh._fc[0].bias.cpu().numpy())
        put(vh._fc[2].weight.cpu().numpy())                    # [256]
        put(np.array([vh._fc[2].bias.item(), 0, 0, 0]))
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
