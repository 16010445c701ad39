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


def _device(d):
    """torch.device with an explicit ordinal ('cuda' means the device current at construction)."""
    d = torch.device(d)
    if d.type == 'cuda' and d.index is None:
        d = torch.device('cuda', torch.cuda.current_device())
    return d


def _launch(device, fn):
    """Call fn(stream) with `device` current: the network entry points launch on the current device,
    so an evaluator built for cuda:1 must not run on whatever device the caller happens to have selected."""
    st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    if torch.cuda.current_device() == device.index:
        return check(fn(st))
    with torch.cuda.device(device):
        return check(fn(st))


def _fold_bn(w, bn, conv_bias=None):
    """Eval-mode BN folded into the preceding conv: returns (w * s[co], shift[co]) in float64."""
    s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
    shift = bn.bias.double() - bn.running_mean.double() * s
    if conv_bias is not None:
        shift = shift + conv_bias.double() * s
    return w.double() * s.view(-1, 1, 1, 1), shift


_FRAG_INDEX = {}


def _frag_index(ksteps, device):
    """Flat gather index of 32x32x16 MFMA A fragments [ksteps][64][8] into a row-major [32][16*ksteps]
    matrix: element j of lane l in k-step s is Wt[row = l & 31][k = 16*s + 8*(l >> 5) + j]."""
    key = (ksteps, str(device))
    if key not in _FRAG_INDEX:
        s = torch.arange(ksteps).view(-1, 1, 1)
        lane = torch.arange(64).view(1, -1, 1)
        j = torch.arange(8).view(1, 1, -1)
        idx = (lane & 31) * (16 * ksteps) + 16 * s + 8 * (lane >> 5) + j
        _FRAG_INDEX[key] = idx.reshape(-1).to(device)
    return _FRAG_INDEX[key]


def _gemm_frags(mat, ksteps):
    """Dense matrix [rows <= 32][K <= 16*ksteps] -> bf16 fragment bytes (uint8 tensor), 32x32x16 MFMA."""
    rows, K = mat.shape
    pad = torch.zeros(32, 16 * ksteps, dtype=torch.float64, device=mat.device)
    pad[:rows, :K] = mat
    fr = pad.reshape(-1)[_frag_index(ksteps, mat.device)]
    return fr.float().to(torch.bfloat16).contiguous().view(torch.uint8)


def _frag16_index(device):
    """Gather index of ONE 16x16x32 MFMA A fragment [64 lanes][8] into a row-major [16][32] matrix:
    element j of lane l is Wt[row = l & 15][k = 8*(l >> 4) + j]."""
    key = ('16', str(device))
    if key not in _FRAG_INDEX:
        lane = torch.arange(64).view(-1, 1)
        j = torch.arange(8).view(1, -1)
        _FRAG_INDEX[key] = ((lane & 15) * 32 + 8 * (lane >> 4) + j).reshape(-1).to(device)
    return _FRAG_INDEX[key]


def _half_rows(device):
    """Output channel computed by row r of MFMA half h in the trunk: 8*(r // 4) + 4*h + r % 4, so that the
    MFMA result lands in the lanes' own B-fragment layout (csrc/az_net_dev.h).  int64 [2][16]."""
    r = torch.arange(16, device=device)
    return torch.stack([8 * (r // 4) + 4 * h + r % 4 for h in range(2)])


def _frags16(mat):
    """[..., 16 rows, 32 k] float64 -> bf16 fragment bytes [..., 64, 8] (uint8 tensor)."""
    lead = mat.shape[:-2]
    flat = mat.reshape(-1, 16 * 32)[:, _frag16_index(mat.device)]
    return flat.reshape(*lead, 64 * 8).float().to(torch.bfloat16).contiguous().view(torch.uint8).reshape(-1)


def _trunk_frags(w32):
    """[..., 32 out channels (zero padded), 32 k] -> fragments [..., 2 halves, 64, 8]: the rows of half h are the
    output channels _half_rows()[h]."""
    rows = _half_rows(w32.device)                                   # [2][16]
    return _frags16(w32[..., rows, :])                              # [..., 2, 16, 32]


def _conv_taps(w):
    """Folded 3x3 conv weight [co][ci][3][3] -> [9 taps][32][32] (tap = ky*3 + kx, zero padded)."""
    co, ci = w.shape[:2]
    mat = torch.zeros(9, 32, 32, dtype=torch.float64, device=w.device)
    mat[:, :co, :ci] = w.permute(2, 3, 0, 1).reshape(9, co, ci)
    return mat


def _f32_bytes(t, pad_to=None):
    t = t.detach().reshape(-1).float()
    if pad_to and t.numel() < pad_to:
        t = torch.cat([t, torch.zeros(pad_to - t.numel(), device=t.device)])
    return t.contiguous().view(torch.uint8)


def pack_c4_blob(model, out=None):
    """AlphaZeroC4 module -> uint8 tensor (on the module's device) in the layout of csrc/az_net.hip; into `out` when
    given.  On a GPU it is one launch (az_c4net_pack_bf16), cheap enough to run after every optimiser step; a module on
    the CPU goes through the tensor operations of _pack_c4_blob_torch, which state the same layout and give the same
    bytes (tests/test_gpu_net.py)."""
    dev = model._cbr[0].weight.device
    if dev.type != 'cuda' or any(t.dtype != torch.float32 or not t.is_contiguous() for t in model.parameters()):
        blob = _pack_c4_blob_torch(model)
        if out is not None:
            out.copy_(blob)
            return out
        return blob
    C_ = model.block_channels
    if C_ > CMAX:
        raise ValueError(f'fused kernel supports up to {CMAX} trunk channels, got {C_}')
    if tuple(model.shape_in) != (2, 6, 7):
        raise ValueError('fused kernel is for Connect Four (2x6x7 input)')
    from .train_c4 import param_struct
    lib = _lib.load()
    if out is None:
        out = torch.empty(lib.az_c4net_blob_bytes(), dtype=torch.uint8, device=dev)
    assert out.is_contiguous() and out.dtype == torch.uint8 and out.numel() == lib.az_c4net_blob_bytes() and out.device == dev
    net = param_struct(model)
    eps = float(model._cbr[1].eps)
    assert all(float(m.eps) == eps for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d))
    _launch(_device(dev), lambda st: lib.az_c4net_pack_bf16(C.byref(net), C_, C.c_double(eps), ptr(out), st))
    return out


def _pack_c4_blob_torch(model):
    """The packed layout of csrc/az_net.hip stated in tensor operations (some 150 small launches on a GPU)."""
    C_ = model.block_channels
    if C_ > CMAX:
        raise ValueError(f'fused kernel supports up to {CMAX} trunk channels, got {C_}')
    if tuple(model.shape_in) != (2, 6, 7):
        raise ValueError('fused kernel is for Connect Four (2x6x7 input)')
    dev = model._cbr[0].weight.device
    parts = []
    with torch.no_grad():
        bias = torch.zeros(3, 32, dtype=torch.float64, device=dev)
        # stem: one k-step, K index = 9 * kx + 3 * plane + ky (18 of 32 used: the order in which the kernel takes
        # the window out of the bitboards)
        w, b = _fold_bn(model._cbr[0].weight, model._cbr[1])
        stem = torch.zeros(32, 32, dtype=torch.float64, device=dev)
        for kx in range(3):
            for plane in range(2):
                for ky in range(3):
                    stem[:C_, 9 * kx + 3 * plane + ky] = w[:, plane, ky, kx]
        parts.append(_trunk_frags(stem))
        bias[0, :C_] = b
        if len(model._tower):
            blk = model._tower[0]
            for i, (cv, bn) in enumerate(((blk._c1, blk._bn1), (blk._c2, blk._bn2))):
                w, b = _fold_bn(cv.weight, bn)
                parts.append(_trunk_frags(_conv_taps(w)))           # [9 taps][2 halves][64][8]
                bias[1 + i, :C_] = b
        else:
            parts.append(torch.zeros(2 * 18 * 1024, dtype=torch.uint8, device=dev))
        # the residual connection as a "tap" of the second conv: identity over the 32 channels
        parts.append(_trunk_frags(torch.eye(32, dtype=torch.float64, device=dev)))
        # heads: 1x1 convs -> rows 0,1 (policy planes) and 2 (value plane) of one [16][32] matrix
        ph, vh = model._policy_head, model._value_head
        wpol, bpol = _fold_bn(ph._cbr[0].weight, ph._cbr[1], ph._cbr[0].bias)
        wval, bval = _fold_bn(vh._cbr[0].weight, vh._cbr[1])
        mat = torch.zeros(16, 32, dtype=torch.float64, device=dev)
        mat[0:2, :C_] = wpol[:, :, 0, 0]
        mat[2, :C_] = wval[0, :, 0, 0]
        parts.append(_frags16(mat))
        w1 = vh._fc[0].weight.double()                          # value FC1 [256][42]: 8 row tiles x 3 k-steps
        for t in range(8):
            parts.append(_gemm_frags(w1[32 * t:32 * t + 32], 3))
        wp = ph._fc.weight.double()                             # policy FC [42][84]: 2 row tiles x 6 k-steps
        parts.append(_gemm_frags(wp[0:32], 6))
        parts.append(_gemm_frags(wp[32:42], 6))
        parts.append(_f32_bytes(torch.cat([bias.reshape(-1), bpol[:2], bval[:1], torch.zeros(1, dtype=torch.float64, device=dev)])))
        parts.append(_f32_bytes(ph._fc.bias, 64))
        parts.append(_f32_bytes(vh._fc[0].bias))
        parts.append(_f32_bytes(vh._fc[2].weight))
        parts.append(_f32_bytes(torch.cat([vh._fc[2].bias.reshape(1), torch.zeros(3, device=dev)])))
        blob = torch.cat(parts)
    assert blob.numel() == _lib.load().az_c4net_blob_bytes(), (blob.numel(), _lib.load().az_c4net_blob_bytes())
    return blob


WIRE_HEADER = 256      # bytes in front of the packed weights in FusedC4Net.wire: int64 {model version, num_blocks}


class FusedC4Net:
    """Evaluator over the packed weights.  `wire` is the one buffer that travels between ranks when new
    weights are published (distributed.broadcast_packed): a 256-byte header and the kernel-ready blob, so a
    receiving rank uses the weights where they land, without re-packing."""

    def __init__(self, blob, num_blocks, device):
        self.lib = _lib.load()
        self.device = _device(device)
        self.num_blocks = int(num_blocks)
        self.wire = torch.zeros(WIRE_HEADER + blob.numel(), dtype=torch.uint8, device=self.device)
        self.blob = self.wire[WIRE_HEADER:]
        self.blob.copy_(blob)
        assert self.blob.data_ptr() % 16 == 0
        self._p = self._v = None

    @classmethod
    def from_module(cls, model, device=None):
        device = device if device is not None else next(model.parameters()).device
        return cls(pack_c4_blob(model), model.num_blocks, device)

    def load_module(self, model):
        """Refresh the packed weights from `model` (after a weight broadcast / optimiser step)."""
        if next(model.parameters()).device == self.blob.device:
            pack_c4_blob(model, out=self.blob)
        else:
            self.blob.copy_(pack_c4_blob(model))
        self.num_blocks = model.num_blocks

    def __call__(self, leaf_bb, leaf_info):
        n = leaf_bb.shape[0]
        if self._p is None or self._p.shape[0] != n:
            self._p = torch.empty(n, 7, dtype=torch.float32, device=self.device)
            self._v = torch.empty(n, dtype=torch.float32, device=self.device)
        _launch(self.device, lambda st: self.lib.az_c4net_bf16_forward(ptr(self.blob), self.num_blocks, ptr(leaf_bb), ptr(leaf_info),
                                             n, ptr(self._p), ptr(self._v), st))
        return self._p, self._v


def pack_c4_blob_f32(model):
    """AlphaZeroC4 module -> float32 tensor in the layout of csrc/az_net_f32.hip (BN folded in
    float64, then rounded once to float32)."""
    C_ = model.block_channels
    if C_ > CMAX or tuple(model.shape_in) != (2, 6, 7):
        raise ValueError('fused fp32 kernel: Connect Four nets with up to 32 trunk channels')
    dev = model._cbr[0].weight.device
    z = lambda *shape: torch.zeros(*shape, dtype=torch.float64, device=dev)
    with torch.no_grad():
        bias = z(3, 32)
        w, b = _fold_bn(model._cbr[0].weight, model._cbr[1])
        stem = z(18, 32)
        stem[:, :C_] = w.permute(2, 3, 1, 0).reshape(18, C_)            # k = tap*2 + plane
        bias[0, :C_] = b
        convs = [z(288, 32), z(288, 32)]
        if len(model._tower):
            blk = model._tower[0]
            for i, (cv, bn) in enumerate(((blk._c1, blk._bn1), (blk._c2, blk._bn2))):
                w, b = _fold_bn(cv.weight, bn)
                m = z(9, 32, 32)
                m[:, :C_, :C_] = w.permute(2, 3, 1, 0).reshape(9, C_, C_)   # [tap][ci][co]
                convs[i] = m.reshape(288, 32)
                bias[1 + i, :C_] = b
        ph, vh = model._policy_head, model._value_head
        wpol, bpol = _fold_bn(ph._cbr[0].weight, ph._cbr[1], ph._cbr[0].bias)
        wval, bval = _fold_bn(vh._cbr[0].weight, vh._cbr[1])
        hw = z(3, 32)
        hw[0:2, :C_] = wpol[:, :, 0, 0]
        hw[2, :C_] = wval[0, :, 0, 0]
        hb = torch.cat([bpol[:2], bval[:1], z(1)])
        bp = torch.cat([ph._fc.bias.double(), z(2)])
        parts = [stem, convs[0], convs[1], bias, hw, hb, ph._fc.weight.double(), bp,
                 vh._fc[0].weight.double().t().contiguous(), vh._fc[0].bias.double(), vh._fc[2].weight.double(),
                 torch.cat([vh._fc[2].bias.double().reshape(1), z(3)])]
        blob = torch.cat([t.reshape(-1) for t in parts]).float().contiguous()
    assert blob.numel() == _lib.load().az_c4net_f32_blob_floats(), blob.numel()
    return blob


class FusedC4NetF32:
    """fp32 parity evaluator: net(leaf_bb, leaf_info) -> (p, v); evaluates every leaf."""

    def __init__(self, model, device=None):
        self.lib = _lib.load()
        self.device = _device(device if device is not None else next(model.parameters()).device)
        self.num_blocks = model.num_blocks
        self.blob = pack_c4_blob_f32(model).to(self.device)
        self._p = self._v = None

    def load_module(self, model):
        self.blob.copy_(pack_c4_blob_f32(model))

    def __call__(self, leaf_bb, leaf_info=None):
        n = leaf_bb.shape[0]
        if self._p is None or self._p.shape[0] != n:
            self._p = torch.empty(n, 7, dtype=torch.float32, device=self.device)
            self._v = torch.empty(n, dtype=torch.float32, device=self.device)
        _launch(self.device, lambda st: self.lib.az_c4net_f32_forward(ptr(self.blob), self.num_blocks, ptr(leaf_bb), n,
                                            ptr(self._p), ptr(self._v), st))
        return self._p, self._v


HX_SCALE = 2048.0     # the low operand half is carried times 2^11 (csrc/az_net_hx_dev.h)


def _frags16_hx(mat):
    """[..., 16 rows, 32 k] float64 -> fp16 fragment bytes [..., 2 (h, l), 64, 8] (uint8 tensor): each weight
    (rounded once to fp32) as h = fp16(w) and l = fp16((w - h) * 2048)."""
    lead = mat.shape[:-2]
    flat = mat.reshape(-1, 16 * 32)[:, _frag16_index(mat.device)].float()          # [n][512]
    if not bool(torch.all(flat.abs() < 6.0e4)):
        raise ValueError('fp16-pair kernel: a folded weight lies beyond the fp16 range')
    h = flat.to(torch.float16)
    l = ((flat - h.float()) * HX_SCALE).to(torch.float16)
    both = torch.stack([h, l], dim=1)                                             # [n][2][512]
    return both.reshape(*lead, 2 * 512).contiguous().view(torch.uint8).reshape(-1)


def _trunk_frags_hx(w32):
    """[..., 32 out channels, 32 k] -> fragments [..., 2 channel halves, 2 (h, l), 64, 8] (rows as _trunk_frags)."""
    rows = _half_rows(w32.device)
    return _frags16_hx(w32[..., rows, :])


def pack_c4_blob_hx(model, out=None, range_error=None):
    """AlphaZeroC4 module -> uint8 tensor in the layout of csrc/az_net_hx_dev.h (BN folded in float64, weights
    rounded once to float32, trunk weights then split into fp16 pairs); into `out` when given.  On a GPU it is one
    launch (az_c4net_pack_hx); range_error = a device int32[1] that the launch sets when a folded weight does not fit
    fp16 (without it the call waits for the launch and raises, like the tensor statement _pack_c4_blob_hx_torch, which
    a module on the CPU goes through and which gives the same bytes)."""
    dev = model._cbr[0].weight.device
    if dev.type != 'cuda' or any(t.dtype != torch.float32 or not t.is_contiguous() for t in model.parameters()):
        blob = _pack_c4_blob_hx_torch(model)
        if out is not None:
            out.copy_(blob)
            return out
        return blob
    C_ = model.block_channels
    if C_ > CMAX or tuple(model.shape_in) != (2, 6, 7):
        raise ValueError('fp16-pair kernel: Connect Four nets with up to 32 trunk channels')
    from .train_c4 import param_struct
    lib = _lib.load()
    nbytes = lib.az_c4net_hx_blob_bytes()
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    assert out.is_contiguous() and out.dtype == torch.uint8 and out.numel() == nbytes and out.device == dev
    wait = range_error is None
    if wait:
        range_error = torch.zeros(1, dtype=torch.int32, device=dev)
    net = param_struct(model)
    eps = float(model._cbr[1].eps)
    assert all(float(m.eps) == eps for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d))
    _launch(_device(dev), lambda st: lib.az_c4net_pack_hx(C.byref(net), C_, C.c_double(eps), ptr(out), ptr(range_error), st))
    if wait and int(range_error.item()):
        raise ValueError('fp16-pair kernel: a folded weight lies beyond the fp16 range')
    return out


def _pack_c4_blob_hx_torch(model):
    """The packed layout of csrc/az_net_hx_dev.h stated in tensor operations."""
    C_ = model.block_channels
    if C_ > CMAX or tuple(model.shape_in) != (2, 6, 7):
        raise ValueError('fp16-pair kernel: Connect Four nets with up to 32 trunk channels')
    dev = model._cbr[0].weight.device
    z = lambda *shape: torch.zeros(*shape, dtype=torch.float64, device=dev)
    parts = []
    with torch.no_grad():
        bias = z(3, 32)
        w, b = _fold_bn(model._cbr[0].weight, model._cbr[1])
        stem = z(32, 32)
        for kx in range(3):
            for plane in range(2):
                for ky in range(3):
                    stem[:C_, 9 * kx + 3 * plane + ky] = w[:, plane, ky, kx]
        parts.append(_trunk_frags_hx(stem))
        bias[0, :C_] = b
        if len(model._tower):
            blk = model._tower[0]
            for i, (cv, bn) in enumerate(((blk._c1, blk._bn1), (blk._c2, blk._bn2))):
                w, b = _fold_bn(cv.weight, bn)
                parts.append(_trunk_frags_hx(_conv_taps(w)))         # [9 taps][2 halves][h, l][64][8]
                bias[1 + i, :C_] = b
        else:
            parts.append(torch.zeros(2 * 9 * 4 * 1024, dtype=torch.uint8, device=dev))
        ident = _trunk_frags_hx(torch.eye(32, dtype=torch.float64, device=dev)).reshape(2, 2, 1024)
        parts.append(ident[:, 0].reshape(-1))                         # the identity has no low half
        ph, vh = model._policy_head, model._value_head
        wpol, bpol = _fold_bn(ph._cbr[0].weight, ph._cbr[1], ph._cbr[0].bias)
        wval, bval = _fold_bn(vh._cbr[0].weight, vh._cbr[1])
        mat = z(16, 32)
        mat[0:2, :C_] = wpol[:, :, 0, 0]
        mat[2, :C_] = wval[0, :, 0, 0]
        parts.append(_frags16_hx(mat))
        parts.append(_f32_bytes(torch.cat([bias.reshape(-1), bpol[:2], bval[:1], vh._fc[2].bias.double().reshape(1)])))
        parts.append(_f32_bytes(ph._fc.bias, 64))
        parts.append(_f32_bytes(vh._fc[0].bias))
        parts.append(_f32_bytes(vh._fc[2].weight))
        parts.append(_f32_bytes(vh._fc[0].weight.t().reshape(42, 64, 4)))      # [cell][lane][u] = W1[4 lane + u][cell]
        parts.append(_f32_bytes(ph._fc.weight))                                # [42 cells][84]
        blob = torch.cat(parts)
    assert blob.numel() == _lib.load().az_c4net_hx_blob_bytes(), (blob.numel(), _lib.load().az_c4net_hx_blob_bytes())
    return blob


class FusedC4NetHX:
    """fp32-grade evaluator on the fp16 matrix cores (csrc/az_net_hx.hip): net(leaf_bb, leaf_info) -> (p, v),
    within 1e-5 relative of the fp32 torch module; with leaf_info only the rows that need the network are
    evaluated and written.  SelfPlayEngine.move_fused(net) runs the one-launch move kernel with these weights."""
    precision = 'hx'

    def __init__(self, model, device=None):
        self.lib = _lib.load()
        self.device = _device(device if device is not None else next(model.parameters()).device)
        self.num_blocks = model.num_blocks
        blob = pack_c4_blob_hx(model)
        # as FusedC4Net: header + kernel-ready blob in one buffer, so that distributed.broadcast_packed can ship it
        self.wire = torch.zeros(WIRE_HEADER + blob.numel(), dtype=torch.uint8, device=self.device)
        self.blob = self.wire[WIRE_HEADER:]
        self.blob.copy_(blob)
        assert self.blob.data_ptr() % 16 == 0
        self._p = self._v = None
        self.range_error = torch.zeros(1, dtype=torch.int32, device=self.device)

    def load_module(self, model, check=False):
        """Refresh the packed weights.  A folded weight beyond the fp16 range sets self.range_error (device int32[1],
        sticky) instead of raising, so that a training loop can hand weights over after every step without waiting
        for the device; check=True waits and raises."""
        if next(model.parameters()).device == self.blob.device:
            pack_c4_blob_hx(model, out=self.blob, range_error=self.range_error)
        else:
            self.blob.copy_(pack_c4_blob_hx(model))
        self.num_blocks = model.num_blocks
        if check and int(self.range_error.item()):
            raise ValueError('fp16-pair kernel: a folded weight lies beyond the fp16 range')

    def __call__(self, leaf_bb, leaf_info=None):
        n = leaf_bb.shape[0]
        if self._p is None or self._p.shape[0] != n:
            self._p = torch.empty(n, 7, dtype=torch.float32, device=self.device)
            self._v = torch.empty(n, dtype=torch.float32, device=self.device)
        _launch(self.device, lambda st: self.lib.az_c4net_hx_forward(ptr(self.blob), self.num_blocks, ptr(leaf_bb), ptr(leaf_info),
                                           n, ptr(self._p), ptr(self._v), st))
        return self._p, self._v


def pack_c4_blob_128(model):
    """AlphaZeroC4 with 128 trunk channels -> uint8 blob in the layout of csrc/az_net128.hip."""
    if model.block_channels != 128 or tuple(model.shape_in) != (2, 6, 7):
        raise ValueError('az_net128 is for Connect Four nets with 128 trunk channels')
    dev = model._cbr[0].weight.device
    with torch.no_grad():
        w, b0 = _fold_bn(model._cbr[0].weight, model._cbr[1])
        stem = w.permute(2, 3, 1, 0).reshape(18, 128)                      # k = tap*2 + plane
        blk = model._tower[0]
        w1, b1 = _fold_bn(blk._c1.weight, blk._bn1)
        w2, b2 = _fold_bn(blk._c2.weight, blk._bn2)
        wts = torch.stack([w1.permute(2, 3, 0, 1).reshape(9, 128, 128),     # [tap][co][ci]
                           w2.permute(2, 3, 0, 1).reshape(9, 128, 128)])
        ph, vh = model._policy_head, model._value_head
        wpol, bpol = _fold_bn(ph._cbr[0].weight, ph._cbr[1], ph._cbr[0].bias)
        wval, bval = _fold_bn(vh._cbr[0].weight, vh._cbr[1])
        hw = torch.cat([wpol[:, :, 0, 0], wval[:, :, 0, 0]])               # [3][128]
        z = lambda k: torch.zeros(k, dtype=torch.float64, device=dev)
        f32 = [stem, torch.stack([b0, b1, b2]), hw, torch.cat([bpol[:2], bval[:1], z(1)]),
               ph._fc.weight.double(), torch.cat([ph._fc.bias.double(), z(2)]),
               vh._fc[0].weight.double().t().contiguous(), vh._fc[0].bias.double(), vh._fc[2].weight.double(),
               torch.cat([vh._fc[2].bias.double().reshape(1), z(3)])]
        # 16x16x32 A-fragment order: [conv][group of 32 co][chunk of 32 ci][tap][half][lane = 16 q + r][8], element
        # j of a lane is W[co = 32 group + 8 (r // 4) + 4 half + r % 4][ci = 32 chunk + 8 q + j]: the row
        # permutation inside a group makes an accumulator land in the next conv's operand layout
        r = torch.arange(16, device=dev)
        co = (32 * torch.arange(4, device=dev).view(4, 1, 1) + 4 * torch.arange(2, device=dev).view(1, 2, 1) +
              (8 * (r // 4) + r % 4).view(1, 1, 16))                          # [group][half][r]
        ci = torch.arange(128, device=dev).view(4, 4, 8)                       # [chunk][q][j]
        wts = wts[:, :, co][:, :, :, :, :, ci]                                 # [conv][tap][group][half][r][chunk][q][j]
        wts = wts.permute(0, 2, 5, 1, 3, 6, 4, 7)                              # [conv][group][chunk][tap][half][q][r][j]
        blob = torch.cat([wts.float().to(torch.bfloat16).contiguous().view(torch.uint8).reshape(-1)] +
                         [t.reshape(-1).float().contiguous().view(torch.uint8) for t in f32])
    assert blob.numel() == _lib.load().az_c4net128_blob_bytes(), blob.numel()
    return blob


class C4Net128:
    """Leaf evaluator for 128-channel Connect Four nets: net(leaf_bb, leaf_info) -> (p, v)."""

    def __init__(self, model, device=None, max_leaves=16384):
        self.lib = _lib.load()
        self.device = _device(device if device is not None else next(model.parameters()).device)
        self.num_blocks = model.num_blocks
        self.blob = pack_c4_blob_128(model).to(self.device)
        self.max_leaves = max_leaves
        nbytes = self.lib.az_c4net128_workspace_bytes(max_leaves)
        self.workspace = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)   # zero rows stay zero
        self._p = self._v = None

    def load_module(self, model):
        self.blob.copy_(pack_c4_blob_128(model))

    def __call__(self, leaf_bb, leaf_info=None):
        n = leaf_bb.shape[0]
        assert n <= self.max_leaves
        if self._p is None or self._p.shape[0] != n:
            self._p = torch.empty(n, 7, dtype=torch.float32, device=self.device)
            self._v = torch.empty(n, dtype=torch.float32, device=self.device)
        _launch(self.device, lambda st: self.lib.az_c4net128_bf16_forward(ptr(self.blob), self.num_blocks, ptr(leaf_bb), ptr(leaf_info), n,
                                                ptr(self.workspace), ptr(self._p), ptr(self._v), st))
        return self._p, self._v


TTT_WIDTHS = (4, 8, 16)      # trunk channels of the tic-tac-toe kernel (csrc/az_net_ttt.hip)


def pack_ttt_blob_f32(model):
    """AlphaZero module with shape_in (2,3,3), shape_out (9,) -> float32 tensor in the layout of
    csrc/az_net_ttt.hip (BN folded in float64, then rounded once to float32)."""
    C_ = model.block_channels
    if tuple(model.shape_in) != (2, 3, 3) or tuple(model.shape_out) != (9,) or C_ not in TTT_WIDTHS:
        raise ValueError('tic-tac-toe kernel: shape_in (2,3,3), shape_out (9,), 4, 8 or 16 trunk channels')
    dev = model._cbr[0].weight.device
    z = lambda *shape: torch.zeros(*shape, dtype=torch.float64, device=dev)

    def taps(w):                                   # [co][ci][3][3] -> [ci][co][12]
        out = z(w.shape[1], w.shape[0], 12)
        out[:, :, :9] = w.permute(1, 0, 2, 3).reshape(w.shape[1], w.shape[0], 9)
        return out

    with torch.no_grad():
        w, b0 = _fold_bn(model._cbr[0].weight, model._cbr[1])
        stem = taps(w)
        c1, c2, b1, b2 = z(C_, C_, 12), z(C_, C_, 12), z(C_), z(C_)
        if len(model._tower):
            blk = model._tower[0]
            w, b1 = _fold_bn(blk._c1.weight, blk._bn1)
            c1 = taps(w)
            w, b2 = _fold_bn(blk._c2.weight, blk._bn2)
            c2 = taps(w)
        ph, vh = model._policy_head, model._value_head
        wpol, bpol = _fold_bn(ph._cbr[0].weight, ph._cbr[1], ph._cbr[0].bias)
        wval, bval = _fold_bn(vh._cbr[0].weight, vh._cbr[1])
        hw = torch.cat([wpol[:, :, 0, 0], wval[:, :, 0, 0]])                      # [3][C]
        hb = torch.cat([bpol, bval, z(1)])
        pfc = z(9, 20)
        pfc[:, :18] = ph._fc.weight.double()
        pb = torch.cat([ph._fc.bias.double(), z(3)])
        v1 = z(256, 12)
        v1[:, :9] = vh._fc[0].weight.double()
        parts = [stem, c1, c2, torch.stack([b0, b1, b2]), hw, hb, pfc, pb, v1, vh._fc[0].bias.double(),
                 vh._fc[2].weight.double(), torch.cat([vh._fc[2].bias.double().reshape(1), z(3)])]
        blob = torch.cat([t.reshape(-1) for t in parts]).float().contiguous()
    assert blob.numel() == _lib.load().az_tttnet_f32_blob_floats(C_), blob.numel()
    return blob


class TTTNetF32:
    """Tic-tac-toe evaluator: net(leaf_bb, leaf_info) -> (p float32[n,9], v float32[n]); with
    leaf_info only the rows that need the network are evaluated and written."""

    def __init__(self, model, device=None):
        self.lib = _lib.load()
        self.device = _device(device if device is not None else next(model.parameters()).device)
        self.num_blocks, self.channels = model.num_blocks, model.block_channels
        self.blob = pack_ttt_blob_f32(model).to(self.device)
        self._p = self._v = None

    @classmethod
    def supports(cls, model):
        return (getattr(model, 'shape_in', None) == (2, 3, 3) and getattr(model, 'shape_out', None) == (9,)
                and getattr(model, 'block_channels', None) in TTT_WIDTHS)

    def load_module(self, model):
        self.blob.copy_(pack_ttt_blob_f32(model))

    def __call__(self, leaf_bb, leaf_info=None):
        n = leaf_bb.shape[0]
        if self._p is None or self._p.shape[0] != n:
            self._p = torch.zeros(n, 9, dtype=torch.float32, device=self.device)
            self._v = torch.zeros(n, dtype=torch.float32, device=self.device)
        _launch(self.device, lambda st: self.lib.az_tttnet_f32_forward(ptr(self.blob), self.channels, self.num_blocks, ptr(leaf_bb),
                                             ptr(leaf_info), n, ptr(self._p), ptr(self._v), st))
        return self._p, self._v
