"""Hand-written forward + loss + backward of the training step for AlphaZeroC4 with 32 trunk channels
(csrc/az_train_c4.hip; replaces model(x, p_valid), AlphaZeroLoss and total.backward() of
alphazero/training.py:57-71).  `C4TrainStep(model)` binds the module's parameter, gradient and batch-norm
buffers (they must stay where they are: FlatSGD's flat views do) and `run(x, p, z, p_valid)` leaves the
gradients in the parameters' .grad tensors, the updated running statistics in the module and returns the
device tensor {policy loss, value loss, max |sum(target) - 1|}."""
import ctypes as C

import torch

from . import _lib

_FP = C.POINTER(C.c_float)


class _Bn(C.Structure):
    _fields_ = [('weight', C.c_void_p), ('bias', C.c_void_p), ('running_mean', C.c_void_p), ('running_var', C.c_void_p),
                ('num_batches_tracked', C.c_void_p), ('weight_grad', C.c_void_p), ('bias_grad', C.c_void_p)]


class TrainNet(C.Structure):
    """az_c4_train_net of include/azhip.h."""
    _fields_ = [('num_blocks', C.c_int32), ('reserved', C.c_int32), ('bn_momentum', C.c_float), ('bn_eps', C.c_float),
                ('stem_w', C.c_void_p), ('stem_w_grad', C.c_void_p), ('stem_bn', _Bn),
                ('c1_w', C.c_void_p), ('c1_w_grad', C.c_void_p), ('bn1', _Bn),
                ('c2_w', C.c_void_p), ('c2_w_grad', C.c_void_p), ('bn2', _Bn),
                ('pol_conv_w', C.c_void_p), ('pol_conv_w_grad', C.c_void_p),
                ('pol_conv_b', C.c_void_p), ('pol_conv_b_grad', C.c_void_p), ('pol_bn', _Bn),
                ('pol_fc_w', C.c_void_p), ('pol_fc_w_grad', C.c_void_p),
                ('pol_fc_b', C.c_void_p), ('pol_fc_b_grad', C.c_void_p),
                ('val_conv_w', C.c_void_p), ('val_conv_w_grad', C.c_void_p), ('val_bn', _Bn),
                ('val_fc1_w', C.c_void_p), ('val_fc1_w_grad', C.c_void_p),
                ('val_fc1_b', C.c_void_p), ('val_fc1_b_grad', C.c_void_p),
                ('val_fc2_w', C.c_void_p), ('val_fc2_w_grad', C.c_void_p),
                ('val_fc2_b', C.c_void_p), ('val_fc2_b_grad', C.c_void_p)]


def param_struct(model):
    """az_c4_train_net with the module's parameters and batch-norm buffers and no gradient pointers: what the
    packing launches (az_c4net_pack_bf16, az_c4net_pack_hx) read.  A tower-less network leaves the block's members null."""
    m = model

    def bn(mod):
        return _Bn(mod.weight.data_ptr(), mod.bias.data_ptr(), mod.running_mean.data_ptr(), mod.running_var.data_ptr(),
                   mod.num_batches_tracked.data_ptr(), 0, 0)
    for t in list(m.parameters()) + list(m.buffers()):
        assert t.is_contiguous()
    n = TrainNet()
    n.num_blocks = m.num_blocks
    some_bn = m._cbr[1]
    n.bn_momentum, n.bn_eps = float(some_bn.momentum or 0.0), float(some_bn.eps)
    n.stem_w, n.stem_bn = m._cbr[0].weight.data_ptr(), bn(m._cbr[1])
    if m.num_blocks:
        blk = m._tower[0]
        n.c1_w, n.bn1 = blk._c1.weight.data_ptr(), bn(blk._bn1)
        n.c2_w, n.bn2 = blk._c2.weight.data_ptr(), bn(blk._bn2)
    ph, vh = m._policy_head, m._value_head
    n.pol_conv_w, n.pol_conv_b, n.pol_bn = ph._cbr[0].weight.data_ptr(), ph._cbr[0].bias.data_ptr(), bn(ph._cbr[1])
    n.pol_fc_w, n.pol_fc_b = ph._fc.weight.data_ptr(), ph._fc.bias.data_ptr()
    n.val_conv_w, n.val_bn = vh._cbr[0].weight.data_ptr(), bn(vh._cbr[1])
    n.val_fc1_w, n.val_fc1_b = vh._fc[0].weight.data_ptr(), vh._fc[0].bias.data_ptr()
    n.val_fc2_w, n.val_fc2_b = vh._fc[2].weight.data_ptr(), vh._fc[2].bias.data_ptr()
    return n


def supports(model):
    """True for the network these kernels are written for: AlphaZeroC4 (2x6x7 -> 6x7) with 32 trunk channels, default
    batch-norm settings, on a GPU."""
    from .models import AlphaZeroC4
    if not isinstance(model, AlphaZeroC4) or model.block_channels != 32:
        return False
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    if any(m.momentum is None or not m.affine or not m.track_running_stats for m in bns):
        return False
    if len({(m.momentum, m.eps) for m in bns}) != 1:
        return False
    return next(model.parameters()).is_cuda


class C4TrainStep:
    """`chain=True` runs the pass as a chain of launches, one step each (az_c4_train_fwd_bwd_chain); by default the
    library takes the one-launch form whenever the batch fits the device (az_c4_train_fwd_bwd).  Same bits either way."""

    def __init__(self, model, chain=False):
        if not supports(model):
            raise ValueError('C4TrainStep: AlphaZeroC4 with 32 trunk channels on a GPU')
        self.lib = _lib.load()
        self._entry = self.lib.az_c4_train_fwd_bwd_chain if chain else self.lib.az_c4_train_fwd_bwd
        self.model = model
        self.device = next(model.parameters()).device
        self._ws = None
        self._batch = 0
        self.out3 = torch.zeros(3, dtype=torch.float32, device=self.device)
        self.logits = self.v_hat = None
        self._bind()

    def _bind(self):
        """(Re)reads the tensors' addresses: parameters, their .grad (created here when missing) and buffers."""
        m = self.model

        def grad(p):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            assert p.is_contiguous() and p.grad.is_contiguous() and p.dtype == torch.float32
            return p.grad.data_ptr()

        def bn(mod):
            return _Bn(mod.weight.data_ptr(), mod.bias.data_ptr(), mod.running_mean.data_ptr(), mod.running_var.data_ptr(),
                       mod.num_batches_tracked.data_ptr(), grad(mod.weight), grad(mod.bias))
        n = TrainNet()
        n.num_blocks = m.num_blocks
        some_bn = m._cbr[1]
        n.bn_momentum, n.bn_eps = float(some_bn.momentum), float(some_bn.eps)
        n.stem_w, n.stem_w_grad, n.stem_bn = m._cbr[0].weight.data_ptr(), grad(m._cbr[0].weight), bn(m._cbr[1])
        if m.num_blocks:
            blk = m._tower[0]
            n.c1_w, n.c1_w_grad, n.bn1 = blk._c1.weight.data_ptr(), grad(blk._c1.weight), bn(blk._bn1)
            n.c2_w, n.c2_w_grad, n.bn2 = blk._c2.weight.data_ptr(), grad(blk._c2.weight), bn(blk._bn2)
        else:
            # no tower: the packing launch still reads two conv-sized tensors; give it zeros
            self._dummy = torch.zeros(2, 32 * 32 * 9 + 64, dtype=torch.float32, device=self.device)
            self._dummy_i = torch.zeros(2, dtype=torch.int64, device=self.device)
            d0, d1 = self._dummy[0], self._dummy[1]
            fake = _Bn(d0.data_ptr(), d0.data_ptr(), d1.data_ptr(), d1.data_ptr(), self._dummy_i.data_ptr(), d1.data_ptr(),
                       d1.data_ptr())
            n.c1_w = n.c2_w = d0.data_ptr()
            n.c1_w_grad = n.c2_w_grad = d1.data_ptr()
            n.bn1 = n.bn2 = fake
        ph, vh = m._policy_head, m._value_head
        n.pol_conv_w, n.pol_conv_w_grad = ph._cbr[0].weight.data_ptr(), grad(ph._cbr[0].weight)
        n.pol_conv_b, n.pol_conv_b_grad = ph._cbr[0].bias.data_ptr(), grad(ph._cbr[0].bias)
        n.pol_bn = bn(ph._cbr[1])
        n.pol_fc_w, n.pol_fc_w_grad = ph._fc.weight.data_ptr(), grad(ph._fc.weight)
        n.pol_fc_b, n.pol_fc_b_grad = ph._fc.bias.data_ptr(), grad(ph._fc.bias)
        n.val_conv_w, n.val_conv_w_grad, n.val_bn = vh._cbr[0].weight.data_ptr(), grad(vh._cbr[0].weight), bn(vh._cbr[1])
        n.val_fc1_w, n.val_fc1_w_grad = vh._fc[0].weight.data_ptr(), grad(vh._fc[0].weight)
        n.val_fc1_b, n.val_fc1_b_grad = vh._fc[0].bias.data_ptr(), grad(vh._fc[0].bias)
        n.val_fc2_w, n.val_fc2_w_grad = vh._fc[2].weight.data_ptr(), grad(vh._fc[2].weight)
        n.val_fc2_b, n.val_fc2_b_grad = vh._fc[2].bias.data_ptr(), grad(vh._fc[2].bias)
        self.net = n

    def one_launch_max_batch(self):
        """Largest batch the one-launch form takes on this device (one board per compute unit)."""
        n = C.c_int32()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.az_c4_train_one_launch_max_batch(self.model.num_blocks, C.byref(n)))
        return int(n.value)

    def run(self, x, p, z, p_valid):
        """x float32 [B,2,6,7], p float32 [B,7], z float32 [B] or [B,1], p_valid bool / uint8 [B,7] (all on the
        module's device, contiguous).  Returns self.out3; self.logits [B,7] and self.v_hat [B] hold the outputs."""
        B = x.shape[0]
        assert x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape[1:]) == (2, 6, 7)
        assert p.dtype == torch.float32 and p.is_contiguous() and tuple(p.shape) == (B, 7)
        assert z.dtype == torch.float32 and z.is_contiguous() and z.numel() == B
        assert p_valid.is_contiguous() and tuple(p_valid.shape) == (B, 7) and p_valid.element_size() == 1
        if B != self._batch:
            nbytes = self.lib.az_c4_train_workspace_bytes(B, self.model.num_blocks)
            self._ws = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
            self.logits = torch.zeros(B, 7, dtype=torch.float32, device=self.device)
            self.v_hat = torch.zeros(B, dtype=torch.float32, device=self.device)
            self._batch = B
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(self._entry(C.byref(self.net), _lib.ptr(x), _lib.ptr(p_valid), _lib.ptr(p), _lib.ptr(z),
                                                    B, _lib.ptr(self._ws), _lib.ptr(self.out3), _lib.ptr(self.logits),
                                                    _lib.ptr(self.v_hat), st))
        return self.out3
