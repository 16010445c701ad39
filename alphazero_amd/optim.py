"""FlatSGD: torch.optim.SGD as the reference configures it (alphazero/training.py:27-29: momentum, weight
decay) with the whole update in ONE launch of az_sgd_momentum (csrc/az_train.hip).

All parameters are moved into one flat fp32 buffer (each parameter becomes a view of it, so the module, its
state_dict and everything that reads the weights keep working), gradients accumulate into a second flat buffer
through persistent `.grad` views, and the momentum is a third.  A step is then a single kernel over n elements
instead of three small launches per tensor.  It is a torch Optimizer, so MultiStepLR drives its learning rate
exactly as it drives SGD's (training.py:30-31)."""
import ctypes as C

import torch

from . import _lib


class FlatSGD(torch.optim.Optimizer):
    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0):
        params = [p for p in params]
        seen, uniq = set(), []
        for p in params:                                   # the tower repeats one block: each tensor once
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        super().__init__(uniq, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        dev = uniq[0].device
        if dev.type != 'cuda':
            raise _lib.AzError('FlatSGD runs on the GPU (az_sgd_momentum); use torch.optim.SGD for CPU tensors')
        assert all(p.device == dev and p.dtype == torch.float32 for p in uniq)
        self.lib = _lib.load()
        self.device = dev
        n = sum(p.numel() for p in uniq)
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in uniq:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_p[off:off + k].view_as(p)
                p.grad = self.flat_g[off:off + k].view_as(p)
                off += k
        self._steps = 0
        self.withheld = torch.zeros(1, dtype=torch.int32, device=dev)   # recorded steps whose update the guard refused
        self._lr_dev = None                                # learning rate in device memory, for recorded steps

    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()                                # the .grad views stay in place

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(self.lib.az_sgd_momentum(_lib.ptr(self.flat_p), _lib.ptr(self.flat_g), _lib.ptr(self.flat_m),
                                                self.flat_p.numel(), float(g['lr']), float(g['momentum']),
                                                float(g['weight_decay']), int(self._steps == 0), st))
        self._steps += 1

    def set_recorded_lr(self):
        """Puts the current learning rate where step_recorded's kernel reads it (a copy only when it changed)."""
        lr = float(self.param_groups[0]['lr'])
        if self._lr_dev is None:
            self._lr_dev = torch.empty(1, dtype=torch.float32, device=self.device)
            self._lr_host = None
        if self._lr_host != lr:
            self._lr_dev.fill_(lr)
            self._lr_host = lr

    @torch.no_grad()
    def step_recorded(self, guard=None, guard_limit=1e-6):
        """The same update with the learning rate read from device memory at run time (az_sgd_momentum_lr_dev):
        the launch can be captured in a HIP graph and replayed while a scheduler changes the rate.  Call
        set_recorded_lr() (outside the capture) before every run.
        guard: a one-element float32 device tensor written earlier in the same recorded sequence; the update is
        applied only while guard < guard_limit (az_sgd_momentum_guarded), otherwise nothing changes and
        self.withheld (device int32[1]) counts the step."""
        g = self.param_groups[0]
        assert self._lr_dev is not None, 'set_recorded_lr() first'
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            if guard is None:
                _lib.check(self.lib.az_sgd_momentum_lr_dev(_lib.ptr(self.flat_p), _lib.ptr(self.flat_g), _lib.ptr(self.flat_m),
                                                           self.flat_p.numel(), _lib.ptr(self._lr_dev), float(g['momentum']),
                                                           float(g['weight_decay']), st))
            else:
                assert guard.dtype == torch.float32 and guard.numel() == 1 and guard.device == self.device
                _lib.check(self.lib.az_sgd_momentum_guarded(_lib.ptr(self.flat_p), _lib.ptr(self.flat_g), _lib.ptr(self.flat_m),
                                                            self.flat_p.numel(), _lib.ptr(self._lr_dev), float(g['momentum']),
                                                            float(g['weight_decay']), _lib.ptr(guard), float(guard_limit),
                                                            _lib.ptr(self.withheld), st))
        self._steps += 1
