"""Training losses with the reference's signatures (alphazero/loss.py:6-63)."""
import torch
import torch.nn.functional as F
from torch import nn


def _masked_log_softmax(logits, mask):
    return F.log_softmax(logits.masked_fill(~mask, float('-inf')), dim=1)


def masked_cross_entropy(input, target, mask):
    """-sum_a target * log_softmax(input over mask); Nx1.  Terms with target == 0 or outside
    the mask contribute exactly 0 (no 0 * -inf)."""
    log_q = _masked_log_softmax(input, mask)
    log_q = torch.where(mask & (target != 0), log_q, torch.zeros_like(log_q))
    return -(target * log_q).sum(dim=1, keepdim=True)


def masked_kl_div(input, target, mask):
    """KL(target || softmax(input over mask)); Nx1."""
    log_q = _masked_log_softmax(input, mask)
    keep = mask & (target != 0)
    log_q = torch.where(keep, log_q, torch.zeros_like(log_q))
    log_p = torch.where(keep, torch.log(torch.where(keep, target, torch.ones_like(target))),
                        torch.zeros_like(target))
    return (target * (log_p - log_q)).sum(dim=1, keepdim=True)


class _HipLoss(torch.autograd.Function):
    """AlphaZeroLoss and its gradients in one launch of az_loss_fwd_bwd (csrc/az_train.hip)."""

    @staticmethod
    def forward(ctx, p_hat, v_hat, p, z, p_valid):
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        B, A = p_hat.shape
        dev = p_hat.device
        logits = p_hat.detach().contiguous().float()
        v = v_hat.detach().reshape(-1).contiguous().float()
        target = p.detach().contiguous().float()
        zz = z.detach().reshape(-1).contiguous().float()
        valid = p_valid.detach().contiguous().to(torch.uint8)
        out = torch.empty(3, dtype=torch.float32, device=dev)
        dlogits = torch.empty(B, A, dtype=torch.float32, device=dev)
        dv = torch.empty(B, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.az_loss_fwd_bwd(A, _lib.ptr(logits), _lib.ptr(v), _lib.ptr(target), _lib.ptr(zz),
                                           _lib.ptr(valid), B, _lib.ptr(out), _lib.ptr(dlogits), _lib.ptr(dv), st))
        ctx.save_for_backward(dlogits, dv)
        ctx.v_shape = v_hat.shape
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_policy, g_value, _g_dev):
        dlogits, dv = ctx.saved_tensors
        return dlogits * g_policy, (dv * g_value).reshape(ctx.v_shape), None, None, None


class AlphaZeroLoss(nn.Module):
    """alphazero/loss.py:50-63.  On the GPU both losses and their gradients come from the hand-written kernel
    (no fallback there: a missing library raises); CPU tensors take the torch statement below, which is the
    one checked bit for bit against the reference's recorded training scalars."""

    def forward(self, p, z, p_hat, v_hat, p_valid):
        if p_hat.is_cuda:
            policy_loss, value_loss, dev = _HipLoss.apply(p_hat, v_hat, p, z, p_valid)
            self.target_deviation = dev.detach()       # max |sum(target) - 1| of this batch, on the device
            if not torch.cuda.is_current_stream_capturing():
                assert float(self.target_deviation) < 1e-6, f'Target policy must be a valid distribution: {p}'
            # (while a HIP graph is being recorded nothing may be read back: whoever replays the graph checks
            # target_deviation afterwards)
            return policy_loss, value_loss
        value_loss = ((z - v_hat) ** 2).mean()
        assert torch.all(torch.abs(p.sum(dim=1) - 1.) < 1e-6), \
            f'Target policy must be a valid distribution: {p}'
        policy_loss = masked_cross_entropy(p_hat, p, p_valid).mean()
        return policy_loss, value_loss
