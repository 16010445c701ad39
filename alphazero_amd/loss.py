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


class AlphaZeroLoss(nn.Module):
    def forward(self, p, z, p_hat, v_hat, p_valid):
        value_loss = ((z - v_hat) ** 2).mean()
        assert torch.all(torch.abs(p.sum(dim=1) - 1.) < 1e-6), \
            f'Target policy must be a valid distribution: {p}'
        policy_loss = masked_cross_entropy(p_hat, p, p_valid).mean()
        return policy_loss, value_loss
