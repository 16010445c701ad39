"""Policy/value networks with the reference's module tree and state_dict keys
(alphazero/models.py:6-95, alphazero/games/connectfour.py:47-72), so checkpoints
and configs written for the reference load unchanged.

The torch modules here define the weights and serve as the fp32 statement of the
network; the self-play engine evaluates leaves with the fused HIP network in
`alphazero_amd.net` when one is attached, else by calling these modules on the
rendered leaf batch.
"""
import math

import torch
from torch import nn


def _conv_bn_relu(c_in, c_out, k, bias=False):
    return nn.Sequential(nn.Conv2d(c_in, c_out, (k, k), 1, k // 2, bias=bias),
                         nn.BatchNorm2d(c_out), nn.ReLU(True))


class ResidualBlock(nn.Module):
    def __init__(self, channels=256):
        super().__init__()
        self._c1 = nn.Conv2d(channels, channels, (3, 3), 1, 1, bias=False)
        self._bn1 = nn.BatchNorm2d(channels)
        self._c2 = nn.Conv2d(channels, channels, (3, 3), 1, 1, bias=False)
        self._bn2 = nn.BatchNorm2d(channels)
        self._relu = nn.ReLU(True)

    def forward(self, x):
        y = self._relu(self._bn1(self._c1(x)))
        y = self._bn2(self._c2(y))
        return self._relu(y + x)


class PolicyHead(nn.Module):
    def __init__(self, shape_in, shape_out):
        super().__init__()
        cells = int(math.prod(shape_in[1:]))
        self._cbr = _conv_bn_relu(shape_in[0], 2, 1, bias=True)     # 1x1 conv WITH bias
        self._fc = nn.Linear(2 * cells, int(math.prod(shape_out)))
        self._shape_out = tuple(shape_out)
        nn.init.zeros_(self._fc.bias)

    def forward(self, x):
        y = self._cbr(x).flatten(1)
        return self._fc(y).view(-1, *self._shape_out)


class ValueHead(nn.Module):
    def __init__(self, shape_in):
        super().__init__()
        cells = int(math.prod(shape_in[1:]))
        self._cbr = _conv_bn_relu(shape_in[0], 1, 1, bias=False)
        self._fc = nn.Sequential(nn.Linear(cells, 256), nn.ReLU(True), nn.Linear(256, 1), nn.Tanh())
        nn.init.zeros_(self._fc[2].bias)

    def forward(self, x):
        return self._fc(self._cbr(x).flatten(1))


class AlphaZero(nn.Module):
    """Stem conv + tower + heads.  As in the reference (models.py:78) the tower is ONE
    residual block applied num_blocks times: `_tower.i.*` keys all alias the same tensors."""

    def __init__(self, shape_in, shape_out, num_blocks=39, block_channels=256):
        super().__init__()
        self.shape_in, self.shape_out = tuple(shape_in), tuple(shape_out)
        self.num_blocks, self.block_channels = num_blocks, block_channels
        self._cbr = _conv_bn_relu(shape_in[0], block_channels, 3)
        shared = ResidualBlock(channels=block_channels)
        self._tower = nn.Sequential(*([shared] * num_blocks))
        self._policy_head = PolicyHead((block_channels, *shape_in[1:]), shape_out)
        self._value_head = ValueHead((block_channels, *shape_in[1:]))

    def forward(self, x, p_valid):
        t = self._tower(self._cbr(x))
        return self._policy_head(t), self._value_head(t)


class AlphaZeroC4(AlphaZero):
    """Connect Four net: the trunk scores the 6x7 cells; a column's logit is the score of
    the cell a stone dropped there would land on (connectfour.py:54-72), gathered in one
    vectorised op instead of a Python loop over batch x columns."""

    def __init__(self, num_blocks, channels_per_block):
        super().__init__(shape_in=(2, 6, 7), shape_out=(6, 7), num_blocks=num_blocks,
                         block_channels=channels_per_block)

    def forward(self, x, p_valid):
        cells, v = super().forward(x, p_valid)                      # [B,6,7]
        height = x.sum(dim=(1, 2)).long()                           # stones per column [B,7]
        open_col = height < 6
        landing = height.clamp(max=5).unsqueeze(1)                  # [B,1,7]
        logits = cells.gather(1, landing).squeeze(1)
        return torch.where(open_col, logits, torch.zeros_like(logits)), v


class UniformModel(nn.Module):
    def forward(self, x, p_valid):
        B = p_valid.shape[0]
        return (torch.ones(p_valid.shape, dtype=torch.float32, device=p_valid.device),
                torch.zeros(B, 1, dtype=torch.float32, device=x.device))
