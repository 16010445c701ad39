#!/usr/bin/env python3
"""Leaf-eval benchmark of the 10x128 network: 16384 leaves per batch (BASELINE config 4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alphazero_amd import engine as E
from alphazero_amd.models import AlphaZeroC4
from alphazero_amd.net import C4Net128
n, blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 10
rng = np.random.RandomState(0)
hists = []
for _ in range(n):                     # uniform random playouts of depth U[0,30]
    h, cnt = [], [0] * 7
    for _ in range(rng.randint(0, 31)):
        c = int(rng.randint(0, 7))
        if cnt[c] < 6:
            h.append(c); cnt[c] += 1
    hists.append(h)
bb, nm = E.positions_from_history(1, hists)
torch.manual_seed(0)
net = C4Net128(AlphaZeroC4(blocks, 128).eval(), 'cuda', max_leaves=n)
for _ in range(3): net(bb)
torch.cuda.synchronize()
ts = []
for r in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): net(bb)
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 5)
ms = sorted(ts)[len(ts) // 2]
flops = n * (2 * blocks * 2 * 9 * 128 * 128 * 42 + 2 * 9 * 2 * 128 * 42)
print(f'10x128, {n} leaves: {ms:.3f} ms/batch, {n / ms * 1e3:.0f} leaves/s, {flops / ms / 1e9:.1f} TFLOP/s '
      f'({flops / ms / 1e9 / 2500 * 100:.1f}% of 2.5 PF)')
