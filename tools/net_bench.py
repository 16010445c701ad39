#!/usr/bin/env python3
"""Micro-benchmark of the fused network kernel alone: n random positions, `iters` launches.
   python tools/net_bench.py [--n 8192] [--blocks 6] [--ch 32] [--iters 50]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from alphazero_amd import engine as E
from alphazero_amd.models import AlphaZeroC4
from alphazero_amd.net import FusedC4Net

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=str, default='8192')
ap.add_argument('--blocks', type=str, default='6')
ap.add_argument('--ch', type=int, default=32)
ap.add_argument('--iters', type=int, default=50)
ap.add_argument('--active', type=float, default=0.583, help='fraction of leaves that need the net in the sparse case')
args = ap.parse_args()

rng = np.random.RandomState(0)
for n_, blocks in [(int(a), int(b)) for a in args.n.split(',') for b in args.blocks.split(',')]:
    args.n = n_
    hists = [list(rng.randint(0, 7, size=rng.randint(0, 6))) for _ in range(args.n)]
    bb, nm = E.positions_from_history(1, hists)
    legal, _, _ = E.game_query(1, bb, nm)
    info = ((legal << 16) | 1).to(torch.int32)
    torch.manual_seed(0)
    net = FusedC4Net.from_module(AlphaZeroC4(blocks, args.ch).eval(), 'cuda')
    flops = args.n * (2 * blocks * 2 * 9 * args.ch * args.ch * 42 + 2 * 9 * 2 * args.ch * 42)
    # interleaved rounds in one process: dense (no packing scan) vs packed with every leaf active
    act = torch.from_numpy(rng.rand(args.n) < args.active).cuda()
    sparse = torch.where(act, info, torch.zeros_like(info))
    nact = int(act.sum())
    res = {'dense': [], 'packed': [], 'sparse': []}
    for rnd in range(5):
        for name, inf in (('dense', None), ('packed', info), ('sparse', sparse)):
            for _ in range(3):
                net(bb, inf)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.iters):
                net(bb, inf)
            b.record()
            torch.cuda.synchronize()
            res[name].append(a.elapsed_time(b) * 1e3 / args.iters)
    for name, v in res.items():
        us = sorted(v)[len(v) // 2]
        if name == 'sparse':
            print(f'blocks={blocks} ch={args.ch} n={args.n} sparse ({nact} active): median {us:.1f} us (min {min(v):.1f}), '
                  f'{flops * nact / args.n / us / 1e6:.1f} TFLOP/s')
            continue
        print(f'blocks={blocks} ch={args.ch} n={args.n} {name}: median {us:.1f} us (min {min(v):.1f}), '
              f'{flops / us / 1e6:.1f} TFLOP/s')
