#!/usr/bin/env python3
"""How many of the leaves sent to the network are positions that were evaluated before (by any game,
earlier in the same move or in the previous moves)?  Upper bound on what an evaluation cache could save."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_amd import engine as E
from alphazero_amd.models import AlphaZeroC4
from alphazero_amd.net import FusedC4Net

G, S = 8192, 800
torch.manual_seed(0)
net = FusedC4Net.from_module(AlphaZeroC4(6, 32).cuda().eval(), 'cuda')
eng = E.SelfPlayEngine(1, G, S, exploration_depth=4, reuse_search_tree=True, seed=0)
for mv in range(160):
    eng.begin_move_uniform(); eng.run_sims_uniform(S); eng.choose_and_advance(); eng.drain_finished_bytes()
seen = torch.empty(0, dtype=torch.int64, device='cuda')      # sorted unique keys of everything evaluated so far
def key(bb):   # 2 x 49-bit boards -> one int64 hash (collisions negligible for counting)
    return (bb[:, 0] * 0x9E3779B97F4A7C15 + bb[:, 1] * 0xC2B2AE3D27D4EB4F) if False else (bb[:, 0] * 1000003) ^ (bb[:, 1] * 998244353 + (bb[:, 1] >> 7))
for mv in range(4):
    tot = hit_move = hit_prev = 0
    move_keys = []
    bb, info = eng.begin_move()
    p, _ = net(bb, info)
    eng.apply_root(p)
    bb, info = eng.select()
    for s in range(S):
        kind = info & 0xFF
        k = key(bb[kind == 1])
        move_keys.append(k)
        p, v = net(bb, info)
        if s < S - 1:
            bb, info = eng.sim_step(p, v)
        else:
            eng.expand_backup(p, v)
    eng.choose_and_advance(); eng.drain_finished_bytes()
    allk = torch.cat(move_keys)
    uniq = torch.unique(allk)
    in_prev = torch.isin(uniq, seen).sum().item() if seen.numel() else 0
    print(f'move {mv}: {allk.numel()} evaluations, {uniq.numel()} distinct positions ({100 * (1 - uniq.numel() / allk.numel()):.1f}% repeats '
          f'inside the move), {100 * in_prev / uniq.numel():.1f}% of the distinct ones were evaluated in earlier moves', flush=True)
    seen = torch.unique(torch.cat([seen, uniq]))
