#!/usr/bin/env python3
"""Distribution of select depth across games at steady state (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alphazero_amd import engine as E
G, S = 8192, 800
eng = E.SelfPlayEngine(1, G, S, exploration_depth=4, reuse_search_tree=True, seed=0)
for mv in range(24):
    eng.begin_move_uniform(); eng.run_sims_uniform(S); eng.choose_and_advance()
    eng.drain_finished()
eng.begin_move_uniform()
eng.run_sims_uniform(600)
bb, info = eng.begin_move()
root_nm = ((info >> 8) & 0xFF).cpu().numpy()
p = torch.full((G, 7), 1 / 7, device='cuda'); v = torch.zeros(G, device='cuda')
for it in range(5):
    bb, info = eng.sim_step(p, v) if it else eng.select()
    i = info.cpu().numpy()
    d = ((i >> 8) & 0xFF) - root_nm
    kind = i & 0xFF
    print('iter', it, 'mean depth %.2f' % d.mean(), 'max', d.max(), 'p99', np.percentile(d, 99), 'p90', np.percentile(d, 90),
          'kinds', np.bincount(kind, minlength=4), 'hist', np.bincount(d, minlength=20)[:24])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
