#!/usr/bin/env python3
"""How the per-simulation search kernel's duration depends on the deepest select of the launch and on
the number of games (tuning aid):  python tools/tree_latency_probe.py [G ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alphazero_amd import engine as E

S = 800
for G in [int(a) for a in sys.argv[1:]] or [8192, 512]:
    eng = E.SelfPlayEngine(1, G, S, exploration_depth=4, reuse_search_tree=True, seed=0)
    for mv in range(20):
        eng.begin_move_uniform(); eng.run_sims_uniform(S); eng.choose_and_advance()
        eng.drain_finished_bytes()
    eng.begin_move_uniform()
    eng.run_sims_uniform(300)
    bb, info = eng.begin_move()
    root_nm = ((info >> 8) & 0xFF).cpu().numpy()
    p = torch.full((G, 7), 1 / 7, device='cuda'); v = torch.zeros(G, device='cuda')
    bb, info = eng.select()
    rows = []
    for it in range(300):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        bb, info = eng.sim_step(p, v)
        b.record()
        torch.cuda.synchronize()
        i = info.cpu().numpy()
        d = ((i >> 8) & 0xFF) - root_nm
        rows.append((d.max(), d.mean(), a.elapsed_time(b) * 1e3))
    r = np.array(rows)
    A = np.stack([np.ones(len(r)), r[:, 0]], 1)
    coef, *_ = np.linalg.lstsq(A, r[:, 2], rcond=None)
    print(f'G={G}: mean time {r[:,2].mean():.1f} us, max depth mean {r[:,0].mean():.1f} (min {r[:,0].min():.0f}, max {r[:,0].max():.0f}), '
          f'mean depth {r[:,1].mean():.2f}; fit time = {coef[0]:.1f} + {coef[1]:.2f} * max_depth us')
    for dmax in sorted(set(r[:, 0].astype(int))):
        sel = r[r[:, 0] == dmax]
        print(f'   max depth {dmax}: {len(sel)} launches, {sel[:,2].mean():.1f} us')
    del eng
