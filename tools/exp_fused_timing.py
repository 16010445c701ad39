#!/usr/bin/env python3
"""Phase split of the one-launch-per-move kernel (tuning aid).  Needs a library built with the cycle
counters compiled in:
    make -C alphazero_amd/csrc HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DAZ_EXP_TIMING"
and prints, per move, the shader cycles a workgroup spends per simulation in the search phase, in
ranking its leaves and in the network phase.  Rebuild without the flag afterwards."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_amd import engine as E
from alphazero_amd.models import AlphaZeroC4
from alphazero_amd.net import FusedC4Net
G, S = int(os.environ.get('AZ_G', 8192)), int(os.environ.get('AZ_S', 800))
CAP = int(os.environ.get('AZ_CAP', 0))      # nodes per game pool (0 = the engine's bound)
eng = E.SelfPlayEngine(E._lib.GAME_C4, G, S, exploration_depth=4, reuse_search_tree=True, seed=0, device=0, node_capacity=CAP)
torch.manual_seed(0)
net = FusedC4Net.from_module(AlphaZeroC4(6, 32).eval().cuda(), 'cuda')
for i in range(160):
    eng.begin_move_uniform(); eng.run_sims_uniform(S); eng.choose_and_advance()
    if i % 2: eng.drain_finished()
out = (C.c_ulonglong * 12)()
for i in range(12):
    bb, info = eng.begin_move(); p, _ = net(bb, info); eng.apply_root(p)
    eng.lib.az_debug_counters(eng.h, out)
    eng.move_fused(net)
    eng.choose_and_advance(); eng.drain_finished()
    eng.lib.az_debug_counters(eng.h, out)
    n = out[3]
    print(f'move {i}: per WG per sim, shader cycles: tree {out[0]/n/S:.0f}  list {out[1]/n/S:.0f}  net {out[2]/n/S:.0f}  (WGs {n})')
    d = out[4:12]
    tasks, busy, idle = max(d[5], 1), max(d[6], 1), max(d[7], 1)
    print(f'   network phase, per wave with work: stem {d[0]/tasks:.0f}/task  tower {d[1]/tasks:.0f}/task  head convs {d[2]/busy:.0f}  '
          f'wait {d[3]/busy:.0f}  fc+softmax {d[4]/(busy+idle):.0f};  tasks per WG-sim {tasks/n/S:.2f}, waves with work {busy/n/S:.2f}, '
          f'idle waves wait {d[7]/idle:.0f}')
