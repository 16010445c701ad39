#!/usr/bin/env python3
"""Phase split of the one-launch-per-move kernel (tuning aid).  Needs a library built with the cycle
counters compiled in:
    make -C alphazero_amd/csrc -j6 OUT=../variants/timing.so EXTRA=-DAZ_EXP_TIMING ALLOW_SCRATCH=1
    AZHIP_LIBRARY=alphazero_amd/variants/timing.so python tools/exp_fused_timing.py
and prints, per move, how the waves' cycles divide into network tasks, search steps and idling, and how many
tasks had two leaves."""
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
last_boards = None
for i in range(12):
    bb, info = eng.begin_move(); p, _ = net(bb, info); eng.apply_root(p)
    eng.lib.az_debug_counters(eng.h, out)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); eng.move_fused(net); ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1)
    if hasattr(eng.lib, 'az_debug_workgroups') and i >= 2:
        import numpy as np
        wg = (C.c_ulonglong * 2048)()
        eng.lib.az_debug_workgroups.argtypes = [C.c_void_p]
        eng.lib.az_debug_workgroups(wg)
        a = np.frombuffer(wg, dtype=np.uint64).reshape(1024, 2)[:256].astype(np.float64)
        t, b = a[:, 0], a[:, 1]
        if last_boards is not None:
            slope, icpt = np.polyfit(b, t, 1)
            resid = t - (slope * b + icpt)
            print(f'   workgroups: cycles mean {t.mean():.3e} max/mean {t.max() / t.mean():.4f}; boards mean {b.mean():.0f} max/mean {b.max() / b.mean():.4f} '
                  f'std/mean {b.std() / b.mean():.4f}; corr(cycles, boards) {np.corrcoef(t, b)[0, 1]:.3f}, {slope:.0f} cycles per board, '
                  f'residual std/mean {resid.std() / t.mean():.4f} max/mean {resid.max() / t.mean():.4f}')
        last_boards = b
    eng.choose_and_advance(); eng.drain_finished()
    eng.lib.az_debug_counters(eng.h, out)
    n = out[3]
    d = out[4:12]
    tot = (out[0] + out[1] + out[2] + d[5] + d[6]) / n
    print(f'move {i}: launch {ms:.2f} ms; per wave: total {tot:.0f} cycles ({tot / (ms * 1e3):.0f} MHz), slowest {d[4]}; '
          f'towers {out[0] / n / tot:.1%} heads {d[5] / n / tot:.1%} hand-over {d[6] / n / tot:.1%} search {out[1] / n / tot:.1%} idle {out[2] / n / tot:.1%}')
    print(f'   tasks: {d[0]} pairs ({out[0] / max(d[0] + d[1], 1):.0f} + {d[5] / max(d[0] + d[1], 1):.0f} cycles per task incl. singles), {d[1]} singles; '
          f'head batches {d[7]} with {(2 * d[0] + d[1]) / max(d[7], 1):.1f} boards each, {d[5] / max(d[7], 1):.0f} cycles per batch; search steps {d[2]} with {d[3] / max(d[2], 1):.2f} games each, {out[1] / max(d[2], 1):.0f} cycles per step')

ww = (C.c_ulonglong * 10)()
if hasattr(eng.lib, 'az_debug_waits'):
    eng.lib.az_debug_waits.argtypes = [C.c_void_p]
    eng.lib.az_debug_waits(ww)
    print('mean cycles a game waits in the leaf queue (search step done -> network task starts) / plane queue (task done -> '
          'head batch starts) / ready queue (heads or search step done -> next search step starts): '
          f'{ww[0] / max(ww[3], 1):.0f} / {ww[1] / max(ww[4], 1):.0f} / {ww[2] / max(ww[5], 1):.0f}  (waits counted: {ww[3]}, {ww[4]}, {ww[5]})')
    print(f'two-board tasks with no other task on their SIMD when they began and when they ended: {ww[7]}, {ww[6] / max(ww[7], 1):.0f} cycles each; '
          f'with another task there at both ends: {ww[9]}, {ww[8] / max(ww[9], 1):.0f} cycles each')
dd = (C.c_ulonglong * 8)()
if hasattr(eng.lib, 'az_debug_done'):
    eng.lib.az_debug_done.argtypes = [C.c_void_p]
    eng.lib.az_debug_done(dd)
    n = max(dd[4], 1)
    print('cycles from the start of a workgroup until its first game / 50 % / 75 % / 90 % / all of its games are done '
          f'(mean over workgroups and moves): {dd[5] / n:.0f} / {dd[0] / n:.0f} / {dd[1] / n:.0f} / {dd[2] / n:.0f} / {dd[3] / n:.0f}')
