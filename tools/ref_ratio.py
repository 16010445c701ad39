#!/usr/bin/env python3
"""Speed of the CPU oracle relative to the reference's own Python, one core each, same workload
(Connect4, 6x32 net, 800 simulations per move, tree reuse).  Runs only where the reference is
importable (the build container):

    PYTHONPATH=/root/reference python tools/ref_ratio.py [seconds]

and writes profiles/r01_ref_vs_oracle_cpu.json, which bench.py quotes next to its cpu_baseline so that
GPU-vs-CPU ratios can be restated against the reference's Python speed."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
os.environ['OMP_NUM_THREADS'] = '1'

import numpy as np
import torch

torch.set_num_threads(1)
S = 800
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0

# ---- the reference, driven the way its train.py drives it (selfplay.py:12-42), one move at a time
import alphazero.mcts as ref_mcts
from alphazero.games.connectfour import AlphaZeroC4, ConnectFour

torch.manual_seed(0)
model = AlphaZeroC4(num_blocks=6, channels_per_block=32).eval()
np.random.seed(0)
t0 = time.time()
ref_moves = 0
while time.time() - t0 < seconds:
    game = ConnectFour()
    tree = ref_mcts.MCTreeNode()
    while not game.terminal and time.time() - t0 < seconds:
        ref_mcts.run_mcts(game, tree, model, S)
        action_index = tree.sample_action() if len(game.history) < 4 else tree.greedy_action()
        game.take_action(game.valid_actions[action_index])
        tree = tree.traverse(action_index)
        tree.kill_siblings()
        ref_moves += 1
ref_wall = time.time() - t0

# ---- the oracle on the same workload (bench.py's CPU baseline worker, one worker)
import bench
pos, sims, wall = bench._cpu_worker('c4_6x32', S, seconds, 0)

out = {'workload': 'Connect4 self-play, 6x32 net (torch CPU, 1 thread), 800 sims/move, reuse, one core',
       'reference_positions_per_s_per_core': ref_moves / ref_wall,
       'oracle_positions_per_s_per_core': pos / wall,
       'oracle_over_reference': (pos / wall) / (ref_moves / ref_wall),
       'reference_moves': ref_moves, 'oracle_moves': pos, 'seconds_each': seconds,
       'where': 'build container (no GPU), measured by tools/ref_ratio.py'}
path = os.path.join(ROOT, 'profiles', 'r01_ref_vs_oracle_cpu.json')
json.dump(out, open(path, 'w'), indent=1)
print(json.dumps(out, indent=1))
