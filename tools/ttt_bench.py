#!/usr/bin/env python3
"""Tic-tac-toe self-play (alphazero/config/tictactoe.yaml: 3x16 net, 32 simulations per move,
exploration_depth 3) on one GPU: positions/s with the HIP network kernel and with the torch module."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd.config import AlphaZeroConfig       # noqa: E402
from alphazero_amd.selfplay import SelfPlayWorker      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--games', type=int, default=8192)
ap.add_argument('--moves', type=int, default=30)
args = ap.parse_args()


class Server:
    def __init__(self, state):
        self.state = state

    def latest(self, model):
        model.load_state_dict(self.state)
        return 1


class Sink:
    def add_game(self, game, version):
        pass


cfg = AlphaZeroConfig(conf={
    'game': {'module': 'alphazero.games.tictactoe', 'name': 'TicTacToe'},
    'model': {'module': 'alphazero.models', 'name': 'AlphaZero',
              'kwargs': {'shape_in': [2, 3, 3], 'shape_out': [9], 'num_blocks': 3, 'block_channels': 16}},
    'selfplay': {'num_expansions_per_sim': 32, 'exploration_depth': 3, 'reuse_search_tree': True}})
torch.manual_seed(0)
state = cfg.Model().state_dict()
for impl in ('hip', 'torch'):
    w = SelfPlayWorker(cfg, Server(state), Sink(), 'cuda:0', num_games=args.games, seed=0)
    if impl == 'torch':
        w._fused = None
    w.refresh_model()
    for _ in range(5):
        w.play_move(False)
    w.engine.drain_finished_bytes()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.moves):
        w.play_move(False)
        w.engine.drain_finished_bytes()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{impl}: {args.games * args.moves / dt:,.0f} positions/s, {1e3 * dt / args.moves:.2f} ms per move '
          f'({args.games} games x 32 simulations)')
