#!/usr/bin/env python3
"""Golden vectors for the host-side formats next to the self-play path, produced by running the
reference itself (this container only).

Usage:  PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_host.py

  * alphazero.replay_buffer.SelfPlayDatabase.write_game   -> the rows of both SQLite tables for a
    few finished games (taken from selfplay_exact.json), blobs as hex
  * alphazero.replay_buffer.ReplayBufferDataset.__getitem__ -> every sample dict of that database
  * alphazero.games.tictactoe.TicTacToe.solve             -> (scores, v) for a set of positions
  * alphazero.training.TrainingWorker.process_batch       -> four SGD steps on the CPU: every logged
    scalar and the weights before and after
Only data is written (host.json); no reference source is copied.
"""
import json
import os
import sqlite3
import sys
import tempfile
import warnings

import numpy as np

from alphazero.games.connectfour import ConnectFour
from alphazero.games.tictactoe import TicTacToe
from alphazero.replay_buffer import ReplayBufferDataset

HERE = os.path.dirname(os.path.abspath(__file__))
GAMES = {'ttt': TicTacToe, 'c4': ConnectFour}


class _Cfg:
    def __init__(self, cls):
        self.Game = cls


def fhex(a):
    return np.ascontiguousarray(a, dtype='<f4').tobytes().hex()


def replay_case(game_name, cases, versions):
    cls = GAMES[game_name]
    path = os.path.join(tempfile.mkdtemp(), 'replay.sqlite')
    ds = ReplayBufferDataset(_Cfg(cls), path)
    games = []
    for case, ver in zip(cases, versions):
        g = cls()
        for mv in case['moves']:
            g.take_action(mv['action'], np.array(mv['N'], dtype=np.uint32))
        ds.add_game(g, ver)
        games.append({'history': list(g.history), 'visits': [list(map(int, mv['N'])) for mv in case['moves']],
                      'model_version': ver})
    con = sqlite3.connect(path)
    rows_g = [[r[0], r[1], r[2], bytes(r[3]).hex()] for r in
              con.execute('SELECT id, model_version, history, search_statistics FROM selfplay_games ORDER BY id')]
    rows_s = [list(r) for r in con.execute('SELECT id, game_id, move_index FROM game_states ORDER BY id')]
    schema = [r[0] for r in con.execute("SELECT sql FROM sqlite_master WHERE type='table' AND name IN "
                                        "('selfplay_games','game_states') ORDER BY name")]
    con.close()
    items = []
    for i in range(len(ds)):
        try:
            it = ds[i]
        except FileNotFoundError:
            # ConnectFour.solve (every tenth state) shells out to /usr/local/bin/c4solver, which this
            # container does not have: no reference output exists for those states
            items.append(None)
            continue
        items.append({'x': fhex(it['x']), 'z': fhex(it['z']), 'p': fhex(it['p']),
                      'p_valid': [int(b) for b in it['p_valid']], 'game_index': int(it['game_index']),
                      'move_index': int(it['move_index']), 'model_version': int(it['model_version']),
                      'solved': bool(it['solved']), 'v*': int(it['v*']), 'move_scores': fhex(it['move_scores'])})
    return {'game': game_name, 'games': games, 'selfplay_games': rows_g, 'game_states': rows_s, 'schema': schema,
            'items': items, 'num_games': ds.db.num_games(), 'num_states': ds.db.num_states(),
            'num_distinct_games': ds.db.num_distinct_games()}


def solve_cases():
    rng = np.random.RandomState(5)
    hists = [[]] + [[a] for a in range(9)] + [[a, b] for a in range(9) for b in range(9) if a != b]
    kat = [0, 2, 8, 4, 6, 7, 3]              # the line the reference's own solver test walks (test_games.py:70)
    hists += [kat[:i] for i in range(3, len(kat))]
    for _ in range(120):
        g = TicTacToe()
        n = rng.randint(3, 9)
        while len(g) < n and not g.terminal:
            g.take_action(int(rng.choice(g.valid_actions)))
        if not g.terminal:
            hists.append(list(g.history))
    out = []
    for h in hists:
        scores, v = TicTacToe(history=list(h)).solve()
        out.append({'history': h, 'scores': [int(s) for s in scores], 'v': int(v)})
    return out


TRAIN_CONF = {
    'game': {'module': 'alphazero.games.tictactoe', 'name': 'TicTacToe'},
    'model': {'module': 'alphazero.models', 'name': 'AlphaZero',
              'kwargs': {'shape_in': [2, 3, 3], 'shape_out': [9], 'num_blocks': 2, 'block_channels': 4}},
    'loss': {'module': 'alphazero.loss', 'name': 'AlphaZeroLoss'},
    'training': {'num_steps': 4, 'batch_size': 8, 'lr': 0.05, 'momentum': 0.9, 'weight_decay': 1.0e-4,
                 'lr_schedule_gamma': 0.1, 'lr_schedule': [2]},
}


def training_case(cases):
    """alphazero.training.TrainingWorker.process_batch x 4 on the CPU: the four tic-tac-toe games in a
    SQLite replay buffer, seeded model init and sampler; every logged scalar and the final weights."""
    import torch
    import yaml
    from alphazero.config import AlphaZeroConfig
    from alphazero.training import TrainingWorker
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, 'config.yaml'), 'w') as f:
        yaml.safe_dump(TRAIN_CONF, f)
    cfg = AlphaZeroConfig(os.path.join(tmp, 'config.yaml'))
    ds = ReplayBufferDataset(cfg, os.path.join(tmp, 'replay.sqlite'))
    for case in cases:
        g = TicTacToe()
        for mv in case['moves']:
            g.take_action(mv['action'], np.array(mv['N'], dtype=np.uint32))
        ds.add_game(g, 1)

    class Server:
        versions = []

        def update(self, model, version):
            self.versions.append(int(version))

    class Writer:
        def __init__(self):
            self.rows = []

        def add_scalar(self, tag, value, step):
            self.rows.append([tag, float(value), int(step)])

    torch.manual_seed(11)
    np.random.seed(12)
    server, writer = Server(), Writer()
    tw = TrainingWorker(cfg, server, ds, writer, 'cpu', 0)
    init = {k: fhex(v.detach().float().numpy()) for k, v in tw._model.state_dict().items() if v.dtype.is_floating_point}
    returns = [int(tw.process_batch()) for _ in range(4)]
    final = {k: fhex(v.detach().float().numpy()) for k, v in tw._model.state_dict().items() if v.dtype.is_floating_point}
    return {'conf': TRAIN_CONF, 'model_seed': 11, 'sampler_seed': 12, 'returns': returns, 'versions': server.versions,
            'scalars': writer.rows, 'init': init, 'final': final,
            'games': [{'history': list(c['history']), 'visits': [list(map(int, mv['N'])) for mv in c['moves']]}
                      for c in cases]}


def main():
    warnings.simplefilter('ignore')
    cases = json.load(open(os.path.join(HERE, 'selfplay_exact.json')))
    ttt = [c for c in cases if c['game'] == 'ttt'][:4]
    c4 = [c for c in cases if c['game'] == 'c4' and c['S'] <= 200][:3] + [c for c in cases if c['game'] == 'c4' and c['S'] == 800][:1]
    out = {'replay': [replay_case('ttt', ttt, [0, 3, 3, 7]), replay_case('c4', c4, [1, 2, 2, 40000])],
           'ttt_solve': solve_cases(),
           'training': training_case(ttt)}
    with open(os.path.join(HERE, 'host.json'), 'w') as f:
        json.dump(out, f, separators=(',', ':'))
    print('written', os.path.join(HERE, 'host.json'))


if __name__ == '__main__':
    sys.exit(main())
