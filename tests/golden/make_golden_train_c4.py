#!/usr/bin/env python3
"""Golden vectors for the training step of the Connect Four network, produced by running the reference itself
(this container only).

Usage:  PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_train_c4.py

alphazero.training.TrainingWorker.process_batch x 3 on the CPU with AlphaZeroC4(2 blocks, 32 channels), batch 8, on
four recorded Connect Four games in the reference's own SQLite replay buffer: every logged scalar, the weights before
and after.  The reference's dataset asks an external solver binary (not in this image) about every tenth state
(alphazero/replay_buffer.py:143); the sampler seed recorded here is the first one whose three batches contain no
such state, so the reference runs unmodified.  Only data is written (training_c4.json).
"""
import json
import os
import sys
import tempfile
import warnings

import numpy as np

from alphazero.games.connectfour import ConnectFour
from alphazero.replay_buffer import ReplayBufferDataset

HERE = os.path.dirname(os.path.abspath(__file__))

CONF = {
    'game': {'module': 'alphazero.games.connectfour', 'name': 'ConnectFour'},
    'model': {'module': 'alphazero.games.connectfour', 'name': 'AlphaZeroC4',
              'kwargs': {'num_blocks': 2, 'channels_per_block': 32}},
    'loss': {'module': 'alphazero.loss', 'name': 'AlphaZeroLoss'},
    'training': {'num_steps': 3, 'batch_size': 8, 'lr': 0.05, 'momentum': 0.9, 'weight_decay': 1.0e-4,
                 'lr_schedule_gamma': 0.1, 'lr_schedule': [2]},
}
STEPS = 3


def fhex(a):
    return np.ascontiguousarray(a, dtype='<f4').tobytes().hex()


def run(cases, sampler_seed):
    import torch
    import yaml
    from alphazero.config import AlphaZeroConfig
    from alphazero.training import TrainingWorker
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, 'config.yaml'), 'w') as f:
        yaml.safe_dump(CONF, f)
    cfg = AlphaZeroConfig(os.path.join(tmp, 'config.yaml'))
    ds = ReplayBufferDataset(cfg, os.path.join(tmp, 'replay.sqlite'))
    for case in cases:
        g = ConnectFour()
        for mv in case['moves']:
            g.take_action(mv['action'], np.array(mv['N'], dtype=np.uint32))
        ds.add_game(g, 1)

    class Server:
        def __init__(self):
            self.versions = []

        def update(self, model, version):
            self.versions.append(int(version))

    class Writer:
        def __init__(self):
            self.rows = []

        def add_scalar(self, tag, value, step):
            self.rows.append([tag, float(value), int(step)])

    torch.manual_seed(31)
    np.random.seed(sampler_seed)
    server, writer = Server(), Writer()
    tw = TrainingWorker(cfg, server, ds, writer, 'cpu', 0)
    # (the initial weights follow from the recorded seed: one tensor of them is kept as a check)
    init = {k: fhex(v.detach().float().numpy()) for k, v in tw._model.state_dict().items() if k == '_tower.0._c1.weight'}
    returns = [int(tw.process_batch()) for _ in range(STEPS)]
    state = tw._model.state_dict()
    final = {k: fhex(v.detach().float().numpy()) for k, v in state.items() if v.dtype.is_floating_point}
    counters = {k: int(v) for k, v in state.items() if not v.dtype.is_floating_point}
    return {'conf': CONF, 'model_seed': 31, 'sampler_seed': sampler_seed, 'returns': returns, 'versions': server.versions,
            'scalars': writer.rows, 'init': init, 'final': final, 'counters': counters,
            'games': [{'history': list(c['history']), 'visits': [list(map(int, mv['N'])) for mv in c['moves']]}
                      for c in cases]}


def main():
    warnings.simplefilter('ignore')
    cases = json.load(open(os.path.join(HERE, 'selfplay_exact.json')))
    c4 = [c for c in cases if c['game'] == 'c4'][:4]
    for seed in range(1000):
        try:
            out = run(c4, seed)
        except FileNotFoundError:
            continue                      # a batch asked for the external solver: the next seed
        with open(os.path.join(HERE, 'training_c4.json'), 'w') as f:
            json.dump(out, f, separators=(',', ':'))
        print('written training_c4.json with sampler seed', seed, [r for r in out['scalars'] if r[0] == 'loss/total'])
        return 0
    return 1


if __name__ == '__main__':
    sys.exit(main())
