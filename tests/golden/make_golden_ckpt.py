#!/usr/bin/env python3
"""Golden vectors for the checkpoint format, produced with the reference's own loader (this container only).

Usage:  PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_ckpt.py

For Connect Four (the shipped 6 x 32 network) and tic-tac-toe (3 x 16):
  * this repository's ModelServer.checkpoint writes LOGDIR/ckpt/000007.pt next to a copy of the reference's
    config file as LOGDIR/config.yaml -- the layout alphazero/train.py:26-31 + model_server.py:42-48 leave behind;
  * the reference's alphazero.eval.load_ckpt (eval.py:135-141: config lookup, config.Model(), torch.load, STRICT
    load_state_dict into the reference's module with its aliased _tower.i.* keys) reads it back;
  * recorded: the reference module's state_dict keys / shapes / dtypes in order, which keys share storage (the
    tower's one block), and the reference module's outputs on eight positions.
The CPU test (tests/test_host_cpu.py) writes the same checkpoint again from the recorded seed and checks this
repository's side against all of it.  Only data is written (checkpoint.json); no reference source is copied.
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/alphazero'
SEED, VERSION = 23, 7


def fhex(a):
    return np.ascontiguousarray(a).tobytes().hex()


def build(game):
    """this repository's module with non-trivial batch-norm statistics, deterministically"""
    from alphazero_amd.models import AlphaZero, AlphaZeroC4
    torch.manual_seed(SEED)
    if game == 'c4':
        m, shape = AlphaZeroC4(6, 32), (2, 6, 7)
    else:
        m, shape = AlphaZero([2, 3, 3], [9], 3, 16), (2, 3, 3)
    m.train()
    with torch.no_grad():
        for _ in range(3):                      # moves the running statistics away from (0, 1)
            x = (torch.rand(32, *shape) < 0.3).float()
            m(x, torch.ones(32, 7 if game == 'c4' else 9, dtype=torch.bool))
    return m.eval(), shape


def positions(game, shape):
    rng = np.random.RandomState(SEED + 1)
    n = 8
    A = 7 if game == 'c4' else 9
    x = np.zeros((n,) + shape, np.float32)
    valid = np.zeros((n, A), bool)
    for i in range(n):
        if game == 'c4':
            heights = rng.randint(0, 7, size=7)
            for c in range(7):
                for r in range(heights[c]):
                    x[i, rng.randint(2), r, c] = 1
                valid[i, c] = heights[c] < 6
        else:
            for sq in range(9):
                k = rng.randint(3)
                if k < 2:
                    x[i, k, sq // 3, sq % 3] = 1
                valid[i, sq] = k == 2
            if not valid[i].any():
                x[i, :, 0, 0] = 0
                valid[i, 0] = True
    return x, valid


def case(game, config_name):
    from alphazero.eval import load_ckpt                 # the reference's loader
    from alphazero_amd.model_server import ModelServer
    model, shape = build(game)
    logdir = tempfile.mkdtemp()
    shutil.copy(os.path.join(REF, 'config', config_name), os.path.join(logdir, 'config.yaml'))
    ModelServer(logdir, ckpt_period=1).update(model, VERSION)
    path = os.path.join(logdir, 'ckpt', f'{VERSION:06d}.pt')
    ref_model, _ = load_ckpt(path)
    ref_model.eval()
    sd = ref_model.state_dict()
    storage = {}
    for k, v in sd.items():
        storage.setdefault(v.data_ptr(), []).append(k)
    x, valid = positions(game, shape)
    with torch.no_grad():
        logits, v = ref_model(torch.from_numpy(x), torch.from_numpy(valid))
    raw = torch.load(path, weights_only=True)
    return {'game': game, 'seed': SEED, 'version': VERSION, 'file_version': int(raw['model_version']),
            'reference_config': config_name,
            'keys': [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()],
            'shared': sorted(sorted(g) for g in storage.values() if len(g) > 1),
            'x': fhex(x), 'valid': fhex(valid.astype(np.uint8)), 'n': int(x.shape[0]),
            'logits': fhex(logits.numpy()), 'v': fhex(v.numpy()),
            'tensor_sums': {k: float(v.double().sum()) for k, v in sd.items() if v.dtype.is_floating_point}}


def main():
    out = [case('c4', 'connectfour.yaml'), case('ttt', 'tictactoe.yaml')]
    with open(os.path.join(HERE, 'checkpoint.json'), 'w') as f:
        json.dump(out, f, separators=(',', ':'))
    print('written', os.path.join(HERE, 'checkpoint.json'), [len(c['keys']) for c in out])


if __name__ == '__main__':
    sys.exit(main())
