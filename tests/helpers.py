"""Shared helpers for the parity tests (golden loading, hex <-> numpy)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def unhex(s, dtype):
    return np.frombuffer(bytes.fromhex(s), dtype=dtype).copy()


def load_lut():
    """float64 [n,2]: (Cs, sqrt(Ns)) as the reference's numpy computed them."""
    return np.load(os.path.join(GOLDEN, 'puct_lut.npy'))


def render_key(x):
    return ''.join('1' if t else '0' for t in np.asarray(x).reshape(-1))


def eval_table(case):
    """selfplay_net.json case -> {render-bitstring: (p float32[], v float32)}."""
    return {k: (unhex(p, '<f4'), unhex(v, '<f4')[0]) for k, (p, v) in case['evals'].items()}
