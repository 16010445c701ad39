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


# ---- host-side evaluators over engine leaf batches (tests only) ---------------------
def planes_hash(x):
    """Position hash of the exact 'subset net' (same function as oracle/az_oracle.c
    orc_position_hash and tests/golden/make_golden.py SubsetModel): x float[n,2,H,W]."""
    flat = np.asarray(x).reshape(x.shape[0], -1) != 0
    idx = np.arange(flat.shape[1], dtype=np.uint64)
    w = (idx * idx * 31 + 7 * idx + 1)
    acc = (flat * w[None, :]).sum(axis=1).astype(np.uint64) & np.uint64(0xFFFFFFFF)
    return (acc * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)


def subset_eval(x, valid):
    """(p float32[n,A] action-indexed, v float32[n]) of the subset net for planes x, legal mask valid."""
    h = planes_hash(x)
    n, A = valid.shape
    bits = ((h[:, None] >> (8 + np.arange(A, dtype=np.uint64))[None, :]) & np.uint64(1)).astype(bool)
    sel = bits & valid
    k = sel.sum(axis=1)
    nv = valid.sum(axis=1)
    p = np.zeros((n, A), np.float32)
    for i in range(n):
        if nv[i] == 0:
            continue
        if k[i] == 0:
            p[i, valid[i]] = np.float32(1.0) / np.float32(nv[i])
        else:
            p[i, sel[i]] = np.float32(1.0) / np.float32(k[i])
    v = ((h % np.uint64(17)).astype(np.int64) - 8).astype(np.float32) / np.float32(8.0)
    return p, v


def uniform_eval(x, valid):
    n, A = valid.shape
    nv = valid.sum(axis=1)
    p = np.zeros((n, A), np.float32)
    for i in range(n):
        if nv[i]:
            p[i, valid[i]] = np.float32(1.0) / np.float32(nv[i])
    return p, np.zeros(n, np.float32)


def table_eval(table):
    """Evaluator from a recorded {render-bitstring: (p_compact, v)} table."""
    def fn(x, valid):
        n, A = valid.shape
        p = np.zeros((n, A), np.float32)
        v = np.zeros(n, np.float32)
        for i in range(n):
            key = render_key(x[i])
            if key in table:
                pc, vv = table[key]
                p[i, valid[i]] = pc
                v[i] = vv
        return p, v
    return fn


def golden_net_model(case, cls):
    """The network of a selfplay_net.json case, rebuilt the way tests/golden/make_golden.py built the
    reference's: same seed, same constructor arguments, same randomised batch-norm statistics."""
    import torch
    torch.manual_seed(case['seed'] + 7)
    m = cls(**case['net_kwargs'])
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.uniform_(-0.2, 0.2)
            mod.running_var.uniform_(0.5, 1.5)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.uniform_(-0.2, 0.2)
    return m.eval()


def golden_net_positions(case, shape):
    """(keys, x float32 [n, 2, H, W]) of every position the reference evaluated in that case."""
    keys = sorted(case['evals'])
    x = np.array([[float(c) for c in k] for k in keys], np.float32).reshape((-1,) + tuple(shape))
    return keys, x
