#!/usr/bin/env python3
"""Generate golden vectors by running the reference itself (this container only).

Usage:  PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py

The reference (mtb2718/alphazero) is imported read-only from /root/reference and
driven through its own public entry points:
  * alphazero.games.{connectfour,tictactoe}  -> games.json   (rules, render)
  * alphazero.mcts.MCTreeNode                -> nodes.json   (Q / PUCT / pi / noise / sampling)
  * alphazero.selfplay.SelfPlayWorker.play_game and alphazero.mcts.run_mcts
                                             -> selfplay_*.json (whole seeded games)
Only DATA (inputs and the reference's outputs) is written; no reference source is
copied.  Floats are stored as little-endian hex so comparisons are bit-exact.
The numpy here is 2.2.x: ucb_score is float64 under its promotion rules.

Randomness: the reference uses the global legacy numpy RandomState
(mcts.py:71,90).  We seed it per game and record, by wrapping numpy's own
functions (never the reference's code), every Dirichlet draw and every uniform
consumed by RandomState.choice, so that a replay can inject the same numbers.
"""
import json
import os
import sys

import numpy as np
import torch

import alphazero.mcts as ref_mcts
import alphazero.selfplay as ref_selfplay
from alphazero.games.connectfour import ConnectFour, AlphaZeroC4, WINNING_POSITIONS
from alphazero.games.tictactoe import TicTacToe
from alphazero.models import AlphaZero, UniformModel

HERE = os.path.dirname(os.path.abspath(__file__))
GAMES = {'ttt': TicTacToe, 'c4': ConnectFour}


def hx(a, dtype):
    return np.ascontiguousarray(np.asarray(a, dtype=dtype)).tobytes().hex()


# --------------------------------------------------------------------------
# games.json
# --------------------------------------------------------------------------
C4_PERFECT = '44444666641267222263721335117357537731155'      # test_games.py:12
C4_WIN_POSITIONS = [                                            # test_games.py:34-57
    (0, [0, 1, 0, 1, 0, 1, 0]), (1, [3, 0, 1, 0, 1, 0, 1, 0]), (0, [6, 1, 6, 1, 6, 1, 6]),
    (1, [3, 6, 1, 6, 1, 6, 1, 6]), (0, [3, 3, 3, 2, 3, 2, 3, 2, 3]),
    (0, [0, 0, 1, 1, 2, 2, 3]), (1, [4, 0, 0, 1, 1, 2, 2, 3]), (0, [6, 6, 5, 5, 4, 4, 3]),
    (1, [0, 6, 6, 5, 5, 4, 4, 3]),
    (0, [6, 6, 5, 5, 4, 4, 0, 0, 1, 1, 2, 2, 3]), (1, [6, 6, 6, 5, 5, 4, 4, 0, 0, 1, 1, 2, 2, 3]),
    (0, [0, 1, 1, 3, 2, 2, 2, 2, 3, 3, 3]), (1, [4, 0, 1, 1, 3, 2, 2, 2, 2, 3, 3, 3]),
    (0, [3, 4, 4, 6, 5, 5, 5, 5, 6, 6, 6]), (1, [1, 3, 4, 4, 6, 5, 5, 5, 5, 6, 6, 6]),
    (0, [1, 0, 0, 0, 0, 2, 2, 1, 1, 5, 3]), (1, [0, 0, 0, 0, 1, 1, 2, 1, 5, 2, 5, 3]),
]


def describe(game):
    w = game.winner
    return {
        'history': [int(a) for a in game.history],
        'valid': [int(a) for a in game.valid_actions],
        'terminal': bool(game.terminal),
        'winner': -2 if w is None else int(w),
        'render': ''.join('1' if v else '0' for v in game.render().reshape(-1)),
    }


def make_games():
    rng = np.random.RandomState(1234)
    out = {'c4': [], 'ttt': [], 'c4_kat_wins': [], 'c4_perfect': C4_PERFECT}
    # every prefix of the perfect game
    acts = [int(c) - 1 for c in C4_PERFECT]
    for i in range(len(acts) + 1):
        out['c4'].append(describe(ConnectFour(list(acts[:i]))))
    # known-answer wins: position before and after the last move
    for winner, hist in C4_WIN_POSITIONS:
        out['c4'].append(describe(ConnectFour(list(hist[:-1]))))
        d = describe(ConnectFour(list(hist)))
        assert d['winner'] == winner and d['terminal']
        out['c4'].append(d)
        out['c4_kat_wins'].append({'winner': winner, 'history': hist})
    # random playouts, every prefix, played on past wins until the board is full
    # (valid_actions ignores terminality, connectfour.py:80-81)
    for name, cls, nplay in (('c4', ConnectFour, 40), ('ttt', TicTacToe, 60)):
        for _ in range(nplay):
            g = cls()
            out[name].append(describe(g))
            stop_at_terminal = rng.rand() < 0.5
            while g.valid_actions:
                if stop_at_terminal and g.terminal:
                    break
                g.take_action(int(rng.choice(g.valid_actions)))
                out[name].append(describe(g))
    out['c4_winning_positions'] = WINNING_POSITIONS.astype(np.uint8).tobytes().hex()
    return out


# --------------------------------------------------------------------------
# nodes.json : MCTreeNode numerics
# --------------------------------------------------------------------------
def make_nodes():
    rng = np.random.RandomState(99)
    cases = []
    for i in range(300):
        n = int(rng.randint(1, 10))
        p = rng.dirichlet([0.7] * n).astype(np.float32)
        node = ref_mcts.MCTreeNode()
        node.expand(np.full(n, 1.0 / n, dtype=np.float32))
        node._edges['P'] = p
        scale = int(rng.choice([1, 10, 100, 3000]))
        N = rng.randint(0, scale + 1, size=n).astype(np.uint32)
        if i % 7 == 0:
            N[:] = 0
        W = (rng.uniform(-1, 1, size=n) * N).astype(np.float32)
        node._edges['N'] = N
        node._edges['W'] = W
        case = {
            'n': n, 'N': [int(x) for x in N], 'W': hx(W, '<f4'), 'P': hx(p, '<f4'),
            'Q': hx(node.mean_value, '<f4'),
            'ucb': hx(node.ucb_score(), '<f8'),
            'argmax': int(np.argmax(node.ucb_score())),
            'pi': hx(node.pi(), '<f8'),
            'greedy': int(node.greedy_action()),
        }
        assert node.ucb_score().dtype == np.float64
        # sampling: recover the uniform RandomState.choice consumes
        np.random.seed(1000 + i)
        st = np.random.get_state()
        idx = int(node.sample_action())
        np.random.set_state(st)
        u = np.random.random_sample()
        case['sample_u'] = hx([u], '<f8')
        case['sample'] = idx
        # noise
        eps = float(rng.choice([0.25, 0.1, 0.0]))
        st = np.random.get_state()
        node.add_exploration_noise(0.3, eps)
        np.random.set_state(st)
        eta = np.random.dirichlet([0.3] * n)
        case['eps'] = eps
        case['eta'] = hx(eta, '<f8')
        case['P_noised'] = hx(node.action_prior, '<f4')
        cases.append(case)
    return cases


def make_lut(n=8192, c_base=19652, c_init=1.25):
    # exactly the expressions of mcts.py:55-57 evaluated for integer Ns
    cs = np.zeros(n, np.float64)
    sq = np.zeros(n, np.float64)
    for i in range(n):
        Ns = np.uint64(i)
        cs[i] = np.log((1 + Ns + c_base) / c_base) + c_init
        sq[i] = np.sqrt(Ns)
    return cs, sq


# --------------------------------------------------------------------------
# models for whole-game goldens
# --------------------------------------------------------------------------
class SubsetModel:
    """Exact 'subset net': logits 0 / -inf from a position hash; v a multiple of 1/8."""

    def __init__(self, num_actions):
        self.A = num_actions

    def eval(self):
        pass

    @staticmethod
    def position_hash(x):
        idx = np.flatnonzero(np.asarray(x).reshape(-1)).astype(np.uint64)
        acc = int(np.sum(idx * idx * 31 + 7 * idx + 1)) & 0xFFFFFFFF
        return (acc * 2654435761) & 0xFFFFFFFF

    def __call__(self, x, p_valid):
        h = self.position_hash(x[0].numpy())
        valid = p_valid[0].numpy()
        bits = np.array([(h >> (8 + a)) & 1 for a in range(self.A)], dtype=bool)
        if not np.any(bits & valid):
            logits = np.zeros(self.A, np.float32)
        else:
            logits = np.where(bits, 0.0, -np.inf).astype(np.float32)
        v = np.float32((int(h % 17) - 8) / 8.0)
        return torch.from_numpy(logits[None]), torch.tensor([[v]], dtype=torch.float32)


class StubConfig:
    def __init__(self, game_cls, model_fn, selfplay):
        self._game_cls, self._model_fn, self.selfplay = game_cls, model_fn, selfplay

    def Game(self, **kw):
        return self._game_cls(**kw)

    def Model(self):
        return self._model_fn()


class StubModelServer:
    def __init__(self, version):
        self.version = version

    def latest(self, model):
        return self.version


class StubDataset:
    def __init__(self):
        self.games = []

    def add_game(self, game, version):
        self.games.append((game, version))


class TorchHolder(torch.nn.Module):
    """So that config.Model().to(device) works for non-module callables."""

    def __init__(self, fn):
        super().__init__()
        self._fn = fn

    def eval(self):
        return self

    def forward(self, x, p_valid):
        return self._fn(x, p_valid)


def play_reference_game(game_name, model_kind, S, reuse, depth, seed, net_kwargs=None):
    game_cls = GAMES[game_name]
    A = game_cls.NUM_ACTIONS
    log = {'eta': [], 'u': [], 'roots': [], 'evals': {}}

    # --- wrap numpy RNG entry points (numpy's, not the reference's) --------
    real_dirichlet, real_choice = np.random.dirichlet, np.random.choice

    def rec_dirichlet(alpha, size=None):
        eta = real_dirichlet(alpha, size)
        log['eta'].append(eta.copy())
        return eta

    def rec_choice(a, size=None, replace=True, p=None):
        st = np.random.get_state()
        idx = real_choice(a, size, replace, p)
        after = np.random.get_state()
        np.random.set_state(st)
        u = np.random.random_sample()
        np.random.set_state(after)
        log['u'].append(float(u))
        return idx

    # --- wrap the reference's module-level functions to observe them -------
    real_evaluate, real_run = ref_mcts.evaluate, ref_selfplay.run_mcts

    def rec_evaluate(game, model):
        p, v = real_evaluate(game, model)
        x = game.render()
        key = ''.join('1' if t else '0' for t in x.reshape(-1))
        ent = (hx(p, '<f4'), hx(v, '<f4'))
        if key in log['evals']:
            assert log['evals'][key] == ent
        log['evals'][key] = ent
        return p, v

    def rec_run(game, root, model, n, *a, **k):
        real_run(game, root, model, n, *a, **k)
        e = root._edges
        log['roots'].append({'N': [int(x) for x in e['N']], 'W': hx(e['W'], '<f4'),
                             'P': hx(e['P'], '<f4'), 'len': len(game)})

    if model_kind == 'uniform':
        version, model_fn = 0, (lambda: TorchHolder(lambda x, pv: None))
    elif model_kind == 'subset':
        version, model_fn = 1, (lambda: TorchHolder(SubsetModel(A)))
    elif model_kind == 'net':
        version = 1

        def model_fn():
            torch.manual_seed(seed + 7)
            if game_name == 'c4':
                m = AlphaZeroC4(**net_kwargs)
            else:
                m = AlphaZero(**net_kwargs)
            # non-trivial BN statistics so eval-mode BN is exercised
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.running_mean.uniform_(-0.2, 0.2)
                    mod.running_var.uniform_(0.5, 1.5)
                    mod.weight.data.uniform_(0.5, 1.5)
                    mod.bias.data.uniform_(-0.2, 0.2)
            return m
    else:
        raise ValueError(model_kind)

    cfg = StubConfig(game_cls, model_fn, {'num_expansions_per_sim': S,
                                          'exploration_depth': depth,
                                          'reuse_search_tree': reuse})
    ds = StubDataset()
    worker = ref_selfplay.SelfPlayWorker(cfg, StubModelServer(version), ds, 'cpu')
    np.random.dirichlet, np.random.choice = rec_dirichlet, rec_choice
    ref_mcts.evaluate, ref_selfplay.run_mcts = rec_evaluate, rec_run
    try:
        np.random.seed(seed)
        torch.set_num_threads(1)
        worker.play_game()
    finally:
        np.random.dirichlet, np.random.choice = real_dirichlet, real_choice
        ref_mcts.evaluate, ref_selfplay.run_mcts = real_evaluate, real_run

    game, _ = ds.games[0]
    moves = []
    ui = 0
    for i, root in enumerate(log['roots']):
        assert root['len'] == i
        stats = game.search_statistics[i]
        assert [int(x) for x in stats] == root['N']
        m = {'N': root['N'], 'W': root['W'], 'P': root['P'], 'eta': hx(log['eta'][i], '<f8'),
             'action': int(game.history[i])}
        if i < depth:
            m['u'] = hx([log['u'][ui]], '<f8')
            ui += 1
        moves.append(m)
    assert ui == len(log['u']) and len(log['eta']) == len(moves)
    case = {'game': game_name, 'model': model_kind, 'S': S, 'reuse': bool(reuse),
            'exploration_depth': depth, 'seed': seed, 'alpha': 0.3, 'epsilon': 0.25,
            'history': [int(a) for a in game.history],
            'winner': int(game.winner), 'moves': moves}
    if model_kind == 'net':
        case['evals'] = log['evals']
        case['net_kwargs'] = net_kwargs
    return case


def run_mcts_eps0_case(game_name, S, seed, prefix):
    """eval.py:52-58 style caller: run_mcts(..., epsilon=0) on a fresh root each move."""
    game_cls = GAMES[game_name]
    np.random.seed(seed)
    game = game_cls(list(prefix))
    model = UniformModel()
    moves = []
    while not game.terminal and len(moves) < 6:
        root = ref_mcts.MCTreeNode()
        ref_mcts.run_mcts(game, root, model, S, epsilon=0)
        e = root._edges
        ai = int(root.greedy_action())
        moves.append({'history': [int(a) for a in game.history], 'N': [int(x) for x in e['N']],
                      'W': hx(e['W'], '<f4'), 'P': hx(e['P'], '<f4'), 'greedy': ai})
        game.take_action(game.valid_actions[ai])
    return {'game': game_name, 'S': S, 'moves': moves}


def main():
    with open(os.path.join(HERE, 'games.json'), 'w') as f:
        json.dump(make_games(), f, separators=(',', ':'))
    with open(os.path.join(HERE, 'nodes.json'), 'w') as f:
        json.dump(make_nodes(), f, separators=(',', ':'))
    cs, sq = make_lut()
    np.save(os.path.join(HERE, 'puct_lut.npy'), np.stack([cs, sq], axis=1))

    cases = []
    for seed in range(3):
        cases.append(play_reference_game('ttt', 'uniform', 32, True, 3, seed))
    cases.append(play_reference_game('ttt', 'uniform', 64, False, 3, 10))
    cases.append(play_reference_game('ttt', 'subset', 32, True, 3, 11))
    cases.append(play_reference_game('ttt', 'subset', 50, False, 9, 12))
    for S, seed in ((32, 20), (64, 21), (200, 22), (800, 23)):
        cases.append(play_reference_game('c4', 'uniform', S, True, 4, seed))
    cases.append(play_reference_game('c4', 'uniform', 200, False, 4, 24))
    cases.append(play_reference_game('c4', 'subset', 200, True, 4, 25))
    cases.append(play_reference_game('c4', 'subset', 64, False, 4, 26))
    cases.append(play_reference_game('c4', 'subset', 100, True, 42, 27))
    cases.append(play_reference_game('c4', 'subset', 800, True, 4, 28))
    with open(os.path.join(HERE, 'selfplay_exact.json'), 'w') as f:
        json.dump(cases, f, separators=(',', ':'))

    net_cases = [
        play_reference_game('c4', 'net', 48, True, 4, 30,
                            {'num_blocks': 2, 'channels_per_block': 8}),
        play_reference_game('c4', 'net', 32, False, 4, 31,
                            {'num_blocks': 1, 'channels_per_block': 4}),
        play_reference_game('ttt', 'net', 32, True, 3, 32,
                            {'shape_in': [2, 3, 3], 'shape_out': [9], 'num_blocks': 1,
                             'block_channels': 4}),
        # the shipped architecture (alphazero/config/connectfour.yaml:4-9)
        play_reference_game('c4', 'net', 32, True, 4, 33,
                            {'num_blocks': 6, 'channels_per_block': 32}),
    ]
    with open(os.path.join(HERE, 'selfplay_net.json'), 'w') as f:
        json.dump(net_cases, f, separators=(',', ':'))

    eps0 = [run_mcts_eps0_case('c4', 128, 40, []), run_mcts_eps0_case('ttt', 128, 41, [4]),
            run_mcts_eps0_case('c4', 128, 42, [3, 3, 3, 3, 3, 3, 2, 4])]
    with open(os.path.join(HERE, 'run_mcts_eps0.json'), 'w') as f:
        json.dump(eps0, f, separators=(',', ':'))
    print('golden vectors written to', HERE)


if __name__ == '__main__':
    sys.exit(main())
