"""The CPU oracle (oracle/az_oracle.c) against vectors produced by the reference itself.

These pin the oracle: every later HIP-vs-oracle comparison inherits its meaning
from the bit-exact agreement checked here.
"""
import numpy as np
import pytest

import oracle as orc
from helpers import eval_table, load_json, load_lut, unhex


@pytest.fixture(scope='module')
def lut():
    return orc.make_lut_array(load_lut())


@pytest.mark.parametrize('name', ['c4', 'ttt'])
def test_game_rules_match_reference(name):
    g = orc.GAME_ID[name]
    cases = load_json('games.json')[name]
    assert len(cases) > 300
    for c in cases:
        h = c['history']
        assert orc.valid_actions(g, h) == c['valid'], h
        assert orc.terminal(g, h) == c['terminal'], h
        assert orc.winner(g, h) == c['winner'], h
        x = orc.render(g, h)
        assert ''.join('1' if t else '0' for t in x.reshape(-1)) == c['render'], h


def test_c4_known_answer_wins():
    # alphazero/test/test_games.py:34-66
    for pos in load_json('games.json')['c4_kat_wins']:
        h = pos['history']
        assert not orc.terminal(orc.C4, h[:-1]) and orc.winner(orc.C4, h[:-1]) == orc.WINNER_NONE
        assert orc.terminal(orc.C4, h) and orc.winner(orc.C4, h) == pos['winner']


def test_c4_perfect_game():
    # alphazero/test/test_games.py:12-31
    acts = [int(c) - 1 for c in load_json('games.json')['c4_perfect']]
    for i in range(len(acts)):
        h = acts[:i]
        assert acts[i] in orc.valid_actions(orc.C4, h)
        assert not orc.terminal(orc.C4, h) and orc.winner(orc.C4, h) == orc.WINNER_NONE
        b = orc.render(orc.C4, h)
        assert b.sum() == i and b[1].sum() - b[0].sum() == i % 2
    assert orc.terminal(orc.C4, acts) and orc.winner(orc.C4, acts) == 0


def test_winning_position_table():
    ref = unhex(load_json('games.json')['c4_winning_positions'], np.uint8).reshape(69, 42)
    assert np.array_equal(orc.winning_positions(orc.C4), ref)


def test_node_numerics_match_reference(lut):
    for c in load_json('nodes.json'):
        node = orc.Node()
        P = unhex(c['P'], '<f4')
        node.expand(P)
        node.set_edges(N=np.array(c['N'], np.uint32), W=unhex(c['W'], '<f4'))
        assert node.mean_value().tobytes() == unhex(c['Q'], '<f4').tobytes()
        s = node.ucb_score(lut)
        assert s.tobytes() == unhex(c['ucb'], '<f8').tobytes()
        assert int(np.argmax(s)) == c['argmax']
        assert node.pi().tobytes() == unhex(c['pi'], '<f8').tobytes()
        assert node.greedy_action() == c['greedy']
        assert node.sample_action(unhex(c['sample_u'], '<f8')[0]) == c['sample']
        node.add_noise(unhex(c['eta'], '<f8'), c['eps'])
        assert node.edges()[2].tobytes() == unhex(c['P_noised'], '<f4').tobytes()


def test_lut_matches_host_numpy():
    # the table is the reference's own expression (mcts.py:55-57) on integer Ns
    lut = load_lut()
    ns = np.arange(lut.shape[0], dtype=np.uint64)
    assert np.array_equal(lut[:, 1], np.sqrt(ns))
    cs = np.array([np.log((1 + n + 19652) / 19652) + 1.25 for n in ns])
    assert cs.tobytes() == lut[:, 0].tobytes()


def test_production_lut_equals_the_golden_table_bit_for_bit():
    # what SelfPlayEngine installs by default (vectorised numpy) against the table the goldens were made with
    # (the reference's scalar expression, one Ns at a time): the same bits, row by row
    from alphazero_amd.engine import make_puct_lut
    lut = load_lut()
    assert make_puct_lut(lut.shape[0]).tobytes() == lut.tobytes()


def _replay(case, lut, evaluator):
    g = orc.GAME_ID[case['game']]
    cfg = orc.make_config(g, case['S'], case['exploration_depth'], case['reuse'], lut,
                          alpha=case['alpha'], epsilon=case['epsilon'])
    etas = [unhex(m['eta'], '<f8') for m in case['moves']]
    us = [unhex(m['u'], '<f8')[0] for m in case['moves'] if 'u' in m]
    stats = orc.Stats()
    rec = orc.play_game(cfg, evaluator, orc.dirichlet_from_list(etas), orc.uniform_from_list(us), stats)
    assert rec.num_moves == len(case['history'])
    assert list(rec.history[:rec.num_moves]) == case['history']
    assert rec.winner == case['winner']
    for i, m in enumerate(case['moves']):
        n = rec.num_edges[i]
        assert n == len(m['N'])
        assert list(rec.visits[i][:n]) == m['N'], (i, list(rec.visits[i][:n]), m['N'])
        assert np.array(rec.root_W[i][:n], np.float32).tobytes() == unhex(m['W'], '<f4').tobytes(), i
        assert np.array(rec.root_P[i][:n], np.float32).tobytes() == unhex(m['P'], '<f4').tobytes(), i
    assert stats.sims == case['S'] * rec.num_moves
    return rec, stats


def test_selfplay_exact_models(lut):
    cases = load_json('selfplay_exact.json')
    assert len(cases) >= 12
    for case in cases:
        ev = orc.eval_uniform() if case['model'] == 'uniform' else orc.eval_subset()
        _replay(case, lut, ev)


def test_selfplay_recorded_net(lut):
    for case in load_json('selfplay_net.json'):
        ev = orc.eval_from_table(orc.GAME_ID[case['game']], eval_table(case))
        _replay(case, lut, ev)
        assert not ev.misses


def test_run_mcts_epsilon_zero(lut):
    # eval.py:52-58 style caller: fresh root per move, epsilon = 0, greedy move
    for case in load_json('run_mcts_eps0.json'):
        g = orc.GAME_ID[case['game']]
        cfg = orc.make_config(g, case['S'], 0, False, lut, epsilon=0.0)
        for m in case['moves']:
            root = orc.Node()
            n_valid = len(orc.valid_actions(g, m['history']))
            eta = [np.full(n_valid, 1.0 / n_valid)]
            orc.run_mcts(cfg, m['history'], root, orc.eval_uniform(), orc.dirichlet_from_list(eta))
            N, W, P = root.edges()
            assert list(N) == m['N'] and int(N.sum()) == case['S']
            assert W.tobytes() == unhex(m['W'], '<f4').tobytes()
            assert P.tobytes() == unhex(m['P'], '<f4').tobytes()
            assert root.greedy_action() == m['greedy']


def test_visit_sum_equals_sims(lut):
    # alphazero/test/test_mcts.py:16-17,38-45: 64 sims from a fresh root, forced history
    cfg = orc.make_config(orc.TTT, 64, 0, False, lut, epsilon=0.25)
    hist = []
    for a in [0, 2, 8, 4, 6, 7, 3]:
        root = orc.Node()
        rng = np.random.RandomState(len(hist))
        n = len(orc.valid_actions(orc.TTT, hist))
        orc.run_mcts(cfg, hist, root, orc.eval_uniform(), orc.dirichlet_from_list([rng.dirichlet([0.3] * n)]))
        assert int(root.edges()[0].sum()) == 64
        hist.append(a)
    assert orc.terminal(orc.TTT, hist) and orc.winner(orc.TTT, hist) == 0
