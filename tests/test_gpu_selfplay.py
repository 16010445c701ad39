"""Search parity on the GPU, through the C ABI: visit counts, W and P bit-exact with the
reference's golden games (fixed Dirichlet draws / uniforms, identical leaf evaluations)
and with the CPU oracle on batches of concurrent games."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle as orc
from helpers import (eval_table, load_json, load_lut, subset_eval, table_eval, unhex, uniform_eval)

pytestmark = pytest.mark.gpu

GAME = {'ttt': 0, 'c4': 1}
NUM_A = {0: 9, 1: 7}


def _host_eval(eng, evaluator, bb, info):
    """Evaluate engine leaves with a numpy evaluator; returns device tensors (p, v)."""
    from alphazero_amd import engine as E
    x, valid = E.render(eng.game, bb)
    p, v = evaluator(x.cpu().numpy(), valid.cpu().numpy())
    return torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda()


def _search_move(eng, evaluator, eta, fused):
    """One move's search (root expansion + noise + S simulations) for every slot."""
    bb, info = eng.begin_move()
    p, _ = _host_eval(eng, evaluator, bb, info)
    eng.apply_root(p, eta)
    if eng.S == 0:
        return
    if fused:
        bb, info = eng.select()
        for _ in range(eng.S - 1):
            p, v = _host_eval(eng, evaluator, bb, info)
            bb, info = eng.sim_step(p, v)
        p, v = _host_eval(eng, evaluator, bb, info)
        eng.expand_backup(p, v)
    else:
        for _ in range(eng.S):
            bb, info = eng.select()
            p, v = _host_eval(eng, evaluator, bb, info)
            eng.expand_backup(p, v)


def _root_edges(eng, g=0):
    img, root = eng.get_tree(g)
    n = int(img[root, eng.A, 0])
    assert int(img[root, eng.A, 1]) == int(img[root, :n, 0].sum())       # cached sum N is consistent
    return img[root, :n, 0].copy(), img[root, :n, 1].copy().view(np.float32), img[root, :n, 2].copy().view(np.float32)


def _replay_case(case, evaluator, mode):
    """mode: 'split' (select + expand_backup), 'fused' (sim_step) or 'inkernel' (uniform only)."""
    from alphazero_amd import engine as E
    g = GAME[case['game']]
    A = NUM_A[g]
    eng = E.SelfPlayEngine(g, 1, case['S'], exploration_depth=case['exploration_depth'],
                           reuse_search_tree=case['reuse'], alpha=case['alpha'], epsilon=case['epsilon'],
                           lut=load_lut())
    for i, m in enumerate(case['moves']):
        eta = torch.zeros(1, A, dtype=torch.float64, device='cuda')
        e = unhex(m['eta'], '<f8')
        eta[0, :len(e)] = torch.from_numpy(e)
        if mode == 'inkernel':
            eng.begin_move_uniform(eta)
            eng.run_sims_uniform(case['S'])
        else:
            _search_move(eng, evaluator, eta, fused=(mode == 'fused'))
        N, W, P = _root_edges(eng)
        assert list(N) == m['N'], (i, list(N), m['N'])
        assert W.tobytes() == unhex(m['W'], '<f4').tobytes(), i
        assert P.tobytes() == unhex(m['P'], '<f4').tobytes(), i
        u = None
        if 'u' in m:
            u = torch.from_numpy(unhex(m['u'], '<f8')).cuda()
        act, ne, vis, win = eng.choose_and_advance(u)
        assert int(act[0]) == m['action'], i
        assert int(ne[0]) == len(m['N'])
        assert list(vis[0, :len(m['N'])].cpu().numpy()) == m['N']
        last = i == len(case['moves']) - 1
        assert int(win[0]) == (case['winner'] if last else -2)
    recs = eng.drain_finished()
    assert len(recs) == 1
    r = recs[0]
    assert r.num_moves == len(case['history']) and r.winner == case['winner'] and r.uid == 0
    assert [r.history[i] for i in range(r.num_moves)] == case['history']
    for i, m in enumerate(case['moves']):
        assert list(r.visits[i][:r.num_edges[i]]) == m['N']
    st = eng.stats()
    assert st['sims'] == case['S'] * len(case['moves']) and st['errors'] == 0
    assert st['games'] == 1 and st['moves'] == len(case['moves'])
    eng.close()
    return st


def test_golden_games_exact_models_split_and_fused():
    for case in load_json('selfplay_exact.json'):
        if case['S'] > 200:
            continue
        ev = uniform_eval if case['model'] == 'uniform' else subset_eval
        _replay_case(case, ev, 'split')
        _replay_case(case, ev, 'fused')


def test_golden_games_800_sims():
    for case in load_json('selfplay_exact.json'):
        if case['S'] == 800:
            ev = uniform_eval if case['model'] == 'uniform' else subset_eval
            _replay_case(case, ev, 'fused')


def test_golden_games_uniform_in_kernel():
    n = 0
    for case in load_json('selfplay_exact.json'):
        if case['model'] == 'uniform':
            _replay_case(case, None, 'inkernel')
            n += 1
    assert n >= 8


def test_golden_games_recorded_net():
    for case in load_json('selfplay_net.json'):
        _replay_case(case, table_eval(eval_table(case)), 'fused')


def test_batch_of_games_matches_oracle():
    """64 concurrent Connect Four games (and 48 tic-tac-toe), each with its own Dirichlet /
    uniform stream, against the oracle playing the same games one at a time."""
    from alphazero_amd import engine as E
    lut_np = load_lut()
    lut = orc.make_lut_array(lut_np)
    for name, G, S, depth, reuse in (('c4', 64, 40, 6, True), ('ttt', 48, 24, 3, True), ('c4', 16, 30, 4, False)):
        g = GAME[name]
        A = NUM_A[g]
        max_moves = 42 if g == 1 else 9
        rng = np.random.RandomState(100 + G)
        etas = rng.dirichlet([0.3] * A, size=(G, max_moves))          # slot-indexed; first n used
        us = rng.random_sample((G, max_moves))
        # oracle: game k uses eta[k, move, :n_edges] renormalised?  No: exactly the first n entries.
        expected = []
        cfg = orc.make_config(g, S, depth, reuse, lut)
        for k in range(G):
            move_eta = iter(range(max_moves))

            def dir_fn(ctx, alpha, n, out, k=k, it=move_eta):
                mv = next(it)
                for j in range(n):
                    out[j] = etas[k, mv, j]
            uit = iter(us[k])
            rec = orc.play_game(cfg, orc.eval_subset(), orc.DIRICHLET_FN(dir_fn),
                                orc.UNIFORM_FN(lambda ctx, uit=uit: float(next(uit))))
            expected.append(rec)
        # the oracle draws one uniform per SAMPLED move only; moves < depth are the first ones
        eng = E.SelfPlayEngine(g, G, S, exploration_depth=depth, reuse_search_tree=reuse, lut=lut_np)
        done = {}
        for mv in range(max_moves):
            eta = torch.from_numpy(np.ascontiguousarray(etas[:, mv, :])).cuda()
            u = torch.from_numpy(np.ascontiguousarray(us[:, min(mv, max_moves - 1)])).cuda()
            _search_move(eng, subset_eval, eta, fused=True)
            eng.choose_and_advance(u)
            for r in eng.drain_finished():
                if r.uid < G and r.uid not in done:
                    done[int(r.uid)] = (r.num_moves, r.winner, [r.history[i] for i in range(r.num_moves)],
                                        [list(r.visits[i][:r.num_edges[i]]) for i in range(r.num_moves)])
            if len(done) == G:
                break
        assert len(done) == G
        for k in range(G):
            e = expected[k]
            nmv, win, hist, vis = done[k]
            assert nmv == e.num_moves and win == e.winner, k
            assert hist == list(e.history[:nmv]), k
            for i in range(nmv):
                assert vis[i] == list(e.visits[i][:e.num_edges[i]]), (k, i)
        assert eng.stats()['errors'] == 0
        eng.close()


def test_whole_tree_matches_oracle():
    """Every node of the search tree (N, W, P by path), not only the root."""
    from alphazero_amd import engine as E
    lut_np = load_lut()
    lut = orc.make_lut_array(lut_np)
    for name, hist, S in (('c4', [3, 3, 2, 4, 2, 2], 300), ('ttt', [4, 0], 200), ('c4', [], 150)):
        g = GAME[name]
        A = NUM_A[g]
        rng = np.random.RandomState(5)
        n_valid = len(orc.valid_actions(g, hist))
        eta_np = rng.dirichlet([0.3] * n_valid)
        cfg = orc.make_config(g, S, 0, True, lut)
        root = orc.Node()
        orc.run_mcts(cfg, hist, root, orc.eval_subset(), orc.dirichlet_from_list([eta_np]))
        want = root.dump()

        eng = E.SelfPlayEngine(g, 1, S, lut=lut_np)
        bb, nm = E.positions_from_history(g, [hist])
        b = bb.cpu().numpy().astype(np.uint64)
        eng.set_position(0, int(b[0, 0]), int(b[0, 1]), len(hist))
        eta = torch.zeros(1, A, dtype=torch.float64, device='cuda')
        eta[0, :n_valid] = torch.from_numpy(eta_np)
        _search_move(eng, subset_eval, eta, fused=True)
        img, root_idx = eng.get_tree(0)

        got = []

        def walk(idx, path):
            n = int(img[idx, A, 0])
            got.append((path, img[idx, :n, 0].copy(), img[idx, :n, 1].copy().view(np.float32),
                        img[idx, :n, 2].copy().view(np.float32)))
            for j in range(n):
                c = int(img[idx, j, 3])
                if c != 0xFFFFFFFF:
                    walk(c, path + (j,))
        walk(root_idx, ())
        assert len(got) == len(want) == root.count()
        for (pa, Na, Wa, Pa), (pb, Nb, Wb, Pb) in zip(got, want):
            assert pa == pb
            assert np.array_equal(Na, Nb), pa
            assert Wa.tobytes() == Wb.tobytes(), pa
            assert Pa.tobytes() == Pb.tobytes(), pa
        eng.close()


def test_run_mcts_epsilon_zero_goldens():
    """Single-game mirror API on the reference's eval.py-style call (fresh root, epsilon=0)."""
    from alphazero_amd.games import BY_NAME
    from alphazero_amd.mcts import MCTreeNode, run_mcts
    from alphazero_amd.models import UniformModel
    for case in load_json('run_mcts_eps0.json'):
        cls = BY_NAME[case['game']]
        for m in case['moves']:
            game = cls(list(m['history']))
            root = MCTreeNode()
            run_mcts(game, root, UniformModel(), case['S'], epsilon=0)
            assert list(root.num_visits) == m['N'] and int(root.num_visits.sum()) == case['S']
            assert root._edges['W'].tobytes() == unhex(m['W'], '<f4').tobytes()
            assert root._edges['P'].tobytes() == unhex(m['P'], '<f4').tobytes()
            assert root.greedy_action() == m['greedy']


def test_reference_play_loop_on_mirror_api():
    """The reference's play_game loop (selfplay.py:24-38) written against this package's
    Game / MCTreeNode / run_mcts with numpy's global RNG seeded like the golden run:
    histories, visit counts, W and P must reproduce the reference bit for bit."""
    from alphazero_amd.games import BY_NAME
    from alphazero_amd.mcts import MCTreeNode, run_mcts
    from alphazero_amd.models import UniformModel
    ran = 0
    for case in load_json('selfplay_exact.json'):
        if case['model'] != 'uniform' or case['S'] > 64:
            continue
        np.random.seed(case['seed'])
        game = BY_NAME[case['game']]()
        tree = MCTreeNode()
        model = UniformModel()
        i = 0
        while not game.terminal:
            run_mcts(game, tree, model, case['S'])
            m = case['moves'][i]
            assert list(tree.num_visits) == m['N'], i
            assert tree._edges['W'].tobytes() == unhex(m['W'], '<f4').tobytes(), i
            assert tree._edges['P'].tobytes() == unhex(m['P'], '<f4').tobytes(), i
            ai = tree.sample_action() if len(game) < case['exploration_depth'] else tree.greedy_action()
            game.take_action(game.valid_actions[ai], search_statistics=tree.num_visits)
            if case['reuse']:
                tree = tree.traverse(ai)
                tree.kill_siblings()
            else:
                tree = MCTreeNode()
            i += 1
        assert game.history == case['history'] and game.winner == case['winner']
        ran += 1
    assert ran >= 4


def test_search_from_child_keeps_ancestors():
    """alphazero/test/test_mcts.py:169-180: searching from a child must not touch the root's edges."""
    from alphazero_amd.games import ConnectFour
    from alphazero_amd.mcts import MCTreeNode, run_mcts
    from alphazero_amd.models import UniformModel
    np.random.seed(0)
    game = ConnectFour()
    root = MCTreeNode()
    run_mcts(game, root, UniformModel(), 64)
    assert int(root.num_visits.sum()) == 64
    before = root._edges.copy()
    ai = root.greedy_action()
    child = root.traverse(ai)
    n_child = int(child.num_visits.sum())
    game.take_action(game.valid_actions[ai])
    run_mcts(game, child, UniformModel(), 32)
    assert root._edges.tobytes() == before.tobytes()
    assert int(child.num_visits.sum()) == n_child + 32
    assert child.parent is root


def test_device_rng_noise_and_sampling_statistics():
    """Production path: Dirichlet(0.3) noise and move sampling from the device Philox RNG."""
    from alphazero_amd import engine as E
    G, A = 4096, 7
    eng = E.SelfPlayEngine(1, G, 1, exploration_depth=42, epsilon=1.0, alpha=0.3, seed=11)
    eng.begin_move_uniform()                       # epsilon = 1 -> P = eta exactly
    P = np.stack([eng.get_tree(g)[0][0, :A, 2].view(np.float32) for g in range(0, G, 8)])
    assert np.allclose(P.sum(axis=1), 1.0, atol=1e-5) and (P >= 0).all()
    # Dirichlet(a,...,a) with K=7: mean 1/K, var = (K-1)/(K^2 (K a + 1))
    K, a = 7, 0.3
    var = (K - 1) / (K * K * (K * a + 1))
    assert abs(P.mean() - 1 / K) < 1e-6
    assert np.all(np.abs(P.mean(axis=0) - 1 / K) < 5 * np.sqrt(var / P.shape[0]))
    assert np.all(np.abs(P.var(axis=0) - var) < 0.25 * var)
    # one simulation each, then sample: chosen actions follow pi = N / sum N = one-hot of slot 0
    eng.run_sims_uniform(1)
    act, _, vis, _ = eng.choose_and_advance()
    vis = vis.cpu().numpy()
    assert (vis.sum(axis=1) == 1).all()
    assert np.array_equal(act.cpu().numpy(), vis.argmax(axis=1))
    # many simulations, uniform prior: sampled move frequencies follow the visit distribution
    eng2 = E.SelfPlayEngine(1, G, 64, exploration_depth=42, epsilon=0.0, seed=3)
    eng2.begin_move_uniform()
    eng2.run_sims_uniform(64)
    act, _, vis, _ = eng2.choose_and_advance()
    vis = vis.cpu().numpy().astype(np.float64)
    assert (vis == vis[0]).all()                   # identical deterministic searches
    freq = np.bincount(act.cpu().numpy(), minlength=A) / G
    pi = vis[0] / vis[0].sum()
    assert np.all(np.abs(freq - pi) < 5 * np.sqrt(pi * (1 - pi) / G) + 1e-3)
    eng.close(); eng2.close()


def test_pool_exhaustion_is_reported():
    from alphazero_amd import engine as E
    eng = E.SelfPlayEngine(1, 2, 64, node_capacity=8)
    eng.begin_move_uniform()
    eng.run_sims_uniform(64)
    assert eng.stats()['errors'] == 2
    eng.close()


def test_search_statistics_survive_the_sqlite_replay_buffer(tmp_path):
    """alphazero/test/test_mcts.py:14-66 on this package: 64 simulations per fresh root with the
    UniformModel, a forced tic-tac-toe line, the SQLite-backed dataset returns exactly the searched
    distribution and the alternating outcome."""
    import os
    from alphazero_amd.config import AlphaZeroConfig
    from alphazero_amd.mcts import MCTreeNode, run_mcts
    from alphazero_amd.replay_buffer import ReplayBufferDataset
    cfg = AlphaZeroConfig(conf={'game': {'module': 'alphazero.games.tictactoe', 'name': 'TicTacToe'},
                                'model': {'module': 'alphazero.models', 'name': 'UniformModel'}})
    model, game = cfg.Model(), cfg.Game()
    ds = ReplayBufferDataset(cfg, os.path.join(str(tmp_path), 'test.sqlite'))
    line = [0, 2, 8, 4, 6, 7, 3]
    stats = np.zeros((len(line), game.NUM_ACTIONS))
    np.random.seed(0)
    for i, action in enumerate(line):
        root = MCTreeNode()
        run_mcts(game, root, model, 64)
        assert np.sum(root.num_visits) == 64
        stats[i, game.valid_actions] = root.num_visits
        game.take_action(action, search_statistics=root.num_visits)
    ds.add_game(game, 0)
    assert len(ds) == len(line)
    for d in (ds, ReplayBufferDataset(cfg, os.path.join(str(tmp_path), 'test.sqlite'))):
        for i in range(len(d)):
            ex = d[i]
            assert np.sum(ex['p_valid']) == 9 - i
            assert np.all(ex['p'] == stats[i] / np.sum(stats[i]))
            assert ex['z'] == (1 if i % 2 == 0 else -1)


def test_golden_net_games_with_the_hip_networks_in_the_loop():
    """The reference's recorded games with real networks, replayed with the fp32 HIP kernels (not the
    recorded outputs) evaluating the leaves: the kernels agree with the reference's torch fp32 to ~1e-7,
    which on these games leaves every PUCT arg-max, hence every visit count and move, unchanged; W and P
    agree within 1e-5."""
    from helpers import golden_net_model
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZero, AlphaZeroC4
    from alphazero_amd.net import FusedC4NetF32, TTTNetF32
    ran = 0
    for case in load_json('selfplay_net.json'):
        g = GAME[case['game']]
        A = NUM_A[g]
        if g == 1:
            net = FusedC4NetF32(golden_net_model(case, AlphaZeroC4), 'cuda')
        else:
            net = TTTNetF32(golden_net_model(case, AlphaZero), 'cuda')
        eng = E.SelfPlayEngine(g, 1, case['S'], exploration_depth=case['exploration_depth'],
                               reuse_search_tree=case['reuse'], alpha=case['alpha'], epsilon=case['epsilon'],
                               lut=load_lut())
        for i, m in enumerate(case['moves']):
            eta = torch.zeros(1, A, dtype=torch.float64, device='cuda')
            e = unhex(m['eta'], '<f8')
            eta[0, :len(e)] = torch.from_numpy(e)
            bb, info = eng.begin_move()
            p, _ = net(bb, None)
            eng.apply_root(p, eta)
            bb, info = eng.select()
            for s in range(case['S']):
                p, v = net(bb, None)
                if s < case['S'] - 1:
                    bb, info = eng.sim_step(p, v)
                else:
                    eng.expand_backup(p, v)
            N, W, P = _root_edges(eng)
            assert list(N) == m['N'], (case['net_kwargs'], i)
            assert np.allclose(W, unhex(m['W'], '<f4'), rtol=1e-5, atol=1e-5)
            assert np.allclose(P, unhex(m['P'], '<f4'), rtol=1e-5, atol=1e-6)
            u = torch.from_numpy(unhex(m['u'], '<f8')).cuda() if 'u' in m else None
            act, _, _, win = eng.choose_and_advance(u)
            assert int(act[0]) == m['action'], i
        eng.close()
        ran += 1
    assert ran >= 4
