"""Host-side logic that needs no GPU: mirror Game classes vs the reference goldens, MCTreeNode
numerics vs goldens, model/state_dict layout, config loader, losses, replay window."""
import os

import numpy as np
import pytest
import torch

from helpers import load_json, unhex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mirror_games_match_reference():
    from alphazero_amd.games import BY_NAME
    g = load_json('games.json')
    for name in ('c4', 'ttt'):
        for c in g[name]:
            o = BY_NAME[name](c['history'])
            w = o.winner
            assert o.valid_actions == c['valid'] and o.terminal == c['terminal']
            assert (-2 if w is None else w) == c['winner']
            assert ''.join('1' if t else '0' for t in o.render().reshape(-1)) == c['render']
            assert o.next_player == len(c['history']) % 2 and len(o) == len(c['history'])


def test_mirror_node_numerics_match_reference():
    from alphazero_amd.mcts import MCTreeNode
    for c in load_json('nodes.json'):
        node = MCTreeNode()
        node.expand(np.full(c['n'], 1.0 / c['n'], np.float32))
        node._edges['P'] = unhex(c['P'], '<f4')
        node._edges['N'] = c['N']
        node._edges['W'] = unhex(c['W'], '<f4')
        assert node.mean_value.tobytes() == unhex(c['Q'], '<f4').tobytes()
        assert node.ucb_score().tobytes() == unhex(c['ucb'], '<f8').tobytes()
        assert node.pi().tobytes() == unhex(c['pi'], '<f8').tobytes()
        assert node.greedy_action() == c['greedy']
        np.random.seed(0)
        st = np.random.get_state()
        node._blend_noise(unhex(c['eta'], '<f8'), c['eps'])
        assert node._edges['P'].tobytes() == unhex(c['P_noised'], '<f4').tobytes()


def test_game_protocol_asserts():
    import pytest
    from alphazero_amd.games import ConnectFour, TicTacToe
    g = ConnectFour([0] * 6)
    with pytest.raises(AssertionError):
        g.take_action(0)
    with pytest.raises(AssertionError):
        TicTacToe(history=[0, 1], search_statistics=[None])
    c = g.clone()
    c.take_action(1)
    assert len(g) == 6 and len(c) == 7
    assert TicTacToe([0, 3, 1, 4, 2]).terminal_value(0) == 1 and TicTacToe([0, 3, 1, 4, 2]).terminal_value(1) == -1


def test_model_state_dict_layout_and_aliasing():
    from alphazero_amd.models import AlphaZero, AlphaZeroC4, UniformModel
    m = AlphaZeroC4(6, 32)
    sd = m.state_dict()
    assert len(sd) == 97 and sum(p.numel() for p in m.parameters()) == 34139
    assert sd['_tower.0._c1.weight'].data_ptr() == sd['_tower.5._c1.weight'].data_ptr()
    assert len({t.data_ptr() for t in sd.values()}) == 37
    t = AlphaZero([2, 3, 3], [9], num_blocks=3, block_channels=16)
    assert sum(p.numel() for p in t.parameters()) == 8036
    x = torch.zeros(2, 2, 3, 3)
    p, v = t.eval()(x, torch.ones(2, 9, dtype=torch.bool))
    assert p.shape == (2, 9) and v.shape == (2, 1)
    p, v = UniformModel()(x, torch.ones(2, 9, dtype=torch.bool))
    assert torch.all(p == 1) and torch.all(v == 0)


def test_config_loader_resolves_reference_module_paths(tmp_path):
    from alphazero_amd.config import AlphaZeroConfig
    from alphazero_amd.games import ConnectFour
    from alphazero_amd.models import AlphaZeroC4
    path = tmp_path / 'c4.yaml'
    path.write_text('game:\n  module: alphazero.games.connectfour\n  name: ConnectFour\n'
                    'model:\n  module: alphazero.games.connectfour\n  name: AlphaZeroC4\n'
                    '  kwargs:\n    num_blocks: 2\n    channels_per_block: 8\n'
                    'loss:\n  module: alphazero.loss\n  name: AlphaZeroLoss\n'
                    'selfplay:\n  num_expansions_per_sim: 8\n  exploration_depth: 4\n  reuse_search_tree: True\n')
    cfg = AlphaZeroConfig(str(path))
    assert isinstance(cfg.Game(), ConnectFour) and isinstance(cfg.Model(), AlphaZeroC4)
    assert cfg.Game(history=[3]).history == [3]
    assert cfg.selfplay['num_expansions_per_sim'] == 8
    assert cfg.Loss() is not None


def test_masked_losses():
    from alphazero_amd.loss import AlphaZeroLoss, masked_cross_entropy, masked_kl_div
    torch.manual_seed(0)
    x = torch.randn(16, 7)
    mask = torch.rand(16, 7) > 0.3
    mask[:, 0] = True
    t = torch.rand(16, 7) * mask
    t[3, 0] = 0
    t = t / t.sum(1, keepdim=True)
    H = masked_cross_entropy(x, t, mask)
    KL = masked_kl_div(x, t, mask)
    for i in range(16):
        lq = torch.log_softmax(x[i][mask[i]], 0)
        ti = t[i][mask[i]]
        assert torch.allclose(H[i, 0], -(ti * lq).sum(), atol=1e-6)
        nz = ti > 0
        assert torch.allclose(KL[i, 0], (ti[nz] * (torch.log(ti[nz]) - lq[nz])).sum(), atol=1e-6)
    pl, vl = AlphaZeroLoss()(t, torch.zeros(16, 1), x, torch.ones(16, 1), mask)
    assert torch.allclose(vl, torch.tensor(1.0)) and torch.isfinite(pl)


def test_replay_window_sample_layout(tmp_path):
    # alphazero/test/test_mcts.py:47-66 semantics on the in-memory window
    from alphazero_amd.config import AlphaZeroConfig
    from alphazero_amd.replay_buffer import ReplayBufferDataset, ReplayBufferSampler
    cfg = AlphaZeroConfig(conf={'game': {'module': 'alphazero.games.tictactoe', 'name': 'TicTacToe'}})
    ds = ReplayBufferDataset(cfg)
    hist = [0, 2, 8, 4, 6, 7, 3]
    game = cfg.Game()
    stats = []
    rng = np.random.RandomState(0)
    for a in hist:
        n = rng.randint(1, 50, size=len(game.valid_actions)).astype(np.uint32)
        stats.append((game.valid_actions, n))
        game.take_action(a, search_statistics=n)
    ds.add_game(game, 3)
    assert len(ds) == len(hist) and ds.num_games() == 1
    for i in range(len(ds)):
        ex = ds[i]
        valid, n = stats[i]
        assert ex['p_valid'].sum() == 9 - i and ex['model_version'] == 3
        want = np.zeros(9, np.float32)
        want[valid] = n.astype(np.int16) / np.sum(n.astype(np.int16))
        assert np.array_equal(ex['p'], want)
        assert ex['z'][0] == (1 if i % 2 == 0 else -1)
        assert ex['x'].shape == (2, 3, 3)
    np.random.seed(0)
    s = ReplayBufferSampler(ds, 4, bufferlen=3)
    idx = next(s)
    assert len(idx) == 4 and all(4 <= j < 7 for j in idx)
    ds.write_sqlite(str(tmp_path / 'dump.sqlite'))
    import sqlite3
    con = sqlite3.connect(str(tmp_path / 'dump.sqlite'))
    row = con.execute('SELECT model_version, history, search_statistics FROM selfplay_games').fetchone()
    assert row[0] == 3 and row[1] == '0,2,8,4,6,7,3'
    assert np.array_equal(np.frombuffer(row[2], np.int16), np.concatenate([n for _, n in stats]).astype(np.int16))
    assert con.execute('SELECT COUNT(*) FROM game_states').fetchone()[0] == 7


def test_model_server_and_checkpoint_format(tmp_path):
    # alphazero/model_server.py:19-48: update / latest / periodic checkpoint {model_version, model_state_dict}
    from alphazero_amd.model_server import ModelServer, load_ckpt
    from alphazero_amd.models import AlphaZeroC4
    torch.manual_seed(0)
    a, b = AlphaZeroC4(3, 8), AlphaZeroC4(3, 8)
    srv = ModelServer(str(tmp_path), ckpt_period=2)
    assert srv.latest(b) == 0
    srv.update(a, 1)
    assert not os.path.exists(tmp_path / 'ckpt' / '000001.pt')
    with torch.no_grad():
        a._cbr[0].weight.add_(1.0)
    srv.update(a, 2)
    assert srv.latest(b) == 2
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(va, vb), k
    path = tmp_path / 'ckpt' / '000002.pt'
    ck = torch.load(path, weights_only=True)
    assert ck['model_version'] == 2 and set(ck['model_state_dict']) == set(a.state_dict())
    assert any(k.startswith('_tower.2.') for k in ck['model_state_dict'])        # aliased block keys, as the reference saves them
    c = AlphaZeroC4(3, 8)
    assert load_ckpt(path, c) == 2 and torch.equal(c._cbr[0].weight, a._cbr[0].weight)
    srv.reset()
    assert srv.latest(b) == 0


@pytest.mark.parametrize('idx', [0, 1])
def test_checkpoint_is_what_the_reference_loader_accepted(tmp_path, idx):
    """tests/golden/checkpoint.json was recorded by handing a checkpoint written by THIS package's
    ModelServer to the reference's own loader (alphazero/eval.py:135-141: strict load_state_dict into the
    reference's module built from its config file).  Written again from the same seed, the file must have
    exactly the keys / shapes / dtypes the reference module took, alias the tower's one block the same
    way, carry the same numbers, and this package's module loaded from it must return what the reference
    module returned on the recorded positions."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import make_golden_ckpt as G
    from alphazero_amd.model_server import ModelServer, load_ckpt
    case = load_json('checkpoint.json')[idx]
    assert (case['seed'], case['version']) == (G.SEED, G.VERSION)
    model, shape = G.build(case['game'])
    ModelServer(str(tmp_path), ckpt_period=1).update(model, case['version'])
    path = tmp_path / 'ckpt' / f"{case['version']:06d}.pt"
    ck = torch.load(path, weights_only=True)
    assert ck['model_version'] == case['file_version'] == case['version']
    sd = ck['model_state_dict']
    assert [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()] == case['keys']
    for k, total in case['tensor_sums'].items():
        assert float(sd[k].double().sum()) == total, k
    # the reference's module aliases every tower block to one set of tensors; so does the file
    storage = {}
    for k, v in sd.items():
        storage.setdefault(v.data_ptr(), []).append(k)
    assert sorted(sorted(g) for g in storage.values() if len(g) > 1) == case['shared']
    # a fresh module of this package loaded from the file evaluates like the reference's module did
    fresh, _ = G.build(case['game'])
    with torch.no_grad():
        for prm in fresh.parameters():
            prm.zero_()
    assert load_ckpt(path, fresh) == case['version']
    n = case['n']
    A = 7 if case['game'] == 'c4' else 9
    x = torch.from_numpy(unhex(case['x'], '<f4').reshape((n,) + shape).copy())
    valid = torch.from_numpy(unhex(case['valid'], 'u1').reshape(n, A).astype(bool))
    with torch.no_grad():
        logits, v = fresh.eval()(x, valid)
    assert np.allclose(logits.numpy().ravel(), unhex(case['logits'], '<f4'), rtol=1e-6, atol=1e-6)
    assert np.allclose(v.numpy().ravel(), unhex(case['v'], '<f4'), rtol=1e-6, atol=1e-6)


def test_mirror_networks_reproduce_the_reference_outputs():
    """Every (p, v) the reference's own networks returned while the golden games were played
    (alphazero/mcts.py:94-107 on alphazero/models.py, connectfour.py:31-72) against the mirror modules
    rebuilt from the same seed: the parity chain for the HIP network kernels starts here."""
    from helpers import eval_table, golden_net_model, golden_net_positions
    from alphazero_amd.models import AlphaZero, AlphaZeroC4
    for case in load_json('selfplay_net.json'):
        c4 = case['game'] == 'c4'
        model = golden_net_model(case, AlphaZeroC4 if c4 else AlphaZero)
        keys, x = golden_net_positions(case, (2, 6, 7) if c4 else (2, 3, 3))
        tab = eval_table(case)
        occ = x[:, 0] + x[:, 1]
        valid = (occ[:, 5, :] == 0) if c4 else (occ.reshape(len(keys), -1) == 0)    # top row / empty cells
        with torch.no_grad():
            logits, v = model(torch.from_numpy(x), torch.from_numpy(valid))
        for i, k in enumerate(keys):
            p_ref, v_ref = tab[k]
            p = torch.softmax(logits[i][torch.from_numpy(valid[i])], 0).numpy()
            assert p.shape == p_ref.shape and np.allclose(p, p_ref, rtol=1e-5, atol=1e-7), (case['game'], i)
            assert abs(float(v[i, 0]) - float(v_ref)) <= 1e-6


def _host_cfg(name):
    from alphazero_amd.config import AlphaZeroConfig
    mod, cls = {'ttt': ('alphazero.games.tictactoe', 'TicTacToe'), 'c4': ('alphazero.games.connectfour', 'ConnectFour')}[name]
    return AlphaZeroConfig(conf={'game': {'module': mod, 'name': cls}})


def _golden_game(cfg, g):
    game = cfg.Game()
    for a, n in zip(g['history'], g['visits']):
        game.take_action(a, np.array(n, dtype=np.uint32))
    return game


def _db_rows(path):
    import sqlite3
    con = sqlite3.connect(path)
    games = [[r[0], r[1], r[2], bytes(r[3]).hex()] for r in
             con.execute('SELECT id, model_version, history, search_statistics FROM selfplay_games ORDER BY id')]
    states = [list(r) for r in con.execute('SELECT id, game_id, move_index FROM game_states ORDER BY id')]
    con.close()
    return games, states


@pytest.mark.parametrize('which', [0, 1])
def test_replay_database_matches_the_reference_rows_and_samples(tmp_path, which):
    # host.json holds what the reference's SelfPlayDatabase.write_game stored for these games and what
    # its ReplayBufferDataset.__getitem__ returned (alphazero/replay_buffer.py:28-47, 115-158)
    from alphazero_amd.replay_buffer import ReplayBufferDataset, SelfPlayDatabase
    case = load_json('host.json')['replay'][which]
    cfg = _host_cfg(case['game'])
    path = str(tmp_path / 'replay.sqlite')
    ds = ReplayBufferDataset(cfg, path)
    for g in case['games']:
        ds.add_game(_golden_game(cfg, g), g['model_version'])
    games, states = _db_rows(path)
    assert games == case['selfplay_games'] and states == case['game_states']
    assert (ds.db.num_games(), ds.db.num_states(), ds.db.num_distinct_games()) == \
        (case['num_games'], case['num_states'], case['num_distinct_games'])

    def check_items(d):
        assert len(d) == len(case['items'])
        for i, want in enumerate(case['items']):
            if want is None:        # Connect Four, every tenth state: the reference needs an external solver binary
                continue
            got = d[i]
            for k, dt in (('x', np.float32), ('z', np.float32), ('p', np.float32), ('move_scores', np.float32)):
                assert np.array_equal(np.asarray(got[k], np.float32).reshape(-1), unhex(want[k], '<f4')), (i, k)
            assert [int(b) for b in got['p_valid']] == want['p_valid']
            for k in ('game_index', 'move_index', 'model_version', 'solved', 'v*'):
                assert got[k] == want[k], (i, k)
    check_items(ds)
    check_items(ReplayBufferDataset(cfg, path))              # an existing database is read back on open
    db = SelfPlayDatabase.connect(path)
    g, version = db.read_game(1, cfg.Game())
    assert g.history == case['games'][1]['history'] and version == case['games'][1]['model_version']
    assert db.read_state(3) == (case['game_states'][3][1] - 1, case['game_states'][3][2])
    with pytest.raises(ValueError):
        db.read_state(10 ** 6)


def test_record_sink_writes_reference_rows(tmp_path):
    import ctypes as C
    from alphazero_amd._lib import GameRecord
    from alphazero_amd.replay_buffer import RecordSink
    case = load_json('host.json')['replay'][1]
    raw = np.zeros((len(case['games']), C.sizeof(GameRecord)), np.uint8)
    for i, g in enumerate(case['games']):
        r = GameRecord()
        r.uid, r.num_moves, r.winner, r.model_version = i, len(g['history']), 0, g['model_version']
        for m, (a, n) in enumerate(zip(g['history'], g['visits'])):
            r.history[m], r.num_edges[m] = a, len(n)
            for j, c in enumerate(n):
                r.visits[m][j] = c
        raw[i] = np.frombuffer(bytes(r), np.uint8)
    path = str(tmp_path / 'sink.sqlite')
    sink = RecordSink(path)
    sink.put(raw[:1])
    sink.put(raw[1:])
    sink.close()
    assert sink.games_written == len(case['games'])
    games, states = _db_rows(path)
    assert games == case['selfplay_games'] and states == case['game_states']


def test_tictactoe_solver_matches_reference():
    from alphazero_amd.games.tictactoe import TicTacToe
    for c in load_json('host.json')['ttt_solve']:
        scores, v = TicTacToe(history=c['history']).solve()
        assert [int(s) for s in scores] == c['scores'] and v == c['v'], c['history']


def test_solver_players_always_draw_tictactoe():
    # alphazero/eval.py:79-132 with both sides on the solver: perfect play is a draw
    from alphazero_amd.eval import SolverPlayer, match, play
    from alphazero_amd.games.tictactoe import TicTacToe
    np.random.seed(1)
    assert match(TicTacToe, [SolverPlayer(), SolverPlayer()], 6) == (0, 6, 0)
    g = TicTacToe(history=[4, 1])                      # a lost position for the side that took an edge
    assert play(g, [SolverPlayer(), SolverPlayer()]) and g.winner == 0


def test_training_steps_match_the_reference_on_cpu():
    # host.json 'training': four process_batch() calls of the reference's TrainingWorker
    # (alphazero/training.py:42-125) on the CPU with seeded init and sampler; same seeds here
    from alphazero_amd.config import AlphaZeroConfig
    from alphazero_amd.replay_buffer import ReplayBufferDataset
    from alphazero_amd.training import TrainingWorker
    t = load_json('host.json')['training']
    cfg = AlphaZeroConfig(conf=t['conf'])
    ds = ReplayBufferDataset(cfg)
    for g in t['games']:
        ds.add_game(_golden_game(cfg, g), 1)

    class Server:
        versions = []

        def update(self, model, version):
            self.versions.append(int(version))

    class Writer:
        rows = []

        def add_scalar(self, tag, value, step):
            self.rows.append([tag, float(value), int(step)])

    torch.manual_seed(t['model_seed'])
    np.random.seed(t['sampler_seed'])
    server, writer = Server(), Writer()
    tw = TrainingWorker(cfg, server, ds, writer, 'cpu', 0)
    state = tw.model.state_dict()
    for k, h in t['init'].items():
        assert np.array_equal(state[k].detach().numpy().reshape(-1), unhex(h, '<f4')), k     # same initialisation
    assert [tw.process_batch() for _ in range(4)] == t['returns'] and server.versions == t['versions']
    assert [(r[0], r[2]) for r in writer.rows] == [(r[0], r[2]) for r in t['scalars']]
    for got, want in zip(writer.rows, t['scalars']):
        assert abs(got[1] - want[1]) <= 1e-6 * max(1.0, abs(want[1])), (got, want)
    state = tw.model.state_dict()
    exact = 0
    for k, h in t['final'].items():
        a, b = state[k].detach().numpy().reshape(-1), unhex(h, '<f4')
        assert np.allclose(a, b, rtol=1e-6, atol=1e-7), k
        exact += int(np.array_equal(a, b))
    assert exact == len(t['final'])         # in fact bit-identical on this torch build


def test_connect_four_training_steps_match_the_reference_on_cpu():
    # training_c4.json: three process_batch() calls of the reference's TrainingWorker (alphazero/training.py:42-125)
    # with AlphaZeroC4(2, 32) on four recorded Connect Four games; the same seeds here, on the CPU (PyTorch's
    # forward and backward, as in the reference)
    from alphazero_amd.config import AlphaZeroConfig
    from alphazero_amd.replay_buffer import ReplayBufferDataset
    from alphazero_amd.training import TrainingWorker
    t = load_json('training_c4.json')
    cfg = AlphaZeroConfig(conf=t['conf'])
    ds = ReplayBufferDataset(cfg)
    for g in t['games']:
        ds.add_game(_golden_game(cfg, g), 1)

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

    torch.manual_seed(t['model_seed'])
    np.random.seed(t['sampler_seed'])
    server, writer = Server(), Writer()
    tw = TrainingWorker(cfg, server, ds, writer, 'cpu', 0)
    state = tw.model.state_dict()
    for k, h in t['init'].items():
        assert np.array_equal(state[k].detach().numpy().reshape(-1), unhex(h, '<f4')), k     # same initialisation
    assert [tw.process_batch() for _ in range(3)] == t['returns'] and server.versions == t['versions']
    assert [(r[0], r[2]) for r in writer.rows] == [(r[0], r[2]) for r in t['scalars']]
    for got, want in zip(writer.rows, t['scalars']):
        assert abs(got[1] - want[1]) <= 1e-6 * max(1.0, abs(want[1])), (got, want)
    state = tw.model.state_dict()
    for k, h in t['final'].items():
        assert np.allclose(state[k].detach().numpy().reshape(-1), unhex(h, '<f4'), rtol=1e-5, atol=1e-6), k
    for k, v in t['counters'].items():
        assert int(state[k]) == v, k


def test_tictactoe_solver_known_line():
    # the positions of alphazero/test/test_games.py:69-92, answers as the reference's solver gave them
    from alphazero_amd.games.tictactoe import TicTacToe
    line = [0, 2, 8, 4, 6, 7, 3]
    table = {tuple(c['history']): c for c in load_json('host.json')['ttt_solve']}
    for i in range(len(line)):
        game = TicTacToe(history=line[:i])
        want = table[tuple(line[:i])]
        scores, v = game.solve()
        assert len(game.valid_actions) == len(scores) and scores == want['scores'] and v == want['v']
    assert table[(0, 2)]['scores'] == [-1, 2, 0, 0, 2, 0, 2] and table[(0, 2, 8, 4, 6, 7)]['scores'] == [0, 2, -1]


def test_worker_refuses_the_eval_cache_with_a_callers_evaluator():
    """az_sim_step answers cached leaves by writing rows of the evaluator's own buffers, a contract only the
    built-in HIP networks keep: the combination is refused before anything is built."""
    import pytest
    from alphazero_amd.selfplay import SelfPlayWorker
    with pytest.raises(ValueError, match='built-in HIP networks'):
        SelfPlayWorker(None, None, None, 'cpu', evaluator=lambda bb, info: None, eval_cache_log2=12)
