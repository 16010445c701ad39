"""Two engine features that must not change a single visit count: the evaluation cache (a leaf answered
from the table gives the search exactly what the evaluator would have given it; the reference evaluates
every leaf: alphazero/mcts.py:139-142) and the one-launch-per-move kernel az_c4_move_fused (a workgroup
owns its games; same trees as the stepwise calls), alone, together and at the benchmark's full size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _play(net_modules, log2_entries, G=512, S=48, moves=14, game='c4'):
    from alphazero_amd import engine as E
    from alphazero_amd.net import C4Net128, FusedC4Net
    dev = torch.device('cuda', 0)
    eng = E.SelfPlayEngine(E._lib.GAME_C4, G, S, exploration_depth=4, reuse_search_tree=True,
                           alpha=0.3, epsilon=0.25, seed=11, device=0)
    if log2_entries:
        eng.enable_eval_cache(log2_entries)
    fused = [FusedC4Net.from_module(m.to(dev), dev) if m.block_channels <= 32 else C4Net128(m.to(dev), dev, max_leaves=G)
             for m in net_modules]
    out = []
    kinds = set()
    for mv in range(moves):
        net = fused[(mv // 5) % len(fused)]          # new weights (and version) every five moves
        eng.set_model_version(1 + mv // 5)
        bb, info = eng.begin_move()
        p, _ = net(bb, info)
        eng.apply_root(p)
        bb, info = eng.select()
        for s in range(S):
            p, v = net(bb, info)
            if s < S - 1:
                bb, info = eng.sim_step(p, v)
                if mv == moves - 1 and s % 8 == 0:
                    kinds |= set((info & 0xFF).unique().tolist())
            else:
                eng.expand_backup(p, v)
        a, ne, vis, w = eng.choose_and_advance()
        out.append((a.cpu().numpy().copy(), vis.cpu().numpy().copy(), w.cpu().numpy().copy()))
    st = eng.stats()
    hits = eng.eval_cache_hits() if log2_entries else 0
    recs = eng.drain_finished_bytes().copy()
    recs = recs[np.argsort(recs[:, :8].copy().view(np.uint64)[:, 0])]     # ring order is arrival order
    eng.close()
    return out, st, hits, recs, kinds


def test_cache_on_off_identical():
    torch.manual_seed(3)
    from alphazero_amd.models import AlphaZeroC4
    nets = [AlphaZeroC4(6, 32).eval(), AlphaZeroC4(2, 16).eval()]
    for m in nets:
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.uniform_(-0.2, 0.2)
                mod.running_var.uniform_(0.5, 1.5)
    ref, st0, _, rec0, kinds0 = _play(nets, 0)
    assert 4 not in kinds0
    for log2 in (20, 10):                              # roomy, and tiny (constant eviction)
        got, st1, hits, rec1, kinds1 = _play(nets, log2)
        for (a0, v0, w0), (a1, v1, w1) in zip(ref, got):
            np.testing.assert_array_equal(a0, a1)
            np.testing.assert_array_equal(v0, v1)
            np.testing.assert_array_equal(w0, w1)
        assert st0 == st1
        np.testing.assert_array_equal(rec0, rec1)
        assert hits > 0
        if log2 == 20:
            assert 4 in kinds1 and hits > st1['leaf_evals'] // 10


def test_cache_on_off_identical_128_channel_network():
    torch.manual_seed(4)
    from alphazero_amd.models import AlphaZeroC4
    nets = [AlphaZeroC4(2, 128).eval()]
    ref, st0, _, rec0, _ = _play(nets, 0, G=192, S=24, moves=8)
    got, st1, hits, rec1, kinds = _play(nets, 16, G=192, S=24, moves=8)
    for (a0, v0, w0), (a1, v1, w1) in zip(ref, got):
        np.testing.assert_array_equal(a0, a1)
        np.testing.assert_array_equal(v0, v1)
    assert st0 == st1 and hits > 0 and 4 in kinds
    np.testing.assert_array_equal(rec0, rec1)


def _hash_evaluator(G, A, dev):
    """(p, v) as an elementwise function of the position; rows the engine answered from its cache
    are left exactly as the engine wrote them."""
    pbuf = torch.zeros(G, A, dtype=torch.float32, device=dev)
    vbuf = torch.zeros(G, dtype=torch.float32, device=dev)
    acts = torch.arange(A, device=dev, dtype=torch.int64)

    def ev(bb, info):
        h = (bb[:, 0].to(torch.int64) * 1000003 + bb[:, 1].to(torch.int64) * 7919) % 65521
        legal = ((info.to(torch.int64)[:, None] >> (16 + acts[None, :])) & 1).to(torch.float32)
        w = (((h[:, None] + 1) * (acts[None, :] + 3)) % 251 + 1).to(torch.float32) * legal
        p = w / w.sum(1, keepdim=True).clamp_min(1.0)
        v = ((h % 201).to(torch.float32) - 100.0) / 100.0
        fresh = (info & 0xFF) != 4
        pbuf[fresh] = p[fresh]
        vbuf[fresh] = v[fresh]
        return pbuf, vbuf
    return ev


@pytest.mark.parametrize('game', ['ttt', 'c4'])
def test_cache_generic_evaluator(game):
    from alphazero_amd import engine as E
    gid = E._lib.GAME_TTT if game == 'ttt' else E._lib.GAME_C4
    G, S, moves = 1024, 40, 6
    dev = torch.device('cuda', 0)
    res = []
    for log2 in (0, 18):
        eng = E.SelfPlayEngine(gid, G, S, exploration_depth=3, reuse_search_tree=True, seed=5, device=0)
        if log2:
            eng.enable_eval_cache(log2)
        ev = _hash_evaluator(G, eng.A, dev)
        out = []
        for mv in range(moves):
            bb, info = eng.begin_move()
            p, _ = ev(bb, info)
            eng.apply_root(p)
            bb, info = eng.select()
            for s in range(S):
                p, v = ev(bb, info)
                if s < S - 1:
                    bb, info = eng.sim_step(p, v)
                else:
                    eng.expand_backup(p, v)
            a, ne, vis, w = eng.choose_and_advance()
            out.append((a.cpu().numpy().copy(), vis.cpu().numpy().copy()))
        res.append((out, eng.stats(), eng.eval_cache_hits() if log2 else 0))
        eng.close()
    for (a0, v0), (a1, v1) in zip(res[0][0], res[1][0]):
        np.testing.assert_array_equal(a0, a1)
        np.testing.assert_array_equal(v0, v1)
    assert res[0][1] == res[1][1]
    assert res[1][2] > 0


def test_cache_bad_size_rejected():
    from alphazero_amd import engine as E
    from alphazero_amd._lib import AzError
    eng = E.SelfPlayEngine(E._lib.GAME_C4, 8, 4, device=0)
    with pytest.raises(AzError):
        eng.enable_eval_cache(5)
    eng.enable_eval_cache(12)
    eng.enable_eval_cache(0)
    eng.close()


@pytest.mark.parametrize('G,S', [(512, 48), (8192, 24), (100, 33), (8, 20), (257, 12), (1, 40), (4000, 7), (16384, 5)])
def test_fused_move_equals_the_stepwise_loop(G, S):
    """az_c4_move_fused (a workgroup owns its games for the whole move, search and network in one launch)
    against the stepwise loop with az_c4net_bf16_forward: identical moves, visit counts, records, counters."""
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import FusedC4Net
    torch.manual_seed(8)
    model = AlphaZeroC4(6, 32).eval().cuda()
    net = FusedC4Net.from_module(model, 'cuda')
    out = []
    hits = 0
    for fused in (False, True, 'cache'):
        eng = E.SelfPlayEngine(E._lib.GAME_C4, G, S, exploration_depth=4, reuse_search_tree=True, seed=21, device=0)
        if fused == 'cache':
            eng.enable_eval_cache(16)
        res = []
        for mv in range(10):
            bb, info = eng.begin_move()
            p, _ = net(bb, info)
            eng.apply_root(p)
            if fused:
                eng.move_fused(net)
            else:
                bb, info = eng.select()
                for s in range(S):
                    p, v = net(bb, info)
                    if s < S - 1:
                        bb, info = eng.sim_step(p, v)
                    else:
                        eng.expand_backup(p, v)
            a, ne, vis, w = eng.choose_and_advance()
            res.append((a.cpu().numpy().copy(), vis.cpu().numpy().copy(), w.cpu().numpy().copy()))
        recs = eng.drain_finished_bytes().copy()
        recs = recs[np.argsort(recs[:, :8].copy().view(np.uint64)[:, 0])]
        out.append((res, eng.stats(), recs))
        if fused == 'cache':
            hits = eng.eval_cache_hits()
        eng.close()
    for other in (1, 2):                               # one launch per move; the same with the evaluation cache
        for (a0, v0, w0), (a1, v1, w1) in zip(out[0][0], out[other][0]):
            np.testing.assert_array_equal(a0, a1)
            np.testing.assert_array_equal(v0, v1)
            np.testing.assert_array_equal(w0, w1)
        assert out[0][1] == out[other][1]
        np.testing.assert_array_equal(out[0][2], out[other][2])
    assert hits > 0 or G == 1              # (a single small tree may hold no transposition)


@pytest.mark.parametrize('G,S', [(8192, 800), (1024, 200)])
def test_full_size_move_properties(G, S):
    """The benchmark's own shapes (8 192 games x 800 simulations, and 1 024 x 200, the single-GPU
    configuration; 6x32 network, one launch per move): the
    size-independent properties of a search -- every simulation is accounted for, a fresh root's visit
    counts sum to S and a re-used root's to at least S, no pool is exhausted -- and the stepwise loop
    agrees bit for bit on the last move."""
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import FusedC4Net
    torch.manual_seed(0)
    net = FusedC4Net.from_module(AlphaZeroC4(6, 32).eval().cuda(), 'cuda')
    engs = [E.SelfPlayEngine(E._lib.GAME_C4, G, S, exploration_depth=4, reuse_search_tree=True, seed=3, device=0)
            for _ in range(2)]
    last = []
    for k, eng in enumerate(engs):
        for mv in range(3):
            bb, info = eng.begin_move()
            fresh = ((info & 0xFF) == 3).cpu().numpy()                   # roots that had to be expanded
            p, _ = net(bb, info)
            eng.apply_root(p)
            if k == 0 or mv < 2:
                eng.move_fused(net)
            else:
                bb, info = eng.select()
                for s in range(S):
                    p, v = net(bb, info)
                    if s < S - 1:
                        bb, info = eng.sim_step(p, v)
                    else:
                        eng.expand_backup(p, v)
            a, ne, vis, w = eng.choose_and_advance()
            tot = vis.sum(dim=1).cpu().numpy()
            assert np.all(tot[fresh] == S) and np.all(tot[~fresh] >= S)
            assert np.all((a.cpu().numpy() >= 0) & (a.cpu().numpy() < 7))
        st = eng.stats()
        assert st['sims'] == G * S * 3 and st['errors'] == 0 and st['moves'] == G * 3
        last.append((a.cpu().numpy().copy(), vis.cpu().numpy().copy()))
        eng.close()
    np.testing.assert_array_equal(last[0][0], last[1][0])
    np.testing.assert_array_equal(last[0][1], last[1][1])
