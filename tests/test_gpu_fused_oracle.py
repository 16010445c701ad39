"""The one-launch-per-move kernel (az_c4_move_fused: what bench.py times) against the CPU oracle
directly, with the network in the loop on both sides.

HIP side: az_apply_root(eta) -> az_c4_move_fused -> az_choose_and_advance(u) for G concurrent games.
Oracle side: the same games one at a time (oracle/az_oracle.c, alphazero/mcts.py:110-158 and
selfplay.py:24-38 restated), every leaf evaluated by the same HIP network kernel on a batch of one --
its rows are pure functions of the position (test_gpu_net.py::test_fused_forward_is_independent_of_batch_shape),
so both sides see identical leaf evaluations.  Done = every move, every visit vector and the bits of the
root's W and P are equal."""
import numpy as np
import pytest
import torch

import oracle as orc
from helpers import load_lut

pytestmark = pytest.mark.gpu


def _single_leaf_evaluator(net, memo):
    """oracle evaluator: (game, history, valid) -> (p over the legal moves, v) from the HIP network."""
    from alphazero_amd import engine as E

    def fn(game, hist, valid):
        key = bytes(hist)
        if key not in memo:
            bb, _ = E.positions_from_history(1, [hist])
            p, v = net(bb, None)
            memo[key] = (p[0].cpu().numpy().copy(), float(v[0].item()))
        p, v = memo[key]
        return p[valid], v
    return fn


# wgs = workgroups the launch may use (0 = one per compute unit, i.e. one game per workgroup for these sizes);
# regime = how a workgroup with that many games evaluates its leaves (csrc/az_engine.hip, FUSED_FEW_GAMES /
# FUSED_OWN_HEADS): 'own' = one board per task, heads by the same wave; 'single' = one board per task, heads in
# shared batches; 'pairs' = two boards per task, heads in shared batches (what 8 192 games on 256 compute units run)
# prec = 'bf16' (az_c4_move_fused) or 'hx' (az_c4_move_fused_hx: the network at fp32 grade, fp16-pair operands; every
# leaf a one-board task whose worker runs its heads, whatever the number of games per workgroup)
@pytest.mark.parametrize('G,S,moves,depth,wgs,regime,prec', [
    (64, 48, 9, 4, 0, 'own', 'bf16'), (64, 200, 4, 2, 0, 'own', 'bf16'), (100, 33, 5, 3, 0, 'own', 'bf16'),
    (96, 48, 5, 3, 8, 'single', 'bf16'),         # 12 games per workgroup
    (256, 48, 5, 4, 8, 'pairs', 'bf16'),         # 32 games per workgroup
    (256, 200, 2, 2, 4, 'pairs', 'bf16'),        # 64 games per workgroup, the most a workgroup takes
    (250, 33, 3, 2, 5, 'pairs', 'bf16'),         # 50 per workgroup, the last one not full
    (512, 200, 3, 3, 16, 'pairs', 'bf16'),       # 32 per workgroup on 16 workgroups, 200 simulations
    (64, 48, 9, 4, 0, 'own', 'hx'), (100, 200, 3, 2, 0, 'own', 'hx'),
    (256, 48, 5, 4, 8, 'own', 'hx'),             # 32 games per workgroup
    (256, 200, 2, 2, 4, 'own', 'hx'),            # 64 games per workgroup
    (250, 33, 3, 2, 5, 'own', 'hx'),
])
def test_fused_move_kernel_matches_the_oracle(G, S, moves, depth, wgs, regime, prec):
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import FusedC4Net, FusedC4NetHX
    A = 7
    torch.manual_seed(17)
    model = AlphaZeroC4(6, 32).eval().cuda()
    net = FusedC4NetHX(model, 'cuda') if prec == 'hx' else FusedC4Net.from_module(model, 'cuda')
    lut_np = load_lut()
    rng = np.random.RandomState(1000 + G + S)
    etas = rng.dirichlet([0.3] * A, size=(G, moves))          # slot-indexed; the first n_edges are used
    us = rng.random_sample((G, moves))

    # ---- HIP: all games at once
    eng = E.SelfPlayEngine(E._lib.GAME_C4, G, S, exploration_depth=depth, reuse_search_tree=True, lut=lut_np, device=0)
    eng.limit_fused_workgroups(wgs)
    eng.fused_counters()
    got = []
    for mv in range(moves):
        bb, info = eng.begin_move()
        p, _ = net(bb, info)
        eng.apply_root(p, torch.from_numpy(np.ascontiguousarray(etas[:, mv])).cuda())
        eng.move_fused(net)
        roots = []
        for g in range(G):
            img, root = eng.get_tree(g)
            n = int(img[root, A, 0])
            roots.append((img[root, :n, 0].copy(), img[root, :n, 1].copy(), img[root, :n, 2].copy()))
        act, ne, vis, win = eng.choose_and_advance(torch.from_numpy(np.ascontiguousarray(us[:, mv])).cuda())
        got.append((act.cpu().numpy().copy(), ne.cpu().numpy().copy(), vis.cpu().numpy().copy(), roots))
    assert eng.stats()['errors'] == 0 and eng.error_flags() == 0
    # the launches really ran in the regime this case is for
    did = eng.fused_counters()
    if regime == 'own':
        assert did['pair_tasks'] == 0 and did['single_tasks'] > 0 and did['head_boards'] == did['head_batches']
    elif regime == 'single':
        assert did['pair_tasks'] == 0 and did['single_tasks'] > 0 and did['head_boards'] > did['head_batches'], did
    else:
        assert did['pair_tasks'] > 4 * did['single_tasks'] and did['head_boards'] > 1.5 * did['head_batches'], did
    assert did['head_boards'] == 2 * did['pair_tasks'] + did['single_tasks']
    eng.close()

    # ---- oracle: one game at a time, the same eta / u per (game, move)
    lut = orc.make_lut_array(lut_np)
    cfg = orc.make_config(orc.C4, S, depth, True, lut)
    memo = {}
    ev = orc.eval_from_python(_single_leaf_evaluator(net, memo))
    checked = 0
    for k in range(G):
        hist, tree = [], orc.Node()
        top = tree
        for mv in range(moves):
            if orc.terminal(orc.C4, hist):
                break                                           # the slot restarted with a new game
            nv = len(orc.valid_actions(orc.C4, hist))
            orc.run_mcts(cfg, hist, tree, ev, orc.dirichlet_from_list([etas[k, mv, :nv]]))
            N, W, P = tree.edges()
            act, ne, vis, roots = got[mv]
            assert int(ne[k]) == len(N), (k, mv)
            assert list(vis[k, :len(N)]) == list(N), (k, mv, list(N), list(vis[k]))
            assert roots[k][0].tolist() == list(N), (k, mv)
            assert roots[k][1].tobytes() == W.view(np.uint32).tobytes(), (k, mv, 'W')
            assert roots[k][2].tobytes() == P.view(np.uint32).tobytes(), (k, mv, 'P')
            ai = tree.sample_action(us[k, mv]) if len(hist) < depth else tree.greedy_action()
            a = orc.valid_actions(orc.C4, hist)[ai]
            assert a == int(act[k]), (k, mv)
            hist.append(a)
            tree = tree.traverse(ai)
            tree.kill_siblings()
            checked += 1
        del top
    assert checked >= G * min(moves, 7) * 0.9


@pytest.mark.gpu
def test_long_launch_with_idle_waves_is_not_abandoned():
    """One game per workgroup and 8 000 simulations: six or seven of a workgroup's eight waves have nothing to do
    for the whole launch (> 100 ms).  The kernel's watchdog must count only polls during which the workgroup as a
    whole made no progress: every simulation is done and no error bit is raised."""
    import torch
    from alphazero_amd import engine as E
    from alphazero_amd.models import AlphaZeroC4
    from alphazero_amd.net import FusedC4Net
    G, S = 8, 8000
    eng = E.SelfPlayEngine(E._lib.GAME_C4, G, S, exploration_depth=4, reuse_search_tree=False, seed=0, device=0)
    torch.manual_seed(0)
    net = FusedC4Net.from_module(AlphaZeroC4(6, 32).eval().cuda(), 'cuda')
    bb, info = eng.begin_move()
    p, _ = net(bb, info)
    eng.apply_root(p)
    eng.move_fused(net)
    a, ne, vis, w = eng.choose_and_advance()
    eng.drain_finished()
    assert vis.sum(1).tolist() == [S] * G
    assert eng.error_flags() == 0
    eng.close()
